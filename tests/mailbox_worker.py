#!/usr/bin/env python3
"""Ranks of a gloo group hammer slab.DistTransport's host-scalar exchanges (shared-memory mailbox on one node)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap  # noqa: E402

_bootstrap.load()
import torch.distributed as dist  # noqa: E402

from chrono_wisc_amd import slab  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
tr = slab.DistTransport(dist, None)
assert (tr.mailbox is not None) == (os.environ.get("CHSPH_NO_HOST_MAILBOX") is None)
peers = [p for p in (rank - 1, rank + 1) if 0 <= p < world]
for it in range(3000):
    got = tr.exchange_ints({p: 1000003 * it + 17 * rank + p for p in peers}, peers)
    assert got == {p: 1000003 * it + 17 * p + rank for p in peers}, (rank, it, got)
    if it % 10 == 0:
        v = tr.allreduce_min_float(1.0 / (1 + ((it + rank) % world)) + it)
        assert v == 1.0 / world + it, (rank, it, v)
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("mailbox ok")

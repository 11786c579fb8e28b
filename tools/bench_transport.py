#!/usr/bin/env python3
"""Per-call cost of slab.DistTransport over RCCL on one GPU (world_size 1, rank sends to itself): host + launch overhead of
the count exchange, of a record-sized device message and of the all-reduces.  No xGMI link is involved."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import torch, torch.distributed as dist
from chrono_wisc_amd import slab
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:29541")
tr = slab.DistTransport(dist, dev)
tr.shared_stream = True
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("exchange_ints (one peer): %.1f us" % timeit(lambda: tr.exchange_ints({0: 7}, [0])))
for mb in (0.01, 1, 5, 10):
    n = int(mb * 1e6) // 4
    a = torch.zeros(n, dtype=torch.int32, device=dev); b = torch.zeros_like(a)
    us = timeit(lambda: tr.exchange({0: (a.data_ptr(), n * 4, 1)}, {0: (b.data_ptr(), n * 4, 1)}))
    print("exchange %.2f MB device message to self: %.1f us (%.1f GB/s)" % (mb, us, n * 4 / us / 1e3))
w = torch.zeros(2048, dtype=torch.int32, device=dev)
print("allreduce_max_words(2048): %.1f us" % timeit(lambda: tr.allreduce_max_words(w.data_ptr(), 2048, 1)))
print("allreduce_min_float: %.1f us" % timeit(lambda: tr.allreduce_min_float(0.5)))
dist.destroy_process_group()

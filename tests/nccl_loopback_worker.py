#!/usr/bin/env python3
"""RCCL plumbing of slab.DistTransport on ONE GPU: a world_size-1 "nccl" group whose only rank sends to itself.  Covers what
a single-GPU box can cover of the multi-GPU transport: zero-copy tensor views of the library's device buffers, grouped
isend/irecv on device memory, the count exchange, the max / min all-reduces, and stream ordering with the solver running on
torch's current stream (pack -> send/recv -> unpack without host synchronisation)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap  # noqa: E402

_bootstrap.load()
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from chrono_wisc_amd import capi, scenes, slab  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:%s" % os.environ.get("MASTER_PORT", "29533"))
tr = slab.DistTransport(dist, dev)
assert tr.backend == "nccl"

# raw device buffers
a = torch.arange(1 << 20, dtype=torch.int32, device=dev)
b = torch.zeros_like(a)
tr.exchange({0: (a.data_ptr(), a.numel() * 4, 1)}, {0: (b.data_ptr(), b.numel() * 4, 1)})
torch.cuda.synchronize()
assert torch.equal(a, b), "device send/recv to self"
assert tr.exchange_ints({0: 12345}, [0]) == {0: 12345}
w = torch.tensor([3, 9, 1, 7], dtype=torch.int32, device=dev)
tr.allreduce_max_words(w.data_ptr(), 4, 1)
assert w.tolist() == [3, 9, 1, 7]
assert tr.allreduce_min_float(0.25) == 0.25
# the all-reduces return early in a group of one: run the RCCL calls themselves (int32 views of a device buffer, MAX and SUM)
tr.nranks = 2
tr.allreduce_max_words(w.data_ptr(), 4, 1)
tr.allreduce_sum_words(w.data_ptr(), 4, 1)
tr.nranks = 1
torch.cuda.synchronize()
assert w.tolist() == [3, 9, 1, 7]

# the library's own buffers, solver on torch's stream: halo pack -> send to self -> unpack into the ghost rows of a second
# context that holds the mirrored slab
api = capi.CApi()
ranks = []
for r in range(2):
    s = scenes.dam_break_named("tiny", api, initialize=False, device=0)
    cuts = s.InitializeSlab(r, 2, [0, 13, 40])
    ranks.append(slab.SlabRank(s, tr, r, 2, cuts))
assert all(r.shared_stream for r in ranks)


class Pair:
    """Both slab ranks live in this process; every exchange is one grouped batch of sends and receives to self, in the
    fixed order rank 0 -> rank 1, then rank 1 -> rank 0 (NCCL matches sends and receives of a peer in issue order)."""

    def __init__(self):
        self.pending = {}

    def run(self, h, nsteps):
        import threading
        bar = threading.Barrier(2)
        box = {}
        errs = []

        class Ep:
            def __init__(ep, rank):
                ep.rank, ep.nranks, ep.backend, ep.device, ep.shared_stream = rank, 2, "nccl", dev, True

            def stream_handle(ep):
                return tr.stream_handle()

            def exchange_ints(ep, sends, peers):
                box[("i", ep.rank)] = sends
                bar.wait()
                out = {p: box[("i", p)][ep.rank] for p in peers}
                bar.wait()
                return out

            def exchange(ep, sends, recvs):
                box[("x", ep.rank)] = (sends, recvs)
                bar.wait()
                if ep.rank == 0:  # one thread issues the whole group
                    ops, keep = [], []
                    for src in (0, 1):
                        s_src, _ = box[("x", src)]
                        for dst, (ptr, nb, devflag) in s_src.items():
                            rptr, rnb, _d = box[("x", dst)][1][src]
                            assert rnb == nb, "message size mismatch %d vs %d" % (nb, rnb)
                            if nb == 0:
                                continue
                            ts, _ = tr._tensor(ptr, nb, devflag, False)
                            trv, _ = tr._tensor(rptr, rnb, 1, True)
                            keep += [ts, trv]
                            ops.append(dist.P2POp(dist.isend, ts, 0))
                            ops.append(dist.P2POp(dist.irecv, trv, 0))
                    if ops:
                        for wk in dist.batch_isend_irecv(ops):
                            wk.wait()
                bar.wait()

        eps = [Ep(0), Ep(1)]
        for r, ep in zip(ranks, eps):
            r.ep = ep

        def work(r):
            try:
                for _ in range(nsteps):
                    r.step(h)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
                bar.abort()

        ts = [threading.Thread(target=work, args=(r,)) for r in ranks]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]


h = 2e-4
Pair().run(h, 40)
got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
one = scenes.dam_break_named("tiny", api, device=0)
for _ in range(40):
    one.DoStepDynamics(h)
sv = one.solver
sv.synchronize()
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives()}
for k in ("pos", "vel", "rpm", "deriv"):
    np.testing.assert_array_equal(got[k], ref[k], err_msg=k)

# the same with the CRM soil column: the stress companions of every refresh and halo message travel through RCCL as well
kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})
ranks = []
for r in range(2):
    s = scenes.granular_column(api, dt=1e-4, initialize=False, device=0, **kw)
    ranks.append(slab.SlabRank(s, tr, r, 2, s.InitializeSlab(r, 2)))
assert all(r.shared_stream and r.crm for r in ranks)
Pair().run(1e-4, 30)
pieces = []
for r in ranks:
    pc = r.owned()
    pc["stress"], pc["pc"] = r.sys.solver.stress(), r.sys.solver.consolidation()
    pieces.append(pc)
got = slab.gather_global(pieces, ranks[0].sys.n_global, extra=("stress", "pc"))
one = scenes.granular_column(api, dt=1e-4, device=0, **kw)
for _ in range(30):
    one.DoStepDynamics(1e-4)
sv = one.solver
sv.synchronize()
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives(), "stress": sv.stress(), "pc": sv.consolidation()}
for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
    np.testing.assert_array_equal(got[k], ref[k], err_msg="crm " + k)
dist.destroy_process_group()
print("nccl loopback ok")

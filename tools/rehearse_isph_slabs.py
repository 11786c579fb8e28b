#!/usr/bin/env python3
"""Scale rehearsal of I2SPH under slabs on ONE GPU: the implicit dam break (label, Jacobi cap) cut into N slab contexts that live in
one process (in-process transport), compared bit for bit -- fields and sweep counts -- with the single-context run.
usage: rehearse_isph_slabs.py [label=C5geom] [nranks=8] [max_iters=50] [steps=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab
label = sys.argv[1] if len(sys.argv) > 1 else "C5geom"
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
max_iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
api = capi.CApi()
dt = 5e-4 * scenes.DAM_BREAK_CONFIGS[label][1] / 0.05
t0 = time.time()
tr = slab.ThreadTransport(nranks, device_direct=("direct" in sys.argv))
ranks = []
for r in range(nranks):
    s = scenes.dam_break_implicit(api, label, max_num_iters=max_iters, initialize=False, check_errors=True)
    cuts = s.InitializeSlab(r, nranks)
    ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, cuts))
    print("rank %d: columns [%d, %d), %d owned markers (%.0f s)" % (r, cuts[r], cuts[r + 1], s.GetCounters().numAllMarkers, time.time() - t0), flush=True)
t1 = time.time()
slab.run_threads(ranks, steps, dt)
for r in ranks:
    r.sys.Synchronize()
print("%d lockstep steps of all %d ranks on one GPU: %.1f ms per step (sum over ranks), sweeps %s" %
      (steps, nranks, (time.time() - t1) / steps * 1e3, ranks[0].sys.solver.isph_iterations()), flush=True)
got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
its = [r.sys.solver.isph_iterations() for r in ranks]
del ranks
one = scenes.dam_break_implicit(api, label, max_num_iters=max_iters, check_errors=True)
t2 = time.time()
for _ in range(steps):
    one.DoStepDynamics(dt)
sv = one.solver
sv.synchronize()
print("single context: %.1f ms per step, sweeps %s" % ((time.time() - t2) / steps * 1e3, sv.isph_iterations()), flush=True)
assert all(i == sv.isph_iterations() for i in its), its
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives()}
for k in ("pos", "vel", "rpm", "deriv"):
    bad = int((got[k] != ref[k]).any(axis=1).sum())
    print("%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref[k])), flush=True)
    assert bad == 0
print("I2SPH slab rehearsal ok (%.0f s)" % (time.time() - t0))

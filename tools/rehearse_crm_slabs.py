#!/usr/bin/env python3
"""Scale rehearsal of the granular multi-GPU workload on ONE GPU: the CRM soil column of demo_FSI-SPH_AngleRepose.cpp (1.5 M markers
at its own size) cut into N slab contexts in one process (in-process transport, device-to-device messages), stepped in lockstep
and compared bit for bit -- state, stress tensor, consolidation state -- with the single-context run.
usage: rehearse_crm_slabs.py [d0=0.01] [nranks=4] [steps=10]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab
d0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dt = 1e-4 * d0 / 0.01
kw = dict(radius=0.5, height=1.0, d0=d0, dt=dt, check_errors=True)
api = capi.CApi()
t0 = time.time()
tr = slab.ThreadTransport(nranks, device_direct=True)
ranks = []
for r in range(nranks):
    s = scenes.granular_column(api, initialize=False, **kw)
    cuts = s.InitializeSlab(r, nranks)
    ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, cuts))
    print("rank %d: columns [%d, %d), %d owned markers (%.0f s)" % (r, cuts[r], cuts[r + 1], s.GetCounters().numAllMarkers, time.time() - t0), flush=True)
slab.run_threads(ranks, steps, dt)
for r in ranks:
    r.sys.Synchronize()
t1 = time.time()
slab.run_threads(ranks, steps, dt)
for r in ranks:
    r.sys.Synchronize()
print("%d lockstep CRM steps of all %d ranks on one GPU: %.2f ms per step (sum over ranks), one-round %s" % (steps, nranks, (time.time() - t1) / steps * 1e3,
      [r.api.slab_one_round(r.ctx) for r in ranks]), flush=True)
pieces = []
for r in ranks:
    pc = r.owned()
    pc["stress"], pc["pc"] = r.sys.solver.stress(), r.sys.solver.consolidation()
    pieces.append(pc)
got = slab.gather_global(pieces, ranks[0].sys.n_global, extra=("stress", "pc"))
del ranks
one = scenes.granular_column(api, **kw)
for _ in range(steps):
    one.DoStepDynamics(dt)
one.Synchronize()
t1 = time.time()
for _ in range(steps):
    one.DoStepDynamics(dt)
one.Synchronize()
print("single context: %.2f ms per step, %d markers" % ((time.time() - t1) / steps * 1e3, one.GetCounters().numAllMarkers), flush=True)
sv = one.solver
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives(), "stress": sv.stress(), "pc": sv.consolidation()}
for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
    bad = int((got[k] != ref[k]).any(axis=1).sum())
    print("%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref[k])), flush=True)
    assert bad == 0
print("CRM slab rehearsal ok (%.0f s)" % (time.time() - t0))

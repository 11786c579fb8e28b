#!/usr/bin/env python3
"""Scale rehearsal of the coupled multi-GPU workload on ONE GPU: the C4 cylinder drop cut into N slab contexts that live in one
process (in-process transport), each advancing its own copy of the body from the all-reduced wrench table, stepped in lockstep
and compared bit for bit (fields, wrench, body state) with the single-context run.  The cylinder starts partly submerged so the
wrench is non-zero from the first step.
usage: rehearse_c4_slabs.py [d0=0.0054] [nranks=8] [steps=20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab
d0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0054
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
h = 2e-5 * d0 / 0.0054
kw = dict(initial_spacing=d0, variable_dt=False, step_size=h, check_errors=True, initial_height=0.9)
push = np.array([-0.5, 0.0, -1.0])
api = capi.CApi()


class FsiRank:
    def __init__(self, fsys, rank):
        self.fsys, self.rank, self.ep = fsys, rank, rank.ep
        fsys.advance_cfd = rank.step

    def step(self, hh):
        self.fsys.DoStepDynamics(hh)


t0 = time.time()
tr = slab.ThreadTransport(nranks)
ranks = []
for r in range(nranks):
    s, f = scenes.object_drop(api, initialize=False, **kw)
    f.bodies[0].lin_vel = push.copy()
    cuts = s.InitializeSlab(r, nranks, body_states=[b.state() for b in f.bodies])
    ranks.append(FsiRank(f, slab.SlabRank(s, tr.endpoint(r), r, nranks, cuts)))
    rigid0 = s.n_global - len(s._bodies[0][0])
    print("rank %d: columns [%d, %d), %d owned markers, %d of them rigid (%.0f s)" % (r, cuts[r], cuts[r + 1], s.GetCounters().numAllMarkers,
          int((s.solver.global_ids() >= rigid0).sum()), time.time() - t0), flush=True)
slab.run_threads(ranks, steps, h)
for r in ranks:
    r.rank.sys.Synchronize()
t1 = time.time()
slab.run_threads(ranks, steps, h)
for r in ranks:
    r.rank.sys.Synchronize()
print("%d lockstep coupled steps of all %d ranks on one GPU: %.2f ms per step (sum over ranks)" % (steps, nranks, (time.time() - t1) / steps * 1e3), flush=True)
got = slab.gather_global([r.rank.owned() for r in ranks], ranks[0].rank.sys.n_global)
wr = [np.concatenate(r.rank.sys.GetBodyWrenches()[0]) for r in ranks]
bodies = [(r.fsys.bodies[0].pos.copy(), r.fsys.bodies[0].rot.copy()) for r in ranks]
del ranks
s, f = scenes.object_drop(api, initialize=False, **kw)
f.bodies[0].lin_vel = push.copy()
s.Initialize([b.state() for b in f.bodies])
for _ in range(steps):
    f.DoStepDynamics(h)
s.Synchronize()
t1 = time.time()
for _ in range(steps):
    f.DoStepDynamics(h)
s.Synchronize()
print("single context: %.2f ms per coupled step, %d markers" % ((time.time() - t1) / steps * 1e3, s.GetCounters().numAllMarkers), flush=True)
sv = s.solver
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives()}
for k in ("pos", "vel", "rpm", "deriv"):
    bad = int((got[k] != ref[k]).any(axis=1).sum())
    print("%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref[k])), flush=True)
    assert bad == 0
w1 = np.concatenate(s.GetBodyWrenches()[0])
print("wrench (single context):", w1, flush=True)
assert np.abs(w1).max() > 0
for w, (bp, br) in zip(wr, bodies):
    assert np.array_equal(w, w1) and np.array_equal(bp, f.bodies[0].pos) and np.array_equal(br, f.bodies[0].rot)
print("coupled slab rehearsal ok: wrench and body state bit-identical on all %d ranks (%.0f s)" % (nranks, time.time() - t0))

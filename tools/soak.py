#!/usr/bin/env python3
"""Soak runs on the GPU: long single-device and slab runs of the dam break (the wave front crosses the slab cuts, ghosts and owners
change all the time) compared bit for bit at the end, plus a long CRM collapse.  usage: soak.py [steps=3000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
api = capi.CApi()
label, h = "C1", 2e-4   # the reference demo verbatim (30 327 markers), a step near its stability limit: fast collapse
t0 = time.time()
one = scenes.dam_break_named(label, api, check_errors=True)
for k in range(steps):
    one.DoStepDynamics(h)
sv = one.solver; sv.synchronize()
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives()}
print("single device: %d steps, max |v| %.2f, front x max %.3f (%.0f s)" % (steps, np.abs(ref["vel"]).max(), ref["pos"][:one.GetCounters().numFluidMarkers, 0].max(), time.time() - t0), flush=True)
for nranks in (2, 4):
    t1 = time.time()
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named(label, api, initialize=False, check_errors=True)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks)))
    slab.run_threads(ranks, steps, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    owned = [r.sys.GetCounters().numAllMarkers for r in ranks]
    for k in ("pos", "vel", "rpm", "deriv"):
        assert int((got[k] != ref[k]).any(axis=1).sum()) == 0, (nranks, k)
    print("%d slabs, static cuts: bit-identical after %d steps; owned markers per rank now %s" % (nranks, steps, owned), flush=True)
    # the same run with the cuts re-balanced every 500 steps
    import threading
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named(label, api, initialize=False, check_errors=True)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks)))

    def work(r):
        for k in range(steps):
            if k and k % 500 == 0:
                r.recut()
            r.step(h)

    ts = [threading.Thread(target=work, args=(r,)) for r in ranks]
    for t in ts: t.start()
    for t in ts: t.join()
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    owned = [r.sys.GetCounters().numAllMarkers for r in ranks]
    for k in ("pos", "vel", "rpm", "deriv"):
        bad = int((got[k] != ref[k]).any(axis=1).sum())
        assert bad == 0, (nranks, k, bad)
    print("%d slabs, re-cut every 500 steps: bit-identical after %d steps; owned markers per rank now %s, cuts %s (%.0f s)" % (nranks, steps, owned, ranks[0].cuts, time.time() - t1), flush=True)
    del ranks
t2 = time.time()
g = scenes.granular_column(api, radius=0.1, height=0.2, d0=0.01, dt=1e-4, box_factor=6, check_errors=True)
n = g.GetCounters().numFluidMarkers
for k in range(steps):
    g.DoStepDynamics(1e-4)
p = g.GetParticlePositions()[:n]
print("CRM column: %d markers, %d steps, spread radius %.3f, max |v| %.3f, finite %s (%.0f s)" %
      (g.GetCounters().numAllMarkers, steps, np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2).max(), np.abs(g.GetParticleVelocities()[:n]).max(), bool(np.isfinite(g.GetParticleStress()).all()), time.time() - t2), flush=True)
print("soak ok")

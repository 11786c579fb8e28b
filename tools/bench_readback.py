#!/usr/bin/env python3
"""Step time when the caller reads the whole marker state back to host memory after EVERY step (positions, velocities, rho / p / mu / type: 44 B per marker
over PCIe, plus the copy from cell order to marker order that the resident state otherwise skips): bench_readback.py [label=C2] [steps=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
api = capi.CApi()
sc, d0, dt = scenes.DAM_BREAK_CONFIGS[label]
s = scenes.dam_break(api, s=sc, d0=d0, dt=dt, check_errors=False)
for _ in range(20): s.DoStepDynamics(dt)
s.Synchronize()
t0 = time.perf_counter()
for _ in range(nsteps): s.DoStepDynamics(dt)
s.Synchronize()
a = (time.perf_counter() - t0) / nsteps * 1e3
t0 = time.perf_counter()
nbytes = 0
for _ in range(nsteps):
    s.DoStepDynamics(dt)
    p, v, r = s.solver.positions(), s.solver.velocities(), s.solver.rho_pres_mu_type()
    nbytes = p.nbytes + v.nbytes + r.nbytes
b = (time.perf_counter() - t0) / nsteps * 1e3
n = s.GetCounters().numAllMarkers
print("%s: %d markers; %.3f ms/step resident (%.3g particle-steps/s); %.3f ms/step with %.1f MB read back per step (%.3g particle-steps/s)" % (label, n, a, n / a * 1e3, b, nbytes / 1e6, n / b * 1e3))

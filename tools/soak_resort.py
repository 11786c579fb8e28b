#!/usr/bin/env python3
"""Long run of the C2 dam break with the default re-sort / search (one scan launch, arrays left out, packed y-edge scan) against the round-3 sequence
(CHSPH_NO_SCAN_SINGLE, CHSPH_EAGER_MAP, CHSPH_EAGER_SXT, CHSPH_NO_PACKED_YWRAP), two contexts in one process stepped side by side, every field compared
bit for bit every `every` steps:  soak_resort.py [label=C2] [steps=5000] [every=1000] [exact]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
api = capi.CApi()
sc, d0, dt = scenes.DAM_BREAK_CONFIGS[label]
a = scenes.dam_break(api, s=sc, d0=d0, dt=dt, check_errors=False)
OLD = {"CHSPH_NO_SCAN_SINGLE": "1", "CHSPH_EAGER_MAP": "1", "CHSPH_EAGER_SXT": "1", "CHSPH_NO_PACKED_YWRAP": "1", "CHSPH_NO_ADAMI_WET": "1", "CHSPH_NO_ADAMI_PUBLISH": "1"}
os.environ.update(OLD)
b = scenes.dam_break(api, s=sc, d0=d0, dt=dt, check_errors=False)
for k in OLD: del os.environ[k]
if "exact" in sys.argv[4:]:
    a.solver.set_arithmetic(0); b.solver.set_arithmetic(0)
t0 = time.perf_counter()
bad = 0
for k in range(1, nsteps + 1):
    a.DoStepDynamics(dt); b.DoStepDynamics(dt)
    if k % every == 0 or k == nsteps:
        for name, fa, fb in (("pos", a.solver.positions(), b.solver.positions()), ("vel", a.solver.velocities(), b.solver.velocities()),
                             ("rpm", a.solver.rho_pres_mu_type(), b.solver.rho_pres_mu_type())):
            n = int((fa != fb).any(axis=1).sum())
            bad += n
            if n: print("step %d: %s differs on %d markers" % (k, name, n), flush=True)
        p = a.solver.positions()
        print("step %d: compared, x front %.4f m, z max %.4f m, %.1f s" % (k, float(p[:a.GetCounters().numFluidMarkers, 0].max()),
              float(p[:a.GetCounters().numFluidMarkers, 2].max()), time.perf_counter() - t0), flush=True)
print("soak_resort %s %d steps: %s" % (label, nsteps, "every field bit-identical" if bad == 0 else "%d differences" % bad))
sys.exit(0 if bad == 0 else 1)

#!/usr/bin/env python3
"""CRM granular path timing: the soil column of demo_FSI-SPH_AngleRepose.cpp (radius 0.5 m, height 1 m, spacing 0.01 m by default).
usage: bench_crm.py [d0=0.01] [steps=100]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
d0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
api = capi.CApi(os.environ["CHSPH_LIB"]) if os.environ.get("CHSPH_LIB") else capi.CApi()  # CHSPH_LIB: a variant build (tools/build_variant.sh)
dt = 1e-4 * d0 / 0.01
s = scenes.granular_column(api, radius=0.5, height=1.0, d0=d0, dt=dt, check_errors=False)
c = s.GetCounters()
for _ in range(10): s.DoStepDynamics(dt)
s.Synchronize(); t0 = time.perf_counter()
for _ in range(nsteps): s.DoStepDynamics(dt)
s.Synchronize(); ms = (time.perf_counter() - t0) / nsteps * 1e3
api.profile_enable(s.solver.ctx, 1)
for _ in range(10): s.DoStepDynamics(dt)
s.Synchronize()
parts = []
for name in ("sort", "neigh", "adami", "force", "integrate", "copy", "copyback"):
    t, k = C.c_double(), C.c_uint64()
    if api.profile_get(s.solver.ctx, name.encode(), C.byref(t), C.byref(k)) == 0 and k.value:
        parts.append("%s %.0f us" % (name, t.value / 10 * 1e3))
print("CRM soil column d0=%g: %d markers (%d soil), %.3f ms/step = %.3g particle-steps/s  [%s]" % (d0, c.numAllMarkers, c.numFluidMarkers, ms, c.numAllMarkers / ms * 1e3, ", ".join(parts)))

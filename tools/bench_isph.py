#!/usr/bin/env python3
"""I2SPH timing harness: dam-break input run with the implicit scheme (label, Jacobi cap) or the Couette demo.
Prints ms per step and the per-phase averages of the solver's built-in event profile."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, fsi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
max_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
api = capi.CApi()
if label == "couette":
    s = scenes.couette(api, max_num_iters=max_iters, check_errors=False)
    dt = scenes.COUETTE["step_size"]
    step = lambda: scenes.couette_step(s)
else:
    dt = 5e-4 * scenes.DAM_BREAK_CONFIGS[label][1] / 0.05
    s = scenes.dam_break_implicit(api, label, max_num_iters=max_iters, check_errors=False)
    step = lambda: s.DoStepDynamics(dt)
n = s.GetCounters().numAllMarkers
step(); s.Synchronize()
t0 = time.perf_counter()
for _ in range(nsteps): step()
s.Synchronize()
ms_step = (time.perf_counter() - t0) / nsteps * 1e3
api.profile_enable(s.solver.ctx, 1)
for _ in range(2): step()
out = {}
for name in ("step", "sort", "neigh", "isph_operators", "isph_vstar", "isph_jacobi_v", "isph_ppe", "isph_jacobi_p", "isph_correct", "isph_operators2", "isph_shift", "copyback"):
    tot = C.c_double(); k = C.c_uint64()
    api.profile_get(s.solver.ctx, name.encode(), C.byref(tot), C.byref(k))
    out[name] = round(tot.value / max(k.value, 1), 3)
it = s.solver.isph_iterations()
print("%s: %d markers, %.2f ms/step (%.3g marker-steps/s), sweeps %s, per-phase ms: %s" % (label, n, ms_step, n / ms_step * 1e3, it, out))

#!/usr/bin/env python3
"""Host/device overhead of the slab stepping path: a dam break on ONE rank (no neighbours) through SlabRank.step vs sph_step,
with the per-phase device times of the library's profile scopes."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes, slab
api = capi.CApi()
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
dt = scenes.DAM_BREAK_CONFIGS[label][2]
SCOPES = ["step", "sort", "neigh", "adami", "force", "integrate", "copy", "copyback"]
def timeit(sysS, step, n=100):
    for _ in range(10): step()
    sysS.Synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    sysS.Synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
    api.profile_enable(sysS.solver.ctx, 1)
    for _ in range(20): step()
    sysS.Synchronize()
    parts = []
    for name in SCOPES:
        ms, k = C.c_double(), C.c_uint64()
        if api.profile_get(sysS.solver.ctx, name.encode(), C.byref(ms), C.byref(k)) == 0 and k.value:
            parts.append("%s %.0f us" % (name, ms.value / 20 * 1e3))
    api.profile_enable(sysS.solver.ctx, 0)
    return "%.3f ms/step  [%s]" % (wall, ", ".join(parts))
a = scenes.dam_break_named(label, api, check_errors=False)
print("sph_step:", timeit(a, lambda: a.DoStepDynamics(dt)))
os.environ["CHSPH_NO_RESIDENT"] = "1"
b = scenes.dam_break_named(label, api, check_errors=False)
print("sph_step, reference sequence (no resident state):", timeit(b, lambda: b.DoStepDynamics(dt)))
del os.environ["CHSPH_NO_RESIDENT"]
s = scenes.dam_break_named(label, api, check_errors=False, initialize=False)
cuts = s.InitializeSlab(0, 1)
r = slab.SlabRank(s, slab.ThreadTransport(1).endpoint(0), 0, 1, cuts)
print("SlabRank.step, 1 rank:", timeit(s, lambda: r.step(dt)))

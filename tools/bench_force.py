#!/usr/bin/env python3
"""Quick A/B harness: build a dam-break config, run a few steps, time the force kernel and one step."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
api = capi.CApi()
s = scenes.dam_break_named(label, api, check_errors=False)
dt = scenes.DAM_BREAK_CONFIGS[label][2]
for _ in range(5): s.DoStepDynamics(dt)
s.Synchronize()
t0 = time.perf_counter()
for _ in range(nsteps): s.DoStepDynamics(dt)
s.Synchronize()
step_ms = (time.perf_counter() - t0) / nsteps * 1e3
api.profile_enable(s.solver.ctx, 1)
for _ in range(5): s.DoStepDynamics(dt)
s.Synchronize()
ms = C.c_double()
s.solver._ck(api.bench_rhs(s.solver.ctx, 30, C.byref(ms)))
out = {}
for name in ("step", "sort", "neigh", "adami", "force", "integrate", "copy", "copyback"):
    tot = C.c_double(); n = C.c_uint64()
    api.profile_get(s.solver.ctx, name.encode(), C.byref(tot), C.byref(n))
    out[name] = (round(tot.value / max(n.value, 1) * 1e3, 1), n.value)
print("%s: step %.3f ms; force kernel %.1f us; per-phase us (avg, launches): %s" % (label, step_ms, ms.value * 1e3, out))

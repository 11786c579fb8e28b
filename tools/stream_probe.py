#!/usr/bin/env python3
"""Does the force kernel's per-process speed mode depend on the HIP stream (hardware queue) it runs on?  One context, the isolated
force launch timed on the context's own stream and on n freshly created streams (sph_set_stream)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
api = capi.CApi()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hip = C.CDLL((capi.mapped_runtimes()["libamdhip64.so"] or ["libamdhip64.so"])[0])  # the runtime the solver already uses, never a second copy
s = scenes.dam_break_named("C2", api, check_errors=False)
for _ in range(10): s.DoStepDynamics(1e-5)
s.Synchronize()
ms = C.c_double()
s.solver._ck(api.bench_rhs(s.solver.ctx, 30, C.byref(ms)))
print("own stream: %.1f us" % (ms.value * 1e3), flush=True)
for k in range(n):
    st = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(st), C.c_uint(1)) == 0
    s.solver._ck(api.set_stream(s.solver.ctx, st))
    for _ in range(3): s.DoStepDynamics(1e-5)
    s.Synchronize()
    s.solver._ck(api.bench_rhs(s.solver.ctx, 30, C.byref(ms)))
    print("stream %d: %.1f us" % (k, ms.value * 1e3), flush=True)

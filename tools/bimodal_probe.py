#!/usr/bin/env python3
"""Is the ~10 % bimodality of the force kernel per process or per allocation?  Builds the C2 dam break several times in ONE process
(each context allocates all its buffers anew) and times the isolated force launch of each; with a second argument, an unrelated
device allocation of that many MiB is made (and kept) before every other context, so that the next context lands on other pages."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
api = capi.CApi()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shift_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hip = C.CDLL((capi.mapped_runtimes()["libamdhip64.so"] or ["libamdhip64.so"])[0])  # the runtime the solver already uses, never a second copy
for k in range(n):
    note = ""
    if shift_mib and k % 2 == 1:
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), C.c_size_t(shift_mib << 20))
        note = " (after an unrelated %d MiB allocation, rc %d)" % (shift_mib, rc)
    s = scenes.dam_break_named("C2", api, check_errors=False)
    for _ in range(10): s.DoStepDynamics(1e-5)
    s.Synchronize()
    ms = C.c_double()
    s.solver._ck(api.bench_rhs(s.solver.ctx, 30, C.byref(ms)))
    print("context %d: isolated force launch %.1f us%s" % (k, ms.value * 1e3, note), flush=True)
    del s

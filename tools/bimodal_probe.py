#!/usr/bin/env python3
"""Is the ~10 % bimodality of the force kernel per process or per allocation?  Builds the C2 dam break several times in ONE process
(each context allocates all its buffers anew) and times the isolated force launch of each."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
api = capi.CApi()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
keep = []
for k in range(n):
    s = scenes.dam_break_named("C2", api, check_errors=False)
    for _ in range(10): s.DoStepDynamics(1e-5)
    s.Synchronize()
    ms = C.c_double()
    s.solver._ck(api.bench_rhs(s.solver.ctx, 30, C.byref(ms)))
    print("context %d: isolated force launch %.1f us" % (k, ms.value * 1e3), flush=True)
    if k % 2 == 0: keep.append(s)   # keep every other context alive so that the next one lands elsewhere
    else: del s

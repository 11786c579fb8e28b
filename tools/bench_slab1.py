#!/usr/bin/env python3
"""Host/device overhead of the slab stepping path: C2 on ONE rank (no neighbours) through SlabRank.step vs sph_step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes, slab
api = capi.CApi()
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
dt = scenes.DAM_BREAK_CONFIGS[label][2]
def timeit(step, sync, n=100):
    for _ in range(10): step()
    sync(); t0 = time.perf_counter()
    for _ in range(n): step()
    sync(); return (time.perf_counter() - t0) / n * 1e3
a = scenes.dam_break_named(label, api, check_errors=False)
print("sph_step: %.3f ms" % timeit(lambda: a.DoStepDynamics(dt), a.Synchronize))
os.environ["CHSPH_NO_RESIDENT"] = "1"
b = scenes.dam_break_named(label, api, check_errors=False)
print("sph_step, reference sequence (no resident state): %.3f ms" % timeit(lambda: b.DoStepDynamics(dt), b.Synchronize))
s = scenes.dam_break_named(label, api, check_errors=False, initialize=False)
cuts = s.InitializeSlab(0, 1)
r = slab.SlabRank(s, slab.ThreadTransport(1).endpoint(0), 0, 1, cuts)
print("SlabRank.step, 1 rank: %.3f ms" % timeit(lambda: r.step(dt), s.Synchronize))

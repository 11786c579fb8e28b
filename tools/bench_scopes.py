#!/usr/bin/env python3
"""Per-phase HIP-event times of the dam-break step (library profile scopes): bench_scopes.py [label=C2] [steps=100] [d0_multiplier=1]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, fsi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mult = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
api = capi.CApi(os.environ["CHSPH_LIB"]) if os.environ.get("CHSPH_LIB") else capi.CApi()  # CHSPH_LIB: a variant build (tools/build_variant.sh)
sc, d0, dt = scenes.DAM_BREAK_CONFIGS[label]
def tweak(sp):
    sp.d0_multiplier = mult
s = scenes.dam_break(api, s=sc, d0=d0, dt=dt, tweak=tweak if mult != 1.0 else None, check_errors=False)
for _ in range(30): s.DoStepDynamics(dt)
s.Synchronize()
t0 = time.perf_counter()
for _ in range(nsteps): s.DoStepDynamics(dt)
s.Synchronize()
ms = (time.perf_counter() - t0) / nsteps * 1e3
api.profile_enable(s.solver.ctx, 1)
for _ in range(nsteps): s.DoStepDynamics(dt)
s.Synchronize()
out = []
for name in (b"sort", b"neigh", b"adami", b"force", b"integrate", b"holmes"):
    tot, cnt = C.c_double(), C.c_uint64()
    api.profile_get(s.solver.ctx, name, C.byref(tot), C.byref(cnt))
    if cnt.value:
        out.append("%s %.1f us x %.1f/step" % (name.decode(), tot.value / cnt.value * 1e3, cnt.value / nsteps))
print("%s (x%g): %d markers, %.4f ms/step | %s | env %s" % (label, mult, s.GetCounters().numAllMarkers, ms, "; ".join(out),
      {k: v for k, v in os.environ.items() if k.startswith("CHSPH_")}))

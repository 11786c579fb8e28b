#!/usr/bin/env python3
"""A/B of library experiment switches inside ONE process (the force kernel's speed mode belongs to the process, NOTEBOOK.md 3.3.1):
ab_exp.py [label=C2] [steps=300] [flags=0,8,0,8] [exact]  -- one context per entry of `flags`, CHSPH_EXP set before it is created
(bit 3 = 8: the force kernel walks its tiles forwards)."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
flags = (sys.argv[3] if len(sys.argv) > 3 else "0,8,0,8").split(",")  # an integer: CHSPH_EXP; NAME=VALUE: the switch CHSPH_NAME (read by sph_create)
api = capi.CApi(os.environ["CHSPH_LIB"]) if os.environ.get("CHSPH_LIB") else capi.CApi()  # CHSPH_LIB: a variant build (tools/build_variant.sh)
sc, d0, dt = scenes.DAM_BREAK_CONFIGS[label]
for f in flags:
    sw, val = ("EXP", f) if "=" not in f else f.split("=", 1)
    os.environ["CHSPH_" + sw] = val
    s = scenes.dam_break(api, s=sc, d0=d0, dt=dt, check_errors=False)
    if "exact" in sys.argv[4:]: s.solver.set_arithmetic(0)
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
    for name in (b"sort", b"neigh", b"stlists", b"adami", b"force"):
        tot, cnt = C.c_double(), C.c_uint64()
        api.profile_get(s.solver.ctx, name, C.byref(tot), C.byref(cnt))
        if cnt.value: out.append("%s %.1f" % (name.decode(), tot.value / cnt.value * 1e3))
    st = (C.c_uint32 * 4)()
    api.staged_tile_stats(s.solver.ctx, st)
    if st[0] + st[1]: out.append("tiles staged %d / gather %d / idle %d, max slots %d" % (st[0], st[1], st[2], st[3]))
    print("CHSPH_%s=%s: %.4f ms/step | %s" % (sw, val, ms, "; ".join(out)), flush=True)
    del s
    del os.environ["CHSPH_" + sw]

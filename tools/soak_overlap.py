#!/usr/bin/env python3
"""Soak of sph_slab_step's overlapped halo path: N ranks on one GPU (threads, device-to-device callbacks that declare themselves stream-ordered) for many
steps, owned results compared bit for bit with the single-context run at the end.  soak_overlap.py [label=tiny] [nranks=3] [steps=1000] [h=2e-4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("CHSPH_ARITH", "0")
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab
import util
label = sys.argv[1] if len(sys.argv) > 1 else "tiny"
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
h = float(sys.argv[4]) if len(sys.argv) > 4 else 2e-4
api = capi.CApi()
t0 = time.time()
tr = slab.ThreadTransport(nranks, device_direct=True)
ranks = []
for r in range(nranks):
    s = scenes.dam_break_named(label, api, initialize=False)
    ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks)))
    ranks[-1].native_ordered = True
done = 0
while done < steps:
    n = min(100, steps - done)
    slab.run_threads(ranks, n, h, native=True)
    done += n
    print("%d steps, overlapped halos per rank %s (%.0f s)" % (done, [api.slab_overlapped_halos(r.ctx) for r in ranks], time.time() - t0), flush=True)
got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
one = scenes.dam_break_named(label, api)
for _ in range(steps):
    one.DoStepDynamics(h)
ref = util.snapshot(one)
for k in ("pos", "vel", "rpm", "deriv"):
    bad = int((got[k] != ref[k]).any(axis=1).sum()) if got[k].ndim > 1 else int((got[k] != ref[k]).sum())
    print("%s: %d markers differ" % (k, bad), flush=True)
    assert bad == 0
print("soak ok (%.0f s)" % (time.time() - t0))

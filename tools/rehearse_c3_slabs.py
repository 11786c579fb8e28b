#!/usr/bin/env python3
"""Scale rehearsal of the multi-GPU workload on ONE GPU: the C3 dam break (17.3 M markers) cut into N slab contexts that live
in one process (in-process transport), stepped in lockstep and compared bit for bit with the single-context run.
usage: rehearse_c3_slabs.py [label=C3] [nranks=8] [steps=6] [direct] [native] [ordered] [noref]
   direct: device-to-device messages instead of host staging; native: the step orchestrated by the library (sph_slab_step);
   ordered: the callbacks declare themselves stream-ordered, so sph_slab_step overlaps the inter-stage halo with the force kernel
   (CHSPH_SLAB_NO_OVERLAP=1 switches that off again)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes, slab
label = sys.argv[1] if len(sys.argv) > 1 else "C3"
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
api = capi.CApi()
dt = scenes.DAM_BREAK_CONFIGS[label][2]
t0 = time.time()
native = "native" in sys.argv[4:]
tr = slab.ThreadTransport(nranks, device_direct=("direct" in sys.argv[4:]))
ranks = []
for r in range(nranks):
    s = scenes.dam_break_named(label, api, initialize=False, check_errors=True)
    cuts = s.InitializeSlab(r, nranks)
    ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, cuts))
    ranks[-1].native_ordered = "ordered" in sys.argv[4:]
    print("rank %d: columns [%d, %d), %d owned markers (%.0f s)" % (r, cuts[r], cuts[r + 1], s.GetCounters().numAllMarkers, time.time() - t0), flush=True)
slab.run_threads(ranks, steps, dt, native=native)
for r in ranks:
    r.sys.Synchronize()
t1 = time.time()
slab.run_threads(ranks, steps, dt, native=native)
for r in ranks:
    r.sys.Synchronize()
print("%d lockstep steps of all %d ranks on one GPU (%s driver): %.1f ms per step (sum over ranks) = %.2f ms per rank" % (steps, nranks, "library" if native else "slab.py", (time.time() - t1) / steps * 1e3, (time.time() - t1) / steps * 1e3 / nranks), flush=True)
for r in ranks:
    c = r.sys.GetCounters()
    print("rank %d: owned %d, extended rows %d, one-round %d, overlapped halos %d" % (r.rank, c.numAllMarkers, c.numExtendedParticles, r.api.slab_one_round(r.ctx), r.api.slab_overlapped_halos(r.ctx)), flush=True)
if "noref" in sys.argv[4:]:  # (profiling runs: the slab ranks' kernels only)
    print("no single-context reference run (%.0f s)" % (time.time() - t0))
    sys.exit(0)
got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
del ranks
one = scenes.dam_break_named(label, api, check_errors=True)
for _ in range(2 * steps):
    one.DoStepDynamics(dt)
sv = one.solver
sv.synchronize()
ref = {"pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives()}
for k in ("pos", "vel", "rpm", "deriv"):
    bad = int((got[k] != ref[k]).any(axis=1).sum())
    print("%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref[k])), flush=True)
    assert bad == 0
print("slab rehearsal ok (%.0f s)" % (time.time() - t0))

#!/usr/bin/env python3
"""Long-horizon sanity of SPH_ARITH_FAST against SPH_ARITH_EXACT: the same dam break advanced N steps in both modes; prints bulk
quantities (wave-front position, kinetic energy, density extrema) and the pointwise differences.  Pointwise agreement is a short-horizon
statement (10 steps, tests/); over thousands of steps two roundings of a chaotic free-surface flow drift apart marker by marker while
the bulk quantities stay together.  usage: long_run_compare.py [label=C2] [steps=20000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
api = capi.CApi()
dt = scenes.DAM_BREAK_CONFIGS[label][2]
out = {}
for mode in ("exact", "fast"):
    s = scenes.dam_break_named(label, api, check_errors=True)
    s.SetArithmetic(mode)
    t0 = time.time()
    for k in range(steps):
        s.DoStepDynamics(dt)
    s.Synchronize()
    nf = s.GetCounters().numFluidMarkers
    pos = s.GetParticlePositions()[:nf].astype(np.float64)
    vel = s.GetParticleVelocities()[:nf].astype(np.float64)
    rho = s.GetParticleFluidProperties()[:nf, 0].astype(np.float64)
    out[mode] = (pos, vel, rho)
    print("%-5s: %d steps (t = %.3f s) in %.1f s; front x max %.5f, mean x %.6f, kinetic energy / marker %.6e, max |v| %.4f, rho in [%.3f, %.3f], finite %s"
          % (mode, steps, steps * dt, time.time() - t0, pos[:, 0].max(), pos[:, 0].mean(), 0.5 * (vel ** 2).sum() / nf, np.abs(vel).max(), rho.min(), rho.max(),
             bool(np.isfinite(pos).all() and np.isfinite(vel).all())), flush=True)
(pe, ve, re_), (pf, vf, rf) = out["exact"], out["fast"]
print("fast vs exact: front %.2e, mean x %.2e, kinetic energy rel %.2e; pointwise max |dx| %.3e (h = %g), median |dx| %.3e"
      % (abs(pf[:, 0].max() - pe[:, 0].max()), abs(pf[:, 0].mean() - pe[:, 0].mean()), abs((vf ** 2).sum() / (ve ** 2).sum() - 1.0),
         np.abs(pf - pe).max(), scenes.DAM_BREAK_CONFIGS[label][1], np.median(np.abs(pf - pe).max(axis=1))))

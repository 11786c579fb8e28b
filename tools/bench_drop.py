#!/usr/bin/env python3
"""C4 timing: cylinder drop, one rigid-body exchange per fluid step, Chrono-free 6-DoF driver.
usage: bench_drop.py [d0=0.0054] [steps=200] [fixed|variable] [submerged]
   submerged: the cylinder starts partly inside the fluid (initial height 0.9 instead of the demo's 1.05), so the timed window includes the fluid wrench"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, scenes
d0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0054
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variable = (sys.argv[3] if len(sys.argv) > 3 else "fixed") == "variable"
api = capi.CApi()
t0 = time.perf_counter()
s, f = scenes.object_drop(api, initial_spacing=d0, variable_dt=variable, step_size=2e-5 * d0 / 0.0054, check_errors=False,
                           initial_height=0.9 if "submerged" in sys.argv[4:] else None)
c = s.GetCounters()
t_build = time.perf_counter() - t0
h = 2e-5 * d0 / 0.0054
for _ in range(20):
    f.DoStepDynamics(s.GetCurrentStepSize() if variable else h)
s.Synchronize()
t0 = time.perf_counter()
for _ in range(nsteps):
    f.DoStepDynamics(s.GetCurrentStepSize() if variable else h)
s.Synchronize()
ms = (time.perf_counter() - t0) / nsteps * 1e3
b = f.bodies[0]
print("object drop d0=%g (%s dt): %d markers (%d fluid, %d wall BCE, %d rigid BCE), build %.1f s, %.3f ms/step incl. exchange, %.3g particle-steps/s; body z=%.6f vz=%.4f Fz=%.3f"
      % (d0, "variable" if variable else "fixed", c.numAllMarkers, c.numFluidMarkers, c.numBoundaryMarkers, c.numRigidMarkers, t_build, ms, c.numAllMarkers / ms * 1e3,
         b.pos[2], b.lin_vel[2], b.force[2]))

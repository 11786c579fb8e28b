#!/usr/bin/env python3
"""Step time of dam-break variants: bench_variant.py <label> <d0_multiplier> [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
from chrono_wisc_amd import capi, fsi, scenes
label = sys.argv[1] if len(sys.argv) > 1 else "C2"
mult = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
api = capi.CApi()
sc, d0, dt = scenes.DAM_BREAK_CONFIGS[label]
def tweak(sp):
    sp.d0_multiplier = mult
    sp.shifting_method = fsi.ShiftingMethod.XSPH; sp.shifting_xsph_eps = 0.5; sp.use_delta_sph = True; sp.delta_sph_coefficient = 0.1
    sp.artificial_viscosity = 0.03; sp.eos_type = fsi.EosType.TAIT; sp.max_velocity = 10.0; sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL
s = scenes.dam_break(api, s=sc, d0=d0, dt=dt, tweak=tweak, check_errors=False)
for _ in range(5): s.DoStepDynamics(dt)
s.Synchronize()
t0 = time.perf_counter()
for _ in range(nsteps): s.DoStepDynamics(dt)
s.Synchronize()
ms = (time.perf_counter() - t0) / nsteps * 1e3
n = s.GetCounters().numAllMarkers
import ctypes as C
rows = C.c_size_t(); nnz = C.c_size_t()
s.solver._ck(api.get_neighbor_csr(s.solver.ctx, None, 0, None, 0, C.byref(rows), C.byref(nnz)))
print("%s d0_multiplier=%g: %d markers, mean row %.1f, %.3f ms/step, %.3g particle-steps/s" % (label, mult, n, nnz.value / max(rows.value, 1), ms, n / ms * 1e3))

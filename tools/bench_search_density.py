#!/usr/bin/env python3
"""Neighbour search at the reference's default d0_multiplier (1.2) on a DISORDERED marker set: a free block of fluid whose markers are
jittered off the lattice, so that a row's run of three cells holds ~41 candidates instead of the lattice's 28 ... 72.
usage: bench_search_density.py [n_per_edge=100] [multiplier=1.2] [jitter=0.3] [periodic]      (CHSPH_SEARCH_LDS_ONE_WORD=1: the row-per-lane kernel instead)"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _bootstrap; _bootstrap.load()
import numpy as np
from chrono_wisc_amd import capi, fsi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mult = float(sys.argv[2]) if len(sys.argv) > 2 else 1.2
jit = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
api = capi.CApi()
d0 = 0.01
s = fsi.ChFsiFluidSystemSPH(api)
s.EnableErrorChecks(False)
s.SetStepSize(1e-5)
fp = fsi.FluidProperties(); fp.density, fp.viscosity = 1000, 1
s.SetCfdSPH(fp)
s.SetGravitationalAcceleration((0, 0, 0))
sp = fsi.SPHParameters()
sp.integration_scheme = fsi.IntegrationScheme.RK2
sp.initial_spacing = d0
sp.d0_multiplier = mult
sp.max_velocity = 1.0
sp.shifting_method = fsi.ShiftingMethod.XSPH
sp.shifting_xsph_eps = 0.5
sp.artificial_viscosity = 0.03
sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL
sp.eos_type = fsi.EosType.TAIT
sp.use_delta_sph = True
sp.delta_sph_coefficient = 0.1
sp.num_proximity_search_steps = 1
s.SetSPHParameters(sp)
L = n * d0
if "periodic" in sys.argv[4:]:  # the block fills a fully periodic box
    s.SetComputationalDomain((-L / 2,) * 3, (L / 2,) * 3, fsi.BC_ALL_PERIODIC)
else:
    M = L / 2 + 20 * d0
    s.SetComputationalDomain((-M,) * 3, (M,) * 3, fsi.BC_NONE)  # a free block
ax = -L / 2 + d0 * (np.arange(n) + 0.5)
X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
pts = pts + np.random.default_rng(1).uniform(-jit * d0, jit * d0, pts.shape)
rpm = np.stack([np.full(len(pts), 1000.0), np.zeros(len(pts)), np.full(len(pts), 1.0)], axis=1)
s.AddSPHParticles(pts, rpm, np.zeros_like(pts))
s.Initialize()
dt = 1e-6
for _ in range(10): s.DoStepDynamics(dt)
s.Synchronize(); t0 = time.perf_counter()
for _ in range(50): s.DoStepDynamics(dt)
s.Synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
api.profile_enable(s.solver.ctx, 1)
for _ in range(20): s.DoStepDynamics(dt)
s.Synchronize()
out = []
for name in (b"sort", b"neigh", b"adami", b"force"):
    tot, cnt = C.c_double(), C.c_uint64()
    api.profile_get(s.solver.ctx, name, C.byref(tot), C.byref(cnt))
    if cnt.value: out.append("%s %.1f us" % (name.decode(), tot.value / cnt.value * 1e3))
offs, _idx = s.solver.neighbor_csr()
print("%d markers, multiplier %.2f, jitter %.2f d0, %.1f neighbours per row: %.3f ms/step | %s" % (len(pts), mult, jit, np.diff(offs).mean(), ms, "; ".join(out)))

#!/usr/bin/env python3
"""Regenerates tests/golden/tiny_rk2_5steps.npz from the CPU oracle (oracle/libsph_oracle.so).

This fixture is a SELF-GENERATED regression pin of the oracle, not an output of the reference: the reference
FSI-SPH module has no CPU path and cannot run in this pipeline.  Inputs: scenes.dam_break "tiny"
(s=0.25, d0=0.05, dt=1e-4), 5 RK2 steps.  Stored: neighbour CSR of the last proximity search and the
original-order state (pos, vel, rhoPresMu+type, derivVelRho)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _bootstrap  # noqa: E402

_bootstrap.load()
from chrono_wisc_amd import capi, scenes  # noqa: E402

api = capi.CApi(os.path.join(ROOT, "oracle", "libsph_oracle.so"), "ref_")
s = scenes.dam_break_named("tiny", api)
for _ in range(5):
    s.DoStepDynamics(1e-4)
offs, idx = s.solver.neighbor_csr()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_rk2_5steps.npz"), offs=offs, idx=idx, pos=s.solver.positions(),
                    vel=s.solver.velocities(), rpm=s.solver.rho_pres_mu_type(), deriv=s.solver.derivatives())
print("markers", s.GetCounters().numAllMarkers, "nnz", idx.size)

# CRM granular path: soil column (scenes.granular_column at reduced size), mu(I) rheology, 20 RK2 steps; and 20 steps with the
# modified Cam-Clay rheology.  Stored: state, stress, (pc, Ev, Sv).
for name, kw in (("crm_mui_20steps", dict(dt=2e-4)), ("crm_mcc_20steps", dict(dt=1e-4, mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.05, "v_lambda": 2.0, "pc": 2000.0}))):
    g = scenes.granular_column(api, radius=0.1, height=0.2, d0=0.02, box_factor=6, **kw)
    for _ in range(20):
        g.DoStepDynamics(kw["dt"])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"), pos=g.solver.positions(), vel=g.solver.velocities(),
                        rpm=g.solver.rho_pres_mu_type(), deriv=g.solver.derivatives(), stress=g.solver.stress(), pcv=g.solver.consolidation())
    print(name, "markers", g.GetCounters().numAllMarkers)

# Coupled and implicit paths: (1) the cylinder drop (one free FSI body on the Chrono-free 6-DoF driver) started partly submerged,
# 10 fixed steps: state + wrench history + body trajectory; (2) the implicit dam break ("tiny", 40 Jacobi sweeps max), 2 steps:
# state + sweep counts of the last step.
s, f = scenes.object_drop(api, initial_spacing=0.05, variable_dt=False, initial_height=0.9)
wr = []
for _ in range(10):
    f.DoStepDynamics(2e-4)
    wr.append(np.concatenate(s.GetBodyWrenches()[0]))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "object_drop_10steps.npz"), pos=s.solver.positions(), vel=s.solver.velocities(),
                    rpm=s.solver.rho_pres_mu_type(), deriv=s.solver.derivatives(), wrench=np.array(wr), body_pos=f.bodies[0].pos, body_rot=f.bodies[0].rot,
                    body_vel=f.bodies[0].lin_vel)
print("object_drop_10steps markers", s.GetCounters().numAllMarkers, "Fz", wr[-1][2])
i = scenes.dam_break_implicit(api, "tiny", max_num_iters=40)
for _ in range(2):
    i.DoStepDynamics(5e-4)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "isph_tiny_2steps.npz"), pos=i.solver.positions(), vel=i.solver.velocities(),
                    rpm=i.solver.rho_pres_mu_type(), iters=np.array(i.solver.isph_iterations()))
print("isph_tiny_2steps markers", i.GetCounters().numAllMarkers, "sweeps", i.solver.isph_iterations())

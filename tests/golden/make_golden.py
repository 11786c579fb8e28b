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

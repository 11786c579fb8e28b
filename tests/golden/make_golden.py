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

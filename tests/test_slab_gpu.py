"""GPU tests of the slab-decomposed path: several HIP contexts (ranks) on the one test GPU, exchanging halos through
the in-process transport and through torch.distributed (gloo staging, two processes).  Owned results must be
bit-identical to the single-context HIP run, which is itself compared with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from chrono_wisc_amd import scenes, slab

import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _single(api, label, nsteps, h, **kw):
    s = scenes.dam_break_named(label, api, **kw)
    for _ in range(nsteps):
        s.DoStepDynamics(h)
    return util.snapshot(s)


def _slab_threads(api, label, nranks, nsteps, h, cuts=None, **kw):
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named(label, api, initialize=False, **kw)
        used = s.InitializeSlab(r, nranks, cuts)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, used))
    slab.run_threads(ranks, nsteps, h)
    return slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global), ranks


@pytest.mark.parametrize("nranks", [2, 4])
def test_slabs_match_single_context_and_oracle(gpu, oracle, nranks):
    h = 1e-4
    one = _single(gpu, "C1", 5, h)
    got, ranks = _slab_threads(gpu, "C1", nranks, 5, h)
    ref = _single(oracle, "C1", 5, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="slab vs single context: " + k)
        util.assert_fields_close(k, got[k], ref[k])
    assert sum(r.sys.GetCounters().numAllMarkers for r in ranks) == 30327


def test_slab_migration_and_reused_ghost_sets(gpu, oracle):
    h = 2e-4
    got, _ = _slab_threads(gpu, "tiny", 2, 150, h, cuts=[0, 13, 40])
    one = _single(gpu, "tiny", 150, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)
    got2, _ = _slab_threads(gpu, "tiny", 2, 6, 1e-4, ps_freq=2)
    ref2 = _single(oracle, "tiny", 6, 1e-4, ps_freq=2)
    for k in ("pos", "vel", "rpm", "deriv"):
        util.assert_fields_close(k, got2[k], ref2[k])


def test_two_processes_gloo_staging(gpu, oracle, tmp_path):
    out = tmp_path / "slab_gpu.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29519",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "product", "--same-gpu", "--label", "C1", "--steps", "4", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=ROOT)
    got = np.load(out)
    ref = _single(oracle, "C1", 4, 1e-4)
    for k in ("pos", "vel", "rpm", "deriv"):
        util.assert_fields_close(k, got[k], ref[k])

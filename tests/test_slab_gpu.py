"""GPU tests of the slab-decomposed path: several HIP contexts (ranks) on the one test GPU, exchanging halos through
the in-process transport and through torch.distributed (gloo staging, two processes).  Owned results must be
bit-identical to the single-context HIP run, which is itself compared with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from chrono_wisc_amd import capi, scenes, slab

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


def test_one_round_preamble_keeps_the_state_in_cell_order(gpu):
    """From the second step on the ranks exchange one message per neighbour and never rebuild marker order; getters
    materialise the owned markers on request, and stepping continues from the resident state afterwards."""
    h = 2e-4
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s = scenes.dam_break_named("tiny", gpu, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, [0, 13, 40])))
    assert all(r.api.slab_one_round(r.ctx) == 0 for r in ranks)
    slab.run_threads(ranks, 30, h)
    assert all(r.api.slab_one_round(r.ctx) == 1 for r in ranks)
    mid = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)  # materialises; the run goes on from cell order
    slab.run_threads(ranks, 30, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    one = scenes.dam_break_named("tiny", gpu)
    for _ in range(30):
        one.DoStepDynamics(h)
    ref_mid = util.snapshot(one)
    for _ in range(30):
        one.DoStepDynamics(h)
    ref = util.snapshot(one)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(mid[k], ref_mid[k], err_msg="after 30 steps: " + k)
        np.testing.assert_array_equal(got[k], ref[k], err_msg="after 60 steps: " + k)


def test_leaving_and_re_entering_the_resident_mode(gpu):
    """sph_force_proximity_search makes the next step use the two-round preamble (marker order rebuilt from the cell-ordered
    state, ghosts re-imported); the step after that enters the resident mode again."""
    h = 2e-4
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s = scenes.dam_break_named("tiny", gpu, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, [0, 13, 40])))
    one = scenes.dam_break_named("tiny", gpu)
    for phase in range(3):
        slab.run_threads(ranks, 10, h)
        for _ in range(10):
            one.DoStepDynamics(h)
        for r in ranks:
            assert r.api.slab_one_round(r.ctx) == 1
            r.sys.solver._ck(r.api.force_proximity_search(r.ctx))
            assert r.api.slab_one_round(r.ctx) == 0
        one.solver._ck(gpu.force_proximity_search(one.solver.ctx))
    slab.run_threads(ranks, 5, h)
    for _ in range(5):
        one.DoStepDynamics(h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    ref = util.snapshot(one)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


@pytest.mark.parametrize("z_offset_h,vz", [(50.0, 0.0), (-0.2, 40.0)])
def test_markers_outside_the_domain_under_slabs(gpu, z_offset_h, vz):
    """A fluid marker that is outside the domain box from the start (parked when the resident mode is entered) or leaves it
    during the run (parked by the refresh pass) stays with its owner and never reaches the neighbour as a ghost."""
    h = 1e-4
    one, gx = util.slab_scene_with_stray_marker(gpu, z_offset_h, vz)
    for _ in range(6):
        one.DoStepDynamics(h)
    ref = util.snapshot(one)
    assert one.GetCounters().numExtendedParticles == one.GetCounters().numAllMarkers - 1
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s, _ = util.slab_scene_with_stray_marker(gpu, z_offset_h, vz, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, [0, 13, gx])))
    slab.run_threads(ranks, 6, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_two_round_preamble_and_tight_capacity(gpu, monkeypatch):
    """CHSPH_SLAB_TWO_ROUND keeps the two-round preamble on every step; CHSPH_TIGHT_CAPACITY removes the slack of the marker
    arrays, so that the refresh has to grow them while the cell-ordered state is the only valid copy."""
    h = 2e-4
    one = _single(gpu, "tiny", 150, h)
    monkeypatch.setenv("CHSPH_SLAB_TWO_ROUND", "1")
    got, ranks = _slab_threads(gpu, "tiny", 2, 150, h, cuts=[0, 13, 40])
    assert not ranks[0].api.slab_one_round(ranks[0].ctx)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="two rounds: " + k)
    monkeypatch.delenv("CHSPH_SLAB_TWO_ROUND")
    monkeypatch.setenv("CHSPH_TIGHT_CAPACITY", "1")
    got, ranks = _slab_threads(gpu, "tiny", 2, 150, h, cuts=[0, 13, 40])
    assert ranks[0].api.slab_one_round(ranks[0].ctx)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="tight capacity: " + k)


def test_recut_between_steps(gpu):
    """SlabRank.recut() on the device path: the resident state is materialised, the column window and the thin-layer flags are
    rebuilt for the new cuts, one two-round step moves the re-assigned columns, then the run is resident again."""
    h = 2e-4
    tr = slab.ThreadTransport(3)
    ranks = []
    for r in range(3):
        s = scenes.dam_break_named("tiny", gpu, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 3, s.InitializeSlab(r, 3, [0, 26, 33, 40])))
    cuts = []

    def work(r):
        for _ in range(4):
            for _k in range(8):
                r.step(h)
            c = r.recut()
            if r.rank == 0:
                cuts.append(c)

    import threading
    ts = [threading.Thread(target=work, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    one = _single(gpu, "tiny", 32, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)
    assert cuts[-1] != [0, 26, 33, 40]
    owned = [r.sys.GetCounters().numAllMarkers for r in ranks]
    assert max(owned) < 0.6 * sum(owned)


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


def test_isph_slabs_match_single_context_and_oracle(gpu, oracle):
    """I2SPH under slabs on the device: field halos, per-sweep residual reduction feeding the device-side loop condition."""
    kw = dict(max_num_iters=40, atol=0.02)  # early exit of the pressure loop, and still stable over these steps (beta_implicit > 0 blows this scene up)
    h = 5e-4

    def single(api):
        s = scenes.dam_break_implicit(api, "tiny", **kw)
        for _ in range(3):
            s.DoStepDynamics(h)
        return util.snapshot(s), s.solver.isph_iterations()

    def slabs(api, nranks):
        tr = slab.ThreadTransport(nranks)
        ranks = []
        for r in range(nranks):
            s = scenes.dam_break_implicit(api, "tiny", initialize=False, **kw)
            ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, None)))
        slab.run_threads(ranks, 3, h)
        return slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global), ranks

    one, its = single(gpu)
    ref, its_ref = single(oracle)
    assert its == its_ref
    for nranks in (2, 3):
        got, ranks = slabs(gpu, nranks)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], one[k], err_msg="slab vs single context: " + k)
            util.assert_fields_close(k, got[k], ref[k])
        assert all(r.sys.solver.isph_iterations() == its for r in ranks)


def test_isph_two_processes_gloo_staging(gpu, oracle, tmp_path):
    out = tmp_path / "slab_gpu_isph.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29523", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29523",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "product", "--same-gpu", "--label", "tiny", "--steps", "2", "--isph", "24",
           "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=ROOT)
    got = np.load(out)
    s = scenes.dam_break_implicit(oracle, "tiny", max_num_iters=24, atol=0.03)
    for _ in range(2):
        s.DoStepDynamics(5e-4)
    ref = util.snapshot(s)
    for k in ("pos", "vel", "rpm", "deriv"):
        util.assert_fields_close(k, got[k], ref[k])


def test_variable_time_step_under_slabs(gpu):
    """I4 under slabs on the device: k_min2 over owned rows + min over ranks = the single-context step sequence."""
    import threading

    def tweak(sp):
        sp.use_variable_time_step = True
        sp.shifting_method = 2  # XSPH
        sp.shifting_xsph_eps = 0.5
        sp.use_delta_sph = True
        sp.delta_sph_coefficient = 0.1
        sp.artificial_viscosity = 0.03
        sp.eos_type = 0  # TAIT
        sp.max_velocity = 10.0
        sp.viscosity_method = 1  # ARTIFICIAL_UNILATERAL

    one = scenes.dam_break_named("tiny", gpu, tweak=tweak)
    hs = []
    for _ in range(5):
        hs.append(one.GetCurrentStepSize())
        one.DoStepDynamics(hs[-1])
    ref = util.snapshot(one)
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s = scenes.dam_break_named("tiny", gpu, initialize=False, tweak=tweak)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, None)))
    seen = [[], []]

    def work(i):
        for _ in range(5):
            h = ranks[i].current_step()
            seen[i].append(h)
            ranks[i].step(h)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen[0] == hs and seen[1] == hs and hs[1] != hs[0]
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_rccl_transport_loopback_on_one_gpu(gpu):
    """torch.distributed "nccl" (= RCCL) with a single rank sending to itself: the device-buffer plumbing of DistTransport
    and a two-context slab run whose exchanges all travel through RCCL on the solver's stream (tests/nccl_loopback_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_loopback_worker.py")], env=env, timeout=300, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "nccl loopback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


class _FsiRank:
    def __init__(self, fsys, rank):
        self.fsys, self.rank, self.ep = fsys, rank, rank.ep
        fsys.advance_cfd = rank.step

    def step(self, h):
        self.fsys.DoStepDynamics(h)


def _object_drop(api, nsteps, h, push, d0):
    s, f = scenes.object_drop(api, initial_spacing=d0, variable_dt=False, initialize=False)
    f.bodies[0].lin_vel = np.array(push, dtype=np.float64)
    s.Initialize([b.state() for b in f.bodies])
    wr = []
    for _ in range(nsteps):
        f.DoStepDynamics(h)
        wr.append(np.concatenate(s.GetBodyWrenches()[0]))
    return util.snapshot(s), np.array(wr), f, s


@pytest.mark.parametrize("nranks,cuts", [(2, None), (3, [0, 4, 8, 16])])
def test_rigid_body_shared_between_slab_contexts(gpu, oracle, nranks, cuts):
    """FSI body cut by slab boundaries (cylinder drop): rigid markers found by global index, per-marker wrench table summed over
    the ranks as integers, every rank advancing its own copy of the body -- bit-identical to the single-context HIP run (fields,
    wrench history end point, body trajectory), which is compared with the oracle."""
    h, nsteps, push, d0 = 2e-4, 40, (-1.5, 0.0, -1.0), 0.05
    one, wr, f1, s1 = _object_drop(gpu, nsteps, h, push, d0)
    ref, wr_ref, _f, _s = _object_drop(oracle, nsteps, h, push, d0)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(one[k], ref[k], err_msg="HIP vs oracle: " + k)
    np.testing.assert_array_equal(wr, wr_ref)
    assert np.abs(wr).max() > 0
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s, f = scenes.object_drop(gpu, initial_spacing=d0, variable_dt=False, initialize=False)
        f.bodies[0].lin_vel = np.array(push, dtype=np.float64)
        used = s.InitializeSlab(r, nranks, cuts, body_states=[b.state() for b in f.bodies])
        ranks.append(_FsiRank(f, slab.SlabRank(s, tr.endpoint(r), r, nranks, used)))
    slab.run_threads(ranks, nsteps, h)
    got = slab.gather_global([r.rank.owned() for r in ranks], ranks[0].rank.sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="slab vs single context: " + k)
    rigid0 = int(ranks[0].rank.sys.n_global) - len(s1._bodies[0][0])
    owners = [int((r.rank.owned()["gid"] >= rigid0).sum()) for r in ranks]
    assert sum(owners) == len(s1._bodies[0][0]) and sum(1 for o in owners if o) >= 2
    assert all(r.rank.api.slab_one_round(r.rank.ctx) == 1 for r in ranks)  # coupled runs use the one-round preamble too
    for r in ranks:
        np.testing.assert_array_equal(np.concatenate(r.rank.sys.GetBodyWrenches()[0]), wr[-1])
        np.testing.assert_array_equal(r.fsys.bodies[0].pos, f1.bodies[0].pos)
        np.testing.assert_array_equal(r.fsys.bodies[0].rot, f1.bodies[0].rot)


def test_gloo_two_processes_rigid_body_on_one_gpu(gpu, tmp_path):
    """Cylinder drop across two OS processes sharing the GPU (gloo staging of device buffers): the wrench table travels through
    DistTransport.allreduce_sum_words, a collective re-cut runs halfway; bit-identical to the single-context run."""
    out = tmp_path / "slab_gloo_body_gpu.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29529", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29529",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "product", "--same-gpu", "--body", "--steps", "12", "--recut", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    one, wr, f1, _s1 = _object_drop(gpu, 12, 2e-4, (-1.5, 0.0, -1.0), 0.05)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)
    np.testing.assert_array_equal(got["wrench"], wr[-1])
    np.testing.assert_array_equal(got["body_pos"], f1.bodies[0].pos)


def _crm_single(api, nsteps, h, **kw):
    s = scenes.granular_column(api, dt=h, **kw)
    for _ in range(nsteps):
        s.DoStepDynamics(h)
    snap = util.snapshot(s)
    snap["stress"], snap["pc"] = s.solver.stress(), s.solver.consolidation()
    return snap


@pytest.mark.parametrize("nranks,mcc", [(2, None), (3, {"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})])
def test_crm_granular_column_shared_between_slab_contexts(gpu, oracle, nranks, mcc):
    """CRM soil column cut by slab boundaries: stress (and Cam-Clay) state in the companion record streams of migration, ghost
    import and per-stage halo; bit-identical to the single-context HIP run and to the oracle."""
    kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc=mcc)
    h, nsteps = 1e-4, 30
    one = _crm_single(gpu, nsteps, h, **kw)
    ref = _crm_single(oracle, nsteps, h, **kw)
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(one[k], ref[k], err_msg="HIP vs oracle: " + k)
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.granular_column(gpu, dt=h, initialize=False, **kw)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks)))
    slab.run_threads(ranks, nsteps, h)
    pieces = []
    for r in ranks:
        pc = r.owned()
        pc["stress"], pc["pc"] = r.sys.solver.stress(), r.sys.solver.consolidation()
        pieces.append(pc)
    got = slab.gather_global(pieces, ranks[0].sys.n_global, extra=("stress", "pc"))
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="slab vs single context: " + k)
    assert sum(1 for pc in pieces if (pc["rpm"][:, 3] < -0.5).any()) >= 2
    assert all(r.api.slab_one_round(r.ctx) == 1 for r in ranks)  # stress companions ride the one-round refresh records too


class _PressRank:
    """Slab rank of a CRM run with one body in prescribed motion (pressed into the soil while spinning): the exchange sequence of
    ChFsiSystem around SlabRank.step, same body state on every rank."""

    def __init__(self, rank, state):
        self.rank, self.ep, self.state, self.k, self.wr = rank, rank.ep, state, 0, None

    def step(self, h):
        s = self.rank.sys
        self.rank.step(h)
        s.OnExchangeSolidForces()
        self.wr = np.concatenate(s.GetBodyWrenches()[0])
        self.k += 1
        st = dict(self.state)
        st["pos"] = (0.0, 0.0, self.state["pos"][2] - 0.5 * h * self.k)
        s.LoadSolidStates([st])
        s.OnExchangeSolidStates()


def test_crm_with_a_rigid_body_shared_between_slab_contexts(gpu):
    """CRM soil + an FSI body (the wheel / cone use case) under slabs: stress companions and the wrench table together."""
    h, nsteps = 2e-4, 15
    kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0)

    def build(initialize):
        s = scenes.granular_column(gpu, dt=h, initialize=False, **kw)
        d0 = s.GetInitialSpacing()
        g = np.arange(-2, 3) * d0
        X, Y, Z = np.meshgrid(g, g, np.arange(0, 3) * d0, indexing="ij")
        local = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        top = -0.15 + 0.2 + 2 * d0
        state = dict(pos=(0.0, 0.0, top + 0.5 * d0), rot=(1.0, 0.0, 0.0, 0.0), lin_vel=(0.0, 0.0, -0.5), ang_vel=(0.0, 0.0, 2.0), lin_acc=(0.0, 0.0, -9.81),
                     ang_acc=(0.0, 0.0, 0.0))
        s.AddFsiBody(local, state)
        if initialize:
            s.Initialize()
        return s, state

    one, st0 = build(True)
    wr1 = None
    for k in range(nsteps):
        one.DoStepDynamics(h)
        one.OnExchangeSolidForces()
        wr1 = np.concatenate(one.GetBodyWrenches()[0])
        st = dict(st0)
        st["pos"] = (0.0, 0.0, st0["pos"][2] - 0.5 * h * (k + 1))
        one.LoadSolidStates([st])
        one.OnExchangeSolidStates()
    ref = util.snapshot(one)
    ref["stress"] = one.solver.stress()
    assert np.abs(wr1).max() > 0
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s, st = build(False)
        ranks.append(_PressRank(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2)), st))
    slab.run_threads(ranks, nsteps, h)
    pieces = []
    for r in ranks:
        pc = r.rank.owned()
        pc["stress"] = r.rank.sys.solver.stress()
        pieces.append(pc)
        np.testing.assert_array_equal(r.wr, wr1)
    got = slab.gather_global(pieces, ranks[0].rank.sys.n_global, extra=("stress",))
    for k in ("pos", "vel", "rpm", "deriv", "stress"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    rigid0 = ranks[0].rank.sys.n_global - 75
    assert all(int((pc["gid"] >= rigid0).sum()) > 0 for pc in pieces)  # both ranks own part of the body


def test_device_direct_thread_transport(gpu):
    """In-process transport with device-to-device message copies (tools/rehearse_c3_slabs.py ... direct): same results."""
    h = 2e-4
    tr = slab.ThreadTransport(2, device_direct=True)
    ranks = []
    for r in range(2):
        s = scenes.dam_break_named("tiny", gpu, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, [0, 13, 40])))
    slab.run_threads(ranks, 40, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    one = _single(gpu, "tiny", 40, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)


def test_crm_markers_migrate_between_slab_contexts_with_their_stress(gpu):
    """Soil markers cross the cut while the column collapses (600 steps): emigrants kept as ghosts and immigrants carry stress and
    Cam-Clay state through the one-round refresh records; bit-identical to the single-context run."""
    import test_slab_cpu as T
    kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})
    h, nsteps = 2e-4, 600
    cuts, ncross = T.crm_cut_that_soil_crosses(gpu, nsteps, h, kw)
    assert ncross > 0
    one = _crm_single(gpu, nsteps, h, **kw)
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s = scenes.granular_column(gpu, dt=h, initialize=False, **kw)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, cuts)))
    slab.run_threads(ranks, nsteps, h)
    pieces = []
    for r in ranks:
        pc = r.owned()
        pc["stress"], pc["pc"] = r.sys.solver.stress(), r.sys.solver.consolidation()
        pieces.append(pc)
    got = slab.gather_global(pieces, ranks[0].sys.n_global, extra=("stress", "pc"))
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)
    assert all(r.api.slab_one_round(r.ctx) == 1 for r in ranks)


def test_crm_slabs_tight_capacity_and_recut(gpu, monkeypatch):
    """CRM under slabs with no slack in the marker arrays (the refresh has to grow them while the cell-ordered stress columns are the
    only valid copy) and a collective re-cut halfway (materialise incl. stress, two-round migration with companions, resident again)."""
    import threading
    import test_slab_cpu as T
    kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})
    h, nsteps = 2e-4, 600
    cuts, _n = T.crm_cut_that_soil_crosses(gpu, nsteps, h, kw)
    one = _crm_single(gpu, nsteps, h, **kw)
    monkeypatch.setenv("CHSPH_TIGHT_CAPACITY", "1")
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s = scenes.granular_column(gpu, dt=h, initialize=False, **kw)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, cuts)))
    errs = []

    def work(r):
        try:
            for k in range(nsteps):
                if k == nsteps // 2:
                    r.recut()
                r.step(h)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
            tr.failed.set()
    ts = [threading.Thread(target=work, args=(r,)) for r in ranks]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    pieces = []
    for r in ranks:
        pc = r.owned()
        pc["stress"], pc["pc"] = r.sys.solver.stress(), r.sys.solver.consolidation()
        pieces.append(pc)
    got = slab.gather_global(pieces, ranks[0].sys.n_global, extra=("stress", "pc"))
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)


@pytest.mark.parametrize("slab_step", ["python", "native", "both"])
def test_bench_with_two_ranks_started_bare_on_one_gpu(slab_step):
    """The driver's command form for the scaling runs, `python bench.py --gpus N ...` with no launcher around it: bench.py starts its
    own ranks (child processes under torch.distributed.run).  Rehearsed here with two ranks sharing GPU 0, gloo staging the messages
    through the host, on the reference demo's own size; prints ONE JSON line carrying the whole-job rate.  Without --slab-step ("both")
    it runs the slab.py group and the sph_slab_step group one after the other and reports the better, with both under config.slab_step."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-gpu", "--config", "C1", "--steps", "20",
                        "--warmup", "5"] + ([] if slab_step == "both" else ["--slab-step", slab_step]), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["value"] > 0 and rec["unit"] == "particle-steps/s" and rec["scaling"] == "strong"
    assert "30327 markers" in rec["config"]["workload"]
    if slab_step == "both":
        both = rec["config"]["slab_step"]
        assert both["reported"] in ("python", "native") and both["python"]["value"] > 0 and both["native"]["value"] > 0
        assert rec["value"] == max(both["python"]["value"], both["native"]["value"])


# ---- the slab step inside the library (sph_slab_step) ---------------------------------------------------------------------
def _slab_threads_native(api, label, nranks, nsteps, h, cuts=None, device_direct=False, ordered=False, **kw):
    tr = slab.ThreadTransport(nranks, device_direct=device_direct)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named(label, api, initialize=False, **kw)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, cuts)))
        ranks[-1].native_ordered = ordered
    slab.run_threads(ranks, nsteps, h, native=True)
    assert all(r._native_kind == "callbacks" for r in ranks)
    return slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global), ranks


@pytest.mark.parametrize("nranks,direct", [(2, False), (3, True)])
def test_native_slab_step_matches_single_context(gpu, nranks, direct):
    """sph_slab_step (preamble, search, stages, halo between them -- orchestrated by the library, transport through callbacks into the
    in-process endpoints, host-staged and device-to-device): owned results bit-identical to the single-context run, like the Python
    driver's."""
    h = 1e-4
    one = _single(gpu, "C1", 6, h)
    got, ranks = _slab_threads_native(gpu, "C1", nranks, 6, h, device_direct=direct)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="native slab step vs single context: " + k)
    assert sum(r.sys.GetCounters().numAllMarkers for r in ranks) == 30327


def test_native_slab_step_migration_two_round_and_sparse_search(gpu, monkeypatch):
    """Markers crossing the cut over 150 steps (one-round resident preamble), the two-round preamble (CHSPH_SLAB_TWO_ROUND) and a
    search every second step (halo after the final stage too) through sph_slab_step."""
    h = 2e-4
    got, _ = _slab_threads_native(gpu, "tiny", 2, 150, h, cuts=[0, 13, 40])
    one = _single(gpu, "tiny", 150, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)
    got2, _ = _slab_threads_native(gpu, "tiny", 2, 6, 1e-4, ps_freq=2)
    one2 = _single(gpu, "tiny", 6, 1e-4, ps_freq=2)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got2[k], one2[k], err_msg="ps_freq 2: " + k)
    monkeypatch.setenv("CHSPH_SLAB_TWO_ROUND", "1")
    got3, _ = _slab_threads_native(gpu, "tiny", 2, 40, h, cuts=[0, 13, 40])
    one3 = _single(gpu, "tiny", 40, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got3[k], one3[k], err_msg="two-round: " + k)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_native_slab_step_overlapped_halo(gpu, arith, monkeypatch):
    """sph_slab_step's overlapped path (stream-ordered transport): the force kernel of a stage runs the waves that hold halo rows first,
    the halo of the stage's target is packed, moved and unpacked on a second stream behind an event while the remaining waves run, and
    the main stream waits for it before the next stage.  Three ranks on one GPU, callbacks that declare themselves stream-ordered;
    migration across the cuts, a search every step and every second step.  Owned results bit-identical to the single-context run (exact
    arithmetic) / to the same slab run with the overlap switched off (fast arithmetic: the row sums do not depend on the launch split)."""
    h = 2e-4
    if arith == "exact":
        one = _single(gpu, "tiny", 60, h)
        got, ranks = _slab_threads_native(gpu, "tiny", 3, 60, h, cuts=[0, 10, 16, 40], device_direct=True, ordered=True)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], one[k], err_msg="overlapped halo vs single context: " + k)
        assert all(gpu.slab_overlapped_halos(r.ctx) == 60 for r in ranks)  # one halo per step (between the two RK2 stages), all overlapped
        got2, ranks2 = _slab_threads_native(gpu, "C1", 2, 6, 1e-4, ordered=True, ps_freq=2)
        assert all(gpu.slab_overlapped_halos(r.ctx) == 12 for r in ranks2)  # after the final stage too
        one2 = _single(gpu, "C1", 6, 1e-4, ps_freq=2)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got2[k], one2[k], err_msg="overlapped halo, ps_freq 2: " + k)
    else:
        monkeypatch.setenv("CHSPH_ARITH", "1")  # sph_create reads it (conftest.py pins the exact arithmetic for the parity tests)
        got, ranks = _slab_threads_native(gpu, "C1", 3, 8, 1e-4, device_direct=True, ordered=True)
        assert all(gpu.get_arithmetic(r.ctx) == 1 for r in ranks)
        monkeypatch.setenv("CHSPH_SLAB_NO_OVERLAP", "1")
        assert all(gpu.slab_overlapped_halos(r.ctx) == 8 for r in ranks)
        ref, refranks = _slab_threads_native(gpu, "C1", 3, 8, 1e-4, device_direct=True, ordered=True)
        assert all(gpu.slab_overlapped_halos(r.ctx) == 0 for r in refranks)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="overlapped vs serial halo (fast arithmetic): " + k)


def test_builtin_rccl_transport_single_rank(gpu):
    """The library's own RCCL transport (ncclGetUniqueId -> ncclCommInitRank -> stream-ordered sph_slab_step) as far as one GPU can
    take it: a communicator of one rank, a slab that owns every column, 10 steps through sph_slab_step without a single host wait
    around the (empty) exchanges -- bit-identical to the plain single-context run.  ncclSend / ncclRecv between two ranks need two
    GPUs; that leg has never run (NOTEBOOK.md 4)."""
    import ctypes as C
    h = 1e-4
    s = scenes.dam_break_named("C1", gpu, initialize=False)
    cuts = s.InitializeSlab(0, 1)
    cfg = capi.SphSlabConfig(0, 1, cuts[0], cuts[1], 2)
    s.solver._ck(gpu.slab_configure(s.solver.ctx, C.byref(cfg)))
    buf = (C.c_uint8 * 256)()
    n = gpu.slab_rccl_unique_id(buf, 256)
    assert n == 128
    t = gpu.slab_transport_rccl(s.solver.ctx, buf, n, 0, 1)
    assert t and t.contents.stream_ordered == 1 and t.contents.nranks == 1
    try:
        tm = 0.0
        for _ in range(10):
            s.solver._ck(gpu.slab_step(s.solver.ctx, tm, h, t))
            tm += h
        sv = s.solver
        sv.synchronize()
        got = slab.gather_global([{"gid": sv.global_ids(), "pos": sv.positions(), "vel": sv.velocities(), "rpm": sv.rho_pres_mu_type(), "deriv": sv.derivatives()}], s.n_global)
    finally:
        gpu.slab_transport_free(t)
    one = _single(gpu, "C1", 10, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg=k)


# ---- round 4: I2SPH inside sph_slab_step, the in-process stream-ordered transport, native collectives -------------------------
def _inproc_ranks(api, make, nranks, cuts=None, body_states=None):
    """One slab context per rank, members of one sph_slab_group (threads of this process, sph_slab_transport_inproc)."""
    group = slab.InprocGroup(api, nranks)
    ranks, extra = [], []
    for r in range(nranks):
        made = make()
        s = made[0] if isinstance(made, tuple) else made
        kw = {} if body_states is None else {"body_states": body_states(made)}
        ranks.append(slab.SlabRank(s, group.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, cuts, **kw)))
        extra.append(made)
    return group, ranks, extra


def _close(group, ranks):
    for r in ranks:
        r.close()
    assert group.api.slab_group_destroy(group.handle) == 0
    group.handle = None


@pytest.mark.parametrize("transport", ["callbacks", "callbacks_direct", "inproc"])
def test_native_isph_slab_step(gpu, oracle, transport):
    """I2SPH through sph_slab_step: the seven phases, a ghost exchange of the iterate and a max-reduction of the residual record after
    every Jacobi sweep (loops of phys/SphForceISPH.cu:1026-1042 / 1135-1164), all orchestrated by the library.  2 and 3 ranks on one
    GPU over Python callbacks (host-staged / device-to-device) and over the in-process stream-ordered transport: owned fields and both
    sweep counts bit-identical to the single-context run, which equals the oracle's."""
    kw = dict(max_num_iters=40, atol=0.02)
    h = 5e-4

    def single(api):
        s = scenes.dam_break_implicit(api, "tiny", **kw)
        for _ in range(3):
            s.DoStepDynamics(h)
        return util.snapshot(s), s.solver.isph_iterations()

    one, its = single(gpu)
    ref, its_ref = single(oracle)
    assert its == its_ref and its[1] < 40  # the pressure loop leaves early: the reduced residual decides it on every rank
    for nranks in (2, 3):
        make = lambda: scenes.dam_break_implicit(gpu, "tiny", initialize=False, **kw)  # noqa: E731
        if transport == "inproc":
            group, ranks, _ = _inproc_ranks(gpu, make, nranks)
        else:
            tr = slab.ThreadTransport(nranks, device_direct=(transport == "callbacks_direct"))
            ranks = []
            for r in range(nranks):
                s = make()
                ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, None)))
        slab.run_threads(ranks, 3, h, native=True)
        assert all(r._native_kind == ("inproc" if transport == "inproc" else "callbacks") for r in ranks)
        got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], one[k], err_msg="native I2SPH slab step (%s, %d ranks) vs single context: %s" % (transport, nranks, k))
            util.assert_fields_close(k, got[k], ref[k])
        assert all(r.sys.solver.isph_iterations() == its for r in ranks)
        if transport == "inproc":
            _close(group, ranks)


@pytest.mark.parametrize("arith", ["exact", "fast"])
def test_inproc_transport_overlapped_halo_without_host_synchronisation(gpu, arith, monkeypatch):
    """The overlapped halo path of sph_slab_step over a transport that really is stream-ordered (sph_slab_transport_inproc: the receiver
    copies on the stream it is handed behind the sender's event; no hipStreamSynchronize anywhere in the exchange): pack / copy /
    unpack on the context's second stream against the force kernel's non-edge waves on the first, three ranks on one GPU, markers
    migrating across both cuts over 60 steps, a search every step and every second step.  Bit-identical to the single-context run
    (exact arithmetic) / to the serial-halo slab run (fast arithmetic)."""
    h = 2e-4
    make = lambda label, **kw: (lambda: scenes.dam_break_named(label, gpu, initialize=False, **kw))  # noqa: E731
    if arith == "exact":
        one = _single(gpu, "tiny", 60, h)
        group, ranks, _ = _inproc_ranks(gpu, make("tiny"), 3, cuts=[0, 10, 16, 40])
        slab.run_threads(ranks, 60, h, native=True)
        got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], one[k], err_msg="in-process transport, overlapped halo vs single context: " + k)
        assert all(gpu.slab_overlapped_halos(r.ctx) == 60 for r in ranks)
        _close(group, ranks)
        one2 = _single(gpu, "C1", 6, 1e-4, ps_freq=2)
        group, ranks, _ = _inproc_ranks(gpu, make("C1", ps_freq=2), 2)
        slab.run_threads(ranks, 6, 1e-4, native=True)
        got2 = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
        assert all(gpu.slab_overlapped_halos(r.ctx) == 12 for r in ranks)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got2[k], one2[k], err_msg="in-process transport, ps_freq 2: " + k)
        _close(group, ranks)
    else:
        monkeypatch.setenv("CHSPH_ARITH", "1")
        group, ranks, _ = _inproc_ranks(gpu, make("C1"), 3)
        slab.run_threads(ranks, 8, 1e-4, native=True)
        assert all(gpu.get_arithmetic(r.ctx) == 1 and gpu.slab_overlapped_halos(r.ctx) == 8 for r in ranks)
        got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
        _close(group, ranks)
        monkeypatch.setenv("CHSPH_SLAB_NO_OVERLAP", "1")
        group, ranks, _ = _inproc_ranks(gpu, make("C1"), 3)
        slab.run_threads(ranks, 8, 1e-4, native=True)
        assert all(gpu.slab_overlapped_halos(r.ctx) == 0 for r in ranks)
        ref = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
        _close(group, ranks)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="overlapped vs serial halo over the in-process transport (fast arithmetic): " + k)


def test_native_collectives_rigid_body_and_variable_step(gpu):
    """sph_slab_body_wrenches (per-marker wrench table summed over the ranks as integers through transport->allreduce, then the
    single-device block reduction) and sph_slab_current_step (minimum of the two step candidates over the ranks, reduced on the
    device): the cylinder drop cut through the body with the variable step on, 3 ranks over the in-process transport, every rank
    advancing its own copy of the body -- step sizes, fields, wrench and body trajectory bit-identical to the single-context run."""
    import threading
    nsteps, push, d0 = 25, (-1.5, 0.0, -1.0), 0.05

    def make():
        s, f = scenes.object_drop(gpu, initial_spacing=d0, variable_dt=True, initialize=False)
        f.bodies[0].lin_vel = np.array(push, dtype=np.float64)
        return s, f

    s1, f1 = make()
    s1.Initialize([b.state() for b in f1.bodies])
    hs = []
    for _ in range(nsteps):
        hs.append(s1.GetCurrentStepSize())
        f1.DoStepDynamics(hs[-1])
    one, wr1 = util.snapshot(s1), np.concatenate(s1.GetBodyWrenches()[0])
    assert len(set(hs)) > 3 and np.abs(wr1).max() > 0
    group, ranks, made = _inproc_ranks(gpu, make, 3, cuts=[0, 4, 8, 16], body_states=lambda m: [b.state() for b in m[1].bodies])
    seen = [[] for _ in ranks]
    errs = []

    def work(i):
        try:
            rk, f = ranks[i], made[i][1]
            rk.native_collectives = True
            f.advance_cfd = rk.step_native
            for _ in range(nsteps):
                hstep = rk.current_step_native()
                seen[i].append(hstep)
                f.DoStepDynamics(hstep)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(sn == hs for sn in seen)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="native collectives vs single context: " + k)
    for i, r in enumerate(ranks):
        np.testing.assert_array_equal(np.concatenate(r.sys.GetBodyWrenches()[0]), wr1)
        np.testing.assert_array_equal(made[i][1].bodies[0].pos, f1.bodies[0].pos)
        np.testing.assert_array_equal(made[i][1].bodies[0].rot, f1.bodies[0].rot)
    _close(group, ranks)


def test_transports_do_not_outlive_their_context(gpu):
    """A library-owned transport holds its context.  sph_destroy releases the transports still attached to the context it destroys, and
    sph_slab_transport_free of a pointer that is no longer live is a no-op (it was a use-after-free); a group can only go once its
    members have."""
    import ctypes as C
    g = gpu.slab_group_create(2)
    assert g and not gpu.slab_group_create(17)
    ctxs, ts = [], []
    for r in range(2):
        s = scenes.dam_break_named("tiny", gpu, initialize=False)
        cuts = s.InitializeSlab(r, 2)
        cfg = capi.SphSlabConfig(r, 2, cuts[r], cuts[r + 1], 2)
        s.solver._ck(gpu.slab_configure(s.solver.ctx, C.byref(cfg)))
        t = gpu.slab_transport_inproc(s.solver.ctx, g, r)
        assert t and t.contents.stream_ordered == 1 and bool(t.contents.allreduce)
        ctxs.append(s)
        ts.append(t)
    assert gpu.slab_group_destroy(g) == 1          # two members alive
    gpu.slab_transport_free(ts[0])
    gpu.slab_transport_free(ts[0])                 # second free: no-op
    ctxs[1].solver.close()                         # sph_destroy releases rank 1's transport
    gpu.slab_transport_free(ts[1])                 # stale pointer: no-op
    assert gpu.slab_group_destroy(g) == 0
    ctxs[0].solver.close()


def test_several_fluid_steps_between_two_exchanges_under_slabs(gpu):
    """The same under slabs (one-round resident preamble, rigid rows found by global id): three slab steps per exchange, body shared by three
    ranks over the in-process transport -- bit-identical to the single-context run (tests/test_parity_gpu.py holds that one against the oracle)."""
    import threading
    h, sub, nex, push, d0 = 2e-4, 3, 8, (-1.5, 0.0, -1.0), 0.05

    def make():
        s, f = scenes.object_drop(gpu, initial_spacing=d0, variable_dt=False, initialize=False)
        f.bodies[0].lin_vel = np.array(push, dtype=np.float64)
        return s, f

    s1, f1 = make()
    s1.Initialize([b.state() for b in f1.bodies])
    f1.advance_cfd = lambda step: [s1.DoStepDynamics(step / sub) for _ in range(sub)]
    for _ in range(nex):
        f1.DoStepDynamics(h)
    one, wr1 = util.snapshot(s1), np.concatenate(s1.GetBodyWrenches()[0])
    group, ranks, made = _inproc_ranks(gpu, make, 3, cuts=[0, 4, 8, 16], body_states=lambda m: [b.state() for b in m[1].bodies])
    errs = []

    def work(i):
        try:
            rk, f = ranks[i], made[i][1]
            rk.native_collectives = True
            f.advance_cfd = lambda step: [rk.step_native(step / sub) for _ in range(sub)]
            for _ in range(nex):
                f.DoStepDynamics(h)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="slabs, several steps per exchange: " + k)
    for i, r in enumerate(ranks):
        np.testing.assert_array_equal(np.concatenate(r.sys.GetBodyWrenches()[0]), wr1)
        np.testing.assert_array_equal(made[i][1].bodies[0].pos, f1.bodies[0].pos)
    _close(group, ranks)


@pytest.mark.parametrize("driver", ["python", "native"])
def test_isph_clamp_pressure_under_slabs(gpu, oracle, driver):
    """ClampPressure under slabs on the device (phase CLAMP + one float MIN over the ranks, through slab.py and through sph_slab_step over the
    in-process transport): owned results bit-identical to the single-context run, which equals the oracle's."""
    h, kw = 5e-4, dict(max_num_iters=20)

    def make(api, initialize):
        s = scenes.dam_break_implicit(api, "tiny", initialize=False, **kw)
        s.p.ClampPressure = 1
        s.p.Conservative_Form = 1
        if initialize:
            s.Initialize()
        return s

    def single(api):
        s = make(api, True)
        for _ in range(3):
            s.DoStepDynamics(h)
        return util.snapshot(s)

    one, ref = single(gpu), single(oracle)
    fluid = one["rpm"][:, 3] == -1
    assert one["rpm"][fluid, 1].min() == 0.0
    for k in ("pos", "vel", "rpm", "deriv"):
        util.assert_fields_close(k, one[k], ref[k])
    if driver == "native":
        group, ranks, _ = _inproc_ranks(gpu, lambda: make(gpu, False), 3)
        slab.run_threads(ranks, 3, h, native=True)
    else:
        tr = slab.ThreadTransport(3)
        ranks = []
        for r in range(3):
            s = make(gpu, False)
            ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 3, s.InitializeSlab(r, 3, None)))
        slab.run_threads(ranks, 3, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="ClampPressure under slabs (%s driver): %s" % (driver, k))
    if driver == "native":
        _close(group, ranks)


@pytest.mark.parametrize("driver", ["python", "native"])
def test_isph_with_fsi_bodies_under_slabs(gpu, oracle, driver):
    """I2SPH with FSI bodies on several ranks (the Couette demo: fixed inner cylinder, spinning outer cylinder, cut along a non-periodic x): BCE_Vel_Acc
    (phys/SphForceISPH.cu:50-165) finds a rigid marker's table entry by its global index, the FSI force of Velocity_Correction_and_update feeds the
    per-marker wrench table, prescribed body states are loaded on every rank.  Fields, sweep counts and wrenches bit-identical to the single-context
    run, which equals the oracle's (tests/test_isph_gpu.py for the periodic original)."""
    import threading
    from chrono_wisc_amd import fsi as F
    kw = dict(scale=0.4, max_num_iters=30, bc=F.BC_NONE)
    nsteps = 3

    def single(api):
        s = scenes.couette(api, **kw)
        for _ in range(nsteps):
            scenes.couette_step(s)
        return util.snapshot(s), s.solver.isph_iterations(), [np.concatenate(w) for w in s.GetBodyWrenches()]

    one, its, wr = single(gpu)
    ref, its_ref, wr_ref = single(oracle)
    assert its == its_ref and np.abs(np.array(wr)).max() > 0
    for k in ("pos", "vel", "rpm", "deriv"):
        util.assert_fields_close(k, one[k], ref[k])
    nranks = 3
    states0 = scenes.couette_body_states(0.0)
    make = lambda: scenes.couette(gpu, initialize=False, **kw)  # noqa: E731
    if driver == "native":
        group, ranks, _ = _inproc_ranks(gpu, make, nranks, body_states=lambda m: states0)
    else:
        tr = slab.ThreadTransport(nranks)
        ranks = []
        for r in range(nranks):
            s = make()
            ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, None, body_states=states0)))
    errs = []

    def work(rk):
        try:
            rk.native_collectives = driver == "native"
            for _ in range(nsteps):
                (rk.step_native if driver == "native" else rk.step)(scenes.COUETTE["step_size"])
                rk.sys.OnExchangeSolidForces()
                rk.sys.LoadSolidStates(scenes.couette_body_states(rk.sys.GetSimTime()))
                rk.sys.OnExchangeSolidStates()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(rk,)) for rk in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="I2SPH + FSI bodies under slabs (%s driver): %s" % (driver, k))
    for r in ranks:
        assert r.sys.solver.isph_iterations() == its
        for a, b in zip([np.concatenate(w) for w in r.sys.GetBodyWrenches()], wr):
            np.testing.assert_array_equal(a, b)
    rigid0 = int(ranks[0].sys.n_global) - int((one["rpm"][:, 3] > 0.5).sum())  # rigid markers carry the last global indices
    owners = [int((r.owned()["gid"] >= rigid0).sum()) for r in ranks]
    assert sum(1 for o in owners if o) >= 2  # the bodies' markers really are spread over several ranks
    if driver == "native":
        _close(group, ranks)


# ---- slabs along a periodic x axis: the ranks form a ring -----------------------------------------------------------------
@pytest.mark.parametrize("driver,nranks", [("python", 3), ("native", 3), ("native", 4)])
def test_poiseuille_in_a_ring_of_slabs(gpu, driver, nranks):
    """The reference's unit-test scene (utest_FSI_Poiseuille_flow.cpp; periodic in x and y) cut into x-slabs: the first and the last rank are
    neighbours across the periodic seam.  A rank hashes consecutive columns of the unrolled axis (the first rank sees the last grid columns as
    -2, -1), positions keep their coordinates, so the pair arithmetic with its half-box wrap (Distance, phys/SphGeneral.cuh) is the single
    device's.  Markers leave through x_max on the last rank and enter the first (ApplyPeriodicBoundaryX_D, phys/SphFluidDynamics.cu:843-870).
    Python-driven over host-staged messages and through sph_slab_step over the in-process stream-ordered transport: every owned marker
    bit-identical to the single-context run (tests/test_parity_gpu.py holds that one against the oracle; tests/test_slab_cpu.py runs the
    same ring on the oracle)."""
    nsteps, bx, dt = 25, 0.4, scenes.POISEUILLE["dt"]
    s1 = scenes.poiseuille(gpu, bxDim=bx, t_start=4.0)
    for _ in range(nsteps):
        s1.DoStepDynamics(dt)
    one = util.snapshot(s1)
    make = lambda: scenes.poiseuille(gpu, bxDim=bx, t_start=4.0, initialize=False)  # noqa: E731
    group = None
    if driver == "python":
        tr = slab.ThreadTransport(nranks)
        ranks = []
        for r in range(nranks):
            s = make()
            ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, None)))
        slab.run_threads(ranks, nsteps, dt)
    else:
        group, ranks, _ = _inproc_ranks(gpu, make, nranks)
        slab.run_threads(ranks, nsteps, dt, native=True)
    assert all(r.ring for r in ranks)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], one[k], err_msg="ring of %d slabs (%s) vs single context: %s" % (nranks, driver, k))
    x_init = np.concatenate([xyz[:, 0] for (_tc, xyz, _v, _r, _h) in make()._pending])
    assert np.count_nonzero(x_init[ranks[0].owned()["gid"]] > 0.1) > 0  # markers of the last rank came round the seam
    if group is not None:
        _close(group, ranks)


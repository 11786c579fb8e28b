"""CPU tests of the N > 1 path: the slab orchestration (chrono-wisc_amd/slab.py) driving CPU ranks backed by the
oracle, in-process (threads) and across processes (torch.distributed, gloo, world_size 2).  Owned results must be
bit-identical to the single-rank run."""
import os
import subprocess
import sys

import numpy as np
import pytest

from chrono_wisc_amd import fsi, scenes, slab

import util

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


@pytest.mark.parametrize("nranks", [2, 3])
def test_threads_slabs_are_bitwise_single_rank(oracle, nranks):
    h = 1e-4
    ref = _single(oracle, "tiny", 6, h)
    got, ranks = _slab_threads(oracle, "tiny", nranks, 6, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    owned = [r.sys.GetCounters().numAllMarkers for r in ranks]
    assert sum(owned) == ranks[0].sys.n_global and min(owned) > 0.5 * max(owned)  # balanced cuts


def test_migration_across_a_cut(oracle):
    """Put the cut on a cell boundary that markers of the collapsing column cross during the run."""
    h = 2e-4
    nsteps = 150
    ref = _single(oracle, "tiny", nsteps, h)
    s0 = scenes.dam_break_named("tiny", oracle, initialize=False)
    dp = s0.DerivedParams()
    gx = int(dp.gridSize[0])
    # markers of the lattice plane x = -0.7 sit on the boundary between cell columns 12 and 13 and drift across it
    cut = 13
    got, ranks = _slab_threads(oracle, "tiny", 2, nsteps, h, cuts=[0, cut, gx])
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    moved = ranks[1].sys.GetCounters().numAllMarkers
    s1 = scenes.dam_break_named("tiny", oracle, initialize=False)
    s1.InitializeSlab(1, 2, [0, cut, gx])
    assert moved != s1.GetCounters().numAllMarkers  # markers did migrate between the ranks


def test_proximity_search_every_second_step_with_slabs(oracle):
    h = 1e-4
    ref = _single(oracle, "tiny", 5, h, ps_freq=2)
    got, _ = _slab_threads(oracle, "tiny", 2, 5, h, ps_freq=2)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def _isph_single(api, nsteps, h, **kw):
    s = scenes.dam_break_implicit(api, "tiny", **kw)
    its = []
    for _ in range(nsteps):
        s.DoStepDynamics(h)
        its.append(s.solver.isph_iterations())
    return util.snapshot(s), its


def _isph_slabs(api, nranks, nsteps, h, **kw):
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_implicit(api, "tiny", initialize=False, **kw)
        used = s.InitializeSlab(r, nranks, None)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, used))
    slab.run_threads(ranks, nsteps, h)
    return slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global), ranks


@pytest.mark.parametrize("nranks,kw", [(2, dict(max_num_iters=20)), (3, dict(max_num_iters=40, atol=0.02))])
def test_isph_slabs_are_bitwise_single_rank(oracle, nranks, kw):
    """I2SPH under slabs: ghost exchange of density/volume, of every Jacobi iterate, of the corrected and the shifted state,
    and a max-reduction of the residual record per sweep; owned results and sweep counts equal the single-rank run."""
    h = 5e-4  # both parameter sets keep the scene stable over these steps (|v| < 1.5 m/s): a marker must not cross a whole slab per step
    ref, its = _isph_single(oracle, 3, h, **kw)
    assert np.abs(ref["vel"]).max() < 5.0
    got, ranks = _isph_slabs(oracle, nranks, 3, h, **kw)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    for r in ranks:
        assert r.sys.solver.isph_iterations() == its[-1]


def test_isph_clamp_pressure_under_slabs(oracle):
    """ClampPressure (phys/SphForceISPH.cu:1223: every pressure relative to the smallest fluid pressure) under slabs: each rank finds the minimum of its own
    markers (phase CLAMP), the minimum over the ranks is taken (one float), FINISH subtracts it -- owned results equal the single-rank run, and the
    clamp really acts (the smallest fluid pressure is exactly zero afterwards)."""
    h, kw = 5e-4, dict(max_num_iters=20)

    def clamp(s):
        s.p.ClampPressure = 1
        s.p.Conservative_Form = 1
        return s

    s1 = clamp(scenes.dam_break_implicit(oracle, "tiny", initialize=False, **kw))
    s1.Initialize()
    for _ in range(3):
        s1.DoStepDynamics(h)
    ref = util.snapshot(s1)
    fluid = ref["rpm"][:, 3] == -1
    assert ref["rpm"][fluid, 1].min() == 0.0 and ref["rpm"][fluid, 1].max() > 0
    for nranks in (2, 3):
        tr = slab.ThreadTransport(nranks)
        ranks = []
        for r in range(nranks):
            s = clamp(scenes.dam_break_implicit(oracle, "tiny", initialize=False, **kw))
            ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, None)))
        slab.run_threads(ranks, 3, h)
        got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
        for k in ("pos", "vel", "rpm", "deriv"):
            np.testing.assert_array_equal(got[k], ref[k], err_msg="%d ranks: %s" % (nranks, k))


def test_marker_outside_the_domain_is_not_exported_as_ghost(oracle):
    """A fluid marker above the domain top (switched off by UpdateActivity, never sorted) sitting in a halo column must
    not reach the neighbour rank: there it would be hashed as an active marker."""
    def scene(initialize):
        s = scenes.dam_break_named("tiny", oracle, initialize=False)
        dp = s.DerivedParams()
        x_cut = float(dp.worldOrigin[0]) + 13 * float(dp.cellSize[0]) - 0.01  # just left of the cut used below
        z_out = float(dp.worldOrigin[2]) + float(dp.boxDims[2]) + 50 * float(dp.h)  # beyond the 40 h tolerance of the hash
        s._pending.insert(1, (s._pending[0][0], np.array([[x_cut, 0.0, z_out]], dtype=np.float32), None,
                              np.array([[1000.0, 0.0, 5.0]], dtype=np.float32), s._pending[0][4]))
        if initialize:
            s.Initialize()
        return s, int(dp.gridSize[0])

    h = 1e-4
    one, gx = scene(True)
    for _ in range(3):
        one.DoStepDynamics(h)
    ref = util.snapshot(one)
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s, _ = scene(False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, [0, 13, gx])))
    slab.run_threads(ranks, 3, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_marker_leaving_the_domain_during_a_slab_run(oracle):
    """One-round preamble: a fluid marker in an exported column crosses the domain top in the second step; its owner keeps it
    (switched off for good), the neighbour's ghost copy disappears."""
    h = 1e-4
    one, gx = util.slab_scene_with_stray_marker(oracle, -0.2, vz=40.0)
    for _ in range(6):
        one.DoStepDynamics(h)
    ref = util.snapshot(one)
    assert one.GetCounters().numExtendedParticles == one.GetCounters().numAllMarkers - 1
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s, _ = util.slab_scene_with_stray_marker(oracle, -0.2, vz=40.0, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, [0, 13, gx])))
    slab.run_threads(ranks, 6, h)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_two_round_preamble_still_matches(oracle, monkeypatch):
    """CHSPH_SLAB_TWO_ROUND forces the two-round preamble (classify / migrate / ghosts) on every step."""
    monkeypatch.setenv("CHSPH_SLAB_TWO_ROUND", "1")
    h = 2e-4
    ref = _single(oracle, "tiny", 40, h)
    got, ranks = _slab_threads(oracle, "tiny", 2, 40, h, cuts=[0, 13, 40])
    assert not ranks[0].api.slab_one_round(ranks[0].ctx)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_one_round_preamble_is_used_from_the_second_step(oracle):
    got, ranks = _slab_threads(oracle, "tiny", 2, 2, 1e-4)
    assert all(r.api.slab_one_round(r.ctx) == 1 for r in ranks)
    fresh, ranks0 = _slab_threads(oracle, "tiny", 2, 0, 1e-4)
    assert all(r.api.slab_one_round(r.ctx) == 0 for r in ranks0)  # no search yet: the first step runs two rounds
    _g, ranks2 = _slab_threads(oracle, "tiny", 2, 2, 1e-4, ps_freq=2)
    assert all(r.api.slab_one_round(r.ctx) == 0 for r in ranks2)


def test_marker_crossing_a_whole_slab_fails_loudly(oracle):
    """Implicit shifting blows this scene up in its second step (markers fly from column 5 to column 39 at km/s): a slab rank
    can only hand a marker to its neighbour, so the run must stop instead of returning fields that differ from one rank's."""
    kw = dict(max_num_iters=40, atol=0.03, beta_implicit=0.05)
    tr = slab.ThreadTransport(3)
    ranks = []
    for r in range(3):
        s = scenes.dam_break_implicit(oracle, "tiny", initialize=False, **kw)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 3, s.InitializeSlab(r, 3, [0, 8, 16, 40])))
    with pytest.raises(Exception, match="crossed a whole slab|peer rank failed"):
        slab.run_threads(ranks, 3, 5e-4)


def _recut_run(api, nranks, cuts, h, blocks):
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named("tiny", api, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, cuts)))
    history = [list(cuts)]
    import threading

    def work(r):
        for nsteps in blocks:
            for _ in range(nsteps):
                r.step(h)
            new = r.recut()
            if r.rank == 0:
                history.append(new)

    ts = [threading.Thread(target=work, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global), ranks, history


def test_recut_moves_the_slab_boundaries_between_steps(oracle):
    """SlabRank.recut(): badly placed cuts are moved towards the balanced ones (each cut stays inside its two old neighbour
    slabs per call), the markers of the re-assigned columns migrate in the next step, results stay those of one rank."""
    h = 2e-4
    got, ranks, history = _recut_run(oracle, 3, [0, 26, 33, 40], h, [6, 6, 6, 6])
    ref = _single(oracle, "tiny", 24, h)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert history[-1] != history[0] and all(b - a >= 4 for a, b in zip(history[-1][:-1], history[-1][1:]))
    owned = [r.sys.GetCounters().numAllMarkers for r in ranks]
    assert max(owned) < 0.6 * sum(owned)  # started with ~90 % on rank 0
    for old, new in zip(history[:-1], history[1:]):
        assert all(old[i - 1] <= new[i] <= old[i + 1] for i in range(1, 3))


def test_variable_time_step_under_slabs(oracle):
    """I4 under slabs: per-rank minima over owned rows + a min over the ranks give the single-rank step sequence."""
    def tweak(sp):
        sp.use_variable_time_step = True
        sp.shifting_method = fsi.ShiftingMethod.XSPH
        sp.shifting_xsph_eps = 0.5
        sp.use_delta_sph = True
        sp.delta_sph_coefficient = 0.1
        sp.artificial_viscosity = 0.03
        sp.eos_type = fsi.EosType.TAIT
        sp.max_velocity = 10.0
        sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL

    one = scenes.dam_break_named("tiny", oracle, tweak=tweak)
    hs = []
    for _ in range(5):
        hs.append(one.GetCurrentStepSize())
        one.DoStepDynamics(hs[-1])
    ref = util.snapshot(one)
    tr = slab.ThreadTransport(2)
    ranks = []
    for r in range(2):
        s = scenes.dam_break_named("tiny", oracle, initialize=False, tweak=tweak)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, 2, s.InitializeSlab(r, 2, None)))
    seen = [[], []]

    def work(i):
        for _ in range(5):
            h = ranks[i].current_step()
            seen[i].append(h)
            ranks[i].step(h)

    import threading
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert seen[0] == hs and seen[1] == hs and hs[1] != hs[0]
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_balanced_cuts_respect_minimum_width():
    cx = np.concatenate([np.full(1000, 3), np.full(10, 30)])
    cuts = slab.balanced_cuts(cx, 40, 4, min_width=4)
    assert cuts[0] == 0 and cuts[-1] == 40 and all(b - a >= 4 for a, b in zip(cuts[:-1], cuts[1:]))
    with pytest.raises(Exception):
        slab.balanced_cuts(cx, 10, 4, min_width=4)


def test_gloo_world_size_2(oracle, tmp_path):
    """Two OS processes, torch.distributed gloo backend, 127.0.0.1 rendezvous."""
    out = tmp_path / "slab_gloo.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "oracle", "--label", "tiny", "--steps", "4", "--recut", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    ref = _single(oracle, "tiny", 4, 1e-4)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


def test_gloo_world_size_2_isph(oracle, tmp_path):
    """I2SPH across two OS processes (gloo): per-sweep ghost exchange + all-reduce(max) of the residual record."""
    out = tmp_path / "slab_gloo_isph.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29521", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29521",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "oracle", "--label", "tiny", "--steps", "2", "--isph", "24", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=ROOT)
    got = np.load(out)
    ref, _its = _isph_single(oracle, 2, 5e-4, max_num_iters=24, atol=0.03)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


@pytest.mark.parametrize("world,mailbox", [(3, True), (2, False)])
def test_host_scalar_exchange_between_processes(world, mailbox, tmp_path):
    """Record counts and the variable time step between ranks of one node: shared-memory mailbox (default) or the collective
    library (CHSPH_NO_HOST_MAILBOX), 3000 rounds with both neighbours."""
    port = "29551" if mailbox else "29552"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    if not mailbox:
        env["CHSPH_NO_HOST_MAILBOX"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "tests", "mailbox_worker.py")]
    r = subprocess.run(cmd, env=env, timeout=600, cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "mailbox ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


class _FsiRank:
    """One slab rank of a coupled run for slab.run_threads: step = the exchange loop of rigid.FsiSystem over SlabRank.step."""

    def __init__(self, fsys, rank):
        self.fsys, self.rank, self.ep = fsys, rank, rank.ep
        fsys.advance_cfd = rank.step

    def step(self, h):
        self.fsys.DoStepDynamics(h)


def _object_drop_slabs(api, nranks, nsteps, h, push, cuts=None, **kw):
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s, f = scenes.object_drop(api, initialize=False, variable_dt=False, **kw)
        f.bodies[0].lin_vel = np.array(push, dtype=np.float64)
        used = s.InitializeSlab(r, nranks, cuts, body_states=[b.state() for b in f.bodies])
        ranks.append(_FsiRank(f, slab.SlabRank(s, tr.endpoint(r), r, nranks, used)))
    slab.run_threads(ranks, nsteps, h)
    return slab.gather_global([r.rank.owned() for r in ranks], ranks[0].rank.sys.n_global), ranks


@pytest.mark.parametrize("nranks", [2, 3])
def test_rigid_body_under_slabs_is_bitwise_single_rank(oracle, nranks):
    """FSI body cut by the slab boundaries (cylinder drop): the per-marker (force, torque) table is summed over the
    ranks and reduced like on a single device, every rank advances its own copy of the body; rigid markers migrate with it."""
    h, nsteps, push = 2e-4, 40, (-1.5, 0.0, -1.0)  # the marker plane x = 0 of the cylinder drifts across the 2-rank cut
    s, f = scenes.object_drop(oracle, initial_spacing=0.05, variable_dt=False, initialize=False)
    f.bodies[0].lin_vel = np.array(push, dtype=np.float64)
    s.Initialize([b.state() for b in f.bodies])
    wr = []
    for _ in range(nsteps):
        f.DoStepDynamics(h)
        wr.append(np.concatenate(s.GetBodyWrenches()[0]))
    ref = util.snapshot(s)
    assert np.abs(np.array(wr)).max() > 0  # the body touches the fluid during the run
    got, ranks = _object_drop_slabs(oracle, nranks, nsteps, h, push, cuts=None if nranks == 2 else [0, 4, 8, 16], initial_spacing=0.05)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert all(r.rank.api.slab_one_round(r.rank.ctx) == 1 for r in ranks)  # coupled runs use the one-round preamble too
    for r in ranks:
        np.testing.assert_array_equal(np.concatenate(r.rank.sys.GetBodyWrenches()[0]), wr[-1])
        np.testing.assert_array_equal(r.fsys.bodies[0].pos, f.bodies[0].pos)
    rigid0 = int(ranks[0].rank.sys.n_global) - len(s._bodies[0][0])
    owners = [int((r.rank.owned()["gid"] >= rigid0).sum()) for r in ranks]
    assert sum(owners) == len(s._bodies[0][0]) and sum(1 for o in owners if o) >= 2  # the body really is shared between ranks
    if nranks == 2:  # and rigid markers changed owner during the run
        s0, f0 = scenes.object_drop(oracle, initial_spacing=0.05, variable_dt=False, initialize=False)
        s0.InitializeSlab(0, 2, None, body_states=[b.state() for b in f0.bodies])
        assert int((s0.solver.global_ids() >= rigid0).sum()) != owners[0]


def test_gloo_world_size_2_rigid_body(oracle, tmp_path):
    """Cylinder drop across two OS processes (gloo): neighbour exchanges + the integer all-reduce of the wrench table + a
    collective re-cut halfway through, with rigid markers on both ranks."""
    out = tmp_path / "slab_gloo_body.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29527", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29527",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "oracle", "--body", "--steps", "12", "--recut", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    s, f = scenes.object_drop(oracle, initial_spacing=0.05, variable_dt=False, initialize=False)
    f.bodies[0].lin_vel = np.array([-1.5, 0.0, -1.0])
    s.Initialize([b.state() for b in f.bodies])
    for _ in range(12):
        f.DoStepDynamics(2e-4)
    ref = util.snapshot(s)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(got["wrench"], np.concatenate(s.GetBodyWrenches()[0]))
    np.testing.assert_array_equal(got["body_pos"], f.bodies[0].pos)


def _crm_single(api, nsteps, h, **kw):
    s = scenes.granular_column(api, dt=h, **kw)
    for _ in range(nsteps):
        s.DoStepDynamics(h)
    snap = util.snapshot(s)
    snap["stress"] = s.solver.stress()
    snap["pc"] = s.solver.consolidation()
    return snap


def _crm_slabs(api, nranks, nsteps, h, cuts=None, **kw):
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.granular_column(api, dt=h, initialize=False, **kw)
        used = s.InitializeSlab(r, nranks, cuts)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, used))
    slab.run_threads(ranks, nsteps, h)
    pieces = []
    for r in ranks:
        pc = r.owned()
        pc["stress"], pc["pc"] = r.sys.solver.stress(), r.sys.solver.consolidation()
        pieces.append(pc)
    return slab.gather_global(pieces, ranks[0].sys.n_global, extra=("stress", "pc")), ranks


@pytest.mark.parametrize("nranks,mcc", [(2, None), (3, {"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})])
def test_crm_granular_column_under_slabs_is_bitwise_single_rank(oracle, nranks, mcc):
    """CRM (elastic_SPH) soil column cut by the slab boundaries: the stress state (and the Cam-Clay state) travels with migrating
    markers, ghosts and the per-stage halo; fields, stress tensor and consolidation state equal the single-rank run."""
    kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc=mcc)
    h, nsteps = 1e-4, 30
    ref = _crm_single(oracle, nsteps, h, **kw)
    assert np.abs(ref["vel"]).max() > 1e-3  # the column is collapsing
    got, ranks = _crm_slabs(oracle, nranks, nsteps, h, **kw)
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert all(r.api.slab_one_round(r.ctx) == 1 for r in ranks)  # the stress state also rides the one-round refresh records
    owned_fluid = [int((r.owned()["rpm"][:, 3] < -0.5).sum()) for r in ranks]
    assert sum(1 for o in owned_fluid if o) >= 2  # the soil really is shared between ranks


def crm_cut_that_soil_crosses(api, nsteps, h, kw):
    """A 2-rank cut (cell boundary) that soil markers of the collapsing column cross within nsteps, and how many do."""
    s0 = scenes.granular_column(api, dt=h, **kw)
    dp = s0.GetParams()
    fluid = s0.solver.rho_pres_mu_type()[:, 3] < -0.5
    c0 = slab.cell_column(s0.solver.positions()[:, 0], dp)
    for _ in range(nsteps):
        s0.DoStepDynamics(h)
    c1 = slab.cell_column(s0.solver.positions()[:, 0], dp)
    moved = fluid & (c1 != c0)
    assert moved.any(), "no soil marker changed its cell column: lengthen the run"
    cross = np.bincount(np.maximum(c0[moved], c1[moved]), minlength=int(dp.gridSize[0]) + 1)  # boundary b separates columns b-1 | b
    cross[:4] = 0
    cross[int(dp.gridSize[0]) - 3:] = 0
    b = int(np.argmax(cross))
    return [0, b, int(dp.gridSize[0])], int(cross[b])


def test_crm_markers_migrate_between_slabs_with_their_stress(oracle):
    """Soil markers cross the cut while the column collapses: as emigrants kept as ghosts and as immigrants they carry stress and
    Cam-Clay state through the one-round refresh records."""
    kw = dict(radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})
    h, nsteps = 2e-4, 600
    cuts, ncross = crm_cut_that_soil_crosses(oracle, nsteps, h, kw)
    assert ncross > 0
    ref = _crm_single(oracle, nsteps, h, **kw)
    got, ranks = _crm_slabs(oracle, 2, nsteps, h, cuts=cuts, **kw)
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    assert all(r.api.slab_one_round(r.ctx) == 1 for r in ranks)


def test_gloo_world_size_2_crm(oracle, tmp_path):
    """CRM soil column across two OS processes (gloo): the stress companions travel through DistTransport with every refresh and halo
    message, and through the two-round records of the collective re-cut."""
    out = tmp_path / "slab_gloo_crm.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "oracle", "--crm", "--steps", "16", "--recut", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    got = np.load(out)
    ref = _crm_single(oracle, 16, 1e-4, radius=0.1, height=0.2, d0=0.02, box_factor=6.0, mcc={"M": 1.02, "kappa": 0.01, "lambda": 0.04, "v_lambda": 1.8, "pc": 2000.0})
    for k in ("pos", "vel", "rpm", "deriv", "stress", "pc"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


# ---- slabs along a periodic x axis: the ranks form a ring -----------------------------------------------------------------
def _poiseuille_single(api, nsteps, bx):
    s = scenes.poiseuille(api, bxDim=bx, t_start=4.0)
    for _ in range(nsteps):
        s.DoStepDynamics(scenes.POISEUILLE["dt"])
    return util.snapshot(s), s


def _poiseuille_ring(api, nranks, nsteps, bx, cuts=None):
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.poiseuille(api, bxDim=bx, t_start=4.0, initialize=False)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, cuts)))
    slab.run_threads(ranks, nsteps, scenes.POISEUILLE["dt"])
    return slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global), ranks


@pytest.mark.parametrize("nranks", [3, 4])
def test_poiseuille_in_a_ring_of_slabs_is_bitwise_single_rank(oracle, nranks):
    """The reference's own unit test scene (utest_FSI_Poiseuille_flow.cpp: periodic in x and y, laminar viscosity, consistent operators)
    cut into x-slabs: with a periodic x axis the first and the last rank are neighbours across the seam (ghosts, halo refreshes and
    emigrants travel around the ring; ApplyPeriodicBoundaryX_D, phys/SphFluidDynamics.cu:843-870, moves the markers that leave through
    one end of the box to the other, i.e. from the last rank to the first).  Owned markers bit-identical to the single-rank run."""
    nsteps, bx = 25, 0.4
    ref, s1 = _poiseuille_single(oracle, nsteps, bx)
    got, ranks = _poiseuille_ring(oracle, nranks, nsteps, bx)
    assert all(r.ring and r.left == (r.rank - 1) % nranks and r.right == (r.rank + 1) % nranks for r in ranks)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    # markers did cross the seam: some of the last rank's initial markers are owned by the first rank now (and wrapped by a period)
    gx = int(s1.DerivedParams().gridSize[0])
    first = ranks[0].owned()
    x0 = scenes.poiseuille(oracle, bxDim=bx, initialize=False)  # the staged coordinates, in global marker order
    x_init = np.concatenate([xyz[:, 0] for (_tc, xyz, _v, _r, _h) in x0._pending])
    came_round = np.count_nonzero(x_init[first["gid"]] > 0.1)
    assert gx >= 4 * nranks and came_round > 0


def test_ring_of_two_slabs_is_rejected(oracle):
    s = scenes.poiseuille(oracle, bxDim=0.4, initialize=False)
    tr = slab.ThreadTransport(2)
    cuts = s.InitializeSlab(0, 2, None)
    with pytest.raises(fsi.SphError, match="at least three ranks"):
        slab.SlabRank(s, tr.endpoint(0), 0, 2, cuts)


def test_gloo_world_size_3_ring(oracle, tmp_path):
    """The ring across three OS processes (torch.distributed, gloo): the seam messages between rank 0 and rank 2 are ordinary point-to-point
    messages between two more neighbours."""
    out = tmp_path / "slab_gloo_ring.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "slab_worker.py"), "--backend", "gloo", "--lib", "oracle", "--poiseuille", "--steps", "12", "--out", str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=ROOT)
    got = np.load(out)
    ref, _s = _poiseuille_single(oracle, 12, 0.4)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)


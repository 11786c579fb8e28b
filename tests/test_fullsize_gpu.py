"""Properties at BASELINE.json's full single-GPU size (configs[1]: the dam break at d0 = 0.01, 1 234 797 markers), where the CPU
oracle is too slow to be the checker: invariants of the neighbour lists and of the sort, agreement of the two arithmetic modes with
each other (SPH_ARITH_EXACT is the one pinned to the oracle bit for bit at the small sizes, tests/test_parity_gpu.py), run-to-run
reproducibility and the slab decomposition against one context.  The other configurations of BASELINE.json at their full sizes follow: C3 (17.3 M
markers, 8 slabs), C4 (cylinder drop, 4 M fluid markers, coupled, 4 slabs) and C5 (I2SPH, 8.8 M markers, 8 slabs), each cut into slab contexts on this
one GPU and held bit for bit against the single-context run.  Everything goes through the C ABI."""
import numpy as np
import pytest

from chrono_wisc_amd import scenes

import util

pytestmark = pytest.mark.gpu

LABEL, DT, STEPS = "C2", 1e-5, 5


def _run(gpu, mode, steps=STEPS):
    s = scenes.dam_break_named(LABEL, gpu)
    s.SetArithmetic(mode)
    for _ in range(steps):
        s.DoStepDynamics(DT)
    return s


@pytest.fixture(scope="module")
def exact(gpu):
    return _run(gpu, "exact")


@pytest.fixture(scope="module")
def fast(gpu):
    return _run(gpu, "fast")


def test_full_size_lists_are_symmetric_sorted_and_self_first(exact):
    """j in N(i) <=> i in N(j) (same distance test both ways), every row starts with itself and lists no marker twice, row lengths add up to NNZ, and interior lattice rows have the lattice's neighbour count."""
    offs, idx = exact.solver.neighbor_csr()
    n = offs.size - 1
    assert n == exact.GetCounters().numAllMarkers == 1234797
    cnt = np.diff(offs.astype(np.int64))
    assert cnt.sum() == idx.size and cnt.min() >= 1
    rows = np.repeat(np.arange(n, dtype=np.int64), cnt)
    first = offs[:-1].astype(np.int64)
    assert np.array_equal(idx[first].astype(np.int64), np.arange(n))                  # self first
    rest = np.ones(idx.size, dtype=bool); rest[first] = False
    fwd = np.sort(rows[rest] * n + idx[rest].astype(np.int64))
    bwd = np.sort(idx[rest].astype(np.int64) * n + rows[rest])
    assert np.all(np.diff(fwd) > 0)                                                    # no neighbour twice in a row, never the row itself again
    assert np.array_equal(fwd, bwd)                                                    # symmetry
    assert 25 <= np.median(cnt) <= 35 and cnt.max() < 64


def test_full_size_staged_search_writes_the_lists_of_the_row_per_lane_search(gpu, exact, monkeypatch):
    """k_neighbors_lds (candidates staged in LDS, bit masks, two-piece windows: the default) against k_neighbors_soa16 (one row per lane, the kernel
    the small-size parity tests pinned to the oracle before the staged one existed; CHSPH_SEARCH_LDS=0 is read by sph_create) on the full-size state
    after 5 steps: the same CSR, entry for entry."""
    monkeypatch.setenv("CHSPH_SEARCH_LDS", "0")
    other = _run(gpu, "exact")
    o1, i1 = exact.solver.neighbor_csr()
    o2, i2 = other.solver.neighbor_csr()
    assert np.array_equal(o1, o2)
    assert np.array_equal(i1, i2)
    a, b = util.snapshot(exact), util.snapshot(other)
    for k in ("pos", "vel", "rpm", "deriv"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("switches", [{"CHSPH_NO_SCAN_SINGLE": "1", "CHSPH_EAGER_MAP": "1", "CHSPH_EAGER_SXT": "1", "CHSPH_NO_PACKED_YWRAP": "1"},
                                      {"CHSPH_SCAN_PREFIX_MIN": "0"}])
def test_full_size_resort_alternatives_give_the_same_order_lists_and_fields(gpu, exact, monkeypatch, switches):
    """The re-sort's one-launch scan (histogram zeroed behind the scan, block totals from k_resident_hash; with and without the one-block prefix
    pass of large windows), the arrays it leaves out and the packed scan of the y-edge waves against the round-3 sequence (three scan launches, every
    array written, scalar edge scan) on the full-size state after 5 steps: same permutation, same CSR, same fields."""
    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    other = _run(gpu, "exact")
    s1, h1 = exact.solver.sorted_index()
    s2, h2 = other.solver.sorted_index()
    assert np.array_equal(s1, s2) and np.array_equal(h1, h2)
    o1, i1 = exact.solver.neighbor_csr()
    o2, i2 = other.solver.neighbor_csr()
    assert np.array_equal(o1, o2)
    assert np.array_equal(i1, i2)
    a, b = util.snapshot(exact), util.snapshot(other)
    for k in ("pos", "vel", "rpm", "deriv"):
        assert np.array_equal(a[k], b[k]), k


def test_full_size_sort_is_a_stable_permutation_by_cell(exact):
    s2o, h = exact.solver.sorted_index()
    n = s2o.size
    assert np.array_equal(np.sort(s2o), np.arange(n, dtype=s2o.dtype))                # a permutation
    assert np.all(np.diff(h.astype(np.int64)) >= 0)                                    # cell ids ascend
    same = np.diff(h.astype(np.int64)) == 0
    assert np.all(np.diff(s2o.astype(np.int64))[same] > 0)                             # ties by original index: stable


def test_full_size_fast_arithmetic_agrees_with_exact(exact, fast):
    """Same bar as tests/test_fast_gpu.py holds against the oracle at the small sizes: lists bit for bit, fields within 1e-4 of their
    inf-norm, pressure within 2 ulps of density x Cs^2."""
    util.compare_csr(fast, exact)
    rep = {}
    util.compare_state_fast(fast, exact, rtol=1e-4, report=rep)
    assert rep["pos"][0] <= 1e-8 and rep["rho"][0] <= 1e-6


def test_full_size_runs_are_reproducible(gpu, fast):
    again = _run(gpu, "fast")
    util.compare_csr(again, fast)
    util.compare_state(again, fast, rtol=0.0)


def test_full_size_bulk_quantities_are_physical(exact):
    """Five steps into the collapse: densities within 1 % of rho0, no marker faster than the model's speed bound, fluid still inside the tank."""
    rpm = exact.solver.rho_pres_mu_type()
    vel = exact.solver.velocities()
    pos = exact.solver.positions()
    P = exact.GetParams()
    fluid = rpm[:, 3] < -0.5
    assert fluid.sum() == 1036521
    assert np.all(np.abs(rpm[fluid, 0] / P.rho0 - 1.0) < 0.01)
    assert np.abs(vel[fluid]).max() < 0.1 * P.Cs
    assert pos[fluid, 2].min() > 0.0 and np.isfinite(pos).all()


def test_full_size_four_slabs_are_bit_identical_to_one_context(gpu, fast):
    """SURVEY 8e at the full single-GPU size: the same dam break cut into four x-slab contexts (in-process transport, all on this
    GPU), stepped in lockstep with the halo / refresh messages of the multi-GPU path -- every owned marker bit for bit as in the
    one-context run (default arithmetic)."""
    from chrono_wisc_amd import slab
    nranks = 4
    tr = slab.ThreadTransport(nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named(LABEL, gpu, initialize=False)
        s.SetArithmetic("fast")
        cuts = s.InitializeSlab(r, nranks)
        ranks.append(slab.SlabRank(s, tr.endpoint(r), r, nranks, cuts))
    slab.run_threads(ranks, STEPS, DT)
    for r in ranks:
        r.sys.Synchronize()
    owned = sum(r.sys.GetCounters().numAllMarkers for r in ranks)
    assert owned == 1234797
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    ref = util.snapshot(fast)
    for k in ("pos", "vel", "rpm", "deriv"):
        assert int((got[k] != ref[k]).any(axis=1).sum()) == 0, k


def test_c3_eight_slabs_with_overlapped_halo_match_one_context(gpu, monkeypatch):
    """BASELINE.json configs[2] at full size (dam break s = 1.0, d0 = 0.01: 17 294 937 markers) cut into the 8 x-slabs the 8-GPU run uses, all
    eight contexts on this one GPU as threads of one process: every step runs through sph_slab_step over the library's in-process stream-ordered
    transport (one-round resident preamble from the second step on, inter-stage halo on the second stream behind the force kernel's edge waves),
    in the library's default fast arithmetic.  Every field of every marker must be bit-identical to the single-context run: each owned row sees the
    same neighbours in the same order, and the split launch runs every wave exactly once."""
    from chrono_wisc_amd import slab
    monkeypatch.setenv("CHSPH_ARITH", "1")
    label, nranks, steps = "C3", 8, 4
    dt = scenes.DAM_BREAK_CONFIGS[label][2]
    group = slab.InprocGroup(gpu, nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_named(label, gpu, initialize=False)
        ranks.append(slab.SlabRank(s, group.endpoint(r), r, nranks, s.InitializeSlab(r, nranks)))
    assert ranks[0].sys.n_global == 17294937
    slab.run_threads(ranks, steps, dt, native=True)
    assert all(gpu.get_arithmetic(r.ctx) == 1 for r in ranks)
    assert all(gpu.slab_one_round(r.ctx) == 1 for r in ranks)
    assert all(gpu.slab_overlapped_halos(r.ctx) == steps for r in ranks)  # one inter-stage halo per RK2 step, all on the overlapped path
    rows = sum(r.sys.GetCounters().numExtendedParticles for r in ranks)
    assert 17294937 < rows < 1.12 * 17294937  # ghost rows: the thin layer keeps them under 12 %
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    for r in ranks:
        r.close()
        r.sys.solver.close()
    del ranks
    assert gpu.slab_group_destroy(group.handle) == 0
    one = scenes.dam_break_named(label, gpu)
    for _ in range(steps):
        one.DoStepDynamics(dt)
    sv = one.solver
    sv.synchronize()
    for k, ref in (("pos", sv.positions()), ("vel", sv.velocities()), ("rpm", sv.rho_pres_mu_type()), ("deriv", sv.derivatives())):
        bad = int((got[k] != ref).any(axis=1).sum())
        assert bad == 0, "%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref))


def test_c4_cylinder_drop_full_size_in_four_slabs_matches_one_context(gpu, monkeypatch):
    """BASELINE.json configs[3] at full size (cylinder drop, spacing 0.0054: 4.0 M fluid markers + the tank's and the body's BCE markers), coupled:
    cut into four x-slab contexts on this GPU (threads, in-process stream-ordered transport), every rank advancing its own copy of the body from
    the wrench that sph_slab_body_wrenches sums over the ranks; fluid steps through sph_slab_step in the default fast arithmetic.  Fields of every
    marker, the wrench and the body state bit-identical to the single-context coupled run (ChFsiSystem::DoStepDynamics, fsi/ChFsiSystem.cpp:264-303;
    Rigid_Forces_Torques, phys/SphBceManager.cu:529-548)."""
    import threading
    from chrono_wisc_amd import slab
    monkeypatch.setenv("CHSPH_ARITH", "1")
    d0, nranks, steps = 0.0054, 4, 6
    h = 2e-5
    kw = dict(initial_spacing=d0, variable_dt=False, step_size=h, check_errors=True, initial_height=0.9)  # partly submerged: the wrench is non-zero from the first step
    push = np.array([-0.5, 0.0, -1.0])

    def make():
        s, f = scenes.object_drop(gpu, initialize=False, **kw)
        f.bodies[0].lin_vel = push.copy()
        return s, f

    group = slab.InprocGroup(gpu, nranks)
    ranks, made = [], []
    for r in range(nranks):
        s, f = make()
        ranks.append(slab.SlabRank(s, group.endpoint(r), r, nranks, s.InitializeSlab(r, nranks, body_states=[b.state() for b in f.bodies])))
        made.append(f)
    assert 3.9e6 < ranks[0].sys.n_global < 6.0e6
    errs = []

    def work(i):
        try:
            rk, f = ranks[i], made[i]
            rk.native_collectives = True
            f.advance_cfd = rk.step_native
            for _ in range(steps):
                f.DoStepDynamics(h)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    wr = [np.concatenate(r.sys.GetBodyWrenches()[0]) for r in ranks]
    bodies = [(f.bodies[0].pos.copy(), f.bodies[0].rot.copy()) for f in made]
    for r in ranks:
        r.close()
        r.sys.solver.close()
    del ranks, made
    assert gpu.slab_group_destroy(group.handle) == 0
    s1, f1 = make()
    s1.Initialize([b.state() for b in f1.bodies])
    for _ in range(steps):
        f1.DoStepDynamics(h)
    sv = s1.solver
    sv.synchronize()
    assert gpu.get_arithmetic(sv.ctx) == 1
    for k, ref in (("pos", sv.positions()), ("vel", sv.velocities()), ("rpm", sv.rho_pres_mu_type()), ("deriv", sv.derivatives())):
        bad = int((got[k] != ref).any(axis=1).sum())
        assert bad == 0, "%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref))
    w1 = np.concatenate(s1.GetBodyWrenches()[0])
    assert np.abs(w1).max() > 0 and np.isfinite(w1).all()
    for w, (bp, br) in zip(wr, bodies):
        np.testing.assert_array_equal(w, w1)
        np.testing.assert_array_equal(bp, f1.bodies[0].pos)
        np.testing.assert_array_equal(br, f1.bodies[0].rot)


def test_c5_implicit_dam_break_full_size_in_eight_slabs_matches_one_context(gpu):
    """BASELINE.json configs[4] at full size (I2SPH dam break, 8.8 M markers) cut into the 8 x-slabs its 8-GPU run uses, all on this GPU: one step
    through sph_slab_step over the in-process transport -- the seven phases, per Jacobi sweep one kernel, the ghost exchange of the iterate and the
    max-reduction of the residual record (phys/SphForceISPH.cu:1026-1042,1135-1164), the sweep cap at 24 so that both loops run to the cap.  Every field
    of every marker and both sweep counts bit-identical to the single-context run."""
    from chrono_wisc_amd import slab
    label, nranks, cap = "C5geom", 8, 24
    dt = 5e-4 * scenes.DAM_BREAK_CONFIGS[label][1] / 0.05
    group = slab.InprocGroup(gpu, nranks)
    ranks = []
    for r in range(nranks):
        s = scenes.dam_break_implicit(gpu, label, max_num_iters=cap, initialize=False, check_errors=True)
        ranks.append(slab.SlabRank(s, group.endpoint(r), r, nranks, s.InitializeSlab(r, nranks)))
    assert 8.0e6 < ranks[0].sys.n_global < 9.5e6
    slab.run_threads(ranks, 1, dt, native=True)
    got = slab.gather_global([r.owned() for r in ranks], ranks[0].sys.n_global)
    its = [r.sys.solver.isph_iterations() for r in ranks]
    for r in ranks:
        r.close()
        r.sys.solver.close()
    del ranks
    assert gpu.slab_group_destroy(group.handle) == 0
    one = scenes.dam_break_implicit(gpu, label, max_num_iters=cap, check_errors=True)
    one.DoStepDynamics(dt)
    sv = one.solver
    sv.synchronize()
    assert all(tuple(i) == tuple(sv.isph_iterations()) for i in its), (its, sv.isph_iterations())
    for k, ref in (("pos", sv.positions()), ("vel", sv.velocities()), ("rpm", sv.rho_pres_mu_type()), ("deriv", sv.derivatives())):
        bad = int((got[k] != ref).any(axis=1).sum())
        assert bad == 0, "%s: %d of %d markers differ from the single-context run" % (k, bad, len(ref))


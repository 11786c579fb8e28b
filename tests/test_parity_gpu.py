"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): integer neighbour lists bit-exact, float fields within 1e-5
relative (to the field's inf-norm) over <= 10 steps from the initial lattice."""
import ctypes as C
import os

import numpy as np
import pytest

from chrono_wisc_amd import fsi, scenes
from chrono_wisc_amd.capi import SphError

import util

pytestmark = pytest.mark.gpu


def _run_pair(gpu, oracle, label, nsteps, dt=None, **kw):
    a, b = util.make_pair(gpu, oracle, label, **kw)
    util.compare_params(a.GetParams(), b.GetParams())
    h = dt if dt is not None else scenes.DAM_BREAK_CONFIGS[label][2]
    rep = {}
    for k in range(nsteps):
        a.DoStepDynamics(h)
        b.DoStepDynamics(h)
        if k == 0:
            util.compare_csr(a, b)
    util.compare_csr(a, b)
    util.compare_state(a, b, report=rep)
    return a, b, rep


def test_derived_parameters_and_counters(gpu, oracle):
    a, b = util.make_pair(gpu, oracle, "C1")
    util.compare_params(a.GetParams(), b.GetParams())
    ca, cb = a.GetCounters(), b.GetCounters()
    for name, _t in ca._fields_:
        assert getattr(ca, name) == getattr(cb, name), name
    assert ca.numAllMarkers == 30327 and ca.numFluidMarkers == 16731
    assert list(a.GetParams().gridSize) == [70, 5, 160]


def test_tiny_rk2_10_steps(gpu, oracle):
    _a, _b, rep = _run_pair(gpu, oracle, "tiny", 10)
    print("tiny RK2 x10:", rep)


def test_c1_dam_break_rk2_10_steps(gpu, oracle):
    """C1 = the reference demo verbatim (30 327 markers)."""
    _a, _b, rep = _run_pair(gpu, oracle, "C1", 10)
    print("C1 RK2 x10:", rep)


@pytest.mark.parametrize("scheme", [fsi.IntegrationScheme.EULER, fsi.IntegrationScheme.SYMPLECTIC])
def test_other_integrators(gpu, oracle, scheme):
    _run_pair(gpu, oracle, "tiny", 5, scheme=scheme)


@pytest.mark.parametrize("kernel", [fsi.KernelType.QUADRATIC, fsi.KernelType.QUINTIC_SPLINE, fsi.KernelType.WENDLAND])
def test_other_kernels(gpu, oracle, kernel):
    def tweak(sp):
        sp.kernel_type = kernel
    _run_pair(gpu, oracle, "tiny", 3, tweak=tweak)


def test_laminar_isothermal_no_delta(gpu, oracle):
    def tweak(sp):
        sp.viscosity_method = fsi.ViscosityMethod.LAMINAR
        sp.eos_type = fsi.EosType.ISOTHERMAL
        sp.use_delta_sph = False
        sp.shifting_method = fsi.ShiftingMethod.NONE
    _run_pair(gpu, oracle, "tiny", 5, tweak=tweak)


@pytest.mark.parametrize("shift", [fsi.ShiftingMethod.PPST, fsi.ShiftingMethod.PPST_XSPH, fsi.ShiftingMethod.DIFFUSION, fsi.ShiftingMethod.DIFFUSION_XSPH])
def test_shifting_variants(gpu, oracle, shift):
    def tweak(sp):
        sp.shifting_method = shift
    _run_pair(gpu, oracle, "tiny", 3, tweak=tweak)


def test_default_support_radius_multiplier(gpu, oracle):
    """d0_multiplier = 1.2 (the reference default): ~57 neighbours per row, off-lattice thresholds."""
    def tweak(sp):
        sp.d0_multiplier = 1.2
    _run_pair(gpu, oracle, "tiny", 3, tweak=tweak)


def test_proximity_search_every_second_step(gpu, oracle):
    _run_pair(gpu, oracle, "tiny", 6, ps_freq=2)


def test_variable_time_step(gpu, oracle):
    def tweak(sp):
        sp.use_variable_time_step = True
    a, b = util.make_pair(gpu, oracle, "tiny", tweak=tweak)
    for _ in range(4):
        ha, hb = a.GetCurrentStepSize(), b.GetCurrentStepSize()
        assert ha == pytest.approx(hb, rel=1e-6)
        a.DoStepDynamics(hb)
        b.DoStepDynamics(hb)
    util.compare_state(a, b)


def _with_extra_fluid(api, extra):
    s = scenes.dam_break_named("tiny", api, initialize=False)
    tc, xyz, vel, rpm, h = s._pending[0]  # the fluid block (fluid must precede BCE)
    s._pending[0] = (tc, np.concatenate([xyz, np.asarray(extra, dtype=np.float32)]), vel, np.concatenate([rpm, np.tile(rpm[:1], (len(extra), 1))]), h)
    s.Initialize()
    return s


def test_zombie_particles_leave_the_sorted_set(gpu, oracle):
    """Fluid markers outside a non-periodic domain side become zombies (activity -1) and drop out of the
    extended set (phys/SphFluidDynamics.cu:241-263): numExtendedParticles < numAllMarkers."""
    hi = float(scenes.dam_break_named("tiny", oracle, initialize=False).p.cMax[2])
    extra = [[0.0, 0.0, hi + 0.01], [0.1, 0.0, hi + 0.02]]
    a, b = _with_extra_fluid(gpu, extra), _with_extra_fluid(oracle, extra)
    for _ in range(3):
        a.DoStepDynamics(1e-4)
        b.DoStepDynamics(1e-4)
    ca, cb = a.GetCounters(), b.GetCounters()
    assert cb.numExtendedParticles == cb.numAllMarkers - 2
    assert ca.numExtendedParticles == cb.numExtendedParticles
    util.compare_csr(a, b)
    util.compare_state(a, b)


def test_marker_several_periods_outside_a_periodic_axis(gpu, oracle):
    """The reference wraps a periodic cell coordinate by one period only; a marker farther out (still inside its 40 h bounds
    test, because the period of this scene is much shorter than that) must not index past the cell arrays: it lands in the edge
    cell, identically in the oracle and on the GPU."""
    def scene(api):
        s = scenes.dam_break_named("tiny", api, initialize=False)
        dp = s.DerivedParams()
        y_far = float(dp.worldOrigin[1]) + 4.4 * float(dp.boxDims[1])
        x = float(dp.worldOrigin[0]) + 0.5 * float(dp.boxDims[0])
        z = float(dp.worldOrigin[2]) + 0.9 * float(dp.boxDims[2])
        s._pending.insert(1, (s._pending[0][0], np.array([[x, y_far, z], [x, -y_far, z]], dtype=np.float32), None,
                              np.array([[1000.0, 0.0, 5.0]] * 2, dtype=np.float32), s._pending[0][4]))
        s.Initialize()
        return s

    a, b = scene(gpu), scene(oracle)
    for _ in range(3):
        a.DoStepDynamics(1e-4)
        b.DoStepDynamics(1e-4)
    # the strays' own rows list their edge cell's markers however often the clamped neighbour cells repeat it (undefined in the
    # reference); every marker's state must agree all the same
    util.compare_state(a, b)


def test_nan_position_raises_like_the_reference(gpu):
    s = scenes.dam_break_named("tiny", gpu, initialize=False)
    first = s._pending[0]
    xyz = first[1].copy()
    xyz[5, 0] = np.nan
    s._pending[0] = (first[0], xyz, first[2], first[3], first[4])
    s.Initialize()
    with pytest.raises(SphError, match="calcHashD"):
        s.DoStepDynamics(1e-4)


def test_unstable_step_raises_nan_error(gpu):
    """An unstable dt trips the NaN scans of EulerStep (phys/SphFluidDynamics.cu:677-684)."""
    s = scenes.dam_break_named("tiny", gpu)
    with pytest.raises(SphError, match="NaN|RhsCFD|calcHashD"):
        for _ in range(200):
            s.DoStepDynamics(5e-2)


def _rigid_scene(api):
    """A rigid box of BCE markers hanging in the dam-break column, driven through the exchange hooks."""
    s = scenes.dam_break_named("tiny", api, initialize=False)
    d0 = s.GetInitialSpacing()
    g = np.arange(-1, 2) * d0
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    local = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    state = dict(pos=(-1.0, 0.0, 0.6), rot=(0.9238795325112867, 0.0, 0.3826834323650898, 0.0), lin_vel=(0.2, 0.0, -0.3), ang_vel=(0.0, 1.5, 0.0),
                 lin_acc=(0.0, 0.0, -9.8), ang_acc=(0.0, 0.5, 0.0))
    s.AddFsiBody(local, state)
    s.Initialize()
    return s, state


def test_rigid_body_bce_coupling(gpu, oracle):
    """B1/S1/S2 rows: BCE acceleration, wrench reduction, marker update (phys/SphBceManager.cu)."""
    a, sta = _rigid_scene(gpu)
    b, _ = _rigid_scene(oracle)
    assert a.GetCounters().numRigidMarkers == 27
    h = 1e-4
    for k in range(4):
        a.DoStepDynamics(h)
        b.DoStepDynamics(h)
        a.OnExchangeSolidForces()
        b.OnExchangeSolidForces()
        wa, wb = a.GetBodyWrenches(), b.GetBodyWrenches()
        for (fa, ta), (fb, tb) in zip(wa, wb):
            scale = max(np.abs(fb).max(), 1e-20)
            assert np.abs(fa - fb).max() <= 1e-5 * scale
            scale = max(np.abs(tb).max(), 1e-20)
            assert np.abs(ta - tb).max() <= 1e-5 * scale
        st = dict(sta)
        st["pos"] = (sta["pos"][0] + 0.2 * h * (k + 1), 0.0, sta["pos"][2] - 0.3 * h * (k + 1))
        a.LoadSolidStates([st])
        b.LoadSolidStates([st])
        a.OnExchangeSolidStates()
        b.OnExchangeSolidStates()
    util.compare_csr(a, b)
    util.compare_state(a, b)
    assert np.abs(wb[0][0]).max() > 0  # the body does feel the fluid


def test_several_fluid_steps_between_two_exchanges(gpu, oracle):
    """ChFsiSystem::AdvanceCFD runs as many fluid steps as fit an exchange interval (variable step: src/chrono_fsi/ChFsiSystem.cpp:229-246), and the
    reference re-sorts each of them from the marker-order arrays, whose rigid rows only change at an exchange: every step starts with the rigid BCE
    markers at the pose and VELOCITY of the last exchange, whatever CfdAdamiBC wrote into the sorted copy (its `prescribedVel = 2 v_marker` reads that
    row).  The product keeps the state in cell order across steps and must put those rows back itself -- without that the extrapolated velocity feeds
    back and doubles every step (found in round 4: NaN after 15 steps).  Cylinder drop, three fluid steps per exchange, 8 exchanges: fields, wrench and
    body trajectory bit-identical to the oracle."""
    h, sub = 2e-4, 3

    def run(api):
        s, f = scenes.object_drop(api, initial_spacing=0.05, variable_dt=False, initial_height=0.9)
        f.advance_cfd = lambda step: [s.DoStepDynamics(step / sub) for _ in range(sub)]
        wr = []
        for _ in range(8):
            f.DoStepDynamics(h)
            wr.append(np.concatenate(s.GetBodyWrenches()[0]))
        return s, f, np.array(wr)

    a, fa, wa = run(gpu)
    b, fb, wb = run(oracle)
    np.testing.assert_array_equal(wa, wb)
    np.testing.assert_array_equal(fa.bodies[0].pos, fb.bodies[0].pos)
    util.compare_csr(a, b)
    util.compare_state(a, b, rtol=0.0)
    assert np.abs(wb).max() > 0 and np.isfinite(wa).all()


def test_getters_match_reference_semantics(gpu, oracle):
    a, b = util.make_pair(gpu, oracle, "tiny")
    a.DoStepDynamics(1e-4)
    b.DoStepDynamics(1e-4)
    np.testing.assert_allclose(a.GetParticleAccelerations(), b.GetParticleAccelerations(), rtol=0, atol=1e-5 * np.abs(b.GetParticleAccelerations()).max())
    pa, va, ra = a.solver.sorted_state()
    pb, vb, rb = b.solver.sorted_state()
    util.assert_fields_close("sorted pos", pa, pb)
    util.assert_fields_close("sorted vel", va, vb)
    util.assert_fields_close("sorted rpm", ra, rb)


def test_cxx_facade_dam_break_matches_oracle(oracle, tmp_path):
    """The C++ host mirror (chrono-wisc_amd/host) builds the reference demo's input itself (grid sampler, box-container
    BCE generator, facade setters) and must land on the same state as the Python mirror driving the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "chrono-wisc_amd", "host", "demo_FSI-SPH_DamBreak_MI")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    dump = str(tmp_path / "c1.bin")
    subprocess.run([exe, "--scale", "1", "--spacing", "0.1", "--step", "1e-4", "--steps", "5", "--dump", dump], check=True, timeout=600)
    raw = np.fromfile(dump, dtype=np.uint8)
    n = int(raw[:8].view(np.uint64)[0])
    f = raw[8:].view(np.float32).reshape(3, n, 3)
    b = scenes.dam_break_named("C1", oracle)
    assert n == b.GetCounters().numAllMarkers
    for _ in range(5):
        b.DoStepDynamics(1e-4)
    util.assert_fields_close("pos", f[0], b.GetParticlePositions())
    util.assert_fields_close("vel", f[1], b.GetParticleVelocities())
    util.assert_fields_close("rho,p,mu", f[2], b.GetParticleFluidProperties())


def test_poiseuille_consistent_operators_parity(gpu, oracle):
    """LAMINAR viscosity + consistent gradient/Laplacian operators + exact 6x6 solve (phys/SphForceWCSPH.cu:40-312,492-531;
    sph/math/SphExactLinearSolvers.cuh:179-990), all-periodic domain, isothermal EOS: the configuration of the reference's
    unit test, GPU against oracle."""
    a = scenes.poiseuille(gpu)
    b = scenes.poiseuille(oracle)
    oracle.set_num_threads(b.solver.ctx, 8)
    util.compare_params(a.GetParams(), b.GetParams())
    rep = {}
    for _ in range(5):
        a.DoStepDynamics(2e-3)
        b.DoStepDynamics(2e-3)
    util.compare_csr(a, b)
    util.compare_state(a, b, report=rep)
    print("Poiseuille x5:", rep)


def test_poiseuille_reference_acceptance_criterion_on_gpu(gpu):
    """The reference's own known-answer test (src/tests/unit_tests/fsi/sph/utest_FSI-SPH_Poiseuille_flow.cpp:41-42,206-221):
    at EVERY one of 500 RK2 steps the velocity RMS error relative to the analytical series stays <= 5e-3 and
    (rho_max - rho_min)/rho0 <= 1e-4."""
    c = scenes.POISEUILLE
    s = scenes.poiseuille(gpu)
    t = c["t_start"]
    worst = [0.0, 0.0]
    for _ in range(int(c["num_steps"])):
        s.DoStepDynamics(c["dt"])
        t += c["dt"]
        v_rel, d_rel = scenes.poiseuille_errors(s, t)
        worst = [max(worst[0], v_rel), max(worst[1], d_rel)]
        assert v_rel <= c["v_tolerance"] and d_rel <= c["d_tolerance"], (v_rel, d_rel)
    print("Poiseuille worst (v_rel, d_rel) over 500 steps:", worst)


def test_object_drop_cylinder_variable_dt(gpu, oracle):
    """C4 input at coarse spacing: ChFsiProblemCartesian-style grid + 4-layer box container + polar cylinder BCE, free rigid
    body driven by the Chrono-free 6-DoF driver, variable time step (I4) on, one exchange per fluid step."""
    a, fa = scenes.object_drop(gpu, initial_spacing=0.05)
    b, fb = scenes.object_drop(oracle, initial_spacing=0.05)
    util.compare_params(a.GetParams(), b.GetParams())
    assert a.GetCounters().numRigidMarkers == b.GetCounters().numRigidMarkers > 50
    steps = []
    for k in range(6):
        ha, hb = a.GetCurrentStepSize(), b.GetCurrentStepSize()
        assert ha == hb, (k, ha, hb)
        steps.append(ha)
        fa.DoStepDynamics(ha)
        fb.DoStepDynamics(hb)
        for x, y in ((fa.bodies[0].force, fb.bodies[0].force), (fa.bodies[0].torque, fb.bodies[0].torque)):
            assert np.abs(x - y).max() <= 1e-5 * max(np.abs(y).max(), 1e-12)
    assert steps[0] == 1e-4 and steps[-1] != steps[0]  # the CFL/acceleration estimate took over after the first step
    util.compare_csr(a, b)
    util.compare_state(a, b)
    assert abs(fa.bodies[0].pos[2] - fb.bodies[0].pos[2]) < 1e-12


def test_holmes_boundary_method(gpu, oracle):
    """calcKernelSupport + CfdHolmesBC (phys/SphForceWCSPH.cu:777-811,907-981) instead of the Adami extrapolation: wall BCE
    on the dam break, rigid BCE (prescribed velocity term) on the cylinder drop."""
    def tweak(sp):
        sp.boundary_method = fsi.BoundaryMethod.HOLMES
        sp.shifting_method = fsi.ShiftingMethod.XSPH
        sp.shifting_xsph_eps = 0.5
        sp.use_delta_sph = True
        sp.delta_sph_coefficient = 0.1
        sp.artificial_viscosity = 0.03
        sp.eos_type = fsi.EosType.TAIT
        sp.max_velocity = 10.0
        sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL

    _a, _b, rep = _run_pair(gpu, oracle, "tiny", 6, tweak=tweak)
    print("Holmes dam break:", rep)

    def holmes(sp):
        sp.boundary_method = fsi.BoundaryMethod.HOLMES

    a, fa = scenes.object_drop(gpu, initial_spacing=0.05, variable_dt=False, tweak=holmes)
    b, fb = scenes.object_drop(oracle, initial_spacing=0.05, variable_dt=False, tweak=holmes)
    assert a.GetParams().boundary_method == 1
    for k in range(5):
        fa.DoStepDynamics(1e-4)
        fb.DoStepDynamics(1e-4)
    util.compare_csr(a, b)
    util.compare_state(a, b)
    assert np.abs(fa.bodies[0].force - fb.bodies[0].force).max() <= 1e-5 * max(np.abs(fb.bodies[0].force).max(), 1e-12)


def test_dense_cluster_grows_rows_and_leaves_the_compact_tile(gpu, oracle):
    """Edge of the native neighbour structures: a blob of 400 fluid markers packed into about one cell makes rows longer than
    the initial slab capacity (deferred-resize protocol, search repeated with a larger CAP) and candidate ranges longer than
    the 8-bit offsets of the compact LDS tile (ERR_SEG_RANGE: the search falls back to the 32-bit tile).  The CSR must still
    equal the oracle's bit for bit."""
    def build(api):
        s = scenes.dam_break_named("tiny", api, initialize=False)
        dp = s.DerivedParams()
        rng = np.random.default_rng(5)
        c = np.array([float(dp.worldOrigin[0]) + 30.5 * float(dp.cellSize[0]), 0.0, float(dp.worldOrigin[2]) + 45.5 * float(dp.cellSize[2])])
        blob = c + (rng.random((400, 3)) - 0.5) * 0.9 * float(dp.cellSize[0])
        rpm = np.tile(np.array([[1000.0, 0.0, 5.0]]), (400, 1))
        s._pending.insert(1, (s._pending[0][0], blob.astype(np.float32), None, rpm.astype(np.float32), s._pending[0][4]))
        s.Initialize()
        return s

    a, b = build(gpu), build(oracle)
    a.solver._ck(gpu.force_proximity_search(a.solver.ctx))
    for _ in range(1):
        a.DoStepDynamics(1e-6)
        b.DoStepDynamics(1e-6)
    offs, idx = util.compare_csr(a, b)
    assert np.diff(offs).max() >= 400  # the blob's rows list the whole blob
    util.compare_state(a, b)


def test_random_cloud_and_markers_on_cell_faces_give_the_oracles_lists(gpu, oracle):
    """A lattice never puts a row at an arbitrary place inside its cell, nor a marker exactly on a cell face: here the fluid of the C1 dam break (5 cell layers across the
    periodic axis, so the inner three take the packed path) is a uniform random cloud -- rows anywhere inside their cells, cell
    occupancies from 0 to ~20 -- plus markers placed exactly on cell faces, edges and corners.  Lists and fields bit for bit."""
    def build(api):
        s = scenes.dam_break_named("C1", api, initialize=False)
        dp = s.DerivedParams()
        o = np.array([float(dp.worldOrigin[k]) for k in range(3)])
        cs = np.array([float(dp.cellSize[k]) for k in range(3)])
        rng = np.random.default_rng(17)
        lo = o + cs * np.array([12.0, 0.0, 82.0])      # a block of 20 x 5 x 12 cells inside the container, across the whole periodic width
        n = 9000
        cloud = lo + rng.random((n, 3)) * cs * np.array([20.0, 5.0, 12.0])
        k = rng.integers(0, [20, 5, 12], size=(600, 3)).astype(np.float64)
        onface = lo + (k + np.where(rng.random((600, 3)) < 0.6, 0.0, rng.random((600, 3)))) * cs   # 60 % of the coordinates on a face
        pts = np.concatenate([cloud, onface]).astype(np.float32)
        # keep the periodic coordinate inside the domain proper
        pts[:, 1] = np.clip(pts[:, 1], np.float32(o[1] + 1e-4), np.float32(o[1] + 5 * cs[1] - 1e-4))
        rpm = np.tile(np.array([[1000.0, 0.0, 5.0]], dtype=np.float32), (len(pts), 1))
        fluid = s._pending[0]
        s._pending[0] = (fluid[0], pts, None, rpm, fluid[4])
        s.Initialize()
        return s

    a, b = build(gpu), build(oracle)
    assert int(a.DerivedParams().gridSize[1]) >= 5
    a.DoStepDynamics(1e-8)
    b.DoStepDynamics(1e-8)
    offs, idx = util.compare_csr(a, b)
    cnt = np.diff(offs)
    assert cnt.max() > 40 and cnt.min() < 12      # crowded rows and lonely ones
    util.compare_state(a, b, rtol=0.0)


@pytest.mark.parametrize("case", ["no_markers", "walls_only", "one_fluid_marker"])
def test_degenerate_marker_sets(gpu, oracle, case):
    """Empty and ragged inputs: nothing at all, BCE walls without fluid, a single fluid marker far from every wall."""
    def build(api):
        s = scenes.dam_break_named("tiny", api, initialize=False)
        fluid = s._pending[0]
        if case == "no_markers":
            s._pending = []
        elif case == "walls_only":
            s._pending = s._pending[1:]
        else:
            s._pending[0] = (fluid[0], np.array([[0.0, 0.0, 1.0]], dtype=np.float32), None, np.array([[1000.0, 0.0, 5.0]], dtype=np.float32), fluid[4])
        s.Initialize()
        return s

    a, b = build(gpu), build(oracle)
    ca, cb = a.GetCounters(), b.GetCounters()
    assert (ca.numAllMarkers, ca.numFluidMarkers, ca.numBoundaryMarkers) == (cb.numAllMarkers, cb.numFluidMarkers, cb.numBoundaryMarkers)
    for _ in range(3):
        a.DoStepDynamics(1e-4)
        b.DoStepDynamics(1e-4)
    if ca.numAllMarkers:
        util.compare_csr(a, b)
        util.compare_state(a, b)
    if case == "one_fluid_marker":
        v = a.GetParticleVelocities()[0]
        assert abs(v[2] - (-9.8 * 3e-4)) < 1e-6 and v[0] == 0  # free fall


def test_active_domain_around_a_body(gpu, oracle):
    """SetActiveDomain / SetActiveDomainDelay (UpdateActivityD, phys/SphFluidDynamics.cu:195-238): only markers within the box
    around the FSI body stay active, a slightly larger shell stays in the sorted set ("extended"), the rest is frozen with zero
    velocity and drops out of the neighbour search; before the delay has passed everything is active."""
    def tweak(sysS):
        sysS.SetActiveDomain((0.5, 0.5, 0.6))
        sysS.SetActiveDomainDelay(2.5e-4)

    def build(api):
        s, f = scenes.object_drop(api, initial_spacing=0.05, variable_dt=False, initialize=False)
        tweak(s)
        s.Initialize()
        return s, f

    (a, fa), (b, fb) = build(gpu), build(oracle)
    n = a.GetCounters().numAllMarkers
    seen = []
    for k in range(6):
        fa.DoStepDynamics(1e-4)
        fb.DoStepDynamics(1e-4)
        assert a.GetCounters().numExtendedParticles == b.GetCounters().numExtendedParticles
        seen.append(a.GetCounters().numExtendedParticles)
    assert seen[0] == n and seen[-1] < n  # all markers sorted before the delay, only the extended shell afterwards
    util.compare_csr(a, b)
    util.compare_state(a, b)


# ---- CRM, the continuum granular model: CrmAdamiBC, CrmRHS + crmDvDt, TauEulerStep (mu(I)) -------------------
def _granular(api, **kw):
    args = dict(radius=0.1, height=0.2, d0=0.02, dt=2e-4, box_factor=6)
    args.update(kw)
    return scenes.granular_column(api, **args)


@pytest.mark.parametrize("scheme", [fsi.IntegrationScheme.RK2, fsi.IntegrationScheme.EULER, fsi.IntegrationScheme.SYMPLECTIC])
def test_crm_granular_column_matches_oracle(gpu, oracle, scheme):
    """Collapse of a soil column (after demo_FSI-SPH_AngleRepose.cpp): state, stress and derivatives of every marker."""
    a, b = _granular(gpu, scheme=scheme), _granular(oracle, scheme=scheme)
    util.compare_params(a.GetParams(), b.GetParams())
    assert a.GetPhysicsProblemString() == "CRM"
    for _ in range(25):
        a.DoStepDynamics(2e-4)
        b.DoStepDynamics(2e-4)
    util.compare_csr(a, b)
    rep = {}
    util.compare_state(a, b, report=rep)
    util.assert_fields_close("stress", a.GetParticleStress(), b.GetParticleStress())
    sa = a.GetParticleStress()
    assert np.abs(sa[: a.GetCounters().numFluidMarkers, 2]).max() > 100.0  # the column carries load
    print("CRM", scheme, rep)


def test_crm_variants_match_oracle(gpu, oracle):
    """Unilateral viscosity, cohesion, delta-SPH off / on, XSPH only, proximity search every second step, variable time step."""
    def tweak(sp):
        sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL
        sp.shifting_method = fsi.ShiftingMethod.XSPH
        sp.use_delta_sph = True
        sp.delta_sph_coefficient = 0.1
        sp.use_variable_time_step = True
        sp.num_proximity_search_steps = 2

    a, b = _granular(gpu, tweak=tweak, ps_freq=2), _granular(oracle, tweak=tweak, ps_freq=2)
    for k in range(12):
        ha, hb = a.GetCurrentStepSize(), b.GetCurrentStepSize()
        assert ha == hb, k
        a.DoStepDynamics(ha)
        b.DoStepDynamics(hb)
    util.compare_state(a, b)
    util.assert_fields_close("stress", a.GetParticleStress(), b.GetParticleStress())


def test_crm_holmes_boundary_matches_oracle(gpu, oracle):
    """calcKernelSupport + CrmHolmesBC (phys/SphForceWCSPH.cu:777-811, 819-905)."""
    def tweak(sp):
        sp.boundary_method = fsi.BoundaryMethod.HOLMES

    a, b = _granular(gpu, tweak=tweak), _granular(oracle, tweak=tweak)
    for _ in range(20):
        a.DoStepDynamics(2e-4)
        b.DoStepDynamics(2e-4)
    util.compare_state(a, b)
    util.assert_fields_close("stress", a.GetParticleStress(), b.GetParticleStress())


def test_crm_initial_stress_and_pressure(gpu, oracle):
    """AddSPHParticle's tauXxYyZz / tauXyXzYz arguments reach the solver; rejections of what is not built."""
    def build(api):
        s = _granular(api, initialize=False)
        blk = s._pending[0]
        n = len(blk[1])
        rng = np.random.default_rng(5)
        td = -np.abs(rng.normal(500.0, 100.0, (n, 3)))
        to = rng.normal(0.0, 20.0, (n, 3))
        s._pending_stress[id(blk[1])] = (td.astype(np.float32), to.astype(np.float32))
        s.Initialize()
        return s, td.astype(np.float32), to.astype(np.float32)

    a, td, to = build(gpu)
    b, _, _ = build(oracle)
    n = len(td)
    np.testing.assert_array_equal(a.GetParticleStress()[:n], np.concatenate([td, to], axis=1))
    np.testing.assert_array_equal(a.GetParticleStress()[n:], 0)
    for _ in range(5):
        a.DoStepDynamics(2e-4)
        b.DoStepDynamics(2e-4)
    util.compare_state(a, b)
    util.assert_fields_close("stress", a.GetParticleStress(), b.GetParticleStress())
    s = _granular(gpu, initialize=False)
    s.p.rheology_model_crm = 7
    with pytest.raises(SphError, match="rheology"):
        s.Initialize()


MCC = {"M": 1.02, "kappa": 0.01, "lambda": 0.05, "v_lambda": 2.0, "pc": 2000.0}


@pytest.mark.parametrize("scheme", [fsi.IntegrationScheme.RK2, fsi.IntegrationScheme.SYMPLECTIC])
def test_crm_modified_cam_clay_matches_oracle(gpu, oracle, scheme):
    """MCC rheology (TauEulerStep :428-562, the moduli of CrmRHS :1513-1531): stress, preconsolidation pressure, volumetric
    strain rate and specific volume of every marker."""
    a, b = _granular(gpu, mcc=MCC, dt=1e-4, scheme=scheme), _granular(oracle, mcc=MCC, dt=1e-4, scheme=scheme)
    util.assert_fields_close("pcEvSv at start", a.GetParticleConsolidation(), b.GetParticleConsolidation())
    for _ in range(60):
        a.DoStepDynamics(1e-4)
        b.DoStepDynamics(1e-4)
    util.compare_state(a, b)
    util.assert_fields_close("stress", a.GetParticleStress(), b.GetParticleStress())
    util.assert_fields_close("pcEvSv", a.GetParticleConsolidation(), b.GetParticleConsolidation())
    n = a.GetCounters().numFluidMarkers
    pc = a.GetParticleConsolidation()[:n]
    assert pc[:, 0].max() > 1.05 * 2000.0 * 1.2 or pc[:, 0].min() < 2000.0  # hardening / softening happened: the plastic branch ran


def test_crm_pow_2p55_is_bitwise_the_oracles(gpu, oracle):
    a, b = _granular(gpu), _granular(oracle)
    rng = np.random.default_rng(11)
    x = np.concatenate([[0.0, 1.0, 0.5, 2.0, 1e-300, 1e300, 1e-5, 0.9999999, 1.0000001], rng.uniform(0, 1.5, 20000), 10.0 ** rng.uniform(-30, 30, 2000)])
    ya, yb = a.solver.pow_2p55(x), b.solver.pow_2p55(x)
    np.testing.assert_array_equal(ya.view(np.uint64), yb.view(np.uint64))
    fin = np.isfinite(yb) & (yb > 0)
    assert np.abs(yb[fin] / np.power(x[fin], 2.55) - 1).max() < 1e-13


def test_crm_rigid_body_on_soil(gpu, oracle):
    """The use CRM exists for: a rigid body (BCE markers with a prescribed motion, like a wheel or a cone) pressed into the soil
    column -- BCE acceleration, CrmAdamiBC on solid markers (2 v_body - extrapolated velocity), wrench, marker update."""
    def build(api):
        s = _granular(api, initialize=False)
        d0 = s.GetInitialSpacing()
        g = np.arange(-2, 3) * d0
        X, Y, Z = np.meshgrid(g, g, np.arange(0, 3) * d0, indexing="ij")
        local = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        top = -0.15 + 0.2 + 2 * d0  # floor + column height + the demo's offset
        state = dict(pos=(0.0, 0.0, top + 0.5 * d0), rot=(1.0, 0.0, 0.0, 0.0), lin_vel=(0.0, 0.0, -0.5), ang_vel=(0.0, 0.0, 2.0), lin_acc=(0.0, 0.0, -9.81),
                     ang_acc=(0.0, 0.0, 0.0))
        s.AddFsiBody(local, state)
        s.Initialize()
        return s, state

    a, sta = build(gpu)
    b, _ = build(oracle)
    assert a.GetCounters().numRigidMarkers == 75
    h = 2e-4
    for k in range(15):
        a.DoStepDynamics(h)
        b.DoStepDynamics(h)
        a.OnExchangeSolidForces()
        b.OnExchangeSolidForces()
        for (fa, ta), (fb, tb) in zip(a.GetBodyWrenches(), b.GetBodyWrenches()):
            assert np.abs(fa - fb).max() <= 1e-5 * max(np.abs(fb).max(), 1e-20)
            assert np.abs(ta - tb).max() <= 1e-5 * max(np.abs(tb).max(), 1e-20)
        st = dict(sta)
        st["pos"] = (0.0, 0.0, sta["pos"][2] - 0.5 * h * (k + 1))
        a.LoadSolidStates([st])
        b.LoadSolidStates([st])
        a.OnExchangeSolidStates()
        b.OnExchangeSolidStates()
    util.compare_state(a, b)
    util.assert_fields_close("stress", a.GetParticleStress(), b.GetParticleStress())
    assert b.GetBodyWrenches()[0][0][2] > 0  # the soil pushes back


@pytest.mark.parametrize("name,kw", [("crm_mui_20steps", dict(dt=2e-4)), ("crm_mcc_20steps", dict(dt=1e-4, mcc=MCC))])
def test_crm_matches_committed_golden_vectors(gpu, name, kw):
    """The committed CRM vectors (tests/golden, generated from the oracle) reproduced by the HIP path bit for bit, without the
    oracle in the loop."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    s = _granular(gpu, **kw)
    for _ in range(20):
        s.DoStepDynamics(kw["dt"])
    got = dict(pos=s.solver.positions(), vel=s.solver.velocities(), rpm=s.solver.rho_pres_mu_type(), deriv=s.solver.derivatives(), stress=s.solver.stress(),
               pcv=s.solver.consolidation())
    for k, v in got.items():
        np.testing.assert_array_equal(v, g[k], err_msg=k)


def test_dam_break_matches_committed_golden_vectors(gpu):
    """tests/golden/tiny_rk2_5steps.npz (generated from the oracle) reproduced by the HIP path without the oracle in the loop:
    neighbour lists bit-exact, fields within 1e-5."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_rk2_5steps.npz"))
    s = scenes.dam_break_named("tiny", gpu)
    for _ in range(5):
        s.DoStepDynamics(1e-4)
    offs, idx = s.solver.neighbor_csr()
    np.testing.assert_array_equal(offs, g["offs"])
    np.testing.assert_array_equal(idx, g["idx"])
    snap = util.snapshot(s)
    for k in ("pos", "vel", "rpm", "deriv"):
        util.assert_fields_close(k, snap[k], g[k])


def test_cxx_facade_angle_of_repose_matches_oracle(oracle, tmp_path):
    """The C++ host mirror builds the CRM angle-of-repose demo itself (SetElasticSPH, grid sampler, plate) and steps it on the
    device; it must land on the state and stress of the Python mirror driving the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "chrono-wisc_amd", "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    dump = str(tmp_path / "repose.bin")
    subprocess.run([os.path.join(host, "demo_FSI-SPH_AngleRepose_MI"), "--radius", "0.1", "--height", "0.2", "--spacing", "0.02", "--box", "6", "--dt", "2e-4", "--steps", "15",
                    "--dump", dump], check=True, timeout=600)
    raw = np.fromfile(dump, dtype=np.uint8)
    n = int(raw[:8].view(np.uint64)[0])
    f = raw[8:8 + 4 * 9 * n].view(np.float32).reshape(3, n, 3)
    st = raw[8 + 4 * 9 * n:].view(np.float32).reshape(n, 6)
    b = _granular(oracle)
    assert n == b.GetCounters().numAllMarkers
    for _ in range(15):
        b.DoStepDynamics(2e-4)
    util.assert_fields_close("pos", f[0], b.GetParticlePositions())
    util.assert_fields_close("vel", f[1], b.GetParticleVelocities())
    util.assert_fields_close("rho,p,mu", f[2], b.GetParticleFluidProperties())
    util.assert_fields_close("stress", st, b.GetParticleStress())


def test_coupled_and_implicit_paths_match_committed_golden_vectors(gpu):
    """tests/golden/object_drop_10steps.npz and isph_tiny_2steps.npz (generated from the oracle) reproduced by the HIP path without
    the oracle in the loop: fields, wrench history and body trajectory of the cylinder drop; fields and sweep counts of I2SPH."""
    import os
    import test_oracle_cpu as T
    here = os.path.dirname(os.path.abspath(__file__))
    for name, got in T._coupled_and_implicit_goldens(gpu).items():
        g = np.load(os.path.join(here, "golden", name + ".npz"))
        for k in g.files:
            np.testing.assert_array_equal(got[k], g[k], err_msg=name + ":" + k)


def test_wavetank_piston_matches_oracle(gpu, oracle):
    """ChFsiProblemWavetank scene (ramp beach, lateral periodicity, piston wavemaker in prescribed motion): HIP vs oracle, every
    field and the wrench on the plate, bit for bit."""
    from chrono_wisc_amd import fsi, problem
    d0, box, depth = 0.05, (1.6, 0.3, 0.6), 0.3

    def run(api):
        pb = problem.ChFsiProblemWavetank(d0, api)
        pb.SetGravitationalAcceleration((0, 0, -9.8))
        pb.SetStepSizeCFD(1e-4)
        fp = fsi.FluidProperties()
        fp.density, fp.viscosity = 1000, 1
        pb.SetCfdSPH(fp)
        sp = fsi.SPHParameters()
        sp.initial_spacing, sp.d0_multiplier, sp.num_bce_layers, sp.max_velocity = d0, 1, 3, 4.0
        sp.integration_scheme = fsi.IntegrationScheme.RK2
        sp.shifting_method, sp.shifting_xsph_eps = fsi.ShiftingMethod.XSPH, 0.5
        pb.SetSPHParameters(sp)
        pb.RegisterParticlePropertiesCallback(problem.depth_pressure_properties(depth))
        pb.SetLateralPeriodicBC(True)
        pb.SetProfile(problem.WaveTankRampBeach(0.8, 0.25), True)
        pb.ConstructWaveTank(problem.WavemakerType.PISTON, (0, 0, 0), box, depth, lambda t: 0.05 * np.sin(2 * np.pi * 5.0 * t))
        f = pb.Initialize()
        wr = []
        for _ in range(40):
            f.DoStepDynamics(1e-4)
            wr.append(np.concatenate(pb.sysSPH.GetBodyWrenches()[0]))
        return util.snapshot(pb.sysSPH), np.array(wr)
    a, wa = run(gpu)
    b, wb = run(oracle)
    for k in ("pos", "vel", "rpm", "deriv"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(wa, wb)
    assert np.abs(wb).max() > 0


def test_marker_device_view_is_zero_copy(gpu):
    """GetMarkerDeviceView: the borrowed device pointers wrapped as torch tensors (CUDA array interface) show the same state the
    host getters return, without a copy through the host."""
    import torch
    s = scenes.dam_break_named("tiny", gpu)
    for _ in range(3):
        s.DoStepDynamics(1e-4)
    pos = s.GetParticlePositions()  # materialises marker order and synchronises
    v = s.GetMarkerDeviceView()
    assert v["n"] == s.GetCounters().numAllMarkers and v["pos4"] and v["vel4"] and v["rpm4"]

    class View:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n, 4), "typestr": "<f4", "data": (int(ptr), False), "version": 2}  # torch refuses the read-only flag; the test only reads
    dp = torch.as_tensor(View(v["pos4"], v["n"]), device="cuda:0")
    dr = torch.as_tensor(View(v["rpm4"], v["n"]), device="cuda:0")
    np.testing.assert_array_equal(dp[:, :3].cpu().numpy(), pos)
    np.testing.assert_array_equal(dr.cpu().numpy(), s.solver.rho_pres_mu_type())


# ---- F4: density re-initialisation (phys/SphForceWCSPH.cu:559-563 counter, :584-646 calcRho_kernel) ----------------------
@pytest.mark.parametrize("steps,scheme", [(0, fsi.IntegrationScheme.RK2), (0, fsi.IntegrationScheme.EULER), (2, fsi.IntegrationScheme.RK2)])
def test_density_reinitialisation(gpu, oracle, steps, scheme):
    """k_density_reinit against the oracle's calcRho_kernel, bit for bit: density_reinit_steps = 0 is the only setting under which
    the reference's kernel writes (its `density_reinit == 0` condition reads the call counter, :636), and only on the first force
    evaluation; a positive period runs the pass on schedule without changing the state.  The CPU-side meaning of the pass (Shepard
    sum over fluid neighbours) is pinned in tests/test_oracle_cpu.py."""
    def tweak(sp):
        sp.density_reinit_steps = steps
    a, b, rep = _run_pair(gpu, oracle, "tiny", 4, scheme=scheme, tweak=tweak)
    assert all(v[0] == 0.0 for v in rep.values()), rep
    plain = scenes.dam_break_named("tiny", gpu, scheme=scheme)
    for _ in range(4):
        plain.DoStepDynamics(1e-4)
    same = np.array_equal(util.snapshot(a)["rpm"].view(np.uint32), util.snapshot(plain)["rpm"].view(np.uint32))
    assert same == (steps != 0)  # steps = 0 really changed the run; a positive period did not


def test_density_reinitialisation_first_evaluation_is_the_shepard_sum(gpu, oracle):
    """One explicit Euler step of length 0 leaves exactly the re-initialised densities in the state: GPU == oracle bit for bit, and
    different from the initial densities."""
    def tweak(sp):
        sp.density_reinit_steps = 0
    a, b = util.make_pair(gpu, oracle, "tiny", scheme=fsi.IntegrationScheme.EULER, tweak=tweak)
    rho0 = a.solver.rho_pres_mu_type()[:, 0].copy()
    a.DoStepDynamics(0.0)
    b.DoStepDynamics(0.0)
    util.compare_state(a, b, rtol=0.0)
    fluid = a.solver.rho_pres_mu_type()[:, 3] < -0.5
    assert np.mean(a.solver.rho_pres_mu_type()[fluid, 0] != rho0[fluid]) > 0.5


def _write_object_drop_scene(path, s, fsys, d0, step, variable_dt):
    scenes.write_object_drop_scene(path, s, fsys, d0, step, variable_dt)


@pytest.mark.parametrize("interface", ["sph", "generic"])
def test_cxx_cosimulation_shell_object_drop(gpu, tmp_path, interface):
    """the C++ twin of ChFsiSystem (host/ChFsiSystem_MI: multibody step on its own thread beside the fluid step, exchange
    hooks after the join, timers / RTF, generic and SPH interfaces) against the Python driver of the same exchange order
    (rigid.FsiSystem) on the same scene and library: body trajectory and fluid wrench agree to rounding of the two 6-DoF
    integrators (double precision), and through them every fluid step saw the same body states."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "chrono-wisc_amd", "host", "demo_FSI-SPH_ObjectDrop_MI")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    d0, step, nsteps = 0.05, 1e-4, 12
    s0, f0 = scenes.object_drop(gpu, initial_spacing=d0, variable_dt=False, step_size=step, initial_height=0.9, initialize=False)
    scene = str(tmp_path / "drop.scn")
    _write_object_drop_scene(scene, s0, f0, d0, step, False)
    dump = str(tmp_path / "drop.bin")
    cmd = [exe, "--scene", scene, "--steps", str(nsteps), "--dump", dump] + (["--generic"] if interface == "generic" else [])
    r = subprocess.run(cmd, check=True, timeout=600, capture_output=True, text=True)
    assert "mean RTF" in r.stdout and ("interface " + ("generic" if interface == "generic" else "SPH")) in r.stdout
    rec = np.fromfile(dump, dtype=np.float64).reshape(nsteps, 16)
    s, fs = scenes.object_drop(gpu, initial_spacing=d0, variable_dt=False, step_size=step, initial_height=0.9)
    for k in range(nsteps):
        fs.DoStepDynamics(step)
        b = fs.bodies[0]
        np.testing.assert_allclose(rec[k, 0:3], b.pos, rtol=0, atol=1e-12)
        np.testing.assert_allclose(rec[k, 3:7], b.rot, rtol=0, atol=1e-12)
        np.testing.assert_allclose(rec[k, 7:10], b.lin_vel, rtol=0, atol=1e-11)
        fscale = max(float(np.abs(b.force).max()), 1e-9)
        np.testing.assert_allclose(rec[k, 10:13], b.force, rtol=0, atol=1e-6 * fscale)
        np.testing.assert_allclose(rec[k, 13:16], b.torque, rtol=0, atol=1e-6 * fscale)
    assert abs(rec[-1, 12]) > 0  # the partly submerged body does feel the fluid


@pytest.mark.parametrize("interface", ["sph", "generic"])
def test_cxx_cosimulation_shell_against_the_oracle_fixture(gpu, tmp_path, interface):
    """SURVEY 8f-3 against the ORACLE, not against another driver of the same library: tests/golden/object_drop_10steps.npz holds the wrench history,
    the body trajectory and the fields the CPU oracle produced for this very scene (cylinder started partly submerged, 10 steps of 2e-4, free body on
    the 6-DoF integrator; tests/golden/make_golden.py).  The C++ shell (host/ChFsiSystem_MI: persistent multibody worker beside the fluid step,
    exchange hooks on the caller's thread) drives the HIP library in SPH_ARITH_EXACT through the same 10 steps: every step's fluid force and torque and
    the body's final position, attitude and velocity must match the fixture -- the wrenches to rounding of the two double-precision 6-DoF integrators
    feeding float body states, the trajectory to 1e-12."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "chrono-wisc_amd", "host", "demo_FSI-SPH_ObjectDrop_MI")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    fx = np.load(os.path.join(root, "tests", "golden", "object_drop_10steps.npz"))
    d0, step, nsteps = 0.05, 2e-4, 10
    s0, f0 = scenes.object_drop(gpu, initial_spacing=d0, variable_dt=False, initial_height=0.9, initialize=False)
    scene = str(tmp_path / "drop.scn")
    _write_object_drop_scene(scene, s0, f0, d0, step, False)
    dump = str(tmp_path / "drop.bin")
    env = dict(os.environ, CHSPH_ARITH="0")  # the oracle's bits
    cmd = [exe, "--scene", scene, "--steps", str(nsteps), "--dump", dump] + (["--generic"] if interface == "generic" else [])
    subprocess.run(cmd, check=True, timeout=600, capture_output=True, text=True, env=env)
    rec = np.fromfile(dump, dtype=np.float64).reshape(nsteps, 16)
    wr = fx["wrench"]
    assert wr.shape == (nsteps, 6) and np.abs(wr).max() > 0
    scale = float(np.abs(wr).max())
    np.testing.assert_allclose(rec[:, 10:16], wr, rtol=0, atol=1e-6 * scale, err_msg="fluid wrench history, C++ shell on HIP vs oracle fixture")
    np.testing.assert_allclose(rec[-1, 0:3], fx["body_pos"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(rec[-1, 3:7], fx["body_rot"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(rec[-1, 7:10], fx["body_vel"], rtol=0, atol=1e-11)


def test_device_kernel_functions_equal_the_reference_headers_bit_for_bit(gpu):
    """Row K on the device against the reference's own numbers: W3h and GradW3h of all four kernel types as the HIP kernels evaluate them
    (sph_debug_kernel_functions: one thread per sample, sph_device.h's functions) must reproduce tests/golden/kernel_functions.npz -- values the
    reference's `__host__ __device__` functions (sph/physics/SphGeneral.cuh:76-229) produced for 3 kernel lengths x 96 radii x 3 directions
    (tests/golden/make_kernel_golden.sh) -- bit for bit.  The same fixture pins the oracle (tests/test_oracle_cpu.py)."""
    import ctypes as C
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernel_functions.npz"))
    s = scenes.dam_break_named("tiny", gpu)
    ctx = s.solver.ctx
    for t in range(4):
        for ih, h in enumerate(fx["h"]):
            invh = np.float32(1.0) / np.float32(h)
            r = np.ascontiguousarray(np.repeat(fx["r"][t, ih], 3), dtype=np.float32)          # one W sample per (radius, direction)
            d3 = np.ascontiguousarray(fx["d"][t, ih].reshape(-1, 3), dtype=np.float32)
            w = np.empty(r.size, dtype=np.float32)
            g = np.empty((r.size, 3), dtype=np.float32)
            s.solver._ck(gpu.debug_kernel_functions(ctx, t, r.size, r.ctypes.data_as(C.c_void_p), d3.ctypes.data_as(C.c_void_p), C.c_float(float(invh)),
                                                    w.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p)))
            assert w.tobytes() == np.repeat(fx["W"][t, ih], 3).astype(np.float32).tobytes(), "W3h, kernel %d, h %g" % (t, h)
            assert g.tobytes() == np.ascontiguousarray(fx["gradW"][t, ih].reshape(-1, 3)).tobytes(), "GradW3h, kernel %d, h %g" % (t, h)


# ---- neighbour search at the reference's default d0_multiplier: the two-word form of the LDS-staged search -------------------------------
def _free_block(api, n, multiplier, jitter):
    """n^3 fluid markers off their lattice (uniform jitter, in units of the spacing) in a domain without walls or periodic sides."""
    d0 = 0.01
    s = fsi.ChFsiFluidSystemSPH(api)
    s.SetStepSize(1e-5)
    fp = fsi.FluidProperties()
    fp.density, fp.viscosity = 1000, 1
    s.SetCfdSPH(fp)
    s.SetGravitationalAcceleration((0, 0, 0))
    sp = fsi.SPHParameters()
    sp.integration_scheme = fsi.IntegrationScheme.RK2
    sp.initial_spacing = d0
    sp.d0_multiplier = multiplier
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
    M = L / 2 + 20 * d0
    s.SetComputationalDomain((-M,) * 3, (M,) * 3, fsi.BC_NONE)
    ax = -L / 2 + d0 * (np.arange(n) + 0.5)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    pts = pts + np.random.default_rng(7).uniform(-jitter * d0, jitter * d0, pts.shape)
    rpm = np.stack([np.full(len(pts), 1000.0), np.zeros(len(pts)), np.full(len(pts), 1.0)], axis=1)
    s.AddSPHParticles(pts, rpm, np.zeros_like(pts))
    s.Initialize()
    return s


def test_staged_search_with_two_mask_words_at_the_default_multiplier(gpu, oracle, monkeypatch):
    """d0_multiplier 1.2 (the reference's parameter default, sph/ChFsiFluidSystemSPH.cpp:81-176): 13.8 markers per cell, ~41 candidates per run of three
    cells -- beyond the 29 one mask word covers.  k_neighbors_lds<.., MW = 2> scans 16 steps of four candidates into two words per run.  On a disordered
    block most waves take it (on the lattice itself a (3, 3)-marker line of cells carries 63-72 candidates per run and every wave falls back): lists
    bit for bit the oracle's (neighborSearchNum / neighborSearchID, phys/SphCollisionSystem.cu:219-306) and those of the row-per-lane kernel."""
    monkeypatch.setenv("CHSPH_WAVE_ORDER", "1")  # records a class per wave: sph_debug_search_classes
    a, b = _free_block(gpu, 26, 1.2, 0.3), _free_block(oracle, 26, 1.2, 0.3)
    for _ in range(3):
        a.DoStepDynamics(1e-5)
        b.DoStepDynamics(1e-5)
    offs, idx = util.compare_csr(a, b)
    util.compare_state(a, b)
    st, fb = C.c_uint32(), C.c_uint32()
    a.solver._ck(gpu.debug_search_classes(a.solver.ctx, C.byref(st), C.byref(fb)))
    assert st.value > fb.value and st.value > 100 and np.diff(offs).max() > 64, (st.value, fb.value)  # most waves are staged (a block this small is mostly surface); long rows exist
    monkeypatch.setenv("CHSPH_SEARCH_LDS", "0")
    c = _free_block(gpu, 26, 1.2, 0.3)
    for _ in range(3):
        c.DoStepDynamics(1e-5)
    o2, i2 = c.solver.neighbor_csr()
    np.testing.assert_array_equal(offs, o2)
    np.testing.assert_array_equal(idx, i2)
    monkeypatch.delenv("CHSPH_SEARCH_LDS")
    lat = _free_block(gpu, 26, 1.2, 0.0)  # the lattice itself: every wave falls back, same lists as the oracle
    lat_ref = _free_block(oracle, 26, 1.2, 0.0)
    lat.DoStepDynamics(1e-5)
    lat_ref.DoStepDynamics(1e-5)
    util.compare_csr(lat, lat_ref)


def test_long_periodic_channel_mixes_tile_and_cell_by_cell_waves(gpu, oracle, monkeypatch):
    """A periodic x axis on one device: rows of the first / last cell column have their three x cells in two places of the sorted order, so waves that
    touch those columns walk their rows cell by cell (neighbors_row_cellwise) while all other waves use the tile kernels.  The Poiseuille scene stretched
    to 80 cell columns (640 rows per line of cells: ten waves, two of them at the seam), after 4 steps with the flow carrying markers across the seam:
    lists and every field bit for bit the oracle's (calcHashD / neighborSearchNum / neighborSearchID with a periodic x, phys/SphCollisionSystem.cu:60-306),
    and the lists of the cell-by-cell kernel for every row (CHSPH_NO_XPERIODIC_TILES=1)."""
    a = scenes.poiseuille(gpu, bxDim=1.6, t_start=4.0)
    b = scenes.poiseuille(oracle, bxDim=1.6, t_start=4.0)
    assert int(a.DerivedParams().gridSize[0]) >= 64
    for _ in range(4):
        a.DoStepDynamics(scenes.POISEUILLE["dt"])
        b.DoStepDynamics(scenes.POISEUILLE["dt"])
    offs, idx = util.compare_csr(a, b)
    util.compare_state(a, b)
    monkeypatch.setenv("CHSPH_NO_XPERIODIC_TILES", "1")
    c = scenes.poiseuille(gpu, bxDim=1.6, t_start=4.0)
    for _ in range(4):
        c.DoStepDynamics(scenes.POISEUILLE["dt"])
    o2, i2 = c.solver.neighbor_csr()
    np.testing.assert_array_equal(offs, o2)
    np.testing.assert_array_equal(idx, i2)



@pytest.mark.parametrize("switch", ["CHSPH_NO_SCAN_SINGLE=1", "CHSPH_SCAN_PREFIX_MIN=0", "CHSPH_EAGER_MAP=1", "CHSPH_EAGER_SXT=1", "CHSPH_NO_PACKED_YWRAP=1",
                                    "CHSPH_NO_SEARCH_SOA=1", "CHSPH_TILE32=1"])
def test_resort_and_search_alternatives_give_the_same_bits(gpu, oracle, monkeypatch, switch):
    """The sorted-resident re-sort (calcHashD + sort_by_key + findCellStartEndD + reorderDataD, phys/SphCollisionSystem.cu:62-215,347-373) in its
    measured alternatives: three scan launches instead of k_scan_window, the one-block prefix of the block totals (large windows), mapO2S and the
    (x, y, z, type) copy written although nothing reads them -- and the neighbour kernels that DO read that copy after re-sorts that left it out
    (k_build_sxt) -- and the scalar scan of the waves in the first / last y layer: the C1 dam break (y periodic) after 10 RK2 steps, lists and
    every field bit for bit the oracle's in each of them."""
    name, val = switch.split("=")
    monkeypatch.setenv(name, val)
    _a, _b, rep = _run_pair(gpu, oracle, "C1", 10)
    assert rep


def _two_clusters(api, half=1.0):
    """Two 3 x 3 x 3 clusters of fluid markers at opposite corners of a cube of 2 * half metres without walls: far more 1024-cell blocks than markers."""
    d0 = 0.01
    s = fsi.ChFsiFluidSystemSPH(api)
    s.SetStepSize(1e-5)
    fp = fsi.FluidProperties()
    fp.density, fp.viscosity = 1000, 1
    s.SetCfdSPH(fp)
    s.SetGravitationalAcceleration((0, 0, -9.8))
    sp = fsi.SPHParameters()
    sp.integration_scheme = fsi.IntegrationScheme.RK2
    sp.initial_spacing = d0
    sp.d0_multiplier = 1.0
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
    s.SetComputationalDomain((-half,) * 3, (half,) * 3, fsi.BC_NONE)
    ax = d0 * (np.arange(3) - 1.0)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    one = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    off = half - 0.1
    pts = np.concatenate([one - off, one + off])
    rpm = np.stack([np.full(len(pts), 1000.0), np.zeros(len(pts)), np.full(len(pts), 1.0)], axis=1)
    vel = np.zeros_like(pts)
    vel[:, 0] = np.where(np.arange(len(pts)) < len(one), 0.5, -0.5)  # the clusters drift towards each other: cells change every few steps
    s.AddSPHParticles(pts, rpm, vel)
    s.Initialize()
    return s


@pytest.mark.parametrize("window", ["window", "whole_grid", "prefix_pass"])
def test_sparse_scene_with_more_cell_blocks_than_markers(gpu, oracle, monkeypatch, window):
    """54 markers in a grid of 10^6 cells (977 blocks of 1024 cells, the markers near the first and the last of them): k_scan_window's block totals are
    cleared by a launch that has fewer threads (256) than there are blocks once the whole grid is scanned, and the totals in front of the second cluster
    are all zero.  20 RK2 steps with the clusters drifting through the cells, order, lists and fields bit for bit the oracle's (calcHashD / findCellStartEndD /
    reorderDataD, phys/SphCollisionSystem.cu:62-215) -- with the cell window, on the whole grid, and with the one-block prefix pass."""
    if window != "window":
        monkeypatch.setenv("CHSPH_NO_CELL_WINDOW", "1")
    if window == "prefix_pass":
        monkeypatch.setenv("CHSPH_SCAN_PREFIX_MIN", "0")
    a, b = _two_clusters(gpu), _two_clusters(oracle)
    assert a.GetCounters().numAllMarkers == 54
    for k in range(20):
        a.DoStepDynamics(1e-4)
        b.DoStepDynamics(1e-4)
        if k % 5 == 4:
            util.compare_csr(a, b)
    util.compare_csr(a, b)
    util.compare_state(a, b)
    assert a.GetCounters().numExtendedParticles == 54

"""GPU parity for the I2SPH path: HIP through the C ABI against the CPU oracle.

Scenes: the reference's only I2SPH demo (Couette flow, demo_FSI-SPH_CouetteFlow.cpp, two FSI bodies with a
prescribed spin) at reduced size, and the dam-break input run with the implicit scheme.  Bar: neighbour lists
bit-exact, float fields within 1e-5 relative, identical Jacobi sweep counts."""
import numpy as np
import pytest

from chrono_wisc_amd import fsi, scenes

import util

pytestmark = pytest.mark.gpu


def _compare_wrenches(a, b):
    for (fa, ta), (fb, tb) in zip(a.GetBodyWrenches(), b.GetBodyWrenches()):
        for x, y in ((fa, fb), (ta, tb)):
            scale = max(np.abs(y).max(), 1e-20)
            assert np.abs(x - y).max() <= 1e-5 * scale


@pytest.mark.parametrize("dbp", [True, False])
def test_couette_small(gpu, oracle, dbp):
    def tweak(sp):
        sp.use_density_based_projection = dbp

    kw = dict(scale=0.4, max_num_iters=40, tweak=tweak)
    a, b = scenes.couette(gpu, **kw), scenes.couette(oracle, **kw)
    util.compare_params(a.GetParams(), b.GetParams())
    assert a.GetCounters().numRigidMarkers > 0 and a.GetCounters().numFluidMarkers > 1000
    rep = {}
    for k in range(4):
        scenes.couette_step(a)
        scenes.couette_step(b)
        assert a.solver.isph_iterations() == b.solver.isph_iterations()
        _compare_wrenches(a, b)
    util.compare_csr(a, b)
    util.compare_state(a, b, report=rep)
    print("couette dbp=%s:" % dbp, rep, a.solver.isph_iterations())


def _isph_dam_break(api, atol, max_iters, clamp=False, dt=5e-4):
    def tweak(sp):
        sp.shifting_xsph_eps = 0.1
        sp.shifting_beta_implicit = 1.0
        sp.min_distance_coefficient = 0.01

    s = scenes.dam_break(api, s=0.25, d0=0.05, dt=dt, scheme=fsi.IntegrationScheme.IMPLICIT_SPH, tweak=tweak, initialize=False)
    lp = fsi.LinSolverParameters()
    lp.atol = atol
    lp.max_num_iters = max_iters
    s.SetLinSolverParameters(lp)
    s.p.ClampPressure = 1 if clamp else 0
    s.Initialize()
    return s


@pytest.mark.parametrize("atol,max_iters,clamp", [(0.0, 25, False), (0.03, 100, False), (0.0, 10, True), (1e30, 50, False), (0.0, 0, False)])
def test_dam_break_implicit(gpu, oracle, atol, max_iters, clamp):
    """Periodic y axis, implicit shifting (beta 1), the device-side convergence gate (early exit on atol, the 3-sweep
    minimum, Max_Iter 0) and the ClampPressure post-processing."""
    a, b = _isph_dam_break(gpu, atol, max_iters, clamp), _isph_dam_break(oracle, atol, max_iters, clamp)
    rep = {}
    for k in range(3):
        a.DoStepDynamics(5e-4)
        b.DoStepDynamics(5e-4)
        assert a.solver.isph_iterations() == b.solver.isph_iterations(), k
    util.compare_csr(a, b)
    util.compare_state(a, b, report=rep)
    it = a.solver.isph_iterations()
    print("dam break I2SPH atol=%g max=%d clamp=%s:" % (atol, max_iters, clamp), rep, it)
    if atol >= 1e30:
        assert it[1] == 3
    if max_iters == 0:
        assert it == (0, 0)


def test_rejects_unreachable_operator_settings(gpu):
    s = scenes.dam_break(gpu, s=0.25, d0=0.05, scheme=fsi.IntegrationScheme.IMPLICIT_SPH, initialize=False)
    s.p.Conservative_Form = 0
    with pytest.raises(Exception, match="I2SPH"):
        s.Initialize()


def test_cxx_facade_couette_matches_oracle(oracle, tmp_path):
    """The C++ host mirror builds the Couette demo itself (annulus generators, plate, two FSI bodies, linear-solver
    parameters) and drives the exchange hooks; it must land on the state of the Python mirror driving the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "chrono-wisc_amd", "host", "demo_FSI-SPH_CouetteFlow_MI")
    subprocess.check_call(["make", "-C", os.path.dirname(exe)], stdout=subprocess.DEVNULL)  # no-op when up to date; the ABI structs may have changed
    dump = str(tmp_path / "couette.bin")
    out_dir = tmp_path / "csv"
    out_dir.mkdir()
    subprocess.run([exe, "--scale", "0.4", "--iters", "30", "--steps", "3", "--dump", dump, "--out", str(out_dir)], check=True, timeout=600)
    assert sorted(os.listdir(out_dir)) == ["FSI_body0.csv", "FSI_body1.csv", "boundary0.csv", "fluid0.csv", "fluid1.csv", "fluid2.csv", "rigidBCE0.csv",
                                           "rigidBCE1.csv", "rigidBCE2.csv"]
    assert open(out_dir / "fluid2.csv").readline().strip() == "x,y,z,v_x,v_y,v_z,|U|,acc,rho,pressure"
    assert len(open(out_dir / "FSI_body1.csv").read().splitlines()) == 4
    raw = np.fromfile(dump, dtype=np.uint8)
    n = int(raw[:8].view(np.uint64)[0])
    f = raw[8:].view(np.float32).reshape(3, n, 3)
    b = scenes.couette(oracle, scale=0.4, max_num_iters=30)
    assert n == b.GetCounters().numAllMarkers
    for _ in range(3):
        scenes.couette_step(b)
    util.assert_fields_close("pos", f[0], b.GetParticlePositions())
    util.assert_fields_close("vel", f[1], b.GetParticleVelocities())
    util.assert_fields_close("rho,p,mu", f[2], b.GetParticleFluidProperties())


def test_isph_g_tensor_and_normals_match_oracle(gpu, oracle):
    """J2, calcNormalizedRho_Gi_fillInMatrixIndices (phys/SphGeneral.cu:289-391): G tensor and normals of the last PreProcessor call,
    bit for bit; switching the pass on changes nothing else in the step."""
    from chrono_wisc_amd import scenes
    a = scenes.dam_break_implicit(gpu, "tiny", max_num_iters=30)
    b = scenes.dam_break_implicit(oracle, "tiny", max_num_iters=30)
    plain = scenes.dam_break_implicit(gpu, "tiny", max_num_iters=30)
    for s in (a, b):
        s.solver.set_isph_diagnostics(True)
    for _ in range(2):
        for s in (a, b, plain):
            s.DoStepDynamics(5e-4)
    Ga, na = a.solver.isph_g_tensor()
    Gb, nb = b.solver.isph_g_tensor()
    np.testing.assert_array_equal(Ga.view(np.uint32), Gb.view(np.uint32))
    np.testing.assert_array_equal(na.view(np.uint32), nb.view(np.uint32))
    assert np.abs(Ga - np.eye(3, dtype=np.float32)).max() > 1e-3  # not the identity fallback everywhere
    util.compare_state(a, b, rtol=0.0)
    util.compare_state(a, plain, rtol=0.0)

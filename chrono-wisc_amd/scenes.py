"""Inputs of the reference's own demos, rebuilt call by call through the facade mirror.

dam_break(s, d0, dt) is src/demos/fsi/sph/demo_FSI-SPH_DamBreak.cpp:100-192 with every length
scaled by `s` at fixed spacing: s=1,d0=0.1 is the demo verbatim (C1, 30 327
markers); s=0.4,d0=0.01 is the "1 M" configuration C2 (1 234 797 markers); s=1,d0=0.01 is C3.
Deterministic; no RNG.
"""
import numpy as np

from . import fsi

# label -> (s, d0, dt) ; SURVEY.md 8d / BASELINE.md 2
DAM_BREAK_CONFIGS = {
    "C1": (1.0, 0.1, 1e-4),
    "tiny": (0.25, 0.05, 1e-4),
    "C2": (0.4, 0.01, 1e-5),
    "4M": (0.63, 0.01, 1e-5),
    "C5geom": (0.794, 0.01, 1e-4),
    "C3": (1.0, 0.01, 1e-5),
    "64M": (1.5874, 0.01, 1e-5),
}


def dam_break(api=None, s=1.0, d0=0.1, dt=1e-4, ps_freq=1, scheme=fsi.IntegrationScheme.RK2, device=-1, initialize=True,
              check_errors=True, tweak=None):
    """Returns an (optionally initialized) ChFsiFluidSystemSPH set up like the dam-break demo."""
    bxDim, byDim, bzDim = 12.0 * s, 1.0 * s, 8.0 * s
    fxDim, fyDim, fzDim = 4.0 * s, 1.0 * s, 4.0 * s

    sysSPH = fsi.ChFsiFluidSystemSPH(api, device=device)
    sysSPH.EnableErrorChecks(check_errors)
    sysSPH.SetStepSize(dt)  # sysFSI.SetStepSizeCFD

    fluid_props = fsi.FluidProperties()
    fluid_props.density = 1000
    fluid_props.viscosity = 5
    sysSPH.SetCfdSPH(fluid_props)
    sysSPH.SetGravitationalAcceleration((0, 0, -9.8))

    sp = fsi.SPHParameters()
    sp.integration_scheme = scheme
    sp.initial_spacing = d0
    sp.d0_multiplier = 1
    sp.max_velocity = 10.0
    sp.shifting_method = fsi.ShiftingMethod.XSPH
    sp.shifting_xsph_eps = 0.5
    sp.artificial_viscosity = 0.03
    sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL
    sp.eos_type = fsi.EosType.TAIT
    sp.use_consistent_gradient_discretization = False
    sp.use_consistent_laplacian_discretization = False
    sp.num_proximity_search_steps = ps_freq
    sp.use_delta_sph = True
    sp.delta_sph_coefficient = 0.1
    sp.boundary_method = fsi.BoundaryMethod.ADAMI
    if tweak:
        tweak(sp)
    sysSPH.SetSPHParameters(sp)
    sysSPH.SetNumProximitySearchSteps(ps_freq)
    if not tweak:
        sysSPH.SetShiftingMethod(fsi.ShiftingMethod.XSPH)
        sysSPH.SetShiftingXSPHParameters(0.5)

    initSpace0 = sysSPH.GetInitialSpacing()
    cMin = (-bxDim / 2 - 10 * initSpace0, -byDim / 2 - initSpace0 / 2, -2 * bzDim)
    cMax = (+bxDim / 2 + 10 * initSpace0, +byDim / 2 + initSpace0 / 2, +2 * bzDim)
    sysSPH.SetComputationalDomain(cMin, cMax, fsi.BC_Y_PERIODIC)

    boxCenter = (-bxDim / 2 + fxDim / 2, 0.0, fzDim / 2)
    boxHalfDim = (fxDim / 2 - initSpace0, fyDim / 2, fzDim / 2 - initSpace0)
    points = fsi.grid_sampler_box(initSpace0, boxCenter, boxHalfDim)

    gz = abs(sysSPH.GetGravitationalAcceleration()[2])
    rho0 = sysSPH.GetDensity()
    cs = sysSPH.GetSoundSpeed()
    pre_ini = rho0 * gz * (-points[:, 2] + fzDim)
    rho_ini = rho0 + pre_ini / (cs * cs)
    rpm = np.stack([rho_ini, pre_ini, np.full_like(pre_ini, sysSPH.GetViscosity())], axis=1)
    sysSPH.AddSPHParticles(points, rpm)

    ground_bce = sysSPH.CreatePointsBoxContainer((bxDim, byDim, bzDim), (2, 0, 2))
    sysSPH.AddBCEBoundary(ground_bce, (0, 0, bzDim / 2), fsi.QUNIT)

    if initialize:
        sysSPH.Initialize()
    return sysSPH


def dam_break_named(label, api=None, **kw):
    s, d0, dt = DAM_BREAK_CONFIGS[label]
    return dam_break(api, s=s, d0=d0, dt=dt, **kw)


def dam_break_implicit(api=None, label="tiny", max_num_iters=200, atol=0.0, dt=None, beta_implicit=0.0, device=-1, initialize=True, check_errors=True,
                       ps_freq=1):
    """The dam-break input advanced with I2SPH under the solver settings of the reference's I2SPH demo
    (demo_FSI-SPH_CouetteFlow.cpp:139-159; SURVEY.md 8d, C5 = label "C5geom").  dt defaults to 5e-4 * d0 / 0.05."""
    sc, d0, _ = DAM_BREAK_CONFIGS[label]
    dt = 5e-4 * d0 / 0.05 if dt is None else dt

    def tweak(sp):
        sp.shifting_xsph_eps = 0.1
        sp.shifting_beta_implicit = beta_implicit
        sp.min_distance_coefficient = 0.001
        sp.use_density_based_projection = True

    s = dam_break(api, s=sc, d0=d0, dt=dt, ps_freq=ps_freq, scheme=fsi.IntegrationScheme.IMPLICIT_SPH, tweak=tweak, initialize=False, device=device,
                  check_errors=check_errors)
    lp = fsi.LinSolverParameters()
    lp.atol = atol
    lp.max_num_iters = max_num_iters
    s.SetLinSolverParameters(lp)
    if initialize:
        s.Initialize()
    return s


# ---- transient plane Poiseuille flow: the reference's only numeric acceptance test ---------------------------
# src/tests/unit_tests/fsi/sph/utest_FSI-SPH_Poiseuille_flow.cpp (marker grid of ChFsiProblemCartesian::Construct +
# AddBoxContainer, sph/ChFsiProblemSPH.cpp:712-749,937-994)
POISEUILLE = dict(bxDim=0.2, byDim=0.1, bzDim=0.2, t_start=1.0, dt=2e-3, num_steps=500, force=0.05, initial_spacing=0.01,
                  v_tolerance=5e-3, d_tolerance=1e-4)


def poiseuille_analytical(Z, H, time, nu, F, initial_spacing):
    """PoiseuilleAnalytical (utest:61-80): 50-term series; the no-slip plane sits half a spacing inside the BCE layer."""
    Z = np.asarray(Z, dtype=np.float64)
    H = H - initial_spacing
    Z = Z - 0.5 * initial_spacing
    v = 1.0 / (2.0 * nu) * F * Z * (H - Z)
    for n in range(50):
        v = v - 4.0 * F * H ** 2 / (nu * np.pi ** 3 * (2 * n + 1) ** 3) * np.sin(np.pi * Z * (2 * n + 1) / H) * \
            np.exp(-(2 * n + 1) ** 2 * np.pi ** 2 * nu * time / H ** 2)
    return v


def poiseuille(api=None, device=-1, check_errors=True, consistent=True, bxDim=None, initialize=True, t_start=None):
    """The unit test's system (utest:106-190), initialized; markers are staged in grid order (the reference iterates an
    unordered_set, so its marker order is unspecified; the acceptance criterion does not depend on it).
    bxDim: another channel length along the periodic x axis (slab runs need >= 12 cell columns: a ring of three ranks);
    t_start: the analytical profile the markers start with."""
    c = dict(POISEUILLE)
    if bxDim is not None:
        c["bxDim"] = bxDim
    if t_start is not None:  # initial velocity profile of a later time (faster flow)
        c["t_start"] = t_start
    d0 = c["initial_spacing"]
    s = fsi.ChFsiFluidSystemSPH(api, device=device)
    s.EnableErrorChecks(check_errors)
    s.SetGravitationalAcceleration((0, 0, 0))
    fp = fsi.FluidProperties()
    fp.density = 1000
    fp.viscosity = 1
    s.SetCfdSPH(fp)
    s.SetBodyForce((c["force"], 0, 0))
    sp = fsi.SPHParameters()
    sp.integration_scheme = fsi.IntegrationScheme.RK2
    sp.num_bce_layers = 3
    sp.initial_spacing = d0
    sp.d0_multiplier = 1
    sp.max_velocity = 0.1
    sp.shifting_method = fsi.ShiftingMethod.NONE
    sp.density_reinit_steps = 10000
    sp.viscosity_method = fsi.ViscosityMethod.LAMINAR
    sp.use_delta_sph = False
    sp.eos_type = fsi.EosType.ISOTHERMAL
    sp.use_consistent_gradient_discretization = consistent
    sp.use_consistent_laplacian_discretization = consistent
    s.SetSPHParameters(sp)
    s.SetStepSize(c["dt"])
    bx, by, bz = c["bxDim"], c["byDim"], c["bzDim"]
    fsize = (bx, by, bz - 2 * d0)
    pos = (0.0, 0.0, d0)
    Nx, Ny, Nz = (int(np.floor(L / d0 + 0.5)) + 1 for L in fsize)
    off = (pos[0] - fsize[0] / 2, pos[1] - fsize[1] / 2, pos[2])
    Ix, Iy, Iz = np.meshgrid(np.arange(Nx), np.arange(Ny), np.arange(Nz), indexing="ij")
    sph_pts = np.stack([d0 * Ix.ravel() + off[0], d0 * Iy.ravel() + off[1], d0 * Iz.ravel() + off[2]], axis=1)
    layers = 3
    zb = np.concatenate([-np.arange(1, layers + 1), Nz - 1 + np.arange(1, layers + 1)])
    Bx, By, Bz = np.meshgrid(np.arange(Nx), np.arange(Ny), zb, indexing="ij")
    bce_pts = np.stack([d0 * Bx.ravel() + off[0], d0 * By.ravel() + off[1], d0 * Bz.ravel() + off[2]], axis=1)
    nu = s.GetViscosity() / s.GetDensity()
    vx = poiseuille_analytical(sph_pts[:, 2], bz, c["t_start"], nu, c["force"], d0)
    vel = np.stack([vx, np.zeros_like(vx), np.zeros_like(vx)], axis=1)
    rpm = np.tile(np.array([[s.GetDensity(), 0.0, s.GetViscosity()]]), (len(sph_pts), 1))
    s.AddSPHParticles(sph_pts, rpm, vel)
    s.AddBCEBoundary(bce_pts)
    c_min = (-bx / 2 - d0 / 2, -by / 2 - d0 / 2, -10 * d0)
    c_max = (+bx / 2 + d0 / 2, +by / 2 + d0 / 2, bz + 10 * d0)
    s.SetComputationalDomain(c_min, c_max, fsi.BC_ALL_PERIODIC)
    if initialize:
        s.Initialize()
    return s


def poiseuille_errors(s, time):
    """(velocity RMS error / max analytical velocity, (rho_max - rho_min) / rho0) over the SPH particles (utest:182-215)."""
    c = POISEUILLE
    n = s.GetCounters().numFluidMarkers
    pos = s.GetParticlePositions()[:n].astype(np.float64)
    v = s.GetParticleVelocities()[:n, 0].astype(np.float64)
    d = s.GetParticleFluidProperties()[:n, 0].astype(np.float64)
    nu = s.GetViscosity() / s.GetDensity()
    va = poiseuille_analytical(pos[:, 2], c["bzDim"], time, nu, c["force"], c["initial_spacing"])
    v_rel = np.sqrt(np.mean((v - va) ** 2)) / va.max()
    d_rel = (d.max() - d.min()) / s.GetDensity()
    return v_rel, d_rel


# ---- Couette flow between a fixed inner and a spinning outer cylinder: the reference's I2SPH demo -------------
# src/demos/fsi/sph/demo_FSI-SPH_CouetteFlow.cpp:96-230 (units: mm, g, s).  The multibody side of the demo only drives
# the outer cylinder with a constant-speed motor about the y axis, so its state is a closed form of time.
COUETTE = dict(step_size=5e-4, initial_spacing=0.15, density=0.001, viscosity=0.001, g=9810.0, omega=62.0 / 60,
               outer_radius=5.0, inner_radius=3.8, fluid_height=2.0, bxDim=14.0, byDim=10.0, bzDim=14.0)
Q_ROTATE_Z_TO_Y = fsi.quat_from_angle_axis(-np.pi / 2, (1, 0, 0))  # src/chrono/core/ChRotation.cpp: z axis -> y axis


def couette_body_states(time, omega=COUETTE["omega"]):
    """States of the two FSI bodies (inner cylinder fixed, outer cylinder spinning about y) at `time`."""
    inner = dict(pos=(0, 0, 0), rot=fsi.QUNIT)
    outer = dict(pos=(0, 0, 0), rot=fsi.quat_from_angle_axis(omega * time, (0, 1, 0)), ang_vel=(0, omega, 0))
    return [inner, outer]


def couette(api=None, device=-1, initial_spacing=None, max_num_iters=200, scale=1.0, check_errors=True, tweak=None, initialize=True, bc=None):
    """The demo's system; `scale` shrinks the radii/height at fixed spacing for quick parity cases.  bc: boundary conditions of the computational
    domain (default: the demo's BC_ALL_PERIODIC; the fluid never reaches the domain faces, so slab tests cut along a non-periodic x)."""
    c = COUETTE
    d0 = c["initial_spacing"] if initial_spacing is None else initial_spacing
    Ro, Ri, Hf = c["outer_radius"] * scale, c["inner_radius"] * scale + (1 - scale) * 0.0, c["fluid_height"] * scale
    Ri = Ro - (c["outer_radius"] - c["inner_radius"])  # keep the gap width
    Hc = 1.2 * Hf
    bx, by, bz = c["bxDim"] * scale, c["byDim"] * scale, c["bzDim"] * scale
    s = fsi.ChFsiFluidSystemSPH(api, device=device)
    s.EnableErrorChecks(check_errors)
    s.SetGravitationalAcceleration((0, -c["g"], 0))
    s.SetStepSize(c["step_size"])
    fp = fsi.FluidProperties()
    fp.density = c["density"]
    fp.viscosity = c["viscosity"]
    s.SetCfdSPH(fp)
    sp = fsi.SPHParameters()
    sp.integration_scheme = fsi.IntegrationScheme.IMPLICIT_SPH
    sp.num_bce_layers = 3
    sp.initial_spacing = d0
    sp.d0_multiplier = 1
    sp.max_velocity = 1
    sp.shifting_xsph_eps = 0.1
    sp.shifting_beta_implicit = 0.0
    sp.min_distance_coefficient = 0.001
    sp.use_density_based_projection = True
    sp.num_proximity_search_steps = 1
    if tweak:
        tweak(sp)
    s.SetSPHParameters(sp)
    lp = fsi.LinSolverParameters()
    lp.type = 0
    lp.atol = 0
    lp.rtol = 0
    lp.max_num_iters = max_num_iters
    s.SetLinSolverParameters(lp)
    s.SetContainerDim((bx, by, bz))
    s.SetComputationalDomain((-bx / 2 * 1.2, -by * 1.2, -bz / 2 * 1.2), (bx / 2 * 1.2, by * 1.2, bz / 2 * 1.2), fsi.BC_ALL_PERIODIC if bc is None else bc)

    pts = s.CreatePointsCylinderAnnulus(Ri + d0 / 2, Ro - d0 / 2, Hf, True)
    xyz = np.stack([pts[:, 0], pts[:, 2], -pts[:, 1]], axis=1)
    pressure = c["g"] * c["density"] * (Hf - xyz[:, 1])
    rpm = np.stack([np.full_like(pressure, c["density"]), pressure, np.full_like(pressure, c["viscosity"])], axis=1)
    s.AddSPHParticles(xyz, rpm)

    plate = s.CreatePointsPlate((Ro * 2.5, Ro * 2.5))
    s.AddBCEBoundary(plate, (0, -Hf / 2 - d0, 0), Q_ROTATE_Z_TO_Y)

    center = (0, Hc / 2 - Hf / 2, 0)
    states = couette_body_states(0.0)
    bce_inner = s.CreatePointsCylinderAnnulus(Ri - 3 * d0, Ri - d0 / 2, Hc, True)
    s.AddFsiBody(fsi.frame_transform(bce_inner, center, Q_ROTATE_Z_TO_Y), states[0])
    bce_outer = s.CreatePointsCylinderAnnulus(Ro + d0 / 2, Ro + 3 * d0, Hc, True)
    s.AddFsiBody(fsi.frame_transform(bce_outer, center, Q_ROTATE_Z_TO_Y), states[1])
    if initialize:
        s.Initialize()
    return s


def couette_step(s, step=None):
    """ChFsiSystem::DoStepDynamics (src/chrono_fsi/ChFsiSystem.cpp:264-303) for the prescribed-motion bodies: advance the
    fluid, reduce the wrenches, load the body states of the new time, move the BCE markers."""
    step = COUETTE["step_size"] if step is None else step
    s.DoStepDynamics(step)
    s.OnExchangeSolidForces()
    s.LoadSolidStates(couette_body_states(s.GetSimTime()))
    s.OnExchangeSolidStates()


# ---- cylinder drop (C4): src/demos/fsi/sph/demo_FSI-SPH_ObjectDrop.cpp with CYLINDER_PRIMITIVE ------------------------
# built like ChFsiProblemCartesian::Construct / AddBoxContainer / Initialize (sph/ChFsiProblemSPH.cpp:185-298,339-378,
# 712-749,937-1047).  The reference keeps grid points in an unordered_set, so its marker order is unspecified; here the
# markers are staged in grid order (Ix, Iy, Iz nested), which no result depends on.
OBJECT_DROP = dict(fsize=(0.8, 0.8, 1.0), initial_spacing=0.025, step_size=1e-4, radius=0.12, length=0.20, density=500.0,
                   initial_height=1.05, num_bce_layers=4, max_velocity=4.538)
Q_ROTATE_Z_TO_X = fsi.quat_from_angle_axis(np.pi / 2, (0, 1, 0))  # src/chrono/core/ChRotation.cpp: z axis -> x axis


def object_drop(api=None, device=-1, initial_spacing=None, variable_dt=True, step_size=None, check_errors=True, initialize=True, tweak=None,
                initial_height=None):
    """Returns (ChFsiFluidSystemSPH, rigid.FsiSystem) for the cylinder-drop demo at the given spacing
    (demo: 0.025; SURVEY.md 8d C4: 0.0054 for ~4 M fluid markers)."""
    from . import rigid
    c = OBJECT_DROP
    d0 = c["initial_spacing"] if initial_spacing is None else initial_spacing
    h0 = c["step_size"] if step_size is None else step_size
    fsize = c["fsize"]
    layers = c["num_bce_layers"]
    s = fsi.ChFsiFluidSystemSPH(api, device=device)
    s.EnableErrorChecks(check_errors)
    g = (0, 0, -9.8)
    s.SetGravitationalAcceleration(g)
    s.SetStepSize(h0)
    fp = fsi.FluidProperties()
    fp.density, fp.viscosity = 1000, 1
    s.SetCfdSPH(fp)
    sp = fsi.SPHParameters()
    sp.integration_scheme = fsi.IntegrationScheme.RK2
    sp.num_bce_layers = layers
    sp.initial_spacing = d0
    sp.d0_multiplier = 1
    sp.max_velocity = c["max_velocity"]
    sp.shifting_method = fsi.ShiftingMethod.XSPH
    sp.shifting_xsph_eps = 0.5
    sp.artificial_viscosity = 0.03
    sp.eos_type = fsi.EosType.TAIT
    sp.use_consistent_gradient_discretization = False
    sp.use_consistent_laplacian_discretization = False
    sp.num_proximity_search_steps = 1
    sp.use_delta_sph = True
    sp.delta_sph_coefficient = 0.1
    sp.use_variable_time_step = variable_dt
    sp.boundary_method = fsi.BoundaryMethod.ADAMI
    sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_UNILATERAL
    if tweak:
        tweak(sp)
    s.SetSPHParameters(sp)
    # the grid uses the spacing as given (double), like ChFsiProblemSPH::m_spacing; the solver holds its float rounding

    # rigid body (demo :258-292)
    r, L = c["radius"], c["length"]
    mass = c["density"] * np.pi * r * r * L
    Ia, It = 0.5 * mass * r * r, mass * (3 * r * r + L * L) / 12.0  # ChCylinder::GetGyration, axis rotated z -> x
    height = (c["initial_height"] if initial_height is None else initial_height) * fsize[2] + r  # demo: 1.05 (released above the surface)
    body = rigid.RigidBody(mass, np.diag([Ia, It, It]), (0, 0, height), fsi.QUNIT, gravity=g)

    # body BCE markers: CreateBCEFsiBody (sph/ChFsiFluidSystemSPH.cpp:983-1020), polar cylinder shells
    bce_body = fsi.frame_transform(s.CreatePointsCylinderInterior(r, L, True), (0, 0, 0), Q_ROTATE_Z_TO_X)
    bce_abs = fsi.frame_transform(bce_body, body.pos, body.rot)

    # SPH grid (Construct :712-749) and embedded-body pruning (ProcessBody :339-378, InsidePoint :306-336)
    rnd = lambda v: int(np.floor(v + 0.5))
    Nx, Ny, Nz = rnd(fsize[0] / d0) + 1, rnd(fsize[1] / d0) + 1, rnd(fsize[2] / d0) + 1
    off = np.array([-fsize[0] / 2, -fsize[1] / 2, 0.0])
    Ix, Iy, Iz = np.meshgrid(np.arange(Nx), np.arange(Ny), np.arange(Nz), indexing="ij")
    G = np.stack([Ix.ravel(), Iy.ravel(), Iz.ravel()], axis=1)
    snapped = np.floor((bce_abs - off) / d0 + 0.5).astype(np.int64)
    lo, hi = snapped.min(axis=0), snapped.max(axis=0)
    cand = np.all((G >= lo) & (G <= hi), axis=1)
    keep = np.ones(len(G), dtype=bool)
    if cand.any():
        p_abs = d0 * G[cand] + off
        p_loc = (p_abs - body.pos) @ fsi.rotmat_from_quat(body.rot)      # TransformPointParentToLocal
        pp = p_loc @ fsi.rotmat_from_quat(Q_ROTATE_Z_TO_X)               # cyl.rot.RotateBack(p - cyl.pos)
        inside = (pp[:, 0] ** 2 + pp[:, 1] ** 2 <= (r + d0) ** 2) & (np.abs(pp[:, 2]) <= L / 2 + d0)
        idx = np.nonzero(cand)[0]
        keep[idx[inside]] = False
    sph_pts = d0 * G[keep] + off

    # depth-based initial pressure (DepthPressurePropertiesCallback, sph/ChFsiProblemSPH.h:480-495)
    gz, c2 = abs(g[2]), float(s.GetSoundSpeed()) ** 2
    p0 = s.GetDensity() * gz * (fsize[2] - sph_pts[:, 2])
    rpm = np.stack([s.GetDensity() + p0 / c2, p0, np.full_like(p0, s.GetViscosity())], axis=1)
    s.AddSPHParticles(sph_pts, rpm)

    # container BCE: ALL & ~Z_POS (AddBoxContainer :937-1047), set semantics = duplicates dropped
    B = []
    a = np.arange
    def block(ix, iy, iz):
        X, Y, Z = np.meshgrid(ix, iy, iz, indexing="ij")
        B.append(np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1))
    block(a(Nx), a(Ny), -a(1, layers + 1))                                     # z_neg
    block(a(-layers, 0), a(Ny), a(-layers, Nz + layers))                        # x_neg
    block(Nx - 1 - a(-layers, 0), a(Ny), a(-layers, Nz + layers))               # x_pos
    block(a(-layers, Nx + layers), a(-layers, 0), a(-layers, Nz + layers))      # y_neg
    block(a(-layers, Nx + layers), Ny - 1 - a(-layers, 0), a(-layers, Nz + layers))  # y_pos
    Bg = np.concatenate(B, axis=0)
    _, first = np.unique(Bg, axis=0, return_index=True)
    Bg = Bg[np.sort(first)]
    bce_pts = d0 * Bg + off
    s.AddBCEBoundary(bce_pts)

    # rigid BCE markers come after the boundary ones (fluid -> wall -> rigid)
    s.AddFsiBody(bce_body, body.state())

    # computational domain from the AABB of SPH + boundary markers (Initialize :262-271)
    allp = np.concatenate([sph_pts, bce_pts], axis=0)
    s.SetComputationalDomain(tuple(allp.min(axis=0) - layers * d0), tuple(allp.max(axis=0) + layers * d0), fsi.BC_NONE)
    if initialize:
        s.Initialize()
    return s, rigid.FsiSystem(s, [body])


def load_particle_csv(path):
    """Read back a fluid/boundary/rigidBCE CSV written by SaveParticleData: returns (pos (n,3), vel (n,3), rho (n,), p (n,)).
    With AddSPHParticles this is a restart path (6 significant digits, like the reference's files)."""
    d = np.loadtxt(path, delimiter=",", skiprows=1, ndmin=2)
    return d[:, 0:3], d[:, 3:6], d[:, 8], d[:, 9]


# ---- CRM granular material: after demo_FSI-SPH_AngleRepose.cpp (a cylinder of soil released on a plate) -----------------------
ANGLE_REPOSE = dict(bulk_density=1500.0, mu_s=0.3819, particle_diameter=0.002, youngs_modulus=2e6, cylinder_radius=0.5, cylinder_height=1.0,
                    init_spacing=0.01, step_size=1e-4)


def granular_column(api=None, radius=0.5, height=1.0, d0=0.01, dt=1e-4, ps_freq=1, device=-1, initialize=True, check_errors=True, tweak=None, box_factor=10.0,
                    scheme=fsi.IntegrationScheme.RK2, mcc=None):
    """demo_FSI-SPH_AngleRepose.cpp:104-190 with the CRM material of the paper it cites: a soil cylinder (grid points of the
    bounding box inside the radius) standing on the bottom plate of a box container.  The demo attaches the plate's BCE markers
    to a fixed body; here they are wall BCE markers.  `box_factor` = container width / cylinder radius (demo: 10)."""
    c = ANGLE_REPOSE
    s = fsi.ChFsiFluidSystemSPH(api, device=device)
    s.EnableErrorChecks(check_errors)
    s.SetStepSize(dt)
    s.SetGravitationalAcceleration((0, 0, -9.81))
    mat = fsi.ElasticMaterialProperties()
    mat.density = c["bulk_density"]
    mat.Young_modulus = c["youngs_modulus"]
    mat.Poisson_ratio = 0.3
    mat.mu_I0 = 0.03
    mat.mu_fric_s = c["mu_s"]
    mat.mu_fric_2 = c["mu_s"]
    mat.average_diam = c["particle_diameter"]
    if mcc is not None:  # modified Cam-Clay instead of mu(I): dict(M, kappa, lambda, v_lambda, pc)
        mat.rheology_model = fsi.RheologyCRM.MCC
        mat.mcc_M, mat.mcc_kappa, mat.mcc_lambda, mat.mcc_v_lambda = mcc["M"], mcc["kappa"], mcc["lambda"], mcc["v_lambda"]
    s.SetElasticSPH(mat)
    sp = fsi.SPHParameters()
    sp.integration_scheme = scheme
    sp.initial_spacing = d0
    sp.d0_multiplier = 1.2
    sp.artificial_viscosity = 0.5
    sp.shifting_method = fsi.ShiftingMethod.PPST_XSPH
    sp.shifting_xsph_eps = 0.25
    sp.shifting_ppst_pull = 1.0
    sp.shifting_ppst_push = 3.0
    sp.free_surface_threshold = 2.0
    sp.num_proximity_search_steps = ps_freq
    sp.use_variable_time_step = False
    sp.kernel_type = fsi.KernelType.CUBIC_SPLINE
    sp.viscosity_method = fsi.ViscosityMethod.ARTIFICIAL_BILATERAL
    sp.boundary_method = fsi.BoundaryMethod.ADAMI
    if tweak:
        tweak(sp)
    s.SetSPHParameters(sp)
    s.SetNumProximitySearchSteps(sp.num_proximity_search_steps)

    bx = by = box_factor * radius
    bz = 1.5 * height
    i0 = s.GetInitialSpacing()
    cMin = (-bx / 2 - 3 * i0 / 2.0, -by / 2 - 3 * i0 / 2.0, -1.0 * bz - 3 * i0)
    cMax = (bx / 2 + 3 * i0 / 2.0, by / 2 + 3 * i0 / 2.0, 2.0 * bz + 3 * i0)
    s.SetComputationalDomain(cMin, cMax, fsi.BC_NONE)

    center = np.array([0.0, 0.0, -bz / 2 + height / 2 + 2 * d0])
    pts = fsi.grid_sampler_box(d0, center, (radius, radius, height / 2))
    rel = pts - center
    pts = pts[rel[:, 0] ** 2 + rel[:, 1] ** 2 <= (radius + 1e-6) ** 2]
    gz = 9.81
    rho0 = float(s.GetDensity())
    Cs = float(np.sqrt(np.float32(s.p.K_bulk) / np.float32(rho0)))
    pre = rho0 * gz * (-(pts[:, 2] + height / 2) + height)
    rho = rho0 + pre / (Cs * Cs)
    rpm = np.stack([rho, pre, np.full(len(pts), float(s.p.mu0))], axis=1)
    if mcc is not None:
        pre = np.maximum(pre, 1.0)  # the specific volume takes log(pc / pressure)
        rpm[:, 1] = pre
        s.AddSPHParticles(pts, rpm, tau_diag=-np.stack([pre, pre, pre], axis=1), pc=np.maximum(mcc["pc"], 1.2 * pre))
    else:
        s.AddSPHParticles(pts, rpm)
    plate = s.CreatePointsBoxContainer((bx, by, bz), (0, 0, -1))
    s.AddBCEBoundary(plate)
    if initialize:
        s.Initialize()
    return s


def write_object_drop_scene(path, s, fsys, d0, step, variable_dt):
    """Scene file of host/demo_FSI-SPH_ObjectDrop_MI.cpp: the staged marker sets of an un-initialised Python facade + the body."""
    import struct
    fl = [b for b in s._pending if b[0] == fsi.capi.SPH_MARKER_FLUID]
    wl = [b for b in s._pending if b[0] == fsi.capi.SPH_MARKER_BCE_WALL]
    fxyz = np.concatenate([b[1] for b in fl]).astype(np.float32)
    frpm = np.concatenate([b[3] for b in fl]).astype(np.float32)
    wxyz = np.concatenate([b[1] for b in wl]).astype(np.float32)
    bloc, _st = s._bodies[0]
    body = fsys.bodies[0]
    c = OBJECT_DROP
    with open(path, "wb") as f:
        f.write(b"CHMISCN1")
        f.write(struct.pack("<QQQ", len(fxyz), len(wxyz), len(bloc)))
        f.write(struct.pack("<5d", d0, step, c["max_velocity"], 1000.0, 1.0))
        f.write(struct.pack("<3d", 0.0, 0.0, -9.8))
        f.write(struct.pack("<3d", *[float(s.p.cMin[a]) for a in range(3)]))
        f.write(struct.pack("<3d", *[float(s.p.cMax[a]) for a in range(3)]))
        f.write(struct.pack("<d", body.mass))
        f.write(struct.pack("<3d", *np.diag(body.I)))
        f.write(struct.pack("<3d", *body.pos))
        f.write(struct.pack("<4d", *body.rot))
        f.write(struct.pack("<ii", c["num_bce_layers"], 1 if variable_dt else 0))
        for a in (fxyz, frpm, wxyz, np.asarray(bloc, dtype=np.float32)):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())

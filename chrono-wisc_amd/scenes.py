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

"""Python mirror of the reference's SPH fluid-system facade for the hot path.

`ChFsiFluidSystemSPH` reproduces, method for method, the subset of
chrono::fsi::sph::ChFsiFluidSystemSPH (sph/ChFsiFluidSystemSPH.{h,cpp}) that the demos and
tests on the path use: parameter setters, marker creation, Initialize, DoStepDynamics and
the original-order accessors.  It plays the role PyChrono's SWIG wrapper plays for the
reference (src/chrono_swig/interface/fsi), but talks to the MI355X solver through the C ABI
(include/chsph_mi.h).  The C++ twin of this class is chrono-wisc_amd/host/ChFsiFluidSystemSPH_MI.h.

All citations are relative to /root/reference/src/chrono_fsi/.
"""
import math

import numpy as np

from . import capi
from .capi import SphError  # noqa: F401  (re-export)

f32 = np.float32


# ---- enums (sph/ChFsiDefinitionsSPH.h:30-79) ----------------------------------------
class IntegrationScheme:
    EULER, RK2, VERLET, SYMPLECTIC, IMPLICIT_SPH = range(5)


class ShiftingMethod:
    NONE, PPST, XSPH, PPST_XSPH, DIFFUSION, DIFFUSION_XSPH = range(6)


class EosType:
    TAIT, ISOTHERMAL = range(2)


class KernelType:
    QUADRATIC, CUBIC_SPLINE, QUINTIC_SPLINE, WENDLAND = range(4)


class ViscosityMethod:
    LAMINAR, ARTIFICIAL_UNILATERAL, ARTIFICIAL_BILATERAL = range(3)


class BoundaryMethod:
    ADAMI, HOLMES = range(2)


class BCType:
    NONE, PERIODIC, INLET_OUTLET = range(3)


BC_NONE = (BCType.NONE, BCType.NONE, BCType.NONE)
BC_X_PERIODIC = (BCType.PERIODIC, BCType.NONE, BCType.NONE)
BC_Y_PERIODIC = (BCType.NONE, BCType.PERIODIC, BCType.NONE)
BC_Z_PERIODIC = (BCType.NONE, BCType.NONE, BCType.PERIODIC)
BC_ALL_PERIODIC = (BCType.PERIODIC, BCType.PERIODIC, BCType.PERIODIC)


class MarkerGroup:
    FLUID, SOLID, BOUNDARY, NON_FLUID, NON_SOLID, NON_BOUNDARY, ALL = range(7)


class OutputLevel:
    STATE, STATE_PRESSURE, CRM_FULL = range(3)


class RheologyCRM:
    MU_OF_I, MCC = range(2)


class ElasticMaterialProperties:
    """ChFsiFluidSystemSPH::ElasticMaterialProperties (sph/ChFsiFluidSystemSPH.cpp:430-443)."""

    def __init__(self):
        self.density = 1000.0
        self.Young_modulus = 1e6
        self.Poisson_ratio = 0.3
        self.mu_I0 = 0.03
        self.mu_fric_s = 0.7
        self.mu_fric_2 = 0.7
        self.average_diam = 0.005
        self.cohesion_coeff = 0.0
        self.rheology_model = RheologyCRM.MU_OF_I
        self.mcc_M = 0.0
        self.mcc_kappa = 0.0
        self.mcc_lambda = 0.0
        self.mcc_v_lambda = 2.0


class FluidProperties:
    """ChFsiFluidSystemSPH::FluidProperties (sph/ChFsiFluidSystemSPH.cpp:419)."""

    def __init__(self):
        self.density = 1000.0
        self.viscosity = 0.1
        self.char_length = 1.0


class SPHParameters:
    """ChFsiFluidSystemSPH::SPHParameters defaults (sph/ChFsiFluidSystemSPH.cpp:478-505)."""

    def __init__(self):
        self.integration_scheme = IntegrationScheme.RK2
        self.initial_spacing = 0.01
        self.d0_multiplier = 1.2
        self.max_velocity = 1.0
        self.shifting_method = ShiftingMethod.XSPH
        self.shifting_xsph_eps = 0.5
        self.shifting_ppst_push = 3.0
        self.shifting_ppst_pull = 1.0
        self.shifting_beta_implicit = 1.0
        self.shifting_diffusion_A = 1.0
        self.shifting_diffusion_AFSM = 3.0
        self.shifting_diffusion_AFST = 2.0
        self.min_distance_coefficient = 0.01
        self.density_reinit_steps = int(2e8)
        self.use_density_based_projection = False
        self.num_bce_layers = 3
        self.use_consistent_gradient_discretization = False
        self.use_consistent_laplacian_discretization = False
        self.viscosity_method = ViscosityMethod.ARTIFICIAL_UNILATERAL
        self.boundary_method = BoundaryMethod.ADAMI
        self.kernel_type = KernelType.CUBIC_SPLINE
        self.use_delta_sph = True
        self.delta_sph_coefficient = 0.1
        self.artificial_viscosity = 0.02
        self.free_surface_threshold = 2.0
        self.num_proximity_search_steps = 1
        self.eos_type = EosType.ISOTHERMAL
        self.use_variable_time_step = False


class LinSolverParameters:
    def __init__(self):
        self.type = 0  # JACOBI
        self.atol = 0.0
        self.rtol = 0.0
        self.max_num_iters = 1000


def _cube(a):
    a = f32(a)
    return f32(f32(a * a) * a)


# ---- geometry helpers restating Chrono core pieces the marker generators rely on --------
def quat_from_angle_axis(angle, axis):
    """QuatFromAngleAxis (src/chrono/core/ChRotation.cpp:149-163)."""
    half = angle / 2
    s = math.sin(half)
    return (math.cos(half), axis[0] * s, axis[1] * s, axis[2] * s)


QUNIT = (1.0, 0.0, 0.0, 0.0)


def rotmat_from_quat(q):
    """ChMatrix33::SetFromQuaternion (src/chrono/core/ChMatrix33.h:253-274)."""
    e0, e1, e2, e3 = q
    e0e0, e1e1, e2e2, e3e3 = e0 * e0, e1 * e1, e2 * e2, e3 * e3
    e0e1, e0e2, e0e3 = e0 * e1, e0 * e2, e0 * e3
    e1e2, e1e3, e2e3 = e1 * e2, e1 * e3, e2 * e3
    return np.array([[(e0e0 + e1e1) * 2 - 1, (e1e2 - e0e3) * 2, (e1e3 + e0e2) * 2],
                     [(e1e2 + e0e3) * 2, (e0e0 + e2e2) * 2 - 1, (e2e3 - e0e1) * 2],
                     [(e1e3 - e0e2) * 2, (e2e3 + e0e1) * 2, (e0e0 + e3e3) * 2 - 1]], dtype=np.float64)


def frame_transform(points, pos, quat):
    """ChFrame::TransformPointLocalToParent (src/chrono/core/ChFrame.h:243): pos + R v, in double."""
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    R = rotmat_from_quat(quat)
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    out = np.empty_like(points)
    for r in range(3):
        out[:, r] = pos[r] + ((R[r, 0] * x + R[r, 1] * y) + R[r, 2] * z)
    return out


def grid_sampler_box(separation, center, half_dim):
    """utils::ChGridSampler<double>::SampleBox (src/chrono/utils/ChUtilsSamplers.h:488-527,168-197).
    Order: ix outer, iy, iz inner."""
    center = np.asarray(center, dtype=np.float64)
    size = np.asarray(half_dim, dtype=np.float64)
    fuzz = 1e-6 * size[0] if size[0] < 1 else 1e-6
    n = np.zeros(3, dtype=np.int64)
    sep = np.zeros(3)
    for i in range(3):
        d = 2 * size[i]
        n[i] = int(math.floor(d / separation + 0.5))  # std::round, d/sep >= 0
        sep[i] = d / n[i] if n[i] != 0 else float("nan")
    bl = center - size
    ix = np.arange(n[0] + 1, dtype=np.float64)
    iy = np.arange(n[1] + 1, dtype=np.float64)
    iz = np.arange(n[2] + 1, dtype=np.float64)
    # ix * sep is NaN*0 in C++ when n == 0 (0 * (d/0) = 0 * inf or nan); Chrono would emit NaN there.
    px = bl[0] + (ix * sep[0] if n[0] else ix * 0.0)
    py = bl[1] + (iy * sep[1] if n[1] else iy * 0.0)
    pz = bl[2] + (iz * sep[2] if n[2] else iz * 0.0)
    X, Y, Z = np.meshgrid(px, py, pz, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    v = np.abs(pts - center)
    keep = (v[:, 0] <= size[0] + fuzz) & (v[:, 1] <= size[1] + fuzz) & (v[:, 2] <= size[2] + fuzz)
    return pts[keep]


class ChFsiFluidSystemSPH:
    """Mirror of chrono::fsi::sph::ChFsiFluidSystemSPH for the CFD (WCSPH / I2SPH) path.

    `api` is a capi.CApi instance; by default the product library (HIP).  There is no CPU
    fallback: constructing the default api raises SphError when the extension is missing."""

    def __init__(self, api=None, device=-1):
        self.api = api if api is not None else capi.CApi()
        self._device = device
        self.p = self.api.default_params()  # InitParams (sph/ChFsiFluidSystemSPH.cpp:81-176)
        self._solver = None
        self._pending = []  # (type_code, xyz, vel, rho_p_mu)
        self._pending_stress = {}  # id(xyz block) -> (tauXxYyZz, tauXyXzYz) for CRM
        self._pending_pc = {}      # id(xyz block) -> preconsolidation pressure for CRM / MCC
        self._bodies = []  # (local bce coords, initial state dict)
        self._step_log = []  # variable time steps handed out (PrintTimeSteps)
        self.m_step = -1.0
        self.m_time = 0.0
        self.m_frame = 0
        self.m_is_initialized = False
        self.m_verbose = False
        self.m_check_errors = True
        self.m_arithmetic = None  # library default

    # ---- generic ChFsiFluidSystem (ChFsiFluidSystem.cpp:41-69)
    def SetVerbose(self, verbose):
        self.m_verbose = verbose

    def SetStepSize(self, step):
        self.m_step = float(step)

    def GetStepSize(self):
        return self.m_step

    def GetSimTime(self):
        return self.m_time

    def EnableErrorChecks(self, on):
        self.m_check_errors = bool(on)

    def SetArithmetic(self, mode):
        """Native (no reference counterpart): "fast" (default: hardware reciprocals, FMA, packed FP32; one force evaluation within
        1e-5 of the exact path's outputs, 10 steps within 1e-4 of each field's inf-norm: tests/test_fast_gpu.py)
        or "exact" (the reference's expressions in source order, IEEE division / sqrt: bit-identical to the CPU oracle).
        May be called before or after Initialize.  None leaves the library default (env CHSPH_ARITH) in place."""
        self.m_arithmetic = {None: None, "exact": 0, "fast": 1, 0: 0, 1: 1}[mode]
        if self.m_is_initialized and self.m_arithmetic is not None:
            self._solver.set_arithmetic(self.m_arithmetic)

    # ---- setters (sph/ChFsiFluidSystemSPH.cpp:180-328)
    def SetBoundaryType(self, m):
        self.p.boundary_method = m

    def SetViscosityType(self, m):
        self.p.viscosity_method = m

    def SetArtificialViscosityCoefficient(self, c):
        self.p.artificial_viscosity = c

    def SetKernelType(self, k):
        self.p.kernel_type = k
        self.p.h_multiplier = 3 if k == KernelType.QUINTIC_SPLINE else 2

    def SetShiftingMethod(self, m):
        self.p.shifting_method = m

    def SetIntegrationScheme(self, s):
        self.p.integration_scheme = s

    def SetContainerDim(self, box_dim):
        self.p.boxDimX, self.p.boxDimY, self.p.boxDimZ = box_dim

    def SetComputationalDomain(self, cmin, cmax, bc_type=None):
        for a in range(3):
            self.p.cMin[a] = cmin[a]
            self.p.cMax[a] = cmax[a]
        self.p.use_default_limits = 0
        if bc_type is not None:
            for a in range(3):
                self.p.bc_type[a] = bc_type[a]

    def SetActiveDomain(self, box_dim):
        for a in range(3):
            self.p.bodyActiveDomain[a] = box_dim[a] / 2
        self.p.use_active_domain = 1

    def SetActiveDomainDelay(self, duration):
        self.p.settlingTime = duration

    def SetNumBCELayers(self, n):
        self.p.num_bce_layers = n

    def SetInitPressure(self, height):
        self.p.pressure_height = height
        self.p.use_init_pressure = 1

    def SetGravitationalAcceleration(self, g):
        for a in range(3):
            self.p.gravity[a] = g[a]

    def SetBodyForce(self, f):
        for a in range(3):
            self.p.bodyForce3[a] = f[a]

    def SetInitialSpacing(self, spacing):
        p = self.p
        p.d0 = spacing
        p.ood0 = f32(1) / f32(p.d0)
        p.volume0 = _cube(p.d0)
        p.markerMass = f32(p.volume0) * f32(p.rho0)
        p.h = f32(p.d0_multiplier) * f32(p.d0)
        p.ooh = f32(1) / f32(p.h)

    def SetKernelMultiplier(self, m):
        p = self.p
        p.d0_multiplier = m
        p.h = f32(p.d0_multiplier) * f32(p.d0)
        p.ooh = f32(1) / f32(p.h)

    def SetDensity(self, rho0):
        p = self.p
        p.rho0 = rho0
        p.invrho0 = f32(1) / f32(p.rho0)
        p.markerMass = f32(p.volume0) * f32(p.rho0)

    def SetShiftingPPSTParameters(self, push, pull):
        self.p.shifting_ppst_push = push
        self.p.shifting_ppst_pull = pull

    def SetShiftingXSPHParameters(self, eps):
        self.p.shifting_xsph_eps = eps

    def SetShiftingDiffusionParameters(self, A, AFSM, AFST):
        self.p.shifting_diffusion_A = A
        self.p.shifting_diffusion_AFSM = AFSM
        self.p.shifting_diffusion_AFST = AFST

    def SetConsistentDerivativeDiscretization(self, grad, lap):
        self.p.use_consistent_gradient_discretization = int(grad)
        self.p.use_consistent_laplacian_discretization = int(lap)

    def SetNumProximitySearchSteps(self, steps):
        self.p.num_proximity_search_steps = steps

    def SetUseVariableTimeStep(self, on):
        self.p.use_variable_time_step = int(on)

    # ---- SetCfdSPH / SetSPHParameters / SetLinSolverParameters (cpp:421-428, 507-559)
    def SetCfdSPH(self, fluid_props):
        self.SetDensity(fluid_props.density)
        self.p.mu0 = fluid_props.viscosity
        self.p.L_Characteristic = fluid_props.char_length

    def SetElasticSPH(self, mat):
        """cpp:445-476: CRM granular material; the moduli are evaluated in double and stored as Real, like the reference."""
        p = self.p
        p.elastic_SPH = 1
        self.SetDensity(mat.density)
        p.rheology_model_crm = mat.rheology_model
        p.E_young = mat.Young_modulus
        p.Nu_poisson = mat.Poisson_ratio
        p.ave_diam = mat.average_diam
        E, nu = float(f32(p.E_young)), float(f32(p.Nu_poisson))
        p.G_shear = E / (2.0 * (1.0 + nu))
        p.INV_G_shear = 1.0 / float(f32(p.G_shear))
        p.K_bulk = E / (3.0 * (1.0 - 2.0 * nu))
        if mat.rheology_model == RheologyCRM.MCC:
            p.mcc_M, p.mcc_kappa, p.mcc_lambda, p.mcc_v_lambda = mat.mcc_M, mat.mcc_kappa, mat.mcc_lambda, mat.mcc_v_lambda
        elif mat.rheology_model == RheologyCRM.MU_OF_I:
            p.mu_I0 = mat.mu_I0
            p.mu_fric_s = mat.mu_fric_s
            p.mu_fric_2 = mat.mu_fric_2
            p.Coh_coeff = mat.cohesion_coeff
        else:
            raise SphError("Invalid rheology model")

    def GetPhysicsProblemString(self):
        return "CRM" if self.p.elastic_SPH else "CFD"

    def SetSPHParameters(self, sp):
        p = self.p
        p.free_surface_threshold = sp.free_surface_threshold
        p.integration_scheme = sp.integration_scheme
        p.eos_type = sp.eos_type
        p.viscosity_method = sp.viscosity_method
        p.boundary_method = sp.boundary_method
        p.kernel_type = sp.kernel_type
        p.shifting_method = sp.shifting_method
        p.d0 = sp.initial_spacing
        p.volume0 = _cube(p.d0)
        p.markerMass = f32(p.volume0) * f32(p.rho0)
        p.d0_multiplier = sp.d0_multiplier
        p.h = f32(p.d0_multiplier) * f32(p.d0)
        p.ood0 = f32(1) / f32(p.d0)
        p.ooh = f32(1) / f32(p.h)
        p.v_Max = sp.max_velocity
        p.shifting_xsph_eps = sp.shifting_xsph_eps
        p.shifting_ppst_push = sp.shifting_ppst_push
        p.shifting_ppst_pull = sp.shifting_ppst_pull
        p.shifting_beta_implicit = sp.shifting_beta_implicit
        p.shifting_diffusion_A = sp.shifting_diffusion_A
        p.shifting_diffusion_AFSM = sp.shifting_diffusion_AFSM
        p.shifting_diffusion_AFST = sp.shifting_diffusion_AFST
        p.epsMinMarkersDis = sp.min_distance_coefficient
        p.density_reinit_steps = int(sp.density_reinit_steps)
        p.use_density_based_projection = int(sp.use_density_based_projection)
        p.num_bce_layers = sp.num_bce_layers
        p.use_consistent_gradient_discretization = int(sp.use_consistent_gradient_discretization)
        p.use_consistent_laplacian_discretization = int(sp.use_consistent_laplacian_discretization)
        p.artificial_viscosity = sp.artificial_viscosity
        p.use_delta_sph = int(sp.use_delta_sph)
        p.density_delta = sp.delta_sph_coefficient
        p.num_proximity_search_steps = sp.num_proximity_search_steps
        p.use_variable_time_step = int(sp.use_variable_time_step)

    def SetLinSolverParameters(self, lp):
        self.p.LinearSolver = lp.type
        self.p.LinearSolver_Abs_Tol = lp.atol
        self.p.LinearSolver_Rel_Tol = lp.rtol
        self.p.LinearSolver_Max_Iter = lp.max_num_iters

    # ---- getters (cpp:2720-2770)
    def GetKernelLength(self):
        return float(self.p.h)

    def GetInitialSpacing(self):
        return float(self.p.d0)

    def GetNumBCELayers(self):
        return self.p.num_bce_layers

    def GetDensity(self):
        return float(self.p.rho0)

    def GetViscosity(self):
        return float(self.p.mu0)

    def GetBasePressure(self):
        return float(self.p.base_pressure)

    def GetParticleMass(self):
        return float(self.p.markerMass)

    def GetGravitationalAcceleration(self):
        return tuple(float(x) for x in self.p.gravity)

    def GetSoundSpeed(self):
        self.p.Cs = f32(10) * f32(self.p.v_Max)
        return float(self.p.Cs)

    def GetParams(self):
        return self._solver.params() if self._solver else self.p

    def GetCounters(self):
        return self._solver.counters()

    def GetNumFluidMarkers(self):
        return self._solver.counters().numFluidMarkers

    def GetNumBoundaryMarkers(self):
        return self._solver.counters().numBoundaryMarkers

    def GetNumRigidBodyMarkers(self):
        return self._solver.counters().numRigidMarkers

    # ---- markers (cpp:1848-1888; phys/SphDataManager.cu:209-236)
    def AddSPHParticle(self, pos, rho=None, pres=None, mu=None, vel=(0, 0, 0)):
        rho = self.p.rho0 if rho is None else rho
        pres = self.p.base_pressure if pres is None else pres
        mu = self.p.mu0 if mu is None else mu
        self.AddSPHParticles(np.array([pos]), np.array([[rho, pres, mu]]), np.array([vel]))

    def AddSPHParticles(self, pos, rho_p_mu=None, vel=None, tau_diag=None, tau_offdiag=None, pc=None):
        """Vectorised AddSPHParticle: pos (n,3) double, rho_p_mu (n,3), vel (n,3); CRM: tauXxYyZz / tauXyXzYz (n,3), MCC: pc (n)."""
        assert not self.m_is_initialized
        blk = (capi.SPH_MARKER_FLUID, np.asarray(pos, dtype=np.float64).astype(np.float32),
               None if vel is None else np.asarray(vel, dtype=np.float64).astype(np.float32),
               None if rho_p_mu is None else np.asarray(rho_p_mu, dtype=np.float64).astype(np.float32), float(self.p.h))
        self._pending.append(blk)
        if pc is not None:
            self._pending_pc[id(blk[1])] = np.asarray(pc, dtype=np.float64).astype(np.float32)
        if tau_diag is not None or tau_offdiag is not None:
            n = len(blk[1])
            z = np.zeros((n, 3), dtype=np.float32)
            self._pending_stress[id(blk[1])] = (z if tau_diag is None else np.asarray(tau_diag, dtype=np.float64).astype(np.float32),
                                                z if tau_offdiag is None else np.asarray(tau_offdiag, dtype=np.float64).astype(np.float32))

    def AddBoxSPH(self, center, half_dim):
        pts = grid_sampler_box(float(self.p.d0), center, half_dim)
        rpm = np.tile(np.array([[float(self.p.rho0), 0.0, float(self.p.mu0)]]), (len(pts), 1))
        self.AddSPHParticles(pts, rpm)

    def AddBCEBoundary(self, points, pos=(0, 0, 0), quat=QUNIT):
        assert not self.m_is_initialized
        g = frame_transform(points, pos, quat)
        self._pending.append((capi.SPH_MARKER_BCE_WALL, g.astype(np.float32), None, None, float(self.p.h)))

    def AddFsiBody(self, bce_local, state):
        """OnAddFsiBody with explicit body-local BCE coordinates (cpp:971-981)."""
        assert not self.m_is_initialized
        self._bodies.append((np.asarray(bce_local, dtype=np.float64).astype(np.float32), dict(state)))
        return len(self._bodies) - 1

    # ---- BCE point generators (cpp:1890-1970)
    def CreatePointsPlate(self, size):
        spacing = f32(self.p.d0)
        num_layers = self.p.num_bce_layers
        hx, hy = size[0] / 2, size[1] / 2
        npx = int(math.floor(hx / float(spacing) + 0.5))
        npy = int(math.floor(hy / float(spacing) + 0.5))
        dx, dy = hx / npx, hy / npy
        ix = np.arange(-npx, npx + 1, dtype=np.float64) * dx
        iy = np.arange(-npy, npy + 1, dtype=np.float64) * dy
        out = []
        for il in range(num_layers):
            z = float(f32(-il) * spacing)
            X, Y = np.meshgrid(ix, iy, indexing="ij")
            out.append(np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], axis=1))
        return np.concatenate(out, axis=0)

    def CreatePointsBoxContainer(self, size, faces):
        spacing = f32(self.p.d0)
        buffer = float(f32(2 * (self.p.num_bce_layers - 1)) * spacing)
        hs = (size[0] / 2, size[1] / 2, size[2] / 2)
        out = []
        pi = math.pi
        if faces[2] in (-1, 2):
            out.append(frame_transform(self.CreatePointsPlate((size[0], size[1])), (0, 0, -hs[2]), QUNIT))
        if faces[2] in (+1, 2):
            out.append(frame_transform(self.CreatePointsPlate((size[0], size[1])), (0, 0, +hs[2]), quat_from_angle_axis(pi, (1, 0, 0))))
        if faces[0] in (-1, 2):
            out.append(frame_transform(self.CreatePointsPlate((size[2] + buffer, size[1])), (-hs[0], 0, 0), quat_from_angle_axis(+pi / 2, (0, 1, 0))))
        if faces[0] in (+1, 2):
            out.append(frame_transform(self.CreatePointsPlate((size[2] + buffer, size[1])), (+hs[0], 0, 0), quat_from_angle_axis(-pi / 2, (0, 1, 0))))
        if faces[1] in (-1, 2):
            out.append(frame_transform(self.CreatePointsPlate((size[0] + buffer, size[2] + buffer)), (0, -hs[1], 0), quat_from_angle_axis(-pi / 2, (1, 0, 0))))
        if faces[1] in (+1, 2):
            out.append(frame_transform(self.CreatePointsPlate((size[0] + buffer, size[2] + buffer)), (0, +hs[1], 0), quat_from_angle_axis(+pi / 2, (1, 0, 0))))
        return np.concatenate(out, axis=0) if out else np.zeros((0, 3))

    def CreatePointsBoxInterior(self, size):
        """cpp:1972-2042: `num_bce_layers` shells inside a box (edges and corners appear in several faces: duplicates kept)."""
        spacing = float(f32(self.p.d0))
        L = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        np_ = [1 + rnd(size[a] / spacing) for a in range(3)]
        fill = any(n <= 2 * L for n in np_)
        delta = [size[a] / (np_[a] - 1) for a in range(3)]
        ax = [(-size[a] / 2 + np.arange(np_[a], dtype=np.float64) * delta[a]) for a in range(3)]
        if fill:
            X, Y, Z = np.meshgrid(ax[0], ax[1], ax[2], indexing="ij")
            return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        out = []
        for il in range(L):
            xm, xp = -size[0] / 2 + il * delta[0], +size[0] / 2 - il * delta[0]
            for y in ax[1]:
                for z in ax[2]:
                    out.append((xm, y, z)); out.append((xp, y, z))
            ym, yp = -size[1] / 2 + il * delta[1], +size[1] / 2 - il * delta[1]
            for z in ax[2]:
                for x in ax[0]:
                    out.append((x, ym, z)); out.append((x, yp, z))
            zm, zp = -size[2] / 2 + il * delta[2], +size[2] / 2 - il * delta[2]
            for x in ax[0]:
                for y in ax[1]:
                    out.append((x, y, zm)); out.append((x, y, zp))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    def CreatePointsSphereInterior(self, radius, polar=True):
        """cpp:2102-2162 (the polar branch's x = y = r sin(phi) before the theta factors is the reference's)."""
        spacing = float(f32(self.p.d0))
        L = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        out = []
        if polar:
            rad_in = radius - (L - 1) * spacing
            for ir in range(L):
                r = rad_in + ir * spacing
                np_phi = rnd(math.pi * r / spacing)
                delta_phi = math.pi / np_phi
                for ip in range(np_phi):
                    phi = ip * delta_phi
                    cphi, sphi = math.cos(phi), math.sin(phi)
                    x = y = r * sphi
                    z = r * cphi
                    np_th = rnd(2 * math.pi * r * sphi / spacing)
                    delta_th = (2 * math.pi) / np_th if np_th > 0 else 1
                    for it in range(np_th):
                        theta = it * delta_th
                        out.append((x * math.cos(theta), y * math.sin(theta), z))
            return np.array(out, dtype=np.float64).reshape(-1, 3)
        npn = rnd(radius / spacing)
        delta = radius / npn
        for iz in range(npn + 1):
            z = iz * delta
            rz_max = math.sqrt(radius * radius - z * z)
            rz_min = max(rz_max - L * delta, 0.0)
            if iz >= npn - L:
                rz_min = 0
            rz_min2, rz_max2 = rz_min * rz_min, rz_max * rz_max
            nq = rnd(rz_max / spacing)
            for ix in range(-nq, nq + 1):
                x = ix * delta
                for iy in range(-nq, nq + 1):
                    y = iy * delta
                    r2 = x * x + y * y
                    if rz_min2 <= r2 <= rz_max2:
                        out.append((x, y, +z)); out.append((x, y, -z))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    def CreatePointsBoxExterior(self, size):
        """cpp:2044-2100: num_bce_layers shells outside a box, x faces over the inflated y-z extent, then y, then z faces."""
        spacing = float(f32(self.p.d0))
        L = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        n = [1 + rnd(size[a] / spacing) for a in range(3)]
        delta = [size[a] / (n[a] - 1) for a in range(3)]
        N = [n[a] + 2 * (L - 1) for a in range(3)]
        S = [size[a] + 2.0 * (L - 1.0) * delta[a] for a in range(3)]
        out = []
        for il in range(L):
            xm, xp = -S[0] / 2 + il * delta[0], +S[0] / 2 - il * delta[0]
            for iy in range(N[1]):
                y = -S[1] / 2 + iy * delta[1]
                for iz in range(N[2]):
                    z = -S[2] / 2 + iz * delta[2]
                    out.append((xm, y, z)); out.append((xp, y, z))
            ym, yp = -S[1] / 2 + il * delta[1], +S[1] / 2 - il * delta[1]
            for iz in range(N[2]):
                z = -S[2] / 2 + iz * delta[2]
                for ix in range(n[0]):
                    x = -size[0] / 2 + ix * delta[0]
                    out.append((x, ym, z)); out.append((x, yp, z))
            zm, zp = -S[2] / 2 + il * delta[2], +S[2] / 2 - il * delta[2]
            for ix in range(n[0]):
                x = -size[0] / 2 + ix * delta[0]
                for iy in range(n[1]):
                    y = -size[1] / 2 + iy * delta[1]
                    out.append((x, y, zm)); out.append((x, y, zp))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    def CreatePointsSphereExterior(self, radius, polar=True):
        """cpp:2164-2223.  pi is the reference's Real(CH_PI) (its float value) in these generators."""
        spacing = float(f32(self.p.d0))
        L = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        pi = float(np.float32(math.pi))
        out = []
        if polar:
            for ir in range(L):
                r = radius + ir * spacing
                np_phi = rnd(pi * r / spacing)
                delta_phi = pi / np_phi
                for ip in range(np_phi):
                    phi = ip * delta_phi
                    cphi, sphi = math.cos(phi), math.sin(phi)
                    x = y = r * sphi
                    z = r * cphi
                    np_th = rnd(2 * pi * r * sphi / spacing)
                    delta_th = (2 * pi) / np_th if np_th > 0 else 1
                    for it in range(np_th):
                        theta = it * delta_th
                        out.append((x * math.cos(theta), y * math.sin(theta), z))
            return np.array(out, dtype=np.float64).reshape(-1, 3)
        npn = rnd(radius / spacing)
        delta = radius / npn
        npn += L
        radius += L * delta
        for iz in range(npn + 1):
            z = iz * delta
            rz_max = math.sqrt(max(radius * radius - z * z, 0.0))
            rz_min = max(rz_max - L * delta, 0.0)
            if iz >= npn - L:
                rz_min = 0
            rz_min2, rz_max2 = rz_min * rz_min, radius * radius - z * z
            for ix in range(-npn, npn + 1):
                x = ix * delta
                for iy in range(-npn, npn + 1):
                    y = iy * delta
                    r2 = x * x + y * y
                    if rz_min2 <= r2 <= rz_max2:
                        out.append((x, y, +z)); out.append((x, y, -z))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    def CreatePointsCylinderExterior(self, rad, height, polar=True):
        """cpp:2322-2412: shells around a cylinder (axis z) plus its two caps."""
        spacing = float(f32(self.p.d0))
        L = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        pi = float(np.float32(math.pi))
        np_h = rnd(height / spacing)
        delta_h = height / np_h
        np_h += 2 * (L - 1)
        height += 2 * (L - 1) * delta_h
        out = []
        if polar:
            rad_max = rad + L * spacing
            rad_min = rad_max - L * spacing
            np_r = rnd((rad_max - rad_min) / spacing)
            delta_r = (rad_max - rad_min) / np_r
            for ir in range(np_r + 1):
                r = rad_min + ir * delta_r
                np_th = rnd(2 * pi * r / spacing)
                delta_th = (2 * pi) / np_th if np_th > 0 else 1
                for it in range(np_th):
                    theta = it * delta_th
                    x, y = r * math.cos(theta), r * math.sin(theta)
                    for iz in range(np_h + 1):
                        out.append((x, y, -height / 2 + iz * delta_h))
            np_r = rnd(rad_min / spacing)
            delta_r = rad_min / np_r
            for ir in range(np_r):
                r = ir * delta_r
                np_th = max(rnd(2 * pi * r / spacing), 1)
                delta_th = (2 * pi) / np_th
                for it in range(np_th):
                    theta = it * delta_th
                    x, y = r * math.cos(theta), r * math.sin(theta)
                    for iz in range(L):
                        z = -height / 2 + iz * delta_h
                        out.append((x, y, -z)); out.append((x, y, +z))
            return np.array(out, dtype=np.float64).reshape(-1, 3)
        np_r = rnd(rad / spacing)
        delta_r = rad / np_r
        np_r += L
        rad += L * delta_r
        rad_max, rad_min = rad, max(rad - L * delta_r, 0.0)
        r_max2, r_min2 = rad_max * rad_max, rad_min * rad_min
        for ix in range(-np_r, np_r + 1):
            x = ix * delta_r
            for iy in range(-np_r, np_r + 1):
                y = iy * delta_r
                r2 = x * x + y * y
                if r_min2 <= r2 <= r_max2:
                    for iz in range(np_h + 1):
                        out.append((x, y, -height / 2 + iz * delta_h))
                if r2 < r_min2:
                    for iz in range(L):
                        z = -height / 2 + iz * delta_h
                        out.append((x, y, -z)); out.append((x, y, +z))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    def _truncated_cone(self, rad, rad_tip, height, polar, exterior):
        """cpp:2417-2489 (interior) / 2494-2570 (exterior): base of radius rad at z = 0, tip radius rad_tip at z = height.  Where a
        ring degenerates to zero width the reference divides 0 by 0; such rings contribute the single radius rad_in here."""
        spacing = float(f32(self.p.d0))
        L = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        pi = float(np.float32(math.pi))
        np_h = rnd(height / spacing)
        delta_h = height / np_h
        if exterior:
            np_h += 2 * L
            height += 2 * L * delta_h
        grow = L * spacing if exterior else 0.0
        out = []
        if polar:
            for iz in range(np_h):
                z = iz * delta_h
                rz = rad_tip + (rad - rad_tip) * (height - z) / height
                rad_out = rz + grow
                rad_in = max(rad_out - L * spacing, 0.0)
                if iz >= np_h - L:
                    rad_in = 0
                np_r = rnd((rad_out - rad_in) / spacing)
                delta_r = (rad_out - rad_in) / np_r if np_r > 0 else 0.0
                for ir in range(np_r + 1):
                    r = rad_in + ir * delta_r
                    np_th = rnd(2 * pi * r / spacing)
                    if np_th <= 0:
                        continue
                    delta_th = (2 * pi) / np_th
                    for it in range(np_th):
                        theta = it * delta_th
                        out.append((r * math.cos(theta), r * math.sin(theta), z))
            if rad_tip < np.finfo(np.float64).eps:
                out.append((0.0, 0.0, height))
            return np.array(out, dtype=np.float64).reshape(-1, 3)
        np_r = rnd(rad / spacing)
        delta_r = rad / np_r
        for iz in range(np_h + 1):
            z = iz * delta_h
            rz = rad_tip + (rad - rad_tip) * (height - z) / height
            rad_out = rz + grow
            rad_in = max(rad_out - L * spacing, 0.0)
            r_out2, r_in2 = rad_out * rad_out, rad_in * rad_in
            for ix in range(-np_r, np_r + 1):
                x = ix * delta_r
                for iy in range(-np_r, np_r + 1):
                    y = iy * delta_r
                    if r_in2 <= x * x + y * y <= r_out2:
                        out.append((x, y, z))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    def CreatePointsTruncatedConeInterior(self, rad, rad_tip, height, polar=True):
        return self._truncated_cone(rad, rad_tip, height, polar, False)

    def CreatePointsConeInterior(self, rad, height, polar=True):
        return self._truncated_cone(rad, 0.0, height, polar, False)

    def CreatePointsTruncatedConeExterior(self, rad, rad_tip, height, polar=True):
        return self._truncated_cone(rad, rad_tip, height, polar, True)

    def CreatePointsConeExterior(self, rad, height, polar=True):
        return self._truncated_cone(rad, 0.0, height, polar, True)

    def CreatePointsCylinderAnnulus(self, rad_inner, rad_outer, height, polar=True):
        """cpp:2572-2625: rings of ~spacing arc length (polar) or an accept/reject lattice, axis along z."""
        spacing = float(f32(self.p.d0))
        rnd = lambda v: int(math.floor(v + 0.5))  # std::round for the positive arguments used here
        hheight = height / 2
        np_h = rnd(hheight / spacing)
        delta_h = hheight / np_h
        zs = np.arange(-np_h, np_h + 1, dtype=np.float64) * delta_h
        out = []
        if polar:
            np_r = rnd((rad_outer - rad_inner) / spacing)
            delta_r = (rad_outer - rad_inner) / np_r
            for ir in range(np_r + 1):
                r = rad_inner + ir * delta_r
                np_th = rnd(2 * math.pi * r / spacing)
                delta_th = (2 * math.pi) / np_th
                for it in range(np_th):
                    theta = it * delta_th
                    x, y = r * math.cos(theta), r * math.sin(theta)
                    out.append(np.stack([np.full_like(zs, x), np.full_like(zs, y), zs], axis=1))
        else:
            np_r = rnd(rad_outer / spacing)
            delta_r = rad_outer / np_r
            r_in2, r_out2 = rad_inner * rad_inner, rad_outer * rad_outer
            for ix in range(-np_r, np_r + 1):
                x = ix * delta_r
                for iy in range(-np_r, np_r + 1):
                    y = iy * delta_r
                    r2 = x * x + y * y
                    if r_in2 <= r2 <= r_out2:
                        out.append(np.stack([np.full_like(zs, x), np.full_like(zs, y), zs], axis=1))
        return np.concatenate(out, axis=0) if out else np.zeros((0, 3))

    def CreatePointsMesh(self, vertices, faces):
        """BCE markers filling a closed triangle mesh (cpp:2627-2714): the grid of spacing d0 anchored at the mesh's bounding-box minimum, every grid
        point that lies inside the surface.  Inside = an odd number of triangle crossings along BOTH of two fixed oblique rays from the point (shifted by
        1e-9 along every axis); a crossing is the Moeller-Trumbore test without culling, |det| >= 1e-6, hit parameter > 1e-6.
        vertices (n, 3) float, faces (m, 3) int.  Returns the points in x-outer / y / z-inner grid order."""
        V = np.asarray(vertices, dtype=np.float64)
        Fc = np.asarray(faces, dtype=np.int64)
        spacing = float(f32(self.p.d0))
        lo, hi = V.min(axis=0), V.max(axis=0)

        def axis(a, b):  # the reference accumulates x += spacing in double
            n = int(np.ceil((b - a) / spacing)) + 2
            g = a + np.concatenate([[0.0], np.cumsum(np.full(n, spacing))])
            return g[g < b]

        gx, gy, gz = axis(lo[0], hi[0]), axis(lo[1], hi[1]), axis(lo[2], hi[2])
        if not (len(gx) and len(gy) and len(gz)) or not len(Fc):
            return np.zeros((0, 3))
        X, Y, Z = np.meshgrid(gx, gy, gz, indexing="ij")
        P = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        v0 = V[Fc[:, 0]]
        e1, e2 = V[Fc[:, 1]] - v0, V[Fc[:, 2]] - v0
        eps = 1e-6
        inside = np.ones(len(P), dtype=bool)
        for d in (np.array([5.0, 0.5, 0.25]), np.array([-3.0, 0.7, 10.0])):
            pvec = np.cross(d, e2)                      # per face
            det = np.einsum("ij,ij->i", e1, pvec)
            ok = ~((det > -eps) & (det < eps))
            inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
            odd = np.zeros(len(P), dtype=bool)
            for c0 in range(0, len(P), 4096):           # points in chunks: (chunk, faces) temporaries
                O = P[c0:c0 + 4096] + 1e-9
                t = O[:, None, :] - v0[None, :, :]
                uu = np.einsum("pfj,fj->pf", t, pvec) * inv
                q = np.cross(t, e1[None, :, :])
                vv = np.einsum("pfj,j->pf", q, d) * inv
                tt = np.einsum("pfj,fj->pf", q, e2) * inv
                hit = ok[None, :] & ~((uu < 0.0) | (uu > 1.0)) & ~((vv < 0.0) | (uu + vv > 1.0)) & (tt > eps)
                odd[c0:c0 + 4096] = (hit.sum(axis=1) % 2) == 1
            inside &= odd
        return P[inside]

    def CreatePointsCylinderInterior(self, rad, height, polar=True):
        """cpp:2225-2320: `num_bce_layers` shells of a solid cylinder (axis z) plus its caps."""
        spacing = float(f32(self.p.d0))
        num_layers = int(self.p.num_bce_layers)
        rnd = lambda v: int(math.floor(v + 0.5))
        np_r = rnd(rad / spacing)
        delta_r = rad / np_r
        np_h = rnd(height / spacing)
        delta_h = height / np_h
        fill = (np_r <= num_layers - 1) or (np_h <= 2 * num_layers - 1)
        out = []
        if polar:
            rad_min = max(rad - (num_layers - 1) * spacing, 0.0)
            if fill:
                rad_min = 0
            np_r = rnd((rad - rad_min) / spacing)
            delta_r = (rad - rad_min) / np_r
            for ir in range(np_r + 1):
                r = rad_min + ir * delta_r
                np_th = max(rnd(2 * math.pi * r / spacing) - 1, 1)
                delta_th = 2 * math.pi / np_th
                for it in range(np_th):
                    theta = it * delta_th
                    x, y = r * math.cos(theta), r * math.sin(theta)
                    for iz in range(np_h + 1):
                        out.append((x, y, -height / 2 + iz * delta_h))
            if not fill:
                np_r = rnd(rad_min / spacing)
                delta_r = rad_min / np_r
                for ir in range(np_r):
                    r = ir * delta_r
                    np_th = max(rnd(2 * math.pi * r / spacing) - 1, 1)
                    delta_th = 2 * math.pi / np_th
                    for it in range(np_th):
                        theta = it * delta_th
                        x, y = r * math.cos(theta), r * math.sin(theta)
                        for iz in range(num_layers):
                            z = height / 2 - iz * delta_h
                            out.append((x, y, -z))
                            out.append((x, y, +z))
            return np.array(out, dtype=np.float64).reshape(-1, 3)
        r_max = rad
        r_min = max(rad - num_layers * delta_r, 0.0)
        if fill:
            r_min = 0
        r_max2, r_min2 = r_max * r_max, r_min * r_min
        for ix in range(-np_r, np_r + 1):
            x = ix * delta_r
            for iy in range(-np_r, np_r + 1):
                y = iy * delta_r
                r2 = x * x + y * y
                if r_min2 <= r2 <= r_max2:
                    for iz in range(np_h + 1):
                        out.append((x, y, -height / 2 + iz * delta_h))
                if not fill and r2 < r_min2:
                    for iz in range(num_layers):
                        z = height / 2 - iz * delta_h
                        out.append((x, y, -z))
                        out.append((x, y, +z))
        return np.array(out, dtype=np.float64).reshape(-1, 3)

    # ---- Initialize (cpp:1350-1509)
    def Initialize(self, body_states=None):
        assert not self.m_is_initialized
        s = capi.Solver(self.api, self.p, self._device) if self.api.prefix == "sph_" else capi.Solver(self.api, self.p)
        s.set_check_errors(self.m_check_errors)
        if self.m_arithmetic is not None:
            s.set_arithmetic(self.m_arithmetic)
        s.set_step_size(self.m_step)
        # posRad.w = h at the time each marker was added (phys/SphDataManager.cu:210,228): the ABI uses the
        # context's current h, so markers are flushed under the h they were created with.
        self._flush_pending(s)
        for local, _st in self._bodies:
            s.add_rigid_body(local)
        states = [st for _l, st in self._bodies] if body_states is None else body_states
        s.initialize(states)
        self._solver = s
        self._pending = []
        self.p = s.params()
        self.m_is_initialized = True

    def _flush_pending(self, s, select=None):
        """Hand the staged marker blocks to a solver context (optionally only the rows select[block] keeps)."""
        saved = capi.SphParams.from_buffer_copy(self.p)
        for b, (type_code, xyz, vel, rpm, h_at_add) in enumerate(self._pending):
            if select is not None:
                keep = select[b]
                xyz = xyz[keep]
                vel = None if vel is None else vel[keep]
                rpm = None if rpm is None else rpm[keep]
            if float(h_at_add) != float(self.p.h):
                tmp = capi.SphParams.from_buffer_copy(self.p)
                tmp.h = h_at_add
                s.set_params(tmp)
            first = int(s.counters_before_init())
            s.add_markers(type_code, xyz, vel, rpm)
            st = self._pending_stress.get(id(self._pending[b][1]))
            if st is not None:
                td, to = st if select is None else (st[0][select[b]], st[1][select[b]])
                s.set_marker_stress(first, td, to)
            pc = self._pending_pc.get(id(self._pending[b][1]))
            if pc is not None:
                s.set_marker_consolidation(first, pc if select is None else pc[select[b]])
            s.set_params(saved)

    def DerivedParams(self):
        """Parameters as ChFsiFluidSystemSPH::Initialize derives them (grid, cell size, marker mass, ...), obtained from a
        throw-away context with no markers."""
        probe = capi.Solver(self.api, self.p, self._device) if self.api.prefix == "sph_" else capi.Solver(self.api, self.p)
        probe.set_step_size(self.m_step)
        probe.initialize(())
        p = probe.params()
        probe.close()
        return p

    def InitializeSlab(self, rank, nranks, cuts=None, body_states=None):
        """Initialize this process as rank `rank` of an x-slab decomposition (see slab.py): every rank stages the same
        global marker set, keeps the markers whose cell column falls in its slab and tags them with their global index.
        FSI bodies are registered on every rank with all their BCE coordinates; a rank stages the rigid markers whose initial
        position lies in its slab (they migrate like any other marker afterwards)."""
        from . import slab
        assert not self.m_is_initialized
        dp = self.DerivedParams()
        cols, sizes, wts = [], [], []
        for (tc, xyz, _v, _r, _h) in self._pending:
            cols.append(slab.cell_column(xyz[:, 0], dp))
            sizes.append(len(xyz))
            wts.append(np.full(len(xyz), 1.0 if tc == capi.SPH_MARKER_FLUID else slab.WALL_MARKER_WEIGHT))
        states = [st for _l, st in self._bodies] if body_states is None else list(body_states)
        rcols = []
        for (local, _st0), st in zip(self._bodies, states):
            R = rotmat_from_quat(np.asarray(st.get("rot", QUNIT), dtype=np.float64))
            rcols.append(slab.cell_column((np.asarray(st["pos"], dtype=np.float64) + local.astype(np.float64) @ R.T)[:, 0], dp))
        rc = np.concatenate(rcols) if rcols else np.zeros(0, dtype=np.int64)
        cx = np.concatenate(cols + [rc]) if cols else rc
        if cuts is None:  # balance the work, not the marker count: walls are cheap rows
            cuts = slab.balanced_cuts(cx, int(dp.gridSize[0]), nranks, weights=np.concatenate(wts + [np.ones(len(rc))]) if len(cx) else None,
                                      column_cost=slab.CELL_WEIGHT * int(dp.gridSize[1]) * int(dp.gridSize[2]))
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        select, gids = [], []
        for b, c in enumerate(cols):
            keep = np.nonzero((c >= cuts[rank]) & (c < cuts[rank + 1]))[0]
            select.append(keep)
            gids.append(offs[b] + keep)
        s = capi.Solver(self.api, self.p, self._device) if self.api.prefix == "sph_" else capi.Solver(self.api, self.p)
        s.set_check_errors(self.m_check_errors)
        if self.m_arithmetic is not None:
            s.set_arithmetic(self.m_arithmetic)
        s.set_step_size(self.m_step)
        self._flush_pending(s, select)
        s.set_global_ids(np.concatenate(gids).astype(np.uint32) if gids else np.zeros(0, dtype=np.uint32))
        if self._bodies:
            for local, _st in self._bodies:
                s.add_rigid_body(local)
            s.slab_rigid_subset(int(offs[-1]), np.nonzero((rc >= cuts[rank]) & (rc < cuts[rank + 1]))[0])
        s.initialize(states)
        self._solver = s
        self._pending = []
        self.p = s.params()
        self.m_is_initialized = True
        self.n_global = int(offs[-1]) + len(rc)
        self.slab_cuts = list(cuts)
        return self.slab_cuts

    # ---- stepping (ChFsiFluidSystem.cpp:58-69)
    def DoStepDynamics(self, step):
        if not self.m_is_initialized:
            raise SphError("FSI system not initialized!")
        self._solver.time = self.m_time
        self._solver.do_step(step)
        self.m_frame += 1
        self.m_time += step

    def GetCurrentStepSize(self):
        h = self._solver.current_step()
        if bool(self.p.use_variable_time_step):
            self._step_log.append(float(h))  # QuantityLogger "time_step" (phys/SphFluidDynamics.cu:105)
        return h

    def GetMarkerDeviceView(self):
        """MarkerDeviceView (sph/ChFsiFluidSystemSPH.h:398-401): borrowed device pointers of the original-order arrays -- dict with
        'pos4', 'vel4', 'rpm4' (integers) and 'n'; valid while this system lives.  `as_torch=True` consumers can wrap them with
        torch.as_tensor through the CUDA array interface (see tests)."""
        if self.api.prefix != "sph_":
            raise SphError("GetMarkerDeviceView: the device view exists on the HIP solver only")
        import ctypes as C
        capi.assert_single_runtime("GetMarkerDeviceView")  # the pointers only mean something to the runtime that allocated them
        a, b, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._solver._ck(self.api.get_device_view(self._solver.ctx, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return {"pos4": a.value or 0, "vel4": b.value or 0, "rpm4": c.value or 0, "n": int(n.value)}

    # ---- diagnostics (cpp:214-216, 330-404, 1767-1774, 1829-1831)
    def SetSPHLinearSolver(self, lin_solver):
        """SolverType: 0 JACOBI (the only solver the reference instantiates, phys/SphForceISPH.cu:901-918), 1 BICGSTAB, 2 GMRES."""
        self.p.LinearSolver = int(lin_solver)

    def CheckSPHParameters(self):
        """The parameter checks of Initialize: incompatible settings raise SphError with the reference's message, questionable ones
        are reported as Python warnings (the reference prints them to stderr).  Returns the list of warnings."""
        import warnings
        p, out = self.p, []
        if p.elastic_SPH:
            if int(p.integration_scheme) == capi.SPH_SCHEME_IMPLICIT_SPH:
                raise SphError("ISPH not supported for granular CRM problems.")
            if int(p.viscosity_method) == capi.SPH_VISC_LAMINAR:
                raise SphError("Viscosity type LAMINAR not supported for CRM granular.")
            if int(p.viscosity_method) == capi.SPH_VISC_ARTIFICIAL_UNILATERAL:
                out.append("Viscosity type ARTIFICIAL_UNILATERAL may be less stable for CRM granular. Consider using ARTIFICIAL_BILATERAL or ensure the step size is small enough.")
            if float(p.mu0) > float(f32(0.001)):
                out.append("The Laminar viscosity parameter has been set to %g but the viscosity model is not laminar. This parameter will have no influence on simulation." % float(p.mu0))
        elif int(p.viscosity_method) == capi.SPH_VISC_ARTIFICIAL_BILATERAL:
            raise SphError("Viscosity type ARTIFICIAL_BILATERAL not supported for CFD.")
        box = [float(p.boxDims[a]) for a in range(3)]
        dmin = [-2 * box[a] - 10 * float(p.h) for a in range(3)]
        dmax = [+2 * box[a] + 10 * float(p.h) for a in range(3)]
        if any(float(p.cMin[a]) < 2 * dmin[a] or float(p.cMax[a]) > 2 * dmax[a] for a in range(3)):
            out.append("User-defined cMin or cMax is much larger than the default values. This may slow down the simulation.")
        if float(p.d0_multiplier) < 1:
            out.append("Kernel interaction length multiplier is less than 1. This may lead to numerical instability due to poor particle approximation.")
        if int(p.kernel_type) == capi.SPH_KERNEL_CUBIC_SPLINE and float(p.d0_multiplier) > 1.5:
            out.append("Kernel interaction radius multiplier is greater than 1.5 and the cubic spline kernel is used. This may lead to pairing instability. You might want to switch to the Wendland kernel.")
        if int(p.num_bce_layers) < 3:
            out.append("Number of BCE layers is less than 3. This may cause insufficient kernel support at the boundaries and lead to leakage of particles")
        sm = int(p.shifting_method)
        eps, push, pull = float(p.shifting_xsph_eps), float(p.shifting_ppst_push), float(p.shifting_ppst_pull)
        if sm == capi.SPH_SHIFT_NONE and (eps != 0.5 or push != 1.0 or pull != 1.0):
            out.append("Shifting method is NONE, but shifting parameters have been modified. These changes will not take effect.")
        elif sm == capi.SPH_SHIFT_XSPH and (pull != 1.0 or push != 3.0):
            out.append("Shifting method is XSPH, but PPST shifting parameters have been modified. These changes will not take effect.")
        elif sm == capi.SPH_SHIFT_PPST and eps != 0.5:
            out.append("Shifting method is PPST, but XSPH shifting parameters have been modified. These changes will not take effect.")
        for w in out:
            warnings.warn("WARNING: " + w, RuntimeWarning, stacklevel=2)
        return out

    def PrintStats(self):
        """QuantityLogger::PrintStats for the quantity this mirror records: the variable time step."""
        if self._step_log:
            a = np.array(self._step_log)
            print("time_step: count %d  min %g  max %g  mean %g" % (len(a), a.min(), a.max(), a.mean()))

    def PrintTimeSteps(self, path):
        """One line per GetCurrentStepSize() call under the variable time step (the reference also logs the two minima it takes the
        step from; they stay on the device here)."""
        with open(path, "w") as f:
            f.write("time_step\n")
            for h in self._step_log:
                f.write("%g\n" % h)

    def WriteParticleFile(self, filename):
        """writeParticleFileCSV (sph/utils/SphUtilsPrint.cu:503-539): SPH particles only, stream formatting (%g), and the tiny offsets
        the reference adds so that a zero prints as 2e-20."""
        pos, vel, rpm = self._solver.positions(), self._solver.velocities(), self._solver.rho_pres_mu_type()
        eps = 1e-20
        with open(filename, "w") as f:
            f.write("x,y,z,v_x,v_y,v_z,|U|,rho,pressure\n")
            for i in np.nonzero(rpm[:, 3] == -1)[0]:
                v = (vel[i].astype(np.float32) + np.float32(1e-20)).astype(np.float32)
                mag = np.float32(np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])))
                f.write("%g, %g, %g, %g, %g, %g, %g, %g, %g\n" % (pos[i, 0], pos[i, 1], pos[i, 2], float(v[0]) + eps, float(v[1]) + eps, float(v[2]) + eps,
                                                                 float(mag) + eps, rpm[i, 0], float(rpm[i, 1]) + eps))

    def OnExchangeSolidForces(self):
        rank = getattr(self, "_slab_rank", None)
        if rank is not None and rank.nranks > 1:
            # collective: every slab rank calls it (native_collectives: inside the library, through the rank's sph_slab_transport)
            (rank.exchange_body_wrenches_native if getattr(rank, "native_collectives", False) else rank.exchange_body_wrenches)()
        else:
            self._solver.compute_body_wrenches()

    def OnExchangeSolidStates(self):
        self._solver.update_body_markers()

    def LoadSolidStates(self, states):
        self._solver.load_body_states(states)

    def GetBodyWrenches(self):
        return self._solver.get_body_wrenches(len(self._bodies))

    def Synchronize(self):
        self._solver.synchronize()

    # ---- accessors (cpp:2801-2871)
    def GetParticlePositions(self):
        return self._solver.positions()

    def GetParticleVelocities(self):
        return self._solver.velocities()

    def GetParticleFluidProperties(self):
        return self._solver.properties()

    def GetParticleAccelerations(self):
        return self._solver.accelerations()

    # ---- remaining accessors of the reference facade (sph/ChFsiFluidSystemSPH.cpp:2728-2925, phys/SphDataManager.cu:560-803)
    def GetPositions(self, indices=None):
        a = self._solver.positions()
        return a if indices is None else a[np.asarray(indices, dtype=np.int64)]

    def GetVelocities(self, indices=None):
        a = self._solver.velocities()
        return a if indices is None else a[np.asarray(indices, dtype=np.int64)]

    def GetAccelerations(self, indices=None):
        """SPH particles only: the first numFluidMarkers entries of the (sorted) derivative array, like the reference."""
        a = self._solver.accelerations()
        return a if indices is None else a[np.asarray(indices, dtype=np.int64)]

    def GetForces(self, indices=None):
        return (self.GetAccelerations(indices) * f32(self.GetParams().markerMass)).astype(np.float32)

    def GetProperties(self):
        return self._solver.properties()

    def GetParticleForces(self):
        return self.GetForces()

    def FindParticlesInBox(self, pos, rot, size):
        """Indices of the SPH particles inside the oriented box (frame = pos, quaternion rot; full edge lengths `size`):
        FsiDataManager::FindParticlesInBox with the in_box predicate (phys/SphDataManager.cu:742-803), in float like the reference."""
        R = rotmat_from_quat(rot).astype(np.float32)  # columns = box axes in the parent frame
        hs = (0.5 * np.asarray(size, dtype=np.float64)).astype(np.float32)
        n = self.GetCounters().numFluidMarkers
        d = self._solver.positions()[:n].astype(np.float32) - np.asarray(pos, dtype=np.float64).astype(np.float32)
        inside = np.ones(n, dtype=bool)
        for a in range(3):
            w = (R[0, a] * d[:, 0] + R[1, a] * d[:, 1]) + R[2, a] * d[:, 2]
            inside &= (w >= -hs[a]) & (w <= hs[a])
        return np.nonzero(inside)[0].astype(np.int32)

    def GetContainerDim(self):
        return (float(self.p.boxDimX), float(self.p.boxDimY), float(self.p.boxDimZ))

    def GetComputationalDomain(self):
        return tuple(float(v) for v in self.p.cMin), tuple(float(v) for v in self.p.cMax)

    def GetBodyForce(self):
        return tuple(float(v) for v in self.p.bodyForce3)

    def GetNumProximitySearchSteps(self):
        return int(self.p.num_proximity_search_steps)

    def GetUseVariableTimeStep(self):
        return bool(self.p.use_variable_time_step)

    def GetPhysicsProblem(self):
        return "CRM" if self.p.elastic_SPH else "CFD"

    def GetSphIntegrationSchemeString(self):
        return {IntegrationScheme.EULER: "WCSPH_EULER", IntegrationScheme.RK2: "WCSPH_RK2", IntegrationScheme.VERLET: "WC_SPH_VERLET",
                IntegrationScheme.SYMPLECTIC: "WCSPH_SYMPLECTIC", IntegrationScheme.IMPLICIT_SPH: "ISPH"}.get(int(self.p.integration_scheme), "")

    def SetCohesionForce(self, Fc):
        self.p.Coh_coeff = Fc

    def EnableGPUErrorCheck(self, on=True):
        self.EnableErrorChecks(on)

    def GetParticleConsolidation(self):
        """CRM / MCC: (n, 3) = preconsolidation pressure, volumetric strain rate, specific volume per marker."""
        return self._solver.consolidation()

    def GetParticleStress(self):
        """CRM: (n, 6) = tauXxYyZz | tauXyXzYz per marker."""
        return self._solver.stress()

    # ---- on-disk output (sph/utils/SphUtilsPrint.cu:36-200, 385-520), STATE_PRESSURE level
    def SetOutputLevel(self, level):
        """OutputLevel (sph/ChFsiDefinitionsSPH.h): STATE, STATE_PRESSURE (default), CRM_FULL."""
        self._output_level = level

    def SaveParticleData(self, directory):
        """fluid<frame>.csv every call, boundary0.csv on the first call, rigidBCE<frame>.csv when bodies exist; header and
        column order of SaveFileCFD for the three output levels, numbers in the iostream default format (6 significant digits)."""
        import os
        frame = getattr(self, "_out_frame", -1) + 1
        self._out_frame = frame
        c = self.GetCounters()
        s = self._solver
        s.synchronize()
        pos, vel, rpm, acc = s.positions(), s.velocities(), s.rho_pres_mu_type(), s.derivatives()

        level = getattr(self, "_output_level", OutputLevel.STATE_PRESSURE)
        h = float(f32(self.p.h))

        def dump(name, lo, hi):
            with open(os.path.join(directory, name), "w") as f:
                vlen = np.sqrt((vel[lo:hi, 0] * vel[lo:hi, 0] + vel[lo:hi, 1] * vel[lo:hi, 1]) + vel[lo:hi, 2] * vel[lo:hi, 2], dtype=np.float32)
                a = acc[lo:hi]
                alen = np.sqrt((a[:, 0] * a[:, 0] + a[:, 1] * a[:, 1]) + a[:, 2] * a[:, 2], dtype=np.float32)
                if level == OutputLevel.STATE:
                    f.write("x,y,z,|U|,acc\n")
                    for k in range(hi - lo):
                        i = lo + k
                        f.write("%g, %g, %g, %g, %g\n" % (pos[i, 0], pos[i, 1], pos[i, 2], vlen[k], alen[k]))
                elif level == OutputLevel.CRM_FULL:
                    # the header names the stress columns, the explicit-scheme writer stops after mu with a trailing separator
                    # (SaveFileCFD, sph/utils/SphUtilsPrint.cu:56-87)
                    f.write("x,y,z,h,v_x,v_y,v_z,|U|,acc,rho(rpx),p11(tauXxYyZz_11),p22(tauXxYyZz_22),p33(tauXxYyZz_33),shear12(tauXyXzYz_12),"
                            "shear13(tauXyXzYz_13),shear23(tauXyXzYz_23)\n")
                    for k in range(hi - lo):
                        i = lo + k
                        f.write("%g, %g, %g, %g, %g, %g, %g, %g, %g, %g, %g, %g, \n" % (pos[i, 0], pos[i, 1], pos[i, 2], h, vel[i, 0], vel[i, 1], vel[i, 2], vlen[k], alen[k],
                                                                                    rpm[i, 0], rpm[i, 1], rpm[i, 2]))
                else:
                    f.write("x,y,z,v_x,v_y,v_z,|U|,acc,rho,pressure\n")
                    for k in range(hi - lo):
                        i = lo + k
                        f.write("%g, %g, %g, %g, %g, %g, %g, %g, %g, %g\n" % (pos[i, 0], pos[i, 1], pos[i, 2], vel[i, 0], vel[i, 1], vel[i, 2], vlen[k], alen[k],
                                                                              rpm[i, 0], rpm[i, 1]))

        nf, nb, nr = c.numFluidMarkers, c.numBoundaryMarkers, c.numRigidMarkers
        dump("fluid%d.csv" % frame, 0, nf)
        if frame == 0:
            dump("boundary0.csv", nf, nf + nb)
        if nr:
            dump("rigidBCE%d.csv" % frame, nf + nb, nf + nb + nr)

    def SaveSolidData(self, directory, time, body_states):
        """FSI_body<i>.csv, one appended line per call: time, position, quaternion, linear velocity, fluid force, torque."""
        import os
        wr = self.GetBodyWrenches() if self._bodies else []
        for i, ((f, t), st) in enumerate(zip(wr, body_states)):
            name = os.path.join(directory, "FSI_body%d.csv" % i)
            if not getattr(self, "_solid_files", False):
                with open(name, "w") as fh:
                    fh.write("Time,x,y,z,q0,q1,q2,q3,Vx,Vy,Vz,Fx,Fy,Fz,Tx,Ty,Tz\n")
            p, q, v = (np.asarray(st.get(k, d), dtype=np.float32) for k, d in (("pos", (0, 0, 0)), ("rot", QUNIT), ("lin_vel", (0, 0, 0))))
            with open(name, "a") as fh:
                fh.write(",".join("%g" % x for x in [time, *p, *q, *v, *f, *t]) + "\n")
        self._solid_files = True

    @property
    def solver(self):
        return self._solver

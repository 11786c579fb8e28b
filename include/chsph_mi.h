/* =============================================================================
 * chsph_mi.h - C ABI of the MI355X-native Chrono::FSI-SPH per-step solver.
 *
 * This is the drop-in boundary for the hot path named by BASELINE.json: it is
 * exactly what a Chrono-side `ChFsiFluidSystem` subclass binds to in order to
 * replace `chrono::fsi::sph::ChFsiFluidSystemSPH`'s device layer
 * (src/chrono_fsi/sph/physics/ *.cu in the reference).  Plain pointers and sizes
 * only; no C++/torch types cross this boundary.  All citations below are
 * relative to /root/reference/src/chrono_fsi/.
 *
 * Ownership: the context owns all device memory; the caller owns every host
 * buffer passed in or out; getters copy into caller memory (the reference
 * getters return fresh std::vector by value: sph/ChFsiFluidSystemSPH.cpp:2801-2871).
 * Errors: functions returning int return 0 on success and non-zero on failure;
 * `sph_last_error` gives the message (the reference throws std::runtime_error
 * after reading a device flag: sph/utils/SphUtilsDevice.cuh:98-136).  The C++
 * facade re-throws std::runtime_error with the same message prefixes.
 * Threading: a context is not thread-safe; one caller thread drives it, exactly
 * as ChFsiSystem::DoStepDynamics drives the fluid system (ChFsiSystem.cpp:264-303).
 * ============================================================================= */
#ifndef CHSPH_MI_H
#define CHSPH_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums: values identical to sph/ChFsiDefinitionsSPH.h:30-79 ------------ */
enum { SPH_SCHEME_EULER = 0, SPH_SCHEME_RK2 = 1, SPH_SCHEME_VERLET = 2, SPH_SCHEME_SYMPLECTIC = 3, SPH_SCHEME_IMPLICIT_SPH = 4 };
enum { SPH_SHIFT_NONE = 0, SPH_SHIFT_PPST = 1, SPH_SHIFT_XSPH = 2, SPH_SHIFT_PPST_XSPH = 3, SPH_SHIFT_DIFFUSION = 4, SPH_SHIFT_DIFFUSION_XSPH = 5 };
enum { SPH_EOS_TAIT = 0, SPH_EOS_ISOTHERMAL = 1 };
enum { SPH_KERNEL_QUADRATIC = 0, SPH_KERNEL_CUBIC_SPLINE = 1, SPH_KERNEL_QUINTIC_SPLINE = 2, SPH_KERNEL_WENDLAND = 3 };
enum { SPH_VISC_LAMINAR = 0, SPH_VISC_ARTIFICIAL_UNILATERAL = 1, SPH_VISC_ARTIFICIAL_BILATERAL = 2 };
enum { SPH_BOUNDARY_ADAMI = 0, SPH_BOUNDARY_HOLMES = 1 };
enum { SPH_BC_NONE = 0, SPH_BC_PERIODIC = 1, SPH_BC_INLET_OUTLET = 2 };
enum { SPH_RHEOLOGY_MU_OF_I = 0, SPH_RHEOLOGY_MCC = 1 }; /* RheologyCRM, sph/ChFsiDefinitionsSPH.h:79 */
enum { SPH_SOLVER_JACOBI = 0, SPH_SOLVER_BICGSTAB = 1, SPH_SOLVER_GMRES = 2 };

/* marker type codes stored in rhoPresMu.w: physics/SphMarkerType.cuh:56-94 */
enum { SPH_MARKER_HELPER = -3, SPH_MARKER_GHOST = -2, SPH_MARKER_FLUID = -1, SPH_MARKER_BCE_WALL = 0, SPH_MARKER_BCE_RIGID = 1 };

/* marker groups for copy-back / getters: physics/SphMarkerType.cuh:33-53 */
enum { SPH_GROUP_FLUID = 0, SPH_GROUP_SOLID = 1, SPH_GROUP_BOUNDARY = 2, SPH_GROUP_NON_FLUID = 3, SPH_GROUP_NON_SOLID = 4, SPH_GROUP_NON_BOUNDARY = 5, SPH_GROUP_ALL = 6 };

/* ---- parameters ---------------------------------------------------------------
 * POD mirror of the subset of `ChFsiParamsSPH` (sph/ChFsiParamsSPH.h:39-202) that
 * the CFD hot path reads, same field names, Real = float.  Fields marked [derived]
 * are (re)computed by sph_initialize following
 * ChFsiFluidSystemSPH::Initialize (sph/ChFsiFluidSystemSPH.cpp:1369-1465);
 * the facade setters keep the others consistent exactly like the reference
 * setters do (sph/ChFsiFluidSystemSPH.cpp:180-328,421-550). */
typedef struct sph_params {
    int32_t integration_scheme;
    int32_t eos_type;
    int32_t viscosity_method;
    int32_t boundary_method;
    int32_t kernel_type;
    int32_t shifting_method;

    int32_t gridSize[3];   /* [derived] */
    float worldOrigin[3];  /* [derived] = cMin */
    float cellSize[3];     /* [derived] */
    float boxDims[3];      /* [derived] = cMax - cMin */

    float d0;
    float ood0;
    float d0_multiplier;
    float h;
    float ooh;
    float h_multiplier;    /* [derived] from kernel type */
    int32_t num_neighbors; /* [derived] */
    float epsMinMarkersDis;
    int32_t num_bce_layers;

    float base_pressure;
    float delta_pressure[3]; /* [derived] = 0 */

    float gravity[3];
    float bodyForce3[3];

    float rho0;
    float invrho0;
    float volume0;
    float markerMass; /* [derived] = rho0 / sum W over 21^3 lattice */
    float mu0;
    float v_Max;
    float shifting_xsph_eps;
    float shifting_ppst_push;
    float shifting_ppst_pull;
    float shifting_beta_implicit;
    float shifting_diffusion_A;
    float shifting_diffusion_AFSM;
    float shifting_diffusion_AFST;

    float dT; /* < 0: take the step from sph_set_step_size (JSON hack, cpp:1369-1374) */

    int32_t use_default_limits;
    int32_t use_init_pressure;
    double pressure_height;
    float binSize0; /* [derived] */

    int32_t density_reinit_steps;
    int32_t use_consistent_gradient_discretization;
    int32_t use_consistent_laplacian_discretization;
    int32_t use_delta_sph;
    float density_delta;

    /* ISPH (integration_scheme == IMPLICIT_SPH) */
    int32_t use_density_based_projection;
    int32_t Pressure_Constraint;
    int32_t LinearSolver;
    float Alpha;
    float LinearSolver_Abs_Tol;
    float LinearSolver_Rel_Tol;
    int32_t LinearSolver_Max_Iter;
    float Max_Pressure;
    float PPE_relaxation;
    int32_t ClampPressure;
    int32_t Conservative_Form;
    int32_t gradient_type;
    int32_t laplacian_type;

    float Cs; /* [derived] = 10 v_Max */
    float L_Characteristic;
    float artificial_viscosity;

    float boxDimX, boxDimY, boxDimZ; /* SetContainerDim; only used for default limits */

    int32_t bc_type[3];
    int32_t x_periodic, y_periodic, z_periodic; /* [derived] */
    int32_t minBounds[3], maxBounds[3];         /* [derived] */
    float cMin[3];
    float cMax[3];

    float bodyActiveDomain[3];
    int32_t use_active_domain;
    float settlingTime;

    int32_t num_proximity_search_steps;
    int32_t use_variable_time_step;

    /* CRM, the continuum granular model (elastic_SPH; ChFsiFluidSystemSPH::SetElasticSPH, sph/ChFsiFluidSystemSPH.cpp:445-476):
     * material properties and the moduli derived from them.  Cs becomes sqrt(K_bulk / rho0) (cpp:1461-1465). */
    int32_t elastic_SPH;
    int32_t rheology_model_crm; /* SPH_RHEOLOGY_MU_OF_I or SPH_RHEOLOGY_MCC */
    float E_young, Nu_poisson;
    float G_shear, INV_G_shear, K_bulk; /* [derived by SetElasticSPH] */
    float ave_diam, mu_I0, mu_fric_s, mu_fric_2, Coh_coeff;
    float free_surface_threshold;
    float mcc_M, mcc_kappa, mcc_lambda, mcc_v_lambda; /* rheology_model_crm == SPH_RHEOLOGY_MCC (modified Cam-Clay) */
} sph_params;

/* Counters (physics/SphDataManager.cuh:201-224), 64-bit like the reference's size_t. */
typedef struct sph_counters {
    uint64_t numFsiBodies;
    uint64_t numFluidMarkers;
    uint64_t numBoundaryMarkers;
    uint64_t numRigidMarkers;
    uint64_t numBceMarkers;
    uint64_t numAllMarkers;
    uint64_t startBoundaryMarkers;
    uint64_t startRigidMarkers;
    uint64_t numActiveParticles;
    uint64_t numExtendedParticles;
} sph_counters;

/* FsiBodyState (ChFsiDefinitions.h:41-48): doubles; pos/lin_* in the global frame.
 * rot = quaternion (e0,e1,e2,e3).  NOTE: with the custom SPH interface the
 * reference loads ang_vel / ang_acc in the BODY-LOCAL frame
 * (sph/ChFsiInterfaceSPH.cpp:46-47) and the device kernels assume that. */
typedef struct sph_body_state {
    double pos[3];
    double rot[4];
    double lin_vel[3];
    double ang_vel[3];
    double lin_acc[3];
    double ang_acc[3];
} sph_body_state;

/* FsiBodyForce (ChFsiDefinitions.h:51-54): force at COM and torque, global frame. */
typedef struct sph_wrench {
    double force[3];
    double torque[3];
} sph_wrench;

typedef struct sph_ctx sph_ctx;

/* ABI guards for foreign-function bindings: version and struct sizes
 * (which: 0 sph_params, 1 sph_counters, 2 sph_body_state, 3 sph_wrench). */
uint32_t sph_abi_version(void); /* 2: sph_slab_transport grew `allreduce` (round 4) */
size_t sph_abi_sizeof(int which);

/* One ROCm runtime per process.  The library needs libamdhip64.so.7; a process that also holds PyTorch can end up with PyTorch's bundled
 * libamdhip64 / libhsa-runtime64 / librccl mapped NEXT to the ROCm installation's when the solver was loaded first (two HIP and two HSA
 * runtimes: device pointers cross between them by accident only, and the process may abort in its exit handlers).  Returns the larger
 * number of distinct copies of libamdhip64 and of librccl mapped in the calling process (1 = healthy; 0 = none / not readable) and
 * writes the paths of all three libraries, one per line, into `out` (may be NULL).  A second libhsa-runtime64 alone is not counted: a
 * profiler (rocprofv3) maps the ROCm installation's HSA before the program starts.  sph_create and sph_slab_transport_rccl refuse to work
 * when the result is > 1.
 * The reference has no counterpart: it links the CUDA runtime statically into one library (sph/CMakeLists.txt). */
int sph_runtime_census(char* out, size_t capacity);

/* Fill `p` with ChFsiFluidSystemSPH::InitParams defaults (sph/ChFsiFluidSystemSPH.cpp:81-176). */
void sph_params_default(sph_params* p);

/* Construct a solver context (replaces `new FsiDataManager(paramsH)`, sph/ChFsiFluidSystemSPH.cpp:72-75).
 * device < 0: current HIP device. */
sph_ctx* sph_create(const sph_params* p, int device);
void sph_destroy(sph_ctx* ctx);
const char* sph_last_error(const sph_ctx* ctx);
/* message of the last failed sph_create (ctx == NULL case) */
const char* sph_create_error(void);

/* Replace the parameter block before sph_initialize (facade setters call this). */
int sph_set_params(sph_ctx* ctx, const sph_params* p);
/* Read back the parameter block (after sph_initialize it includes the derived fields). */
int sph_get_params(const sph_ctx* ctx, sph_params* out);
int sph_get_counters(const sph_ctx* ctx, sph_counters* out);

/* ChFsiFluidSystem::SetStepSize (ChFsiFluidSystem.cpp:45-47) */
int sph_set_step_size(sph_ctx* ctx, double step);
/* m_check_errors (sph/ChFsiFluidSystemSPH.cpp:70): NaN scans + device error flags read every step. */
int sph_set_check_errors(sph_ctx* ctx, int on);

/* Arithmetic of the pair loops (native; no reference counterpart -- the reference's arithmetic is whatever nvcc / hipcc make of
 * its source: IEEE division and sqrt, fused multiply-adds where the compiler contracts, phys/SphForceWCSPH.cu:1586-1637).
 *  SPH_ARITH_EXACT: the reference's expressions in source order, IEEE-rounded division / sqrt, no contraction, its double
 *                   sub-expressions honoured: bit-identical to the CPU oracle.
 *  SPH_ARITH_FAST : hardware reciprocal / reciprocal square root (1 ulp), fused multiply-adds, shared reciprocals, branch-free
 *                   pair body, float instead of the double sub-expressions the reference's `2.` literals imply; same rows,
 *                   lists, traversal order and formulas.  Differs from EXACT by rounding only.  Tolerance contract (tests/
 *                   test_fast_gpu.py): ONE force evaluation from an identical state is within 1e-5 of each output field's
 *                   inf-norm (the north-star tolerance, where it is attainable); over 10 time steps the fields are within
 *                   1e-4 of their inf-norm (chaotic amplification of the rounding differences: the oracle's own source built
 *                   with FMA contraction moves by more, tests/test_oracle_cpu.py); neighbour lists stay bit-exact.  Physics
 *                   combinations the fast kernels do not cover run the exact kernels.
 * Default: SPH_ARITH_FAST (the environment variable CHSPH_ARITH=0 / 1 overrides the default of new contexts).
 * May be changed between steps. */
#define SPH_ARITH_EXACT 0
#define SPH_ARITH_FAST 1
int sph_set_arithmetic(sph_ctx* ctx, int mode);
int sph_get_arithmetic(const sph_ctx* ctx);

/* Append markers in host staging, before sph_initialize.
 *  type_code SPH_MARKER_FLUID : FsiDataManager::AddSphParticle (physics/SphDataManager.cu:209-225)
 *  type_code SPH_MARKER_BCE_WALL: FsiDataManager::AddBceMarker (physics/SphDataManager.cu:227-236)
 * xyz: n*3 floats.  vel: n*3 floats or NULL (zeros).  rho_p_mu: n*3 floats or NULL
 * (rho0, base_pressure, mu0).  Markers must be added fluid -> wall -> rigid
 * (physics/SphDataManager.cuh:195-200). */
int sph_add_markers(sph_ctx* ctx, int type_code, size_t n, const float* xyz, const float* vel, const float* rho_p_mu);

/* Register a rigid FSI body with `n_bce` BCE markers given in body-local coordinates
 * (ChFsiFluidSystemSPH::AddBCEFsiBody, sph/ChFsiFluidSystemSPH.cpp:1327-1346).  Global
 * marker positions/velocities are set from the body state at sph_initialize
 * (SphBceManager::UpdateBodyMarkerStateInitial, physics/SphBceManager.cu:688-704).
 * Returns the body index (>= 0) or a negative error. */
int sph_add_rigid_body(sph_ctx* ctx, size_t n_bce, const float* local_xyz);

/* CRM: initial stress of `n` markers starting at marker `first` of those added so far (before sph_initialize): the
 * tauXxYyZz / tauXyXzYz arguments of ChFsiFluidSystemSPH::AddSPHParticle (sph/ChFsiFluidSystemSPH.cpp:1848-1861).  Markers
 * without a call hold zero stress (AddBoxSPH :1863-1877, BCE markers physics/SphDataManager.cu:226-236). */
int sph_set_marker_stress(sph_ctx* ctx, size_t first, size_t n, const float* tau_xx_yy_zz, const float* tau_xy_xz_yz);
/* CRM with the MCC rheology: preconsolidation pressure `pc` of `n` markers from marker `first` on (the last argument of
 * AddSPHParticle, default 1e3); the library derives the specific volume like FsiDataManager::AddSphParticle
 * (sph/physics/SphDataManager.cu:217-223) from pc, the marker's pressure and the MCC parameters in force at the call.
 * sph_get_consolidation returns (pc, volumetric strain rate, specific volume) per marker. */
int sph_set_marker_consolidation(sph_ctx* ctx, size_t first, size_t n, const float* pc);
int sph_get_consolidation(sph_ctx* ctx, float* out3, size_t n);
/* CRM: stress of every marker in marker order, 6 floats each (xx yy zz xy xz yz). */
int sph_get_stress(sph_ctx* ctx, float* out6, size_t n);
/* Parity hook: x^2.55 as the CRM force kernel evaluates it (crmDvDt's artificial-pressure term; the reference calls pow(),
 * whose last bits depend on the math library, so oracle and device share one +,*,/ sequence instead). */
int sph_debug_pow_2p55(sph_ctx* ctx, const double* x, double* out, size_t n);
/* Parity hook: the device's kernel functions on caller-supplied inputs -- w[i] = W3h(kernel_type, r[i], invh) and grad3[i] =
 * GradW3h(kernel_type, d3[i], invh) (sph/physics/SphGeneral.cuh:76-229; kernel_type as sph_params.kernel_type) -- evaluated by one thread per
 * sample with the arithmetic every force kernel uses.  tests/golden/kernel_functions.npz holds the reference header's own values. */
/* Waves of the last neighbour search served from LDS-staged candidates / by the row-per-lane fall-back (k_neighbors_lds; needs CHSPH_WAVE_ORDER=1,
 * which makes the kernel record a class per wave). */
int sph_debug_search_classes(sph_ctx* ctx, uint32_t* staged, uint32_t* fallback);
int sph_debug_kernel_functions(sph_ctx* ctx, int kernel_type, size_t n, const float* r, const float* d3, float invh, float* w, float* grad3);

/* ChFsiFluidSystemSPH::Initialize(body_states) (sph/ChFsiFluidSystemSPH.cpp:1350-1509):
 * derive parameters, build counters, upload markers, load initial body states. */
int sph_initialize(sph_ctx* ctx, size_t n_bodies, const sph_body_state* states);

/* ChFsiFluidSystemSPH::OnDoStepDynamics(time, step) (sph/ChFsiFluidSystemSPH.cpp:1776-1804) + the
 * m_frame++ of ChFsiFluidSystem::DoStepDynamics (ChFsiFluidSystem.cpp:58-69).
 * Asynchronous with respect to the host unless check_errors is on; see sph_synchronize. */
int sph_step(sph_ctx* ctx, double time, double step);
/* ChFsiFluidSystemSPH::GetCurrentStepSize (sph/ChFsiFluidSystemSPH.cpp:1757-1765) */
double sph_current_step(sph_ctx* ctx);

/* I2SPH only: Jacobi sweeps the last step spent on the velocity-predictor system and on the pressure Poisson system
 * (the `Iteration` counters of SphForceISPH::ForceSPH, sph/physics/SphForceISPH.cu:1024-1042,1133-1165). */
int sph_get_isph_iterations(sph_ctx* ctx, int* velocity_sweeps, int* pressure_sweeps);
/* I2SPH only: calcNormalizedRho_Gi_fillInMatrixIndices (sph/physics/SphGeneral.cu:289-391), the kernel-gradient renormalisation
 * tensor G_i (3x3, row-major, 9 floats per marker) and the unit normals of SphForceISPH::PreProcessor (sph/physics/
 * SphForceISPH.cu:1262-1266).  The reference computes both in every PreProcessor call, but only its non-conservative operator
 * form reads them (calc_A_tensor / calc_L_tensor :1268-1276, SphGeneral.cu:453,496), which the reference API cannot select
 * (Conservative_Form is fixed, sph/ChFsiFluidSystemSPH.cpp:119-147): the pass is therefore off by default and switched on here.
 * With diagnostics on, every PreProcessor call of a step evaluates it like the reference (after calcRho_kernel, before the
 * operator rows); sph_get_isph_g_tensor returns the arrays of the LAST call in cell (sorted) order, numExtendedParticles rows
 * (pair them with sph_get_sorted_index). */
int sph_set_isph_diagnostics(sph_ctx* ctx, int on);
int sph_get_isph_g_tensor(sph_ctx* ctx, float* G9 /* n*9 */, float* normals3 /* n*3 */, size_t n);
/* Run every kernel and copy of this context on the caller's HIP stream (hipStream_t; null = the default stream) instead of
 * the context's own.  Meant for slab runs whose transport enqueues on that same stream (RCCL through torch.distributed):
 * pack -> send/recv -> unpack are then ordered by the stream and sph_slab_halo_pack / sph_slab_field_pack no longer
 * block the host.  The stream must outlive the context. */
int sph_set_stream(sph_ctx* ctx, void* stream);
/* Note on marker-order getters: while every marker is active and the search runs every step the solver keeps its state in
 * cell order across steps and does not scatter it back after each step (NOTEBOOK.md 3.1); sph_get_positions / velocities /
 * properties / derivatives / rho_pres_mu_type / device_view rebuild the marker-order arrays on demand, so their results are
 * the same as with the reference's per-step copy-back, only the cost moves to the call. */
/* SynchronizeCopyStream + compute stream; surfaces deferred device errors. */
int sph_synchronize(sph_ctx* ctx);
/* force a proximity search at the next step (m_force_proximity_search) */
int sph_force_proximity_search(sph_ctx* ctx);

/* ChFsiFluidSystemSPH::LoadSolidStates (sph/ChFsiFluidSystemSPH.cpp:599-612) /
 * ChFsiInterfaceSPH::ExchangeSolidStates (sph/ChFsiInterfaceSPH.cpp:37-53) */
int sph_load_body_states(sph_ctx* ctx, size_t n, const sph_body_state* states);
/* OnExchangeSolidForces -> SphBceManager::Rigid_Forces_Torques (physics/SphBceManager.cu:529-548) */
int sph_compute_body_wrenches(sph_ctx* ctx);
/* ChFsiInterfaceSPH::ExchangeSolidForces D2H part (sph/ChFsiInterfaceSPH.cpp:109-125) */
int sph_get_body_wrenches(sph_ctx* ctx, size_t n, sph_wrench* out);
/* OnExchangeSolidStates (sph/ChFsiFluidSystemSPH.cpp:1812-1821) */
int sph_update_body_markers(sph_ctx* ctx);

/* Original-order accessors (physics/SphDataManager.cu:560-616).  out: n*3 floats, n = numAllMarkers
 * (accelerations: numFluidMarkers, reproducing the reference quirk of reading the SORTED derivVelRho). */
int sph_get_positions(sph_ctx* ctx, float* out, size_t n);
int sph_get_velocities(sph_ctx* ctx, float* out, size_t n);
int sph_get_properties(sph_ctx* ctx, float* out, size_t n); /* rho, p, mu */
int sph_get_accelerations(sph_ctx* ctx, float* out, size_t n);
/* derivVelRhoOriginalD (x,y,z = dv/dt, w = drho/dt), original order: n*4 floats */
int sph_get_derivatives(sph_ctx* ctx, float* out, size_t n);
/* full rhoPresMu incl. type code, original order: n*4 floats */
int sph_get_rho_pres_mu_type(sph_ctx* ctx, float* out, size_t n);

/* Parity hooks (state of the last proximity search, sorted order):
 *  offsets: numExtendedParticles+1 uint32 (numNeighborsPerPart after the in-place scan,
 *           physics/SphCollisionSystem.cu:385-394)
 *  idx    : NNZ uint32 (neighborList, self first: physics/SphCollisionSystem.cu:262-306)
 * Pass NULL buffers to query sizes only. */
int sph_get_neighbor_csr(sph_ctx* ctx, uint32_t* offsets, size_t n_offsets, uint32_t* idx, size_t n_idx, size_t* n_rows, size_t* nnz);
/* gridMarkerIndexD (sorted -> original) and gridMarkerHashD, numExtendedParticles each */
int sph_get_sorted_index(sph_ctx* ctx, uint32_t* sorted_to_original, uint32_t* hash, size_t n);
/* Sorted-order state right after the last step (sortedSphMarkers2_D): pos4, vel3, rhoPresMu4 */
int sph_get_sorted_state(sph_ctx* ctx, float* pos4, float* vel3, float* rpm4, size_t n);

/* Borrowed device pointers (GetMarkerDeviceView, sph/ChFsiFluidSystemSPH.h:398-401): original-order
 * pos4 / vel4(padded) / rhoPresMu4 arrays, valid until sph_destroy. */
int sph_get_device_view(sph_ctx* ctx, const void** pos4, const void** vel4, const void** rpm4, size_t* n);

/* ---- measurement hooks (bench.py; HIP events on the context's compute stream) ---- */
/* Enable per-kernel HIP-event timing of the force kernel; accumulates over steps. */
int sph_profile_enable(sph_ctx* ctx, int on);
/* name: "rhs", "adami", "shift", "neigh_count", "neigh_fill", "sort", "integrate", "step".
 * Returns accumulated milliseconds and launch count since enable. */
int sph_profile_get(sph_ctx* ctx, const char* name, double* total_ms, uint64_t* launches);
/* Average milliseconds an event pair reports around a kernel that does nothing (the bracketing cost inside sph_profile_get's figures). */
int sph_profile_overhead(sph_ctx* ctx, int reps, double* avg_ms);
/* Time `reps` back-to-back launches of the force (RHS) kernel on the current sorted state
 * with HIP events on the compute stream; returns average milliseconds per launch. */
int sph_bench_rhs(sph_ctx* ctx, int reps, double* avg_ms);
/* device-to-device copy ceiling on this box (bytes moved = 2*nbytes per rep) */
int sph_bench_copy(sph_ctx* ctx, size_t nbytes, int reps, double* avg_ms);
/* 128-row tiles of the LDS-staged force kernel after the last search: out[0] staged, [1] taking the gather loop, [2] without work (walls),
 * [3] largest number of LDS slots a staged tile needs.  All zero when the staged kernel is not in use. */
int sph_staged_tile_stats(sph_ctx* ctx, uint32_t out[4]);

/* ---- step phases ----------------------------------------------------------------
 * sph_step(t, h) == sph_step_begin(t); if (sph_step_needs_search) sph_step_search();
 *                   for s in 0..sph_step_num_stages-1: sph_step_stage(s, h);  sph_step_end();
 * i.e. the body of ChFsiFluidSystemSPH::OnDoStepDynamics (sph/ChFsiFluidSystemSPH.cpp:1776-1804) cut where a
 * slab-decomposed run must exchange halos (see below).  Stage s evaluates the force on one sorted copy and
 * advances another: RK2/SYMPLECTIC stage 0 = copy y->y_tmp, Force(y), half step on y_tmp; stage 1 = Force(y_tmp),
 * full step on y; EULER has the single stage Force(y), step y (phys/SphFluidDynamics.cu:115-167).
 * sph_step_stage_target tells which copy a stage advanced (1 = y / sortedSphMarkers2_D, 0 = y_tmp). */
int sph_step_begin(sph_ctx* ctx, double time);
int sph_step_needs_search(const sph_ctx* ctx);
int sph_step_search(sph_ctx* ctx);
int sph_step_num_stages(const sph_ctx* ctx);
int sph_step_stage(sph_ctx* ctx, int stage, double step);
int sph_step_stage_target(const sph_ctx* ctx, int stage);
int sph_step_end(sph_ctx* ctx);

/* ---- multi-GPU slabs (one process per GPU; transport supplied by the caller) --------
 * The reference is single-device; this is the MI355X-native extension of the path: markers are
 * partitioned into slabs of whole cell columns along x, rank r owning columns [x_lo, x_hi).  Every proximity
 * search, markers that left the slab migrate to the neighbour rank and each rank imports the markers of the
 * `halo_cells` (= 2) columns beyond its cuts as ghosts (read, never integrated: activity 0), so that the Adami
 * extrapolation of first-layer ghost BCE markers and the force on every owned marker see complete
 * neighbourhoods.  Between stages the owners refresh the ghosts' state ("halo").  Cells are hashed with the GLOBAL
 * grid and ties inside a cell are broken by the marker's GLOBAL index, so every owned row lists the same
 * neighbours in the same order as the single-device run and all owned results are bit-identical to it.
 * The library only classifies, packs and unpacks; the caller moves the buffers between ranks (RCCL
 * ncclSend/ncclRecv on device pointers, or any other transport) - see chrono-wisc_amd/slab.py.
 *
 * Record formats (little-endian 32-bit words):
 *   marker record (migrate / ghost), 12 words: x y z h | vx vy vz gid(u32) | rho p mu type
 *   halo record, 8 words:                      x y z rho | vx vy vz p                                  */
typedef struct sph_slab_config {
    int32_t rank, nranks;
    int32_t x_lo, x_hi;     /* owned cell columns [x_lo, x_hi) of the global grid; rank 0 also owns x < 0, the last rank x >= gridSize.x */
    int32_t halo_cells;     /* 2 */
} sph_slab_config;
enum { SPH_SLAB_MIGRATE = 0, SPH_SLAB_GHOST = 1, SPH_SLAB_HALO = 2, SPH_SLAB_REFRESH = 6 /* 3..5: I2SPH fields, below */,
       /* CRM (elastic_SPH) companions: whenever records of kind MIGRATE / GHOST / HALO / REFRESH travel, a buffer of the companion kind
        * with the SAME record count and order travels with them (filled by the same call that fills the base kind, consumed
        * by the same apply / unpack call): the stress state of the marker.
        *   MIGRATE_CRM / GHOST_CRM / REFRESH_CRM, 12 words: tauXx tauYy tauZz pc | tauXy tauXz tauYz Ev | Sv 0 0 0
        *   HALO_CRM, 8 words:                               tauXx tauYy tauZz 0  | tauXy tauXz tauYz 0 */
       SPH_SLAB_MIGRATE_CRM = 7, SPH_SLAB_GHOST_CRM = 8, SPH_SLAB_HALO_CRM = 9, SPH_SLAB_REFRESH_CRM = 10 };
enum { SPH_SLAB_LEFT = 0, SPH_SLAB_RIGHT = 1 };
enum { SPH_SLAB_SEND = 0, SPH_SLAB_RECV = 1 };

/* Global marker indices of the markers added so far (before sph_initialize); default: local index. */
int sph_set_global_ids(sph_ctx* ctx, size_t n, const uint32_t* gid);
/* Global indices of the owned markers, in the local original order used by every getter (n = numAllMarkers). */
int sph_get_global_ids(sph_ctx* ctx, uint32_t* out, size_t n);
/* After sph_initialize: set the owned column range.  With a periodic x axis the ranks form a ring (rank 0 and rank nranks - 1 exchange across the
 * seam: side LEFT of rank 0 is rank nranks - 1); three ranks or more, and x_hi - x_lo + 2 halo_cells <= gridSize.x. */
int sph_slab_configure(sph_ctx* ctx, const sph_slab_config* cfg);
/* FSI bodies under slabs (explicit schemes and, since round 4, I2SPH).  Every rank registers every body with ALL its BCE coordinates
 * (sph_add_rigid_body) and holds every body state (sph_load_body_states), but stages only the rigid markers whose initial
 * position lies in its slab: before sph_initialize, sph_slab_rigid_subset(gid_first_rigid = global index of rigid marker 0 =
 * number of fluid + wall markers of the whole scene, idx[n] = indices into the concatenated BCE list).  Rigid markers then
 * migrate and serve as ghosts like any other marker; a rank finds "its" rigid markers by global index (in marker order
 * under the two-round preamble, among the cell-ordered rows under the one-round preamble).
 * Wrenches (Rigid_Forces_Torques, phys/SphBceManager.cu:307-386,529-548), collective:
 *   1. sph_slab_wrench_table: per-marker (F.xyz, 0, T.xyz, 0) of the rigid markers this rank owns in a table over all rigid
 *      markers, zero bits elsewhere;
 *   2. the caller sums the table over all ranks AS 32-BIT INTEGERS (ncclAllReduce(ncclInt32, ncclSum): one rank contributes
 *      per entry, so the sum reproduces its bits exactly);
 *   3. sph_slab_wrench_reduce: the single-device block reduction on the summed table; sph_get_body_wrenches as usual.
 * Every rank ends up with the same wrenches, bit-identical to the single-device run's, and advances its own copy of the
 * bodies (or rank 0 broadcasts the new states): no further collective. */
int sph_slab_rigid_subset(sph_ctx* ctx, uint32_t gid_first_rigid, size_t n, const uint32_t* idx);
int sph_slab_wrench_table(sph_ctx* ctx, void** ptr, size_t* nbytes, int* on_device);
int sph_slab_wrench_reduce(sph_ctx* ctx);
/* Proximity-search preamble of a slab run, two exchange rounds (a marker that just crossed a cut must be owned by
 * its new rank before that rank decides which markers it exports as ghosts):
 *   1. sph_slab_classify: drop last search's ghosts, find the owned markers whose cell column left [x_lo, x_hi);
 *      their records sit in the MIGRATE send buffers; counts = { to left, to right }.
 *   2. caller exchanges the counts and the MIGRATE records with both neighbours.
 *   3. sph_slab_apply_migration(recv = { from left, from right }): removes the emigrants, appends the immigrants,
 *      then packs the owned markers lying in the halo_cells columns next to each cut into the GHOST send
 *      buffers; ghost_counts = { to left, to right }.
 *   4. caller exchanges the counts and the GHOST records.
 *   5. sph_slab_apply_ghosts(recv = { from left, from right }): appends the ghosts after the owned markers. */
int sph_slab_classify(sph_ctx* ctx, uint32_t counts[2]);
int sph_slab_apply_migration(sph_ctx* ctx, const uint32_t recv_counts[2], uint32_t ghost_counts[2]);
int sph_slab_apply_ghosts(sph_ctx* ctx, const uint32_t recv_counts[2]);
/* Address of a transport buffer.  dir SEND: filled by the calls above / sph_slab_halo_pack.  dir RECV: sized for
 * `nrecords` records which the caller fills before the matching apply / sph_slab_halo_unpack.  on_device tells
 * whether ptr is a device (HIP) or host pointer.  For SPH_SLAB_HALO nrecords is implied by the last ghost exchange
 * (SEND: ghosts exported to `side`; RECV: ghosts imported from `side`) and is returned through nbytes. */
int sph_slab_buffer(sph_ctx* ctx, int dir, int kind, int side, size_t nrecords, void** ptr, size_t* nbytes, int* on_device);
/* Pack the state of the markers exported as ghosts to `side` (from sorted copy y: 1 = y, 0 = y_tmp) / overwrite the
 * ghosts imported from `side` with the received records. */
int sph_slab_halo_pack(sph_ctx* ctx, int y, int side);
int sph_slab_halo_unpack(sph_ctx* ctx, int y, int side);

/* One-round preamble (explicit schemes with a proximity search every step, from the second step on).  A rank's owned
 * rows are the rows of its cell columns, so a marker that crossed a cut simply stays behind as a ghost of its old rank and
 * travels to the new owner in the SAME message as the ghost layer; the receiver tells owned from ghost records by their
 * column.  The product keeps the state resident in cell order across steps (no round trip through marker order), which is
 * why getters under slabs return the owned markers in an order that may change from step to step (pair them with
 * sph_get_global_ids).  Halo messages list the rows of the exported columns in cell order -- the same order on both ranks.
 *   sph_slab_one_round: 1 if this step must use the one-round preamble (same answer on every rank), 0 = two rounds above.
 *   1. sph_slab_refresh_pack: REFRESH send buffers = emigrants + owned rows of the two edge columns; counts per side.
 *   2. caller exchanges the counts and the REFRESH records with both neighbours.
 *   3. sph_slab_refresh_apply(recv = { from left, from right }), then sph_step_begin / sph_step_search as usual.
 * A marker that crosses more than the two ghost columns within one step cannot be handled in one round: the receiving
 * rank fails loudly (sph_step_search returns an error). */
int sph_slab_one_round(const sph_ctx* ctx);
int sph_slab_refresh_pack(sph_ctx* ctx, uint32_t counts[2]);
int sph_slab_refresh_apply(sph_ctx* ctx, const uint32_t recv_counts[2]);

/* ---- the whole slab step inside the library (explicit schemes and I2SPH) --------------------------------------------
 * sph_slab_step runs one OnDoStepDynamics of a slab rank: preamble (one round when sph_slab_one_round says so, else the two
 * rounds above), sph_step_begin / _search, the stages with the halo exchange between them, sph_step_end -- the sequence of
 * chrono-wisc_amd/slab.py::SlabRank.step, without a host-language call per phase.  The caller supplies the transport:
 *   exchange        moves the send buffers to the slab neighbours and fills the receive buffers; side 0 = rank - 1, side 1 =
 *                   rank + 1; a side without neighbour or without data has nbytes 0.  on_device: the pointers are memory of
 *                   this context's GPU.  `stream` is the context's HIP stream.
 *   exchange_counts two record counts to / from the neighbours, blocking (small host integers).
 *   stream_ordered  1: `exchange` only ENQUEUES on `stream` (or on a stream it orders against it) and returns; the library then
 *                   never waits for the GPU around an exchange (RCCL).  0: `exchange` is a host-side copy that completes before
 *                   it returns; the library synchronises the stream before calling it.
 * Overlap (stream-ordered transports, RK2 / symplectic, update fused into the force kernel): the force kernel of a stage is launched
 * twice -- first the waves (64 rows of the cell order) that hold a row of the halo, sent or received, then the others -- and the halo
 * of the stage's target is packed, exchanged and unpacked on a second stream of the context behind an event recorded between the two
 * launches; the main stream waits for it before the next stage.  `stream` handed to `exchange` is then that second stream.  The two
 * launches run every wave exactly once over the same lists: results do not depend on the split.  CHSPH_SLAB_NO_OVERLAP=1 (read by
 * sph_create) keeps the halo behind the whole kernel on the main stream; sph_slab_overlapped_halos counts the halos that took the
 * overlapped path.
 *   allreduce       in-place reduction of `nwords` 32-bit words over ALL ranks (op: SPH_SLAB_REDUCE_*), same stream rules as `exchange`.
 *                   Needed by I2SPH steps (residual record of every Jacobi sweep, MaxVel), sph_slab_body_wrenches and
 *                   sph_slab_current_step; may be NULL otherwise.
 * I2SPH (IMPLICIT_SPH): the phases of "I2SPH under slabs" below with their ghost exchanges, and per Jacobi sweep one sweep kernel,
 * the ghosts of the iterate and one max-reduction of the 8-KB residual record (loops of sph/physics/SphForceISPH.cu:1026-1042,
 * 1135-1164); with a stream-ordered transport the host only looks at the device-side loop condition every 16th sweep.
 * Built-in transports (both stream-ordered; release with sph_slab_transport_free, or let sph_destroy of their context do it -- a
 * transport must not be used after its context is gone, and freeing one twice or after sph_destroy is a no-op):
 *   sph_slab_transport_rccl    one process per GPU: one ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd per exchange and
 *                              ncclAllReduce per reduction on the stream handed over, over a communicator of one rank per GPU
 *                              (nccl_unique_id: the 128 bytes of ncclGetUniqueId from rank 0, distributed by the caller); counts
 *                              as an 8-byte device message + one stream synchronisation.  NULL on failure (sph_last_error).
 *   sph_slab_transport_inproc  the ranks are THREADS of one process (one context each, on one GPU or on several with peer
 *                              access), members of a sph_slab_group: the receiver copies the sender's buffer with
 *                              hipMemcpyAsync on its own stream behind an event the sender recorded on its stream; host
 *                              threads only meet to hand over pointers and events (CHSPH_INPROC_TIMEOUT seconds, default
 *                              120, then every member fails).  All ranks must call sph_slab_step concurrently. */
enum { SPH_SLAB_REDUCE_MAX_U32 = 0, SPH_SLAB_REDUCE_SUM_I32 = 1, SPH_SLAB_REDUCE_MIN_F32 = 2 };
typedef struct sph_slab_transport {
    void* user;
    int32_t rank, nranks;
    int32_t stream_ordered;
    int (*exchange)(void* user, const void* const send[2], const size_t send_bytes[2], void* const recv[2], const size_t recv_bytes[2], int on_device, void* stream);
    int (*exchange_counts)(void* user, const uint32_t send[2], uint32_t recv[2]);
    int (*allreduce)(void* user, void* buf, size_t nwords, int op, int on_device, void* stream);
} sph_slab_transport;
typedef struct sph_slab_group sph_slab_group;
int sph_slab_step(sph_ctx* ctx, double time, double step, const sph_slab_transport* transport);
size_t sph_slab_rccl_unique_id(void* out, size_t capacity); /* rank 0: fills the id (128 bytes), returns its size; 0 on failure */
sph_slab_transport* sph_slab_transport_rccl(sph_ctx* ctx, const void* nccl_unique_id, size_t id_bytes, int rank, int nranks);
sph_slab_group* sph_slab_group_create(int nranks);  /* at most 16 ranks; NULL otherwise */
int sph_slab_group_destroy(sph_slab_group* group);  /* 1 (and nothing happens) while transports of the group are still alive */
sph_slab_transport* sph_slab_transport_inproc(sph_ctx* ctx, sph_slab_group* group, int rank);
void sph_slab_transport_free(sph_slab_transport* transport);
uint64_t sph_slab_overlapped_halos(const sph_ctx* ctx);
/* OnExchangeSolidForces of a slab rank (collective): sph_slab_wrench_table, the integer sum over the ranks through
 * transport->allreduce, sph_slab_wrench_reduce -- then sph_get_body_wrenches as on one device. */
int sph_slab_body_wrenches(sph_ctx* ctx, const sph_slab_transport* transport);
/* GetCurrentStepSize of a slab run (collective when the variable step is on): minimum over the ranks of the two step candidates,
 * reduced on the device; every rank gets the single-device value.  < 0 on failure. */
double sph_slab_current_step(sph_ctx* ctx, const sph_slab_transport* transport);


/* ---- I2SPH under slabs -------------------------------------------------------------------------------------------
 * SphForceISPH::ForceSPH (sph/physics/SphForceISPH.cu:953-1235) cut at the points where owned rows read per-marker values
 * of ghost markers that only the owner can compute.  One step of a slab rank (sph_slab_step runs exactly this; chrono-wisc_amd/slab.py
 * drives the same calls from Python):
 *   phase DENSITY1  -> max-reduce MAXVEL over ranks, exchange field RHO_VOL
 *   phase VSYSTEM   -> for t in 0..Max_Iter-1: sph_isph_sweep(0, t); exchange field VSTAR(t); max-reduce RESIDUAL(t)
 *   phase PSYSTEM   -> for t in 0..Max_Iter-1: sph_isph_sweep(1, t); exchange field Q(t);     max-reduce RESIDUAL(t)
 *   phase CORRECT   -> exchange the halo state (sph_slab_halo_pack / _unpack, y = 1)
 *   phase DENSITY2  -> exchange field RHO_VOL
 *   phase SHIFT     -> exchange the halo state
 *   phase CLAMP     -> (ClampPressure only) MIN-reduce MINPRESSURE over ranks: one float, ordinary float comparison
 *   phase FINISH
 * The loop condition of the Jacobi solves is evaluated on the device from the (reduced) residual record of the previous
 * sweep; sweeps launched after it closed do nothing, so the caller may look at it only now and then (sph_isph_sweep_ran,
 * which synchronises).  On a single device sph_step runs the same phases back to back.                                */
enum { SPH_ISPH_DENSITY1 = 0, SPH_ISPH_VSYSTEM = 1, SPH_ISPH_PSYSTEM = 2, SPH_ISPH_CORRECT = 3, SPH_ISPH_DENSITY2 = 4, SPH_ISPH_SHIFT = 5, SPH_ISPH_FINISH = 6,
       SPH_ISPH_CLAMP = 7 /* runs between SHIFT and FINISH: the smallest fluid pressure of the rank (ClampPressure, sph/physics/SphForceISPH.cu:1223) */ };
enum { SPH_ISPH_REDUCE_MAXVEL = 0, SPH_ISPH_REDUCE_RESIDUAL = 1, SPH_ISPH_REDUCE_MINPRESSURE = 2 /* one float, MIN over the ranks */ };
/* per-ghost-marker fields: RHO_VOL 2 words (density, volume), VSTAR 4 words (x y z 0), Q 1 word */
enum { SPH_FIELD_RHO_VOL = 3, SPH_FIELD_VSTAR = 4, SPH_FIELD_Q = 5 };
int sph_isph_phase(sph_ctx* ctx, int phase, double step);
int sph_isph_sweep(sph_ctx* ctx, int which /* 0 velocity system, 1 pressure system */, uint32_t t);
int sph_isph_sweep_ran(sph_ctx* ctx, int which, uint32_t t); /* 1: sweep t ran, 0: the loop had ended, -1: error */
/* 32-bit words to reduce over all ranks, in place: MAXVEL and RESIDUAL by max as non-negative integers, MINPRESSURE by min as a float */
int sph_isph_reduce_buffer(sph_ctx* ctx, int what, uint32_t t, void** ptr, size_t* nwords, int* on_device);
/* Like sph_slab_halo_pack / _unpack for one field; buffers through sph_slab_buffer(dir, kind = field, side).  `t` is the
 * sweep just executed (VSTAR, Q: selects the ping-pong buffer it wrote); ignored for RHO_VOL. */
/* side < 0: both sides in one call (one launch sequence, one synchronisation) */
int sph_slab_field_pack(sph_ctx* ctx, int field, uint32_t t, int side);
int sph_slab_field_unpack(sph_ctx* ctx, int field, uint32_t t, int side);

/* ---- environment switches (read once, by sph_create unless stated) --------------------------------------------------------------
 * None of them is needed in production; they select measured alternatives for A/B runs and force rarely taken paths in tests.
 *   CHSPH_ARITH=0|1            initial arithmetic of a context: 0 SPH_ARITH_EXACT, 1 SPH_ARITH_FAST (default; sph_set_arithmetic overrides)
 *   CHSPH_NO_RESIDENT=1        always the reference sequence (hash from marker order, copy-back every step) instead of the sorted-resident state
 *   CHSPH_NO_CELL_WINDOW=1     the resident re-sort zeroes and scans the whole grid instead of the window of cells around the markers
 *   CHSPH_NO_SCAN_SINGLE=1     the resident re-sort zeroes the histogram and scans it with three launches (k_zero_cells, k_scan_reduce, k_scan_apply) instead of
 *                              one (k_scan_window: histogram zeroed behind the scan's reads, block totals kept by k_resident_hash)
 *   CHSPH_SCAN_PREFIX_MIN=<n>  k_scan_window's blocks read the prefix of the block totals from a one-block pass (k_coarse_prefix) when the window has more than
 *                              n blocks of 1024 cells (default 2048; 0 forces the pass in tests), and add the totals up themselves below that
 *   CHSPH_EAGER_MAP=1          the resident re-sort writes mapO2S even when nothing can read it (no rigid markers, no CRM columns, no slabs)
 *   CHSPH_EAGER_SXT=1          every re-sort writes the (x, y, z, type) copy sXT, which only k_neighbors_sell / _sell16 read (built on demand otherwise)
 *   CHSPH_NO_DEFER=1           wait for the search's verdict right after the search instead of behind the speculatively queued first stage
 *   CHSPH_NO_FUSE=1            explicit update as its own kernel (k_integrate) instead of the force kernel's epilogue
 *   CHSPH_NO_ADAMI_FAST=1      k_adami_bc (one lane per row, IEEE sqrt) also under SPH_ARITH_FAST
 *   CHSPH_SEARCH_LDS=0|1|2     neighbour search: 0 k_neighbors_soa16 for every wave, 1 k_neighbors_lds with in-kernel fall-back (default where a
 *                              row run fits 29 candidates), 2 fall-back waves in a second launch
 *   CHSPH_EXACT_ALWAYS_WRAPS=1 k_force (exact arithmetic) evaluates Distance()'s half-box wrap for every wave instead of only for waves with a row in
 *                              the first / last cell layer of a wrapped axis (same bits either way)
 *   CHSPH_NO_PACKED_YWRAP=1    search waves with a row in the first / last cell layer of a periodic y axis take the scalar scan (one candidate at a time with
 *                              Distance()'s selects) instead of the packed scan with the y selects written out per component (same bits)
 *   CHSPH_NO_XPERIODIC_TILES=1 a periodic x axis (one device) searches with the cell-by-cell kernel for every row instead of the tile kernels
 *   CHSPH_NO_CRM_FUSE=1        CRM, RK2 / symplectic: the update as its own launch (k_crm_integrate) instead of the right-hand-side kernel's epilogue
 *   CHSPH_NO_CRM_FAST=1        CRM: k_crm_rhs (the exact right-hand side) also under SPH_ARITH_FAST
 *   CHSPH_NO_ADAMI_WET=1       the second stage's k_adami_fast runs over every row again instead of the list of BCE rows with a fluid neighbour that the first stage appended
 *   CHSPH_NO_ADAMI_PUBLISH=1   inside sph_step the search's verdict words always go through k_publish_words instead of block 0 of the boundary kernel queued behind the search
 *   CHSPH_VERDICT_COPY=1       the search's verdict words through hipMemcpyAsync instead of a one-wave kernel storing into mapped pinned memory
 *   CHSPH_SEARCH_LDS_ONE_WORD=1  k_neighbors_lds with one mask word per run only (up to 29 candidates per row and run: d0_multiplier 1); denser lattices then
 *                              take k_neighbors_soa16 instead of the two-word form (61 candidates: the default multiplier 1.2)
 *   CHSPH_WAVE_ORDER=1         k_neighbors_lds deals the fall-back waves of the last search into their own workgroups instead of walking its waves in
 *                              tile order (measured on C2: -2 us per search, +84 us for the ordering kernel every 16th search: off by default)
 *   CHSPH_NO_SEARCH_SOA=1      k_neighbors_sell16 (scalar scan of (x, y, z, type) records) instead of the packed component-array scan
 *   CHSPH_TILE32=1             32-bit list tile in the search from the start (otherwise after the first ERR_SEG_RANGE)
 *   CHSPH_TIGHT_CAPACITY=1     no slack in the marker arrays: every growth path runs (tests)
 *   CHSPH_EXP=<bits>           bit 0: keep the first stage's derivative store inside sph_step; bit 3: force kernels walk their tiles forwards;
 *                              diagnostics in front of every force launch: bit 4 streams the neighbour lists through the caches, bit 5 the gathered state, bit 6 leaves the GPU idle for ~25 us
 *   CHSPH_SLAB_TWO_ROUND=1     slab ranks always use the two-round preamble (migrate, then ghosts) instead of the one-round resident refresh
 *   CHSPH_SLAB_FULL_LAYER=1    two ghost columns everywhere instead of the thin layer
 *   CHSPH_SLAB_GLOBAL_GRID=1   slab ranks hash on the whole grid instead of their column window
 *   CHSPH_SLAB_NO_OVERLAP=1    sph_slab_step: inter-stage halo behind the whole force kernel on the main stream
 *   CHSPH_INPROC_TIMEOUT=<s>   sph_slab_group_create: seconds a member waits for its neighbours before the whole group fails (default 120)
 *   CHSPH_BENCH_THRASH=<MiB>   sph_bench_rhs: overwrite that much unrelated memory between two launches (cold-cache timing; read at the call)
 * Variant builds only (tools/build_variant.sh): CHSPH_FORCE_STAGED=1 selects k_force_staged in a library built with -DCHSPH_WITH_STAGED_FORCE.
 * Python binding (chrono-wisc_amd/): CHSPH_RUNTIME=system (capi: keep the ROCm installation's HIP runtime instead of torch's bundled one),
 * CHSPH_NO_HOST_MAILBOX, CHSPH_SLAB_OWN_STREAM, CHSPH_SLAB_CALLBACK_TRANSPORT (slab.py: collective-library counts / the solver's own stream /
 * Python callbacks instead of the built-in RCCL transport); tools honour CHSPH_LIB=<path to a variant library>. */

#ifdef __cplusplus
}
#endif
#endif /* CHSPH_MI_H */

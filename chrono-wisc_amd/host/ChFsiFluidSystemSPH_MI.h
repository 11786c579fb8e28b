// =============================================================================
// ChFsiFluidSystemSPH_MI.h - C++ host side of the MI355X SPH solver.
//
// Chrono-free mirror of the reference's fluid-solver plug-in interface for the hot path:
//   chrono::fsi::ChFsiFluidSystem           (src/chrono_fsi/ChFsiFluidSystem.h:37-152)
//   chrono::fsi::sph::ChFsiFluidSystemSPH   (src/chrono_fsi/sph/ChFsiFluidSystemSPH.h:48-717)
// with the same method names, argument meaning and error behaviour (std::runtime_error with the
// reference's message prefixes).  It owns no GPU code: every call forwards to the C ABI of
// include/chsph_mi.h.  The handful of Chrono core value types the facade signatures use
// (ChVector3d, ChQuaterniond, ChFramed, ChAABB) are restated here as minimal PODs so that the
// class compiles without Chrono/Eigen; INTEGRATION.md shows the adapter that derives from the real
// chrono::fsi::ChFsiFluidSystem where Chrono is installed.
// =============================================================================
#ifndef CH_FSI_FLUID_SYSTEM_SPH_MI_H
#define CH_FSI_FLUID_SYSTEM_SPH_MI_H

#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/chsph_mi.h"

namespace chrono_mi {

// ---- minimal stand-ins for the Chrono core value types used in the facade signatures ----
struct ChVector3d {
    double v[3];
    ChVector3d(double x = 0, double y = 0, double z = 0) : v{x, y, z} {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    ChVector3d operator+(const ChVector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    ChVector3d operator-(const ChVector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    ChVector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    ChVector3d operator/(double s) const { return {v[0] / s, v[1] / s, v[2] / s}; }
};
struct ChVector2d {
    double v[2];
    ChVector2d(double x = 0, double y = 0) : v{x, y} {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
};
struct ChVector3i {
    int v[3];
    ChVector3i(int x = 0, int y = 0, int z = 0) : v{x, y, z} {}
    int x() const { return v[0]; }
    int y() const { return v[1]; }
    int z() const { return v[2]; }
};
struct ChQuaterniond {
    double e[4];
    ChQuaterniond(double e0 = 1, double e1 = 0, double e2 = 0, double e3 = 0) : e{e0, e1, e2, e3} {}
};
static const ChQuaterniond QUNIT(1, 0, 0, 0);
ChQuaterniond QuatFromAngleAxis(double angle, const ChVector3d& axis);  // src/chrono/core/ChRotation.cpp:149-163
inline ChQuaterniond QuatFromAngleX(double a) { return QuatFromAngleAxis(a, {1, 0, 0}); }
inline ChQuaterniond QuatFromAngleY(double a) { return QuatFromAngleAxis(a, {0, 1, 0}); }
inline ChQuaterniond QuatFromAngleZ(double a) { return QuatFromAngleAxis(a, {0, 0, 1}); }
struct ChFramed {
    ChVector3d pos;
    ChQuaterniond rot;
    ChFramed(const ChVector3d& p = ChVector3d(), const ChQuaterniond& q = QUNIT) : pos(p), rot(q) {}
    ChVector3d TransformPointLocalToParent(const ChVector3d& p) const;  // src/chrono/core/ChFrame.h:243 + ChMatrix33.h:253-274
};
struct ChAABB {
    ChVector3d min, max;
    ChAABB(const ChVector3d& a = ChVector3d(), const ChVector3d& b = ChVector3d()) : min(a), max(b) {}
};

namespace fsi {

// ---- ChFsiDefinitions.h:41-54 ----
struct FsiBodyState {
    ChVector3d pos;
    ChQuaterniond rot;
    ChVector3d lin_vel, ang_vel, lin_acc, ang_acc;
};
struct FsiBodyForce {
    ChVector3d force, torque;
};

/// Base class for an FSI-aware fluid solver (ChFsiFluidSystem.h:37-152).
class ChFsiFluidSystem {
  public:
    virtual ~ChFsiFluidSystem() {}
    void SetVerbose(bool verbose) { m_verbose = verbose; }
    virtual void SetGravitationalAcceleration(const ChVector3d& gravity) = 0;
    void SetStepSize(double step) { m_step = step; }
    virtual void Initialize();
    void DoStepDynamics(double step);  // ChFsiFluidSystem.cpp:58-69
    double GetSimTime() const { return m_time; }
    double GetStepSize() const { return m_step; }
    double GetRtf() const { return m_RTF; }
    double GetTimerStep() const { return m_timer_step; }
    virtual void OnDoStepDynamics(double time, double step) = 0;
    virtual void OnExchangeSolidForces() = 0;
    virtual void OnExchangeSolidStates() = 0;
    virtual void LoadSolidStates(const std::vector<FsiBodyState>& body_states) = 0;
    virtual void StoreSolidForces(std::vector<FsiBodyForce> body_forces) = 0;
    virtual double GetCurrentStepSize() { return GetStepSize(); }
    virtual void Initialize(const std::vector<FsiBodyState>& body_states) = 0;

  protected:
    ChFsiFluidSystem() : m_verbose(false), m_is_initialized(false), m_time(0), m_step(-1), m_frame(0), m_timer_step(0), m_RTF(0) {}
    bool m_verbose;
    bool m_is_initialized;
    double m_time, m_step;
    unsigned int m_frame;
    double m_timer_step, m_RTF;
};

namespace sph {

// ---- ChFsiDefinitionsSPH.h:30-110 (values == the SPH_* constants of the C ABI) ----
enum class IntegrationScheme { EULER, RK2, VERLET, SYMPLECTIC, IMPLICIT_SPH };
enum class ShiftingMethod { NONE, PPST, XSPH, PPST_XSPH, DIFFUSION, DIFFUSION_XSPH };
enum class EosType { TAIT, ISOTHERMAL };
enum class KernelType { QUADRATIC, CUBIC_SPLINE, QUINTIC_SPLINE, WENDLAND };
enum class ViscosityMethod { LAMINAR, ARTIFICIAL_UNILATERAL, ARTIFICIAL_BILATERAL };
enum class BoundaryMethod { ADAMI, HOLMES };
enum class RheologyCRM { MU_OF_I, MCC };
enum class BCType { NONE, PERIODIC, INLET_OUTLET };
enum class SolverType { JACOBI, BICGSTAB, GMRES };  // sph/ChFsiDefinitionsSPH.h; the reference's I2SPH path only runs JACOBI
struct BoundaryConditions {
    BCType x, y, z;
};
constexpr BoundaryConditions BC_NONE = {BCType::NONE, BCType::NONE, BCType::NONE};
constexpr BoundaryConditions BC_X_PERIODIC = {BCType::PERIODIC, BCType::NONE, BCType::NONE};
constexpr BoundaryConditions BC_Y_PERIODIC = {BCType::NONE, BCType::PERIODIC, BCType::NONE};
constexpr BoundaryConditions BC_Z_PERIODIC = {BCType::NONE, BCType::NONE, BCType::PERIODIC};
constexpr BoundaryConditions BC_ALL_PERIODIC = {BCType::PERIODIC, BCType::PERIODIC, BCType::PERIODIC};

/// SPH fluid system on the MI355X solver (mirror of ChFsiFluidSystemSPH, CFD problems: WCSPH and I2SPH).
class ChFsiFluidSystemSPH : public ChFsiFluidSystem {
  public:
    struct ElasticMaterialProperties {  // sph/ChFsiFluidSystemSPH.cpp:430-443 (CRM granular material)
        double density, Young_modulus, Poisson_ratio, mu_I0, mu_fric_s, mu_fric_2, average_diam, cohesion_coeff;
        RheologyCRM rheology_model;
        double mcc_M, mcc_kappa, mcc_lambda, mcc_v_lambda;
        ElasticMaterialProperties()
            : density(1000), Young_modulus(1e6), Poisson_ratio(0.3), mu_I0(0.03), mu_fric_s(0.7), mu_fric_2(0.7), average_diam(0.005), cohesion_coeff(0),
              rheology_model(RheologyCRM::MU_OF_I), mcc_M(0), mcc_kappa(0), mcc_lambda(0), mcc_v_lambda(2.0) {}
    };
    struct FluidProperties {  // sph/ChFsiFluidSystemSPH.cpp:419
        double density, viscosity, char_length;
        FluidProperties() : density(1000), viscosity(0.1), char_length(1) {}
    };
    struct SPHParameters {  // sph/ChFsiFluidSystemSPH.cpp:478-505
        IntegrationScheme integration_scheme;
        double initial_spacing, d0_multiplier, max_velocity;
        ShiftingMethod shifting_method;
        double shifting_xsph_eps, shifting_ppst_push, shifting_ppst_pull, shifting_beta_implicit;
        double shifting_diffusion_A, shifting_diffusion_AFSM, shifting_diffusion_AFST;
        double min_distance_coefficient;
        int density_reinit_steps;
        bool use_density_based_projection;
        int num_bce_layers;
        bool use_consistent_gradient_discretization, use_consistent_laplacian_discretization;
        ViscosityMethod viscosity_method;
        BoundaryMethod boundary_method;
        KernelType kernel_type;
        bool use_delta_sph;
        double delta_sph_coefficient, artificial_viscosity, free_surface_threshold;
        int num_proximity_search_steps;
        EosType eos_type;
        bool use_variable_time_step;
        SPHParameters();
    };

    struct LinSolverParameters {  // sph/ChFsiFluidSystemSPH.cpp:552-559
        SolverType type;
        double atol, rtol;
        int max_num_iters;
        LinSolverParameters() : type(SolverType::JACOBI), atol(0.0), rtol(0.0), max_num_iters(1000) {}
    };

    explicit ChFsiFluidSystemSPH(int device = -1);
    ~ChFsiFluidSystemSPH();

    // ---- setters (sph/ChFsiFluidSystemSPH.cpp:180-328, 421-550)
    void SetBoundaryType(BoundaryMethod m) { m_p.boundary_method = (int)m; }
    void SetViscosityType(ViscosityMethod m) { m_p.viscosity_method = (int)m; }
    void SetArtificialViscosityCoefficient(double c) { m_p.artificial_viscosity = (float)c; }
    void SetKernelType(KernelType k);
    void SetShiftingMethod(ShiftingMethod m) { m_p.shifting_method = (int)m; }
    void SetIntegrationScheme(IntegrationScheme s) { m_p.integration_scheme = (int)s; }
    void SetContainerDim(const ChVector3d& d);
    void SetComputationalDomain(const ChAABB& aabb, BoundaryConditions bc);
    void SetComputationalDomain(const ChAABB& aabb);
    void SetNumBCELayers(int n) { m_p.num_bce_layers = n; }
    void SetInitPressure(double height);
    void SetGravitationalAcceleration(const ChVector3d& g) override;
    void SetBodyForce(const ChVector3d& f);
    void SetInitialSpacing(double spacing);
    void SetKernelMultiplier(double multiplier);
    void SetDensity(double rho0);
    void SetShiftingPPSTParameters(double push, double pull);
    void SetShiftingXSPHParameters(double eps) { m_p.shifting_xsph_eps = (float)eps; }
    void SetShiftingDiffusionParameters(double A, double AFSM, double AFST);
    void SetNumProximitySearchSteps(int steps) { m_p.num_proximity_search_steps = steps; }
    void SetUseVariableTimeStep(bool on) { m_p.use_variable_time_step = on ? 1 : 0; }
    void SetElasticSPH(const ElasticMaterialProperties& mat_props);  // cpp:445-476
    std::string GetPhysicsProblemString() const { return m_p.elastic_SPH ? "CRM" : "CFD"; }
    std::vector<float> GetParticleStress() const;  // n x 6: xx yy zz xy xz yz
    void SetCfdSPH(const FluidProperties& fluid_props);
    void SetSPHParameters(const SPHParameters& sph_params);
    void SetLinSolverParameters(const LinSolverParameters& linsolv_params);  // cpp:552-559
    void SetSPHLinearSolver(SolverType lin_solver) { m_p.LinearSolver = (int)lin_solver; }  // cpp:214-216
    void EnableErrorChecks(bool on) { m_check_errors = on; }  // m_check_errors, cpp:70
    /// Native (no reference counterpart): SPH_ARITH_FAST (library default) or SPH_ARITH_EXACT, see include/chsph_mi.h.
    void SetArithmetic(int mode);

    // ---- getters (cpp:2720-2770)
    double GetKernelLength() const { return m_p.h; }
    double GetInitialSpacing() const { return m_p.d0; }
    int GetNumBCELayers() const { return m_p.num_bce_layers; }
    double GetDensity() const { return m_p.rho0; }
    double GetViscosity() const { return m_p.mu0; }
    float GetBulkModulus() const { return m_p.K_bulk; }  // CRM: K derived by SetElasticSPH
    double GetBasePressure() const { return m_p.base_pressure; }
    double GetParticleMass() const { return m_p.markerMass; }
    ChVector3d GetGravitationalAcceleration() const { return {m_p.gravity[0], m_p.gravity[1], m_p.gravity[2]}; }
    double GetSoundSpeed() const;
    size_t GetNumFluidMarkers() const;
    size_t GetNumBoundaryMarkers() const;
    size_t GetNumRigidBodyMarkers() const;
    const sph_params& GetParams() const { return m_p; }

    // ---- markers (cpp:1848-1970)
    void AddSPHParticle(const ChVector3d& pos, double rho, double pres, double mu, const ChVector3d& vel = ChVector3d(), const ChVector3d& tauXxYyZz = ChVector3d(),
                        const ChVector3d& tauXyXzYz = ChVector3d(), double pc = 1e3);
    void AddSPHParticle(const ChVector3d& pos, const ChVector3d& vel = ChVector3d());
    void AddBoxSPH(const ChVector3d& boxCenter, const ChVector3d& boxHalfDim);
    void AddBCEBoundary(const std::vector<ChVector3d>& points, const ChFramed& frame);
    std::vector<ChVector3d> CreatePointsPlate(const ChVector2d& size) const;
    std::vector<ChVector3d> CreatePointsBoxContainer(const ChVector3d& size, const ChVector3i& faces) const;
    std::vector<ChVector3d> CreatePointsBoxInterior(const ChVector3d& size) const;                            // cpp:1972-2042
    std::vector<ChVector3d> CreatePointsSphereInterior(double radius, bool polar = true) const;                 // cpp:2102-2162
    std::vector<ChVector3d> CreatePointsCylinderInterior(double rad, double height, bool polar = true) const;  // cpp:2225-2320
    /// Markers filling a closed triangle mesh (cpp:2627-2714): grid points of spacing d0 from the bounding-box minimum that lie inside the surface
    /// (odd crossing count along both of two fixed rays).  faces: three vertex indices per triangle.
    std::vector<ChVector3d> CreatePointsMesh(const std::vector<ChVector3d>& vertices, const std::vector<int>& faces) const;
    std::vector<ChVector3d> CreatePointsCylinderAnnulus(double rad_inner, double rad_outer, double height, bool polar = true) const;  // cpp:2572-2625
    /// Jacobi sweeps of the last I2SPH step: (velocity system, pressure system).
    std::pair<int, int> GetIsphIterations();
    /// Register a rigid FSI body through its body-local BCE points (OnAddFsiBody, cpp:971-981). Returns the body index.
    size_t AddFsiBody(const std::vector<ChVector3d>& bce_local, const FsiBodyState& initial_state);

    // ---- ChFsiFluidSystem virtuals
    using ChFsiFluidSystem::Initialize;
    void Initialize(const std::vector<FsiBodyState>& body_states) override;
    void OnDoStepDynamics(double time, double step) override;
    void OnExchangeSolidForces() override;
    void OnExchangeSolidStates() override;
    void LoadSolidStates(const std::vector<FsiBodyState>& body_states) override;
    void StoreSolidForces(std::vector<FsiBodyForce> body_forces) override;  // by value, like the reference: returns nothing
    double GetCurrentStepSize() override;
    /// What ChFsiInterfaceSPH::ExchangeSolidForces reads (sph/ChFsiInterfaceSPH.cpp:109-125).
    std::vector<FsiBodyForce> GetBodyForces();
    void Synchronize();

    // ---- accessors (cpp:2801-2871): original marker order
    std::vector<ChVector3d> GetParticlePositions();
    std::vector<ChVector3d> GetParticleVelocities();
    std::vector<ChVector3d> GetParticleAccelerations();
    std::vector<ChVector3d> GetParticleFluidProperties();

    // ---- on-disk output, OutputLevel::STATE_PRESSURE (sph/utils/SphUtilsPrint.cu:36-200, 385-520)
    void SaveParticleData(const std::string& dir);                 ///< fluid<frame>.csv, boundary0.csv (first call), rigidBCE<frame>.csv
    void SaveSolidData(const std::string& dir, double time);       ///< FSI_body<i>.csv, one appended line per call

    sph_ctx* GetContext() { return m_ctx; }

  private:
    void check(int rc, const char* what);
    std::vector<ChVector3d> get3(int (*fn)(sph_ctx*, float*, size_t), size_t n);
    sph_params m_p;
    sph_ctx* m_ctx;
    int m_device;
    bool m_check_errors;
    int m_arithmetic = -1;  // -1: library default
    struct Block {
        int type;
        float h;
        std::vector<float> xyz, vel, rpm;
        std::vector<float> tauD, tauO, pc;  // CRM: initial stress and preconsolidation pressure, filled alongside xyz
    };
    std::vector<Block> m_blocks;
    std::vector<std::vector<float>> m_body_bce;
    std::vector<FsiBodyState> m_body_states;
    int m_out_frame = -1;
    bool m_solid_files = false;
};

/// utils::ChGridSampler<double>::SampleBox (src/chrono/utils/ChUtilsSamplers.h:488-527)
std::vector<ChVector3d> GridSamplerBox(double separation, const ChVector3d& center, const ChVector3d& halfDim);

}  // namespace sph
}  // namespace fsi
}  // namespace chrono_mi

#endif

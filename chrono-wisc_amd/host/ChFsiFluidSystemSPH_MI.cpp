// =============================================================================
// ChFsiFluidSystemSPH_MI.cpp - see the header.  Pure host C++; forwards to include/chsph_mi.h.
// Citations relative to /root/reference/src/chrono_fsi/ unless a full path is given.
// =============================================================================
#include "ChFsiFluidSystemSPH_MI.h"

#include <fstream>
#include <iostream>
#include <sstream>

#include <chrono>
#include <cstring>

namespace chrono_mi {

ChQuaterniond QuatFromAngleAxis(double angle, const ChVector3d& axis) {
    double halfang = angle / 2;
    double sinhalf = std::sin(halfang);
    return ChQuaterniond(std::cos(halfang), axis.x() * sinhalf, axis.y() * sinhalf, axis.z() * sinhalf);
}

ChVector3d ChFramed::TransformPointLocalToParent(const ChVector3d& p) const {
    const double e0 = rot.e[0], e1 = rot.e[1], e2 = rot.e[2], e3 = rot.e[3];
    double e0e0 = e0 * e0, e1e1 = e1 * e1, e2e2 = e2 * e2, e3e3 = e3 * e3;
    double e0e1 = e0 * e1, e0e2 = e0 * e2, e0e3 = e0 * e3, e1e2 = e1 * e2, e1e3 = e1 * e3, e2e3 = e2 * e3;
    double m[3][3] = {{(e0e0 + e1e1) * 2 - 1, (e1e2 - e0e3) * 2, (e1e3 + e0e2) * 2},
                      {(e1e2 + e0e3) * 2, (e0e0 + e2e2) * 2 - 1, (e2e3 - e0e1) * 2},
                      {(e1e3 - e0e2) * 2, (e2e3 + e0e1) * 2, (e0e0 + e3e3) * 2 - 1}};
    ChVector3d r;
    for (int i = 0; i < 3; i++) r[i] = pos[i] + ((m[i][0] * p.x() + m[i][1] * p.y()) + m[i][2] * p.z());
    return r;
}

namespace fsi {

void ChFsiFluidSystem::Initialize() { Initialize(std::vector<FsiBodyState>()); }

void ChFsiFluidSystem::DoStepDynamics(double step) {
    auto t0 = std::chrono::steady_clock::now();
    OnDoStepDynamics(m_time, step);
    m_timer_step = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    m_RTF = m_timer_step / step;
    m_frame++;
    m_time += step;
}

namespace sph {

static inline float cubef(float a) { return a * a * a; }

ChFsiFluidSystemSPH::SPHParameters::SPHParameters()
    : integration_scheme(IntegrationScheme::RK2), initial_spacing(0.01), d0_multiplier(1.2), max_velocity(1.0), shifting_method(ShiftingMethod::XSPH),
      shifting_xsph_eps(0.5), shifting_ppst_push(3.0), shifting_ppst_pull(1.0), shifting_beta_implicit(1.0), shifting_diffusion_A(1.0),
      shifting_diffusion_AFSM(3.0), shifting_diffusion_AFST(2), min_distance_coefficient(0.01), density_reinit_steps((int)2e8),
      use_density_based_projection(false), num_bce_layers(3), use_consistent_gradient_discretization(false),
      use_consistent_laplacian_discretization(false), viscosity_method(ViscosityMethod::ARTIFICIAL_UNILATERAL), boundary_method(BoundaryMethod::ADAMI),
      kernel_type(KernelType::CUBIC_SPLINE), use_delta_sph(true), delta_sph_coefficient(0.1), artificial_viscosity(0.02), free_surface_threshold(2.0),
      num_proximity_search_steps(1), eos_type(EosType::ISOTHERMAL), use_variable_time_step(false) {}

ChFsiFluidSystemSPH::ChFsiFluidSystemSPH(int device) : m_ctx(nullptr), m_device(device), m_check_errors(true) {
    sph_params_default(&m_p);  // InitParams, cpp:81-176
}
ChFsiFluidSystemSPH::~ChFsiFluidSystemSPH() {
    if (m_ctx) sph_destroy(m_ctx);
}

void ChFsiFluidSystemSPH::check(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string(m_ctx ? sph_last_error(m_ctx) : what));
}

void ChFsiFluidSystemSPH::SetKernelType(KernelType k) {
    m_p.kernel_type = (int)k;
    m_p.h_multiplier = (k == KernelType::QUINTIC_SPLINE) ? 3.0f : 2.0f;
}
void ChFsiFluidSystemSPH::SetContainerDim(const ChVector3d& d) {
    m_p.boxDimX = (float)d.x(); m_p.boxDimY = (float)d.y(); m_p.boxDimZ = (float)d.z();
}
void ChFsiFluidSystemSPH::SetComputationalDomain(const ChAABB& aabb, BoundaryConditions bc) {
    SetComputationalDomain(aabb);
    m_p.bc_type[0] = (int)bc.x; m_p.bc_type[1] = (int)bc.y; m_p.bc_type[2] = (int)bc.z;
}
void ChFsiFluidSystemSPH::SetComputationalDomain(const ChAABB& aabb) {
    for (int a = 0; a < 3; a++) { m_p.cMin[a] = (float)aabb.min[a]; m_p.cMax[a] = (float)aabb.max[a]; }
    m_p.use_default_limits = 0;
}
void ChFsiFluidSystemSPH::SetInitPressure(double height) { m_p.pressure_height = height; m_p.use_init_pressure = 1; }
void ChFsiFluidSystemSPH::SetGravitationalAcceleration(const ChVector3d& g) { for (int a = 0; a < 3; a++) m_p.gravity[a] = (float)g[a]; }
void ChFsiFluidSystemSPH::SetBodyForce(const ChVector3d& f) { for (int a = 0; a < 3; a++) m_p.bodyForce3[a] = (float)f[a]; }
void ChFsiFluidSystemSPH::SetInitialSpacing(double spacing) {
    m_p.d0 = (float)spacing;
    m_p.ood0 = 1 / m_p.d0;
    m_p.volume0 = cubef(m_p.d0);
    m_p.markerMass = m_p.volume0 * m_p.rho0;
    m_p.h = m_p.d0_multiplier * m_p.d0;
    m_p.ooh = 1 / m_p.h;
}
void ChFsiFluidSystemSPH::SetKernelMultiplier(double multiplier) {
    m_p.d0_multiplier = (float)multiplier;
    m_p.h = m_p.d0_multiplier * m_p.d0;
    m_p.ooh = 1 / m_p.h;
}
void ChFsiFluidSystemSPH::SetDensity(double rho0) {
    m_p.rho0 = (float)rho0;
    m_p.invrho0 = 1 / m_p.rho0;
    m_p.markerMass = m_p.volume0 * m_p.rho0;
}
void ChFsiFluidSystemSPH::SetShiftingPPSTParameters(double push, double pull) { m_p.shifting_ppst_push = (float)push; m_p.shifting_ppst_pull = (float)pull; }
void ChFsiFluidSystemSPH::SetShiftingDiffusionParameters(double A, double AFSM, double AFST) {
    m_p.shifting_diffusion_A = (float)A; m_p.shifting_diffusion_AFSM = (float)AFSM; m_p.shifting_diffusion_AFST = (float)AFST;
}
void ChFsiFluidSystemSPH::SetElasticSPH(const ElasticMaterialProperties& mat) {  // cpp:445-476
    m_p.elastic_SPH = 1;
    SetDensity(mat.density);
    m_p.rheology_model_crm = (int)mat.rheology_model;
    m_p.E_young = (float)mat.Young_modulus;
    m_p.Nu_poisson = (float)mat.Poisson_ratio;
    m_p.ave_diam = (float)mat.average_diam;
    m_p.G_shear = (float)(m_p.E_young / (2.0 * (1.0 + m_p.Nu_poisson)));
    m_p.INV_G_shear = (float)(1.0 / m_p.G_shear);
    m_p.K_bulk = (float)(m_p.E_young / (3.0 * (1.0 - 2.0 * m_p.Nu_poisson)));
    switch (mat.rheology_model) {
        case RheologyCRM::MCC:
            m_p.mcc_M = (float)mat.mcc_M; m_p.mcc_kappa = (float)mat.mcc_kappa; m_p.mcc_lambda = (float)mat.mcc_lambda; m_p.mcc_v_lambda = (float)mat.mcc_v_lambda;
            break;
        case RheologyCRM::MU_OF_I:
            m_p.mu_I0 = (float)mat.mu_I0; m_p.mu_fric_s = (float)mat.mu_fric_s; m_p.mu_fric_2 = (float)mat.mu_fric_2; m_p.Coh_coeff = (float)mat.cohesion_coeff;
            break;
        default: throw std::runtime_error("Invalid rheology model");
    }
}
void ChFsiFluidSystemSPH::SetCfdSPH(const FluidProperties& fp) {
    SetDensity(fp.density);
    m_p.mu0 = (float)fp.viscosity;
    m_p.L_Characteristic = (float)fp.char_length;
}
void ChFsiFluidSystemSPH::SetSPHParameters(const SPHParameters& sp) {  // cpp:507-550
    m_p.integration_scheme = (int)sp.integration_scheme;
    m_p.eos_type = (int)sp.eos_type;
    m_p.viscosity_method = (int)sp.viscosity_method;
    m_p.free_surface_threshold = (float)sp.free_surface_threshold;
    m_p.boundary_method = (int)sp.boundary_method;
    m_p.kernel_type = (int)sp.kernel_type;
    m_p.shifting_method = (int)sp.shifting_method;
    m_p.d0 = (float)sp.initial_spacing;
    m_p.volume0 = cubef(m_p.d0);
    m_p.markerMass = m_p.volume0 * m_p.rho0;
    m_p.d0_multiplier = (float)sp.d0_multiplier;
    m_p.h = m_p.d0_multiplier * m_p.d0;
    m_p.ood0 = 1 / m_p.d0;
    m_p.ooh = 1 / m_p.h;
    m_p.v_Max = (float)sp.max_velocity;
    m_p.shifting_xsph_eps = (float)sp.shifting_xsph_eps;
    m_p.shifting_ppst_push = (float)sp.shifting_ppst_push;
    m_p.shifting_ppst_pull = (float)sp.shifting_ppst_pull;
    m_p.shifting_beta_implicit = (float)sp.shifting_beta_implicit;
    m_p.shifting_diffusion_A = (float)sp.shifting_diffusion_A;
    m_p.shifting_diffusion_AFSM = (float)sp.shifting_diffusion_AFSM;
    m_p.shifting_diffusion_AFST = (float)sp.shifting_diffusion_AFST;
    m_p.epsMinMarkersDis = (float)sp.min_distance_coefficient;
    m_p.density_reinit_steps = sp.density_reinit_steps;
    m_p.use_density_based_projection = sp.use_density_based_projection;
    m_p.num_bce_layers = sp.num_bce_layers;
    m_p.use_consistent_gradient_discretization = sp.use_consistent_gradient_discretization;
    m_p.use_consistent_laplacian_discretization = sp.use_consistent_laplacian_discretization;
    m_p.artificial_viscosity = (float)sp.artificial_viscosity;
    m_p.use_delta_sph = sp.use_delta_sph;
    m_p.density_delta = (float)sp.delta_sph_coefficient;
    m_p.num_proximity_search_steps = sp.num_proximity_search_steps;
    m_p.use_variable_time_step = sp.use_variable_time_step;
}
double ChFsiFluidSystemSPH::GetSoundSpeed() const { return 10 * m_p.v_Max; }  // CFD branch of cpp:2760-2770

static sph_counters counters_of(sph_ctx* ctx) {
    sph_counters c;
    std::memset(&c, 0, sizeof(c));
    if (ctx) sph_get_counters(ctx, &c);
    return c;
}
std::vector<float> ChFsiFluidSystemSPH::GetParticleStress() const {
    std::vector<float> out(6 * counters_of(m_ctx).numAllMarkers);
    if (sph_get_stress(m_ctx, out.data(), out.size() / 6)) throw std::runtime_error(sph_last_error(m_ctx));
    return out;
}
size_t ChFsiFluidSystemSPH::GetNumFluidMarkers() const { return counters_of(m_ctx).numFluidMarkers; }
size_t ChFsiFluidSystemSPH::GetNumBoundaryMarkers() const { return counters_of(m_ctx).numBoundaryMarkers; }
size_t ChFsiFluidSystemSPH::GetNumRigidBodyMarkers() const { return counters_of(m_ctx).numRigidMarkers; }

// ---- markers
void ChFsiFluidSystemSPH::AddSPHParticle(const ChVector3d& pos, double rho, double pres, double mu, const ChVector3d& vel, const ChVector3d& tauXxYyZz,
                                         const ChVector3d& tauXyXzYz, double pc) {
    if (m_is_initialized) throw std::runtime_error("AddSPHParticle: system already initialized");
    if (m_blocks.empty() || m_blocks.back().type != SPH_MARKER_FLUID || m_blocks.back().h != m_p.h) m_blocks.push_back(Block{SPH_MARKER_FLUID, m_p.h, {}, {}, {}, {}, {}, {}});
    Block& b = m_blocks.back();
    for (int a = 0; a < 3; a++) { b.xyz.push_back((float)pos[a]); b.vel.push_back((float)vel[a]); b.tauD.push_back((float)tauXxYyZz[a]); b.tauO.push_back((float)tauXyXzYz[a]); }
    b.rpm.push_back((float)rho); b.rpm.push_back((float)pres); b.rpm.push_back((float)mu);
    b.pc.push_back((float)pc);
}
void ChFsiFluidSystemSPH::AddSPHParticle(const ChVector3d& pos, const ChVector3d& vel) { AddSPHParticle(pos, m_p.rho0, m_p.base_pressure, m_p.mu0, vel); }
void ChFsiFluidSystemSPH::AddBoxSPH(const ChVector3d& boxCenter, const ChVector3d& boxHalfDim) {
    for (const auto& p : GridSamplerBox(m_p.d0, boxCenter, boxHalfDim)) AddSPHParticle(p, m_p.rho0, 0, m_p.mu0, ChVector3d(0, 0, 0));
}
void ChFsiFluidSystemSPH::AddBCEBoundary(const std::vector<ChVector3d>& points, const ChFramed& frame) {
    if (m_is_initialized) throw std::runtime_error("AddBCEBoundary: system already initialized");
    m_blocks.push_back(Block{SPH_MARKER_BCE_WALL, m_p.h, {}, {}, {}, {}, {}, {}});
    Block& b = m_blocks.back();
    for (const auto& p : points) {
        ChVector3d g = frame.TransformPointLocalToParent(p);
        for (int a = 0; a < 3; a++) b.xyz.push_back((float)g[a]);
    }
}
size_t ChFsiFluidSystemSPH::AddFsiBody(const std::vector<ChVector3d>& bce_local, const FsiBodyState& state) {
    if (m_is_initialized) throw std::runtime_error("AddFsiBody: system already initialized");
    std::vector<float> loc;
    for (const auto& p : bce_local) for (int a = 0; a < 3; a++) loc.push_back((float)p[a]);
    m_body_bce.push_back(loc);
    m_body_states.push_back(state);
    return m_body_bce.size() - 1;
}

std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsPlate(const ChVector2d& size) const {  // cpp:1890-1910
    std::vector<ChVector3d> bce;
    float spacing = m_p.d0;
    int num_layers = m_p.num_bce_layers;
    double hx = size.x() / 2, hy = size.y() / 2;
    int npx = (int)std::round(hx / spacing), npy = (int)std::round(hy / spacing);
    double dx = hx / npx, dy = hy / npy;
    for (int il = 0; il < num_layers; il++)
        for (int ix = -npx; ix <= npx; ix++)
            for (int iy = -npy; iy <= npy; iy++) bce.push_back({ix * dx, iy * dy, (double)(-il * spacing)});
    return bce;
}
std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsBoxContainer(const ChVector3d& size, const ChVector3i& faces) const {  // cpp:1912-1970
    std::vector<ChVector3d> bce;
    const double pi = 3.14159265358979323846, pi_2 = pi / 2;
    float spacing = m_p.d0;
    float buffer = 2 * (m_p.num_bce_layers - 1) * spacing;
    ChVector3d hs = size / 2;
    auto add = [&bce](const std::vector<ChVector3d>& pts, const ChFramed& X) { for (const auto& v : pts) bce.push_back(X.TransformPointLocalToParent(v)); };
    if (faces.z() == -1 || faces.z() == 2) add(CreatePointsPlate({size.x(), size.y()}), ChFramed({0, 0, -hs.z()}, QUNIT));
    if (faces.z() == +1 || faces.z() == 2) add(CreatePointsPlate({size.x(), size.y()}), ChFramed({0, 0, +hs.z()}, QuatFromAngleX(pi)));
    if (faces.x() == -1 || faces.x() == 2) add(CreatePointsPlate({size.z() + buffer, size.y()}), ChFramed({-hs.x(), 0, 0}, QuatFromAngleY(+pi_2)));
    if (faces.x() == +1 || faces.x() == 2) add(CreatePointsPlate({size.z() + buffer, size.y()}), ChFramed({+hs.x(), 0, 0}, QuatFromAngleY(-pi_2)));
    if (faces.y() == -1 || faces.y() == 2) add(CreatePointsPlate({size.x() + buffer, size.z() + buffer}), ChFramed({0, -hs.y(), 0}, QuatFromAngleX(-pi_2)));
    if (faces.y() == +1 || faces.y() == 2) add(CreatePointsPlate({size.x() + buffer, size.z() + buffer}), ChFramed({0, +hs.y(), 0}, QuatFromAngleX(+pi_2)));
    return bce;
}

// ---- BCE point generators for bodies.  Marker-for-marker the sets (and the order) of the reference's generators
// (sph/ChFsiFluidSystemSPH.cpp:1972-2320, 2572-2625) -- the C ABI takes them as given and the parity tests compare marker by marker
// -- but organised around three small emitters: a circle of points, a column of axial stations, and a pair of mirrored points.
namespace {
const double kPi = 3.14159265358979323846;
inline int nearest_int(double v) { return (int)std::round(v); }

// visits the points of a circle of radius r (z = 0 plane): `count` points starting on the x axis
template <class F>
void for_circle(double r, int count, F&& visit) {
    const double step = 2 * kPi / count;
    for (int k = 0; k < count; k++) {
        const double angle = k * step;
        visit(r * std::cos(angle), r * std::sin(angle));
    }
}
// axial stations z0, z0 + dz, ... (n of them) above the point (x, y)
inline void column(std::vector<ChVector3d>& out, double x, double y, double z0, double dz, int n) {
    for (int k = 0; k < n; k++) out.push_back({x, y, z0 + k * dz});
}
// the `layers` outermost stations of both end caps: (x, y, -z), (x, y, +z) for z = top, top - dz, ...
inline void cap_pair(std::vector<ChVector3d>& out, double x, double y, double top, double dz, int layers) {
    for (int k = 0; k < layers; k++) {
        const double z = top - k * dz;
        out.push_back({x, y, -z});
        out.push_back({x, y, +z});
    }
}
}  // namespace

std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsCylinderInterior(double rad, double height, bool polar) const {
    std::vector<ChVector3d> out;
    const double d0 = m_p.d0;
    const int layers = m_p.num_bce_layers;
    const int n_axial = nearest_int(height / d0);
    const double dz = height / n_axial;
    int n_radial = nearest_int(rad / d0);
    // a cylinder too thin or too short for `layers` shells is filled completely
    const bool solid = (n_radial <= layers - 1) || (n_axial <= 2 * layers - 1);
    if (polar) {
        // lateral shells: circles from the innermost shell radius outwards, one column of axial stations per circle point
        const double shell_r0 = solid ? 0.0 : std::max(rad - (layers - 1) * d0, 0.0);
        n_radial = nearest_int((rad - shell_r0) / d0);
        const double dr = (rad - shell_r0) / n_radial;
        for (int i = 0; i <= n_radial; i++) {
            const double r = shell_r0 + i * dr;
            for_circle(r, std::max(nearest_int(2 * kPi * r / d0) - 1, 1), [&](double x, double y) { column(out, x, y, -height / 2, dz, n_axial + 1); });
        }
        if (solid) return out;
        // end caps: the discs inside the innermost shell, `layers` stations deep at both ends
        const int n_disc = nearest_int(shell_r0 / d0);
        const double dr_disc = shell_r0 / n_disc;
        for (int i = 0; i < n_disc; i++) {
            const double r = i * dr_disc;
            for_circle(r, std::max(nearest_int(2 * kPi * r / d0) - 1, 1), [&](double x, double y) { cap_pair(out, x, y, height / 2, dz, layers); });
        }
        return out;
    }
    // Cartesian lattice clipped by the cylinder: full columns in the shell, cap stations inside it
    const double dr = rad / n_radial;
    const double outer2 = rad * rad;
    const double inner = solid ? 0.0 : std::max(rad - layers * dr, 0.0);
    const double inner2 = inner * inner;
    for (int i = -n_radial; i <= n_radial; i++) {
        const double x = i * dr;
        for (int j = -n_radial; j <= n_radial; j++) {
            const double y = j * dr, rr = x * x + y * y;
            if (rr >= inner2 && rr <= outer2) column(out, x, y, -height / 2, dz, n_axial + 1);
            if (!solid && rr < inner2) cap_pair(out, x, y, height / 2, dz, layers);
        }
    }
    return out;
}

// Triangle-mesh filling.  One crossing test per (point, ray, triangle): Moeller-Trumbore without back-face culling, a near-parallel ray
// (|det| < 1e-6) and a hit at or behind the origin (t <= 1e-6) do not count.  A point is inside when BOTH rays cross the surface an odd number of times.
namespace {
struct Tri { double v0[3], e1[3], e2[3]; };
inline void cross3(const double a[3], const double b[3], double o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
bool crosses(const Tri& T, const double o[3], const double dir[3]) {
    constexpr double eps = 1e-6;
    double pvec[3], tvec[3], qvec[3];
    cross3(dir, T.e2, pvec);
    const double det = dot3(T.e1, pvec);
    if (det > -eps && det < eps) return false;
    const double inv = 1.0 / det;
    for (int k = 0; k < 3; k++) tvec[k] = o[k] - T.v0[k];
    const double uu = dot3(tvec, pvec) * inv;
    if (uu < 0.0 || uu > 1.0) return false;
    cross3(tvec, T.e1, qvec);
    const double vv = dot3(dir, qvec) * inv;
    if (vv < 0.0 || uu + vv > 1.0) return false;
    return dot3(T.e2, qvec) * inv > eps;
}
}  // namespace
std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsMesh(const std::vector<ChVector3d>& vertices, const std::vector<int>& faces) const {
    std::vector<ChVector3d> points;
    if (vertices.empty() || faces.size() < 3) return points;
    const double spacing = (double)m_p.d0;
    double lo[3] = {vertices[0][0], vertices[0][1], vertices[0][2]}, hi[3] = {lo[0], lo[1], lo[2]};
    for (const auto& v : vertices)
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
    std::vector<Tri> tris(faces.size() / 3);
    for (size_t f = 0; f < tris.size(); f++) {
        const ChVector3d &a = vertices[faces[3 * f]], &b = vertices[faces[3 * f + 1]], &c = vertices[faces[3 * f + 2]];
        for (int k = 0; k < 3; k++) { tris[f].v0[k] = a[k]; tris[f].e1[k] = b[k] - a[k]; tris[f].e2[k] = c[k] - a[k]; }
    }
    const double rays[2][3] = {{5, 0.5, 0.25}, {-3, 0.7, 10}};
    for (double x = lo[0]; x < hi[0]; x += spacing)
        for (double y = lo[1]; y < hi[1]; y += spacing)
            for (double z = lo[2]; z < hi[2]; z += spacing) {
                const double o[3] = {x + 1e-9, y + 1e-9, z + 1e-9};
                bool inside = true;
                for (int r = 0; r < 2 && inside; r++) {
                    unsigned n = 0;
                    for (const Tri& T : tris) n += crosses(T, o, rays[r]) ? 1u : 0u;
                    inside = (n & 1u) != 0u;
                }
                if (inside) points.push_back(ChVector3d(x, y, z));
            }
    return points;
}

std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsCylinderAnnulus(double rad_inner, double rad_outer, double height, bool polar) const {
    std::vector<ChVector3d> out;
    const double d0 = m_p.d0, half = height / 2;
    const int n_half = nearest_int(half / d0);  // stations on each side of the mid plane
    const double dz = half / n_half;
    if (polar) {
        const int n_radial = nearest_int((rad_outer - rad_inner) / d0);
        const double dr = (rad_outer - rad_inner) / n_radial;
        for (int i = 0; i <= n_radial; i++) {
            const double r = rad_inner + i * dr;
            for_circle(r, nearest_int(2 * kPi * r / d0), [&](double x, double y) {
                for (int k = -n_half; k <= n_half; k++) out.push_back({x, y, k * dz});
            });
        }
        return out;
    }
    const int n_radial = nearest_int(rad_outer / d0);
    const double dr = rad_outer / n_radial, in2 = rad_inner * rad_inner, out2 = rad_outer * rad_outer;
    for (int i = -n_radial; i <= n_radial; i++)
        for (int j = -n_radial; j <= n_radial; j++) {
            const double x = i * dr, y = j * dr, rr = x * x + y * y;
            if (rr < in2 || rr > out2) continue;
            for (int k = -n_half; k <= n_half; k++) out.push_back({x, y, k * dz});
        }
    return out;
}

std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsBoxInterior(const ChVector3d& size) const {  // the set of sph/ChFsiFluidSystemSPH.cpp:1972-2042
    std::vector<ChVector3d> out;
    const double d0 = m_p.d0;
    const int layers = m_p.num_bce_layers;
    int n[3];
    double step[3];
    bool solid = false;
    for (int a = 0; a < 3; a++) {
        n[a] = 1 + nearest_int(size[a] / d0);
        step[a] = size[a] / (n[a] - 1);
        solid = solid || n[a] <= 2 * layers;
    }
    auto coord = [&](int a, int i) { return -size[a] / 2 + i * step[a]; };
    if (solid) {  // too small for shells: the whole lattice
        for (int i = 0; i < n[0]; i++)
            for (int j = 0; j < n[1]; j++)
                for (int k = 0; k < n[2]; k++) out.push_back({coord(0, i), coord(1, j), coord(2, k)});
        return out;
    }
    // shell by shell, the two faces of each axis in turn; edges and corners appear once per face they belong to (kept, like the reference)
    for (int shell = 0; shell < layers; shell++) {
        const double xm = -size[0] / 2 + shell * step[0], xp = +size[0] / 2 - shell * step[0];
        for (int j = 0; j < n[1]; j++)
            for (int k = 0; k < n[2]; k++) { out.push_back({xm, coord(1, j), coord(2, k)}); out.push_back({xp, coord(1, j), coord(2, k)}); }
        const double ym = -size[1] / 2 + shell * step[1], yp = +size[1] / 2 - shell * step[1];
        for (int k = 0; k < n[2]; k++)
            for (int i = 0; i < n[0]; i++) { out.push_back({coord(0, i), ym, coord(2, k)}); out.push_back({coord(0, i), yp, coord(2, k)}); }
        const double zm = -size[2] / 2 + shell * step[2], zp = +size[2] / 2 - shell * step[2];
        for (int i = 0; i < n[0]; i++)
            for (int j = 0; j < n[1]; j++) { out.push_back({coord(0, i), coord(1, j), zm}); out.push_back({coord(0, i), coord(1, j), zp}); }
    }
    return out;
}

std::vector<ChVector3d> ChFsiFluidSystemSPH::CreatePointsSphereInterior(double radius, bool polar) const {  // the set of sph/ChFsiFluidSystemSPH.cpp:2102-2162
    std::vector<ChVector3d> out;
    const double d0 = m_p.d0;
    const int layers = m_p.num_bce_layers;
    if (polar) {
        // concentric shells; on each, parallels from the north pole down, points spaced ~d0 along each parallel
        const double r_first = radius - (layers - 1) * d0;
        for (int shell = 0; shell < layers; shell++) {
            const double r = r_first + shell * d0;
            const int n_parallels = nearest_int(kPi * r / d0);
            const double dphi = kPi / n_parallels;
            for (int p = 0; p < n_parallels; p++) {
                const double phi = p * dphi, ring = r * std::sin(phi), z = r * std::cos(phi);
                const int count = nearest_int(2 * kPi * ring / d0);
                const double dth = count > 0 ? (2 * kPi) / count : 1;
                for (int t = 0; t < count; t++) out.push_back({ring * std::cos(t * dth), ring * std::sin(t * dth), z});
            }
        }
        return out;
    }
    // Cartesian: horizontal slices of the upper half, mirrored; each slice keeps an annulus `layers` spacings wide (a full disc near the pole)
    const int n = nearest_int(radius / d0);
    const double h = radius / n;
    for (int s = 0; s <= n; s++) {
        const double z = s * h;
        const double r_out = std::sqrt(radius * radius - z * z);
        double r_in = std::max(r_out - layers * h, 0.0);
        if (s >= n - layers) r_in = 0;
        const double in2 = r_in * r_in, out2 = r_out * r_out;
        const int nq = nearest_int(r_out / d0);
        for (int i = -nq; i <= nq; i++)
            for (int j = -nq; j <= nq; j++) {
                const double x = i * h, y = j * h, rr = x * x + y * y;
                if (rr >= in2 && rr <= out2) { out.push_back({x, y, +z}); out.push_back({x, y, -z}); }
            }
    }
    return out;
}
void ChFsiFluidSystemSPH::SetLinSolverParameters(const LinSolverParameters& lp) {  // cpp:552-559
    m_p.LinearSolver = (int)lp.type;
    m_p.LinearSolver_Abs_Tol = (float)lp.atol;
    m_p.LinearSolver_Rel_Tol = (float)lp.rtol;
    m_p.LinearSolver_Max_Iter = lp.max_num_iters;
}
std::pair<int, int> ChFsiFluidSystemSPH::GetIsphIterations() {
    int v = 0, p = 0;
    if (m_ctx) check(sph_get_isph_iterations(m_ctx, &v, &p), "sph_get_isph_iterations");
    return {v, p};
}

void ChFsiFluidSystemSPH::SaveParticleData(const std::string& dir) {  // SaveAllCFD / SaveFileCFD, utils/SphUtilsPrint.cu:36-160
    if (!m_ctx) throw std::runtime_error("SaveParticleData: not initialized");
    m_out_frame++;
    const size_t nf = GetNumFluidMarkers(), nb = GetNumBoundaryMarkers(), nr = GetNumRigidBodyMarkers(), n = nf + nb + nr;
    std::vector<float> pos(3 * n), vel(3 * n), rpm(4 * n), acc(4 * n);
    check(sph_get_positions(m_ctx, pos.data(), n), "sph_get_positions");
    check(sph_get_velocities(m_ctx, vel.data(), n), "sph_get_velocities");
    check(sph_get_rho_pres_mu_type(m_ctx, rpm.data(), n), "sph_get_rho_pres_mu_type");
    check(sph_get_derivatives(m_ctx, acc.data(), n), "sph_get_derivatives");
    auto dump = [&](const std::string& name, size_t lo, size_t hi) {
        std::ofstream file(dir + "/" + name);
        std::stringstream ss;
        ss << "x,y,z,v_x,v_y,v_z,|U|,acc,rho,pressure\n";
        for (size_t i = lo; i < hi; i++) {
            const float* p = &pos[3 * i];
            const float* v = &vel[3 * i];
            const float* a = &acc[4 * i];
            float v_len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), a_len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
            ss << p[0] << ", " << p[1] << ", " << p[2] << ", " << v[0] << ", " << v[1] << ", " << v[2] << ", " << v_len << ", " << a_len << ", " << rpm[4 * i] << ", "
               << rpm[4 * i + 1] << std::endl;
        }
        file << ss.str();
    };
    dump("fluid" + std::to_string(m_out_frame) + ".csv", 0, nf);
    if (m_out_frame == 0) dump("boundary0.csv", nf, nf + nb);
    if (nr) dump("rigidBCE" + std::to_string(m_out_frame) + ".csv", nf + nb, n);
}
void ChFsiFluidSystemSPH::SaveSolidData(const std::string& dir, double time) {  // saveSolidData, utils/SphUtilsPrint.cu:385-470
    auto forces = GetBodyForces();
    for (size_t i = 0; i < forces.size() && i < m_body_states.size(); i++) {
        std::string filename = dir + "/FSI_body" + std::to_string(i) + ".csv";
        if (!m_solid_files) {
            std::ofstream file(filename);
            file << "Time,x,y,z,q0,q1,q2,q3,Vx,Vy,Vz,Fx,Fy,Fz,Tx,Ty,Tz" << std::endl;
        }
        const FsiBodyState& s = m_body_states[i];
        std::ofstream file(filename, std::fstream::app);
        file << time << "," << (float)s.pos.x() << "," << (float)s.pos.y() << "," << (float)s.pos.z() << "," << (float)s.rot.e[0] << "," << (float)s.rot.e[1] << ","
             << (float)s.rot.e[2] << "," << (float)s.rot.e[3] << "," << (float)s.lin_vel.x() << "," << (float)s.lin_vel.y() << "," << (float)s.lin_vel.z() << ","
             << (float)forces[i].force.x() << "," << (float)forces[i].force.y() << "," << (float)forces[i].force.z() << "," << (float)forces[i].torque.x() << ","
             << (float)forces[i].torque.y() << "," << (float)forces[i].torque.z() << std::endl;
    }
    m_solid_files = true;
}

std::vector<ChVector3d> GridSamplerBox(double separation, const ChVector3d& center, const ChVector3d& size) {
    std::vector<ChVector3d> out;
    double fuzz = (size.x() < 1) ? 1e-6 * size.x() : 1e-6;
    int n[3];
    double sep[3];
    for (int i = 0; i < 3; i++) {
        double d = 2 * size[i];
        n[i] = (int)std::round(d / separation);
        sep[i] = d / n[i];
    }
    ChVector3d bl = center - size;
    for (int ix = 0; ix <= n[0]; ix++)
        for (int iy = 0; iy <= n[1]; iy++)
            for (int iz = 0; iz <= n[2]; iz++) {
                ChVector3d p = bl + ChVector3d(ix * sep[0], iy * sep[1], iz * sep[2]);
                ChVector3d v = p - center;
                if (std::abs(v.x()) <= size.x() + fuzz && std::abs(v.y()) <= size.y() + fuzz && std::abs(v.z()) <= size.z() + fuzz) out.push_back(p);
            }
    return out;
}

// ---- Initialize / step / exchange
static void to_state(const FsiBodyState& s, sph_body_state& o) {
    for (int a = 0; a < 3; a++) {
        o.pos[a] = s.pos[a]; o.lin_vel[a] = s.lin_vel[a]; o.ang_vel[a] = s.ang_vel[a]; o.lin_acc[a] = s.lin_acc[a]; o.ang_acc[a] = s.ang_acc[a];
    }
    for (int a = 0; a < 4; a++) o.rot[a] = s.rot.e[a];
}

void ChFsiFluidSystemSPH::SetArithmetic(int mode) {
    m_arithmetic = mode;
    if (m_ctx) check(sph_set_arithmetic(m_ctx, mode), "set_arithmetic");
}

void ChFsiFluidSystemSPH::Initialize(const std::vector<FsiBodyState>& body_states) {
    if (m_is_initialized) throw std::runtime_error("Initialize: system already initialized");
    m_ctx = sph_create(&m_p, m_device);
    if (!m_ctx) throw std::runtime_error(sph_create_error());
    check(sph_set_check_errors(m_ctx, m_check_errors), "set_check_errors");
    if (m_arithmetic >= 0) check(sph_set_arithmetic(m_ctx, m_arithmetic), "set_arithmetic");
    check(sph_set_step_size(m_ctx, m_step), "set_step_size");
    // posRad.w = h at the time each marker was added (phys/SphDataManager.cu:210,228)
    size_t n_added = 0;
    for (auto& b : m_blocks) {
        sph_params tmp = m_p;
        tmp.h = b.h;
        check(sph_set_params(m_ctx, &tmp), "set_params");
        size_t n = b.xyz.size() / 3;
        check(sph_add_markers(m_ctx, b.type, n, b.xyz.data(), b.vel.empty() ? nullptr : b.vel.data(), b.rpm.empty() ? nullptr : b.rpm.data()), "add_markers");
        if (m_p.elastic_SPH && b.tauD.size() == 3 * n) check(sph_set_marker_stress(m_ctx, n_added, n, b.tauD.data(), b.tauO.data()), "set_marker_stress");
        if (m_p.elastic_SPH && m_p.rheology_model_crm == SPH_RHEOLOGY_MCC && b.pc.size() == n) check(sph_set_marker_consolidation(m_ctx, n_added, n, b.pc.data()), "set_marker_consolidation");
        n_added += n;
    }
    check(sph_set_params(m_ctx, &m_p), "set_params");
    for (auto& loc : m_body_bce)
        if (sph_add_rigid_body(m_ctx, loc.size() / 3, loc.data()) < 0) throw std::runtime_error(sph_last_error(m_ctx));
    const std::vector<FsiBodyState>& st = body_states.empty() ? m_body_states : body_states;
    std::vector<sph_body_state> cs(st.size());
    for (size_t i = 0; i < st.size(); i++) to_state(st[i], cs[i]);
    check(sph_initialize(m_ctx, cs.size(), cs.data()), "initialize");
    check(sph_get_params(m_ctx, &m_p), "get_params");
    m_blocks.clear();
    m_is_initialized = true;
}
void ChFsiFluidSystemSPH::OnDoStepDynamics(double time, double step) {
    if (!m_is_initialized) throw std::runtime_error("FSI system not initialized!\n");
    check(sph_step(m_ctx, time, step), "step");
}
void ChFsiFluidSystemSPH::OnExchangeSolidForces() { check(sph_compute_body_wrenches(m_ctx), "wrenches"); }
void ChFsiFluidSystemSPH::OnExchangeSolidStates() { check(sph_update_body_markers(m_ctx), "update_body_markers"); }
void ChFsiFluidSystemSPH::LoadSolidStates(const std::vector<FsiBodyState>& body_states) {
    std::vector<sph_body_state> cs(body_states.size());
    for (size_t i = 0; i < body_states.size(); i++) to_state(body_states[i], cs[i]);
    check(sph_load_body_states(m_ctx, cs.size(), cs.data()), "load_body_states");
    m_body_states = body_states;  // what SaveSolidData reports (fsiBodyState_D in the reference)
}
void ChFsiFluidSystemSPH::StoreSolidForces(std::vector<FsiBodyForce> body_forces) {
    // the reference takes the output vector by value (ChFsiFluidSystem.h:89): nothing reaches the caller. Kept.
    auto f = GetBodyForces();
    for (size_t i = 0; i < body_forces.size() && i < f.size(); i++) body_forces[i] = f[i];
}
std::vector<FsiBodyForce> ChFsiFluidSystemSPH::GetBodyForces() {
    size_t n = m_body_bce.size();
    std::vector<sph_wrench> w(n ? n : 1);
    check(sph_get_body_wrenches(m_ctx, n, w.data()), "get_body_wrenches");
    std::vector<FsiBodyForce> out(n);
    for (size_t i = 0; i < n; i++) {
        out[i].force = ChVector3d(w[i].force[0], w[i].force[1], w[i].force[2]);
        out[i].torque = ChVector3d(w[i].torque[0], w[i].torque[1], w[i].torque[2]);
    }
    return out;
}
double ChFsiFluidSystemSPH::GetCurrentStepSize() { return m_ctx ? sph_current_step(m_ctx) : GetStepSize(); }
void ChFsiFluidSystemSPH::Synchronize() { check(sph_synchronize(m_ctx), "synchronize"); }

std::vector<ChVector3d> ChFsiFluidSystemSPH::get3(int (*fn)(sph_ctx*, float*, size_t), size_t n) {
    std::vector<float> buf(3 * (n ? n : 1));
    check(fn(m_ctx, buf.data(), n), "getter");
    std::vector<ChVector3d> out(n);
    for (size_t i = 0; i < n; i++) out[i] = ChVector3d(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
    return out;
}
std::vector<ChVector3d> ChFsiFluidSystemSPH::GetParticlePositions() { return get3(sph_get_positions, counters_of(m_ctx).numAllMarkers); }
std::vector<ChVector3d> ChFsiFluidSystemSPH::GetParticleVelocities() { return get3(sph_get_velocities, counters_of(m_ctx).numAllMarkers); }
std::vector<ChVector3d> ChFsiFluidSystemSPH::GetParticleFluidProperties() { return get3(sph_get_properties, counters_of(m_ctx).numAllMarkers); }
std::vector<ChVector3d> ChFsiFluidSystemSPH::GetParticleAccelerations() { return get3(sph_get_accelerations, counters_of(m_ctx).numFluidMarkers); }

}  // namespace sph
}  // namespace fsi
}  // namespace chrono_mi

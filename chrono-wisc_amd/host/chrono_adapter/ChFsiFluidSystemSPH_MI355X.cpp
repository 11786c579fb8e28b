// =============================================================================
// ChFsiFluidSystemSPH_MI355X.cpp - see the header.  NOT COMPILED IN THIS BUILD IMAGE (needs Chrono core + Eigen).
// =============================================================================
#include "ChFsiFluidSystemSPH_MI355X.h"

#include <algorithm>
#include <stdexcept>

#include "chrono/physics/ChBody.h"
#include "chrono/utils/ChBodyGeometry.h"

namespace chrono {
namespace fsi {
namespace sph {

ChFsiFluidSystemSPH_MI355X::ChFsiFluidSystemSPH_MI355X(int device) : ChFsiFluidSystem(), m_twin(device) {}
ChFsiFluidSystemSPH_MI355X::~ChFsiFluidSystemSPH_MI355X() {}

chrono_mi::fsi::FsiBodyState ChFsiFluidSystemSPH_MI355X::cs(const FsiBodyState& s) {
    chrono_mi::fsi::FsiBodyState o;
    o.pos = cv(s.pos);
    o.rot = chrono_mi::ChQuaterniond(s.rot.e0(), s.rot.e1(), s.rot.e2(), s.rot.e3());
    o.lin_vel = cv(s.lin_vel);
    o.ang_vel = cv(s.ang_vel);  // body-local frame: the interface fills it from GetAngVelLocal (sph/ChFsiInterfaceSPH.cpp:46)
    o.lin_acc = cv(s.lin_acc);
    o.ang_acc = cv(s.ang_acc);
    return o;
}
std::vector<ChVector3d> ChFsiFluidSystemSPH_MI355X::cvv(const std::vector<chrono_mi::ChVector3d>& v) {
    std::vector<ChVector3d> o;
    o.reserve(v.size());
    for (const auto& p : v) o.push_back(cv(p));
    return o;
}

// ---- facade surface with Chrono types
void ChFsiFluidSystemSPH_MI355X::SetGravitationalAcceleration(const ChVector3d& gravity) { m_twin.SetGravitationalAcceleration(cv(gravity)); }
void ChFsiFluidSystemSPH_MI355X::SetBodyForce(const ChVector3d& force) { m_twin.SetBodyForce(cv(force)); }
void ChFsiFluidSystemSPH_MI355X::SetContainerDim(const ChVector3d& box_dim) { m_twin.SetContainerDim(cv(box_dim)); }
void ChFsiFluidSystemSPH_MI355X::SetComputationalDomain(const ChAABB& aabb, chrono_mi::fsi::sph::BoundaryConditions bc_type) {
    m_twin.SetComputationalDomain(chrono_mi::ChAABB(cv(aabb.min), cv(aabb.max)), bc_type);
}
void ChFsiFluidSystemSPH_MI355X::SetComputationalDomain(const ChAABB& aabb) { m_twin.SetComputationalDomain(chrono_mi::ChAABB(cv(aabb.min), cv(aabb.max))); }
void ChFsiFluidSystemSPH_MI355X::AddSPHParticle(const ChVector3d& pos, double rho, double pres, double mu, const ChVector3d& vel, const ChVector3d& tauXxYyZz,
                                                const ChVector3d& tauXyXzYz, double pc) {
    m_twin.AddSPHParticle(cv(pos), rho, pres, mu, cv(vel), cv(tauXxYyZz), cv(tauXyXzYz), pc);
}
void ChFsiFluidSystemSPH_MI355X::AddSPHParticle(const ChVector3d& pos, const ChVector3d& vel) { m_twin.AddSPHParticle(cv(pos), cv(vel)); }
void ChFsiFluidSystemSPH_MI355X::AddBoxSPH(const ChVector3d& boxCenter, const ChVector3d& boxHalfDim) { m_twin.AddBoxSPH(cv(boxCenter), cv(boxHalfDim)); }
void ChFsiFluidSystemSPH_MI355X::AddBCEBoundary(const std::vector<ChVector3d>& points, const ChFramed& frame) {
    // the frame transform is applied here with Chrono's own ChFrame (double), the twin receives absolute points
    std::vector<chrono_mi::ChVector3d> abs_points;
    abs_points.reserve(points.size());
    for (const auto& p : points) abs_points.push_back(cv(frame.TransformPointLocalToParent(p)));
    m_twin.AddBCEBoundary(abs_points, chrono_mi::ChFramed());
}
std::vector<ChVector3d> ChFsiFluidSystemSPH_MI355X::CreatePointsBoxContainer(const ChVector3d& size, const ChVector3i& faces) const {
    return cvv(m_twin.CreatePointsBoxContainer(cv(size), chrono_mi::ChVector3i(faces.x(), faces.y(), faces.z())));
}
ChVector3d ChFsiFluidSystemSPH_MI355X::GetGravitationalAcceleration() const { return cv(m_twin.GetGravitationalAcceleration()); }
std::vector<ChVector3d> ChFsiFluidSystemSPH_MI355X::GetParticlePositions() { return cvv(m_twin.GetParticlePositions()); }
std::vector<ChVector3d> ChFsiFluidSystemSPH_MI355X::GetParticleVelocities() { return cvv(m_twin.GetParticleVelocities()); }
std::vector<ChVector3d> ChFsiFluidSystemSPH_MI355X::GetParticleAccelerations() { return cvv(m_twin.GetParticleAccelerations()); }
std::vector<ChVector3d> ChFsiFluidSystemSPH_MI355X::GetParticleFluidProperties() { return cvv(m_twin.GetParticleFluidProperties()); }

// ---- ChFsiFluidSystem virtuals: one C-ABI call each (through the twin, which throws std::runtime_error with the solver's message)
void ChFsiFluidSystemSPH_MI355X::OnDoStepDynamics(double time, double step) { m_twin.OnDoStepDynamics(time, step); }  // sph_step
void ChFsiFluidSystemSPH_MI355X::OnExchangeSolidForces() { m_twin.OnExchangeSolidForces(); }                          // sph_compute_body_wrenches
void ChFsiFluidSystemSPH_MI355X::OnExchangeSolidStates() { m_twin.OnExchangeSolidStates(); }                          // sph_update_body_markers
void ChFsiFluidSystemSPH_MI355X::LoadSolidStates(const std::vector<FsiBodyState>& body_states) {                      // sph_load_body_states
    std::vector<chrono_mi::fsi::FsiBodyState> s;
    s.reserve(body_states.size());
    for (const auto& b : body_states) s.push_back(cs(b));
    m_twin.LoadSolidStates(s);
}
void ChFsiFluidSystemSPH_MI355X::StoreSolidForces(std::vector<FsiBodyForce> body_forces) { (void)body_forces; }
double ChFsiFluidSystemSPH_MI355X::GetCurrentStepSize() { return m_twin.GetCurrentStepSize(); }                       // sph_current_step

std::vector<FsiBodyForce> ChFsiFluidSystemSPH_MI355X::GetBodyForces() {                                               // sph_get_body_wrenches
    auto f = m_twin.GetBodyForces();
    std::vector<FsiBodyForce> out(f.size());
    for (size_t i = 0; i < f.size(); i++) {
        out[i].force = cv(f[i].force);
        out[i].torque = cv(f[i].torque);
    }
    return out;
}

// ---- construction hooks called by ChFsiSystem / ChFsiInterface
void ChFsiFluidSystemSPH_MI355X::OnAddFsiBody(std::shared_ptr<FsiBody> fsi_body, bool check_embedded) {
    (void)check_embedded;  // pruning of SPH particles embedded in the body is the job of the problem builder (ChFsiProblemSPH)
    if (m_is_initialized) throw std::runtime_error("OnAddFsiBody: system already initialized");
    // CreateBCEFsiBody (sph/ChFsiFluidSystemSPH.cpp:983-1020): BCE points of every collision shape in the body REFERENCE frame ...
    std::vector<ChVector3d> in_ref;
    const auto& geometry = fsi_body->geometry;
    if (geometry) {
        for (const auto& sphere : geometry->coll_spheres)
            for (const auto& p : m_twin.CreatePointsSphereInterior(sphere.radius, true)) in_ref.push_back(cv(p) + sphere.pos);
        for (const auto& box : geometry->coll_boxes)
            for (const auto& p : m_twin.CreatePointsBoxInterior(cv(box.dims))) in_ref.push_back(box.pos + box.rot.Rotate(cv(p)));
        for (const auto& cyl : geometry->coll_cylinders)
            for (const auto& p : m_twin.CreatePointsCylinderInterior(cyl.radius, cyl.length, true)) in_ref.push_back(cyl.pos + cyl.rot.Rotate(cv(p)));
        for (const auto& mesh : geometry->coll_meshes) {  // sph/ChFsiFluidSystemSPH.cpp:1005-1008: every grid point inside the (closed) surface
            std::vector<chrono_mi::ChVector3d> verts;
            std::vector<int> faces;
            for (const auto& v : mesh.trimesh->GetCoordsVertices()) verts.push_back(cv(v));
            for (const auto& f : mesh.trimesh->GetIndicesVertexes()) { faces.push_back(f.x()); faces.push_back(f.y()); faces.push_back(f.z()); }
            for (const auto& p : m_twin.CreatePointsMesh(verts, faces)) in_ref.push_back(cv(p));
        }
    }
    // ... then relative to the centroidal frame, which is the frame the solver's body states refer to
    const ChFrame<>& X_G_R = fsi_body->body->GetFrameRefToAbs();
    const ChFrame<>& X_G_COM = fsi_body->body->GetFrameCOMToAbs();
    std::vector<chrono_mi::ChVector3d> local;
    local.reserve(in_ref.size());
    for (const auto& p : in_ref) local.push_back(cv(X_G_COM.TransformPointParentToLocal(X_G_R.TransformPointLocalToParent(p))));
    FsiBodyState s0;
    s0.pos = fsi_body->body->GetPos();
    s0.rot = fsi_body->body->GetRot();
    s0.lin_vel = fsi_body->body->GetPosDt();
    s0.ang_vel = fsi_body->body->GetAngVelLocal();
    s0.lin_acc = fsi_body->body->GetPosDt2();
    s0.ang_acc = fsi_body->body->GetAngAccLocal();
    const size_t index = m_twin.AddFsiBody(local, cs(s0));
    if (index != fsi_body->index) throw std::runtime_error("OnAddFsiBody: FSI body index mismatch");
    m_bodies.push_back(fsi_body);
}

void ChFsiFluidSystemSPH_MI355X::Initialize(const std::vector<FsiBodyState>& body_states) {
    if (body_states.size() != m_bodies.size()) throw std::runtime_error("Initialize: number of body states differs from the number of FSI bodies");
    m_twin.SetStepSize(GetStepSize());
    std::vector<chrono_mi::fsi::FsiBodyState> s;
    s.reserve(body_states.size());
    for (const auto& b : body_states) s.push_back(cs(b));
    // sph_create + sph_add_markers (fluid, wall) + sph_add_rigid_body per body + sph_initialize; derived parameters read back
    m_twin.Initialize(s);
    // m_is_initialized of THIS object is set by ChFsiSystem::Initialize through friendship (ChFsiSystem.cpp:215)
}

#ifdef CHRONO_FEA
void ChFsiFluidSystemSPH_MI355X::LoadSolidStates(const std::vector<FsiBodyState>& b, const std::vector<FsiMeshState>& m1, const std::vector<FsiMeshState>& m2) {
    if (!m1.empty() || !m2.empty()) throw std::runtime_error("MI355X SPH solver: flexible FSI solids are not supported");
    LoadSolidStates(b);
}
void ChFsiFluidSystemSPH_MI355X::StoreSolidForces(std::vector<FsiBodyForce>, std::vector<FsiMeshForce>, std::vector<FsiMeshForce>) {}
void ChFsiFluidSystemSPH_MI355X::OnAddFsiMesh1D(std::shared_ptr<FsiMesh1D>, bool) { throw std::runtime_error("MI355X SPH solver: flexible FSI solids are not supported"); }
void ChFsiFluidSystemSPH_MI355X::OnAddFsiMesh2D(std::shared_ptr<FsiMesh2D>, bool) { throw std::runtime_error("MI355X SPH solver: flexible FSI solids are not supported"); }
void ChFsiFluidSystemSPH_MI355X::Initialize(const std::vector<FsiBodyState>& b, const std::vector<FsiMeshState>& m1, const std::vector<FsiMeshState>& m2) {
    if (!m1.empty() || !m2.empty()) throw std::runtime_error("MI355X SPH solver: flexible FSI solids are not supported");
    Initialize(b);
}
#endif

}  // namespace sph
}  // namespace fsi
}  // namespace chrono

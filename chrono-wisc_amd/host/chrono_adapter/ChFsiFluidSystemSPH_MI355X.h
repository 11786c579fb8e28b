// =============================================================================
// ChFsiFluidSystemSPH_MI355X.h - Chrono-side adapter: a chrono::fsi::ChFsiFluidSystem whose device layer is the MI355X solver
// (libchsph_mi.so, C ABI of include/chsph_mi.h).
//
// Written against the reference interfaces
//   src/chrono_fsi/ChFsiFluidSystem.h:37-152     (the virtuals this class overrides)
//   src/chrono_fsi/ChFsiDefinitions.h:41-64      (FsiBodyState, FsiBodyForce, FsiBody)
//   src/chrono_fsi/sph/ChFsiFluidSystemSPH.cpp:971-1020 (OnAddFsiBody / CreateBCEFsiBody)
// It lives in a Chrono source tree (e.g. src/chrono_fsi/sph_mi355x/) and needs Chrono core + Eigen to compile; this build image
// has neither, so THIS FILE IS NOT COMPILED HERE.  What is compiled and tested here is everything it forwards to: the
// Chrono-free twin chrono_mi::fsi::sph::ChFsiFluidSystemSPH (../ChFsiFluidSystemSPH_MI.{h,cpp}: parameter derivation, marker
// staging, BCE generators, exchange hooks; tests/test_parity_gpu.py::test_cxx_*), of which this class is a type-converting shell.
// =============================================================================
#ifndef CH_FSI_FLUID_SYSTEM_SPH_MI355X_H
#define CH_FSI_FLUID_SYSTEM_SPH_MI355X_H

#include <memory>
#include <vector>

#include "chrono/core/ChFrame.h"
#include "chrono/geometry/ChGeometry.h"
#include "chrono_fsi/ChFsiFluidSystem.h"
#include "chrono_fsi/ChFsiDefinitions.h"

#include "ChFsiFluidSystemSPH_MI.h"  // the Chrono-free twin (namespace chrono_mi)

namespace chrono {
namespace fsi {
namespace sph {

class ChFsiInterfaceSPH_MI355X;

/// SPH fluid system running on the MI355X solver.  Same construction sequence as ChFsiFluidSystemSPH:
/// set properties and parameters, add SPH particles and boundary BCE markers, attach to a ChFsiSystem (which adds the FSI bodies and
/// calls Initialize), then ChFsiSystem::DoStepDynamics.
class ChFsiFluidSystemSPH_MI355X : public ChFsiFluidSystem {
  public:
    using Twin = chrono_mi::fsi::sph::ChFsiFluidSystemSPH;
    using FluidProperties = Twin::FluidProperties;
    using ElasticMaterialProperties = Twin::ElasticMaterialProperties;
    using SPHParameters = Twin::SPHParameters;
    using LinSolverParameters = Twin::LinSolverParameters;

    explicit ChFsiFluidSystemSPH_MI355X(int device = -1);
    ~ChFsiFluidSystemSPH_MI355X();

    /// Every setter / getter / generator of the facade that takes no Chrono type (SetSPHParameters, SetCfdSPH, SetElasticSPH,
    /// SetShiftingMethod, SetNumProximitySearchSteps, SetArithmetic, GetKernelLength, GetNumFluidMarkers, SaveParticleData, ...).
    Twin& Facade() { return m_twin; }
    const Twin& Facade() const { return m_twin; }

    // ---- the part of the facade surface that carries Chrono types (sph/ChFsiFluidSystemSPH.h:136-517)
    virtual void SetGravitationalAcceleration(const ChVector3d& gravity) override;
    void SetBodyForce(const ChVector3d& force);
    void SetContainerDim(const ChVector3d& box_dim);
    void SetComputationalDomain(const ChAABB& computational_AABB, chrono_mi::fsi::sph::BoundaryConditions bc_type);
    void SetComputationalDomain(const ChAABB& computational_AABB);
    void AddSPHParticle(const ChVector3d& pos, double rho, double pres, double mu, const ChVector3d& vel = ChVector3d(0),
                        const ChVector3d& tauXxYyZz = ChVector3d(0), const ChVector3d& tauXyXzYz = ChVector3d(0), double pc = 1e3);
    void AddSPHParticle(const ChVector3d& pos, const ChVector3d& vel = ChVector3d(0));
    void AddBoxSPH(const ChVector3d& boxCenter, const ChVector3d& boxHalfDim);
    void AddBCEBoundary(const std::vector<ChVector3d>& points, const ChFramed& frame);
    std::vector<ChVector3d> CreatePointsBoxContainer(const ChVector3d& size, const ChVector3i& faces) const;
    ChVector3d GetGravitationalAcceleration() const;
    std::vector<ChVector3d> GetParticlePositions();
    std::vector<ChVector3d> GetParticleVelocities();
    std::vector<ChVector3d> GetParticleAccelerations();
    std::vector<ChVector3d> GetParticleFluidProperties();

    // ---- ChFsiFluidSystem virtuals (ChFsiFluidSystem.h:69-110)
    virtual void OnDoStepDynamics(double time, double step) override;
    virtual void OnExchangeSolidForces() override;
    virtual void OnExchangeSolidStates() override;
    virtual void LoadSolidStates(const std::vector<FsiBodyState>& body_states) override;
    /// By value, like the reference (ChFsiFluidSystem.h:89): nothing can reach the caller through it.  ChFsiInterfaceSPH_MI355X reads
    /// the wrenches through GetBodyForces() instead (the route the reference's own ChFsiInterfaceSPH takes).
    virtual void StoreSolidForces(std::vector<FsiBodyForce> body_forces) override;
    virtual double GetCurrentStepSize() override;
#ifdef CHRONO_FEA
    // flexible solids are outside the scope of this solver (FEA coupling out of scope)
    virtual void LoadSolidStates(const std::vector<FsiBodyState>&, const std::vector<FsiMeshState>&, const std::vector<FsiMeshState>&) override;
    virtual void StoreSolidForces(std::vector<FsiBodyForce>, std::vector<FsiMeshForce>, std::vector<FsiMeshForce>) override;
#endif

    /// Fluid force at the COM and torque on every FSI body, global frame (what ChFsiInterfaceSPH::ExchangeSolidForces copies back).
    std::vector<FsiBodyForce> GetBodyForces();

  protected:
    /// Body-local BCE markers from the body's collision geometry (spheres, boxes, cylinders), as CreateBCEFsiBody does.
    virtual void OnAddFsiBody(std::shared_ptr<FsiBody> fsi_body, bool check_embedded) override;
    virtual void Initialize(const std::vector<FsiBodyState>& body_states) override;
#ifdef CHRONO_FEA
    virtual void OnAddFsiMesh1D(std::shared_ptr<FsiMesh1D> mesh, bool check_embedded) override;
    virtual void OnAddFsiMesh2D(std::shared_ptr<FsiMesh2D> mesh, bool check_embedded) override;
    virtual void Initialize(const std::vector<FsiBodyState>&, const std::vector<FsiMeshState>&, const std::vector<FsiMeshState>&) override;
#endif

  private:
    static chrono_mi::ChVector3d cv(const ChVector3d& v) { return chrono_mi::ChVector3d(v.x(), v.y(), v.z()); }
    static ChVector3d cv(const chrono_mi::ChVector3d& v) { return ChVector3d(v.x(), v.y(), v.z()); }
    static chrono_mi::fsi::FsiBodyState cs(const FsiBodyState& s);
    static std::vector<ChVector3d> cvv(const std::vector<chrono_mi::ChVector3d>& v);

    Twin m_twin;
    std::vector<std::shared_ptr<FsiBody>> m_bodies;  // in FSI-body index order

    friend class ChFsiInterfaceSPH_MI355X;
};

}  // namespace sph
}  // namespace fsi
}  // namespace chrono

#endif

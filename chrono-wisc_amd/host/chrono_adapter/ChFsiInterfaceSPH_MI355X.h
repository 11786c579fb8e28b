// =============================================================================
// ChFsiInterfaceSPH_MI355X.h - twin of chrono::fsi::sph::ChFsiInterfaceSPH (src/chrono_fsi/sph/ChFsiInterfaceSPH.{h,cpp}) for
// ChFsiFluidSystemSPH_MI355X: the direct exchange between a Chrono multibody system and the MI355X fluid solver.
// NOT COMPILED IN THIS BUILD IMAGE (needs Chrono core + Eigen); its Chrono-free twin is chrono_mi::fsi::sph::ChFsiInterfaceSPH
// (../ChFsiSystem_MI.{h,cpp}), tested in tests/test_parity_gpu.py::test_cxx_cosimulation_shell_object_drop.
// =============================================================================
#ifndef CH_FSI_INTERFACE_SPH_MI355X_H
#define CH_FSI_INTERFACE_SPH_MI355X_H

#include "chrono_fsi/ChFsiInterface.h"

#include "ChFsiFluidSystemSPH_MI355X.h"

namespace chrono {
namespace fsi {
namespace sph {

class ChFsiInterfaceSPH_MI355X : public ChFsiInterface {
  public:
    ChFsiInterfaceSPH_MI355X(ChSystem* sysMBS, ChFsiFluidSystemSPH_MI355X* sysSPH) : ChFsiInterface(sysMBS, sysSPH), m_sysSPH(sysSPH) {}
    ~ChFsiInterfaceSPH_MI355X() {}

    /// MBS -> fluid: body poses, velocities and accelerations into the solver's body-state table (sph_load_body_states).
    /// Angular velocity and acceleration in the BODY-LOCAL frame, as the solver's kernels expect (sph/ChFsiInterfaceSPH.cpp:46-47).
    virtual void ExchangeSolidStates() override;
    /// fluid -> MBS: wrenches out of the solver (sph_get_body_wrenches), applied through each body's FSI force accumulator at the
    /// body position (sph/ChFsiInterfaceSPH.cpp:112-124).
    virtual void ExchangeSolidForces() override;

  private:
    ChFsiFluidSystemSPH_MI355X* m_sysSPH;
};

}  // namespace sph
}  // namespace fsi
}  // namespace chrono

#endif

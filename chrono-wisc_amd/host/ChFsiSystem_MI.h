// =============================================================================
// ChFsiSystem_MI.h - the generic co-simulation shell around the MI355X fluid solver.
//
// Chrono-free mirror of
//   chrono::fsi::ChFsiSystem            (src/chrono_fsi/ChFsiSystem.{h,cpp}: Initialize :176-217, AdvanceCFD :229-246,
//                                        AdvanceMBS :248-262, DoStepDynamics :264-303, timers / RTF ChFsiSystem.h:152-181)
//   chrono::fsi::ChFsiInterface         (src/chrono_fsi/ChFsiInterface.{h,cpp}: state / force vectors, StoreSolidStates,
//                                        LoadSolidForces)
//   chrono::fsi::ChFsiInterfaceGeneric  (ChFsiInterface.cpp:603-651: cached exchange buffers)
//   chrono::fsi::sph::ChFsiInterfaceSPH (sph/ChFsiInterfaceSPH.cpp:37-125: the direct exchange the SPH solver uses)
// with the same method names, step sequence and threading: the multibody step runs on a second std::thread while the fluid
// step runs on the caller's thread; all four exchange hooks run on the caller's thread after the join.
//
// The multibody side is a stand-in (ChSystemMI / ChBodyMI: free rigid bodies, semi-implicit Euler, the arithmetic of
// chrono-wisc_amd/rigid.py) -- Chrono's own ChSystem needs Eigen, which this image does not have; the shell only touches it
// through the same narrow surface the reference uses (DoStepDynamics, GetTimerStep, body pose / velocity getters, force
// accumulators), and a user-supplied MBDCallback (ChFsiSystem.h:60-70) replaces it entirely.  The adapter that derives from
// the real chrono::fsi classes is in chrono_adapter/.
// =============================================================================
#ifndef CH_FSI_SYSTEM_MI_H
#define CH_FSI_SYSTEM_MI_H

#include <chrono>
#include <memory>
#include <vector>

#include "ChFsiFluidSystemSPH_MI.h"

namespace chrono_mi {

/// Stopwatch with the surface of chrono::ChTimer (src/chrono/core/ChTimer.h): reset / start / stop / operator().
class ChTimer {
  public:
    void reset() { m_total = 0; }
    void start() { m_t0 = std::chrono::steady_clock::now(); }
    void stop() { m_total += std::chrono::duration<double>(std::chrono::steady_clock::now() - m_t0).count(); }
    double operator()() const { return m_total; }

  private:
    std::chrono::steady_clock::time_point m_t0;
    double m_total = 0;
};

/// Free rigid body of the multibody stand-in: what ChFsiInterface reads (pose, velocities, accelerations; angular quantities
/// in the BODY-LOCAL frame, sph/ChFsiInterfaceSPH.cpp:46-47) and writes (force / torque accumulators, :120-122).
class ChBodyMI {
  public:
    ChBodyMI(double mass, const ChVector3d& principal_inertia, const ChVector3d& pos, const ChQuaterniond& rot = QUNIT);
    void SetFixed(bool fixed) { m_fixed = fixed; }
    void SetGravity(const ChVector3d& g) { m_gravity = g; }
    // getters named like ChBody's
    const ChVector3d& GetPos() const { return pos; }
    const ChQuaterniond& GetRot() const { return rot; }
    const ChVector3d& GetPosDt() const { return lin_vel; }
    const ChVector3d& GetPosDt2() const { return lin_acc; }
    const ChVector3d& GetAngVelLocal() const { return ang_vel; }
    const ChVector3d& GetAngAccLocal() const { return ang_acc; }
    // ChBody accumulators (ChBody.h: EmptyAccumulator / AccumulateForce / AccumulateTorque), global frame
    void EmptyAccumulator() { acc_force = ChVector3d(); acc_torque = ChVector3d(); }
    void AccumulateForce(const ChVector3d& f) { acc_force = acc_force + f; }  ///< applied at the body position (the COM)
    void AccumulateTorque(const ChVector3d& t) { acc_torque = acc_torque + t; }
    /// semi-implicit Euler under gravity and the accumulated wrench (the arithmetic of rigid.RigidBody.advance)
    void Advance(double h);

    ChVector3d pos;
    ChQuaterniond rot;
    ChVector3d lin_vel, ang_vel, lin_acc, ang_acc;
    ChVector3d acc_force, acc_torque;

  private:
    double m_mass;
    ChVector3d m_inertia;  // principal moments, body frame
    ChVector3d m_gravity;
    bool m_fixed;
};

/// Multibody system stand-in with the surface ChFsiSystem uses of chrono::ChSystem.
class ChSystemMI {
  public:
    void AddBody(std::shared_ptr<ChBodyMI> body) { m_bodies.push_back(body); }
    void SetGravitationalAcceleration(const ChVector3d& g) { for (auto& b : m_bodies) b->SetGravity(g); m_gravity = g; }
    void DoStepDynamics(double h);
    double GetChTime() const { return m_time; }
    double GetTimerStep() const { return m_timer_step; }
    const std::vector<std::shared_ptr<ChBodyMI>>& GetBodies() const { return m_bodies; }

  private:
    std::vector<std::shared_ptr<ChBodyMI>> m_bodies;
    ChVector3d m_gravity = ChVector3d(0, 0, -9.8);
    double m_time = 0, m_timer_step = 0;
};

namespace fsi {

/// ChFsiDefinitions.h:57-64
struct FsiBody {
    size_t index;
    std::shared_ptr<ChBodyMI> body;
    ChVector3d fsi_force, fsi_torque;
};

/// Base class of the exchange between the two phases (ChFsiInterface.h).
class ChFsiInterface {
  public:
    ChFsiInterface(ChSystemMI* sysMBS, ChFsiFluidSystem* sysCFD) : m_sysMBS(sysMBS), m_sysCFD(sysCFD) {}
    virtual ~ChFsiInterface() {}
    std::shared_ptr<FsiBody> AddFsiBody(std::shared_ptr<ChBodyMI> body);  // ChFsiInterface.cpp:54-70
    size_t GetNumBodies() const { return m_fsi_bodies.size(); }
    const ChVector3d& GetFsiBodyForce(size_t i) const { return m_fsi_bodies[i]->fsi_force; }
    const ChVector3d& GetFsiBodyTorque(size_t i) const { return m_fsi_bodies[i]->fsi_torque; }
    virtual void Initialize() {}
    void AllocateStateVectors(std::vector<FsiBodyState>& body_states) const { body_states.resize(m_fsi_bodies.size()); }
    void AllocateForceVectors(std::vector<FsiBodyForce>& body_forces) const { body_forces.resize(m_fsi_bodies.size()); }
    /// MBS -> records (ChFsiInterface.cpp: StoreSolidStates)
    void StoreSolidStates(std::vector<FsiBodyState>& body_states) const;
    /// records -> MBS as external loads (ChFsiInterface.cpp: LoadSolidForces)
    void LoadSolidForces(const std::vector<FsiBodyForce>& body_forces);
    virtual void ExchangeSolidStates() = 0;
    virtual void ExchangeSolidForces() = 0;

  protected:
    ChSystemMI* m_sysMBS;
    ChFsiFluidSystem* m_sysCFD;
    std::vector<std::shared_ptr<FsiBody>> m_fsi_bodies;
};

/// Exchange through cached vectors and the fluid system's generic virtuals (ChFsiInterface.cpp:603-651).  The reference's
/// StoreSolidForces takes its vector BY VALUE (ChFsiFluidSystem.h:89), so nothing can come back through it; like the reference's
/// SPH solver this build therefore reads the forces through the solver-specific accessor when the fluid system offers one.
class ChFsiInterfaceGeneric : public ChFsiInterface {
  public:
    ChFsiInterfaceGeneric(ChSystemMI* sysMBS, ChFsiFluidSystem* sysCFD) : ChFsiInterface(sysMBS, sysCFD) {}
    void Initialize() override;
    void ExchangeSolidStates() override;
    void ExchangeSolidForces() override;

  private:
    std::vector<FsiBodyState> m_body_states;
    std::vector<FsiBodyForce> m_body_forces;
};

namespace sph {
/// The direct exchange of the SPH solver (sph/ChFsiInterfaceSPH.cpp:37-125): states straight into the solver's body-state
/// table, forces straight out of its wrench table.
class ChFsiInterfaceSPH : public ChFsiInterface {
  public:
    ChFsiInterfaceSPH(ChSystemMI* sysMBS, ChFsiFluidSystemSPH* sysSPH) : ChFsiInterface(sysMBS, sysSPH), m_sysSPH(sysSPH) {}
    void ExchangeSolidStates() override;
    void ExchangeSolidForces() override;

  private:
    ChFsiFluidSystemSPH* m_sysSPH;
};
}  // namespace sph

/// The co-simulation driver (ChFsiSystem.h:36-200).
class ChFsiSystem {
  public:
    /// User-supplied advance of the multibody phase (ChFsiSystem.h:60-70).
    class MBDCallback {
      public:
        virtual ~MBDCallback() {}
        virtual void Advance(double step, double threshold) = 0;
    };

    ChFsiSystem(ChSystemMI* sysMBS, ChFsiFluidSystem* sysCFD, std::shared_ptr<ChFsiInterface> fsi_interface);
    virtual ~ChFsiSystem();
    void SetGravitationalAcceleration(const ChVector3d& gravity);  // both phases (ChFsiSystem.cpp:76-82)
    void SetStepSizeCFD(double step) { m_step_CFD = step; }
    void SetStepsizeMBD(double step) { m_step_MBD = step; }
    void RegisterMBDCallback(std::shared_ptr<MBDCallback> callback) { m_MBD_callback = callback; }
    std::shared_ptr<FsiBody> AddFsiBody(std::shared_ptr<ChBodyMI> body) { return m_fsi_interface->AddFsiBody(body); }
    void Initialize();
    void DoStepDynamics(double step);
    double GetSimTime() const { return m_time; }
    double GetStepSizeCFD() const { return m_step_CFD; }
    double GetStepSizeMBD() const { return m_step_MBD; }
    double GetRtf() const { return m_RTF; }                 ///< wall time of the two phases / simulated time
    double GetRatioMBD() const { return m_ratio_MBD; }      ///< MBS timer / CFD timer
    double GetTimerStep() const { return m_timer_step(); }  ///< both phases (concurrent), last step
    double GetTimerFSI() const { return m_timer_FSI(); }    ///< the four exchange hooks, last step
    double GetTimerCFD() const { return m_timer_CFD; }
    double GetTimerMBD() const { return m_timer_MBD; }
    const ChVector3d& GetFsiBodyForce(size_t i) const { return m_fsi_interface->GetFsiBodyForce(i); }
    const ChVector3d& GetFsiBodyTorque(size_t i) const { return m_fsi_interface->GetFsiBodyTorque(i); }

  private:
    void AdvanceCFD(double step, double threshold);
    void AdvanceMBS(double step, double threshold);
    struct MbsWorker;  ///< the multibody phase's persistent thread (condition-variable hand-off; the reference spawns a thread per step)
    std::unique_ptr<MbsWorker> m_worker;

    ChSystemMI* m_sysMBS;
    ChFsiFluidSystem* m_sysCFD;
    std::shared_ptr<ChFsiInterface> m_fsi_interface;
    std::shared_ptr<MBDCallback> m_MBD_callback;
    bool m_is_initialized = false;
    double m_step_MBD = -1, m_step_CFD = -1, m_time = 0;
    ChTimer m_timer_step, m_timer_FSI;
    double m_timer_CFD = 0, m_timer_MBD = 0, m_RTF = 0, m_ratio_MBD = 0;
};

}  // namespace fsi
}  // namespace chrono_mi

#endif

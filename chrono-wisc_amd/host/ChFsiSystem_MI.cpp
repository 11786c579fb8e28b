// =============================================================================
// ChFsiSystem_MI.cpp - generic co-simulation shell (see ChFsiSystem_MI.h).
// Reference for the behaviour: src/chrono_fsi/ChFsiSystem.cpp:176-303, ChFsiInterface.cpp, sph/ChFsiInterfaceSPH.cpp:37-125.
// =============================================================================
#include "ChFsiSystem_MI.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

namespace chrono_mi {

// ---- multibody stand-in ------------------------------------------------------------------------------------------------
static void rotmat(const ChQuaterniond& q, double R[3][3]) {  // ChMatrix33(quaternion), src/chrono/core/ChMatrix33.h:253-274
    const double e0 = q.e[0], e1 = q.e[1], e2 = q.e[2], e3 = q.e[3];
    R[0][0] = (e0 * e0 + e1 * e1) * 2 - 1; R[0][1] = (e1 * e2 - e0 * e3) * 2;     R[0][2] = (e1 * e3 + e0 * e2) * 2;
    R[1][0] = (e1 * e2 + e0 * e3) * 2;     R[1][1] = (e0 * e0 + e2 * e2) * 2 - 1; R[1][2] = (e2 * e3 - e0 * e1) * 2;
    R[2][0] = (e1 * e3 - e0 * e2) * 2;     R[2][1] = (e2 * e3 + e0 * e1) * 2;     R[2][2] = (e0 * e0 + e3 * e3) * 2 - 1;
}

ChBodyMI::ChBodyMI(double mass, const ChVector3d& principal_inertia, const ChVector3d& p, const ChQuaterniond& q)
    : pos(p), rot(q), m_mass(mass), m_inertia(principal_inertia), m_gravity(0, 0, -9.8), m_fixed(false) {}

void ChBodyMI::Advance(double h) {
    if (m_fixed) return;
    double R[3][3];
    rotmat(rot, R);
    lin_acc = m_gravity + acc_force / m_mass;
    // torque in the body frame, Euler's equations with a principal inertia tensor
    ChVector3d tl(R[0][0] * acc_torque[0] + R[1][0] * acc_torque[1] + R[2][0] * acc_torque[2],
                  R[0][1] * acc_torque[0] + R[1][1] * acc_torque[1] + R[2][1] * acc_torque[2],
                  R[0][2] * acc_torque[0] + R[1][2] * acc_torque[1] + R[2][2] * acc_torque[2]);
    const ChVector3d Iw(m_inertia[0] * ang_vel[0], m_inertia[1] * ang_vel[1], m_inertia[2] * ang_vel[2]);
    const ChVector3d gyro(ang_vel[1] * Iw[2] - ang_vel[2] * Iw[1], ang_vel[2] * Iw[0] - ang_vel[0] * Iw[2], ang_vel[0] * Iw[1] - ang_vel[1] * Iw[0]);
    ang_acc = ChVector3d((tl[0] - gyro[0]) / m_inertia[0], (tl[1] - gyro[1]) / m_inertia[1], (tl[2] - gyro[2]) / m_inertia[2]);
    lin_vel = lin_vel + lin_acc * h;
    ang_vel = ang_vel + ang_acc * h;
    pos = pos + lin_vel * h;
    // q' = q + h/2 q (x) (0, w_local), renormalised
    const double* a = rot.e;
    const double b[4] = {0.0, ang_vel[0], ang_vel[1], ang_vel[2]};
    const double dq[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                          a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    double q[4], nrm = 0;
    for (int k = 0; k < 4; k++) { q[k] = a[k] + dq[k] * (0.5 * h); nrm += q[k] * q[k]; }
    nrm = std::sqrt(nrm);
    rot = ChQuaterniond(q[0] / nrm, q[1] / nrm, q[2] / nrm, q[3] / nrm);
}

void ChSystemMI::DoStepDynamics(double h) {
    auto t0 = std::chrono::steady_clock::now();
    for (auto& b : m_bodies) b->Advance(h);
    m_time += h;
    m_timer_step = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

namespace fsi {

// ---- ChFsiInterface ----------------------------------------------------------------------------------------------------
std::shared_ptr<FsiBody> ChFsiInterface::AddFsiBody(std::shared_ptr<ChBodyMI> body) {
    auto fsi_body = std::make_shared<FsiBody>();
    fsi_body->index = m_fsi_bodies.size();
    fsi_body->body = body;
    m_fsi_bodies.push_back(fsi_body);
    return fsi_body;
}

void ChFsiInterface::StoreSolidStates(std::vector<FsiBodyState>& body_states) const {
    if (body_states.size() != m_fsi_bodies.size()) throw std::runtime_error("ChFsiInterface::StoreSolidStates: incorrect size of state vector");
    size_t i = 0;
    for (const auto& fb : m_fsi_bodies) {
        body_states[i].pos = fb->body->GetPos();
        body_states[i].rot = fb->body->GetRot();
        body_states[i].lin_vel = fb->body->GetPosDt();
        body_states[i].lin_acc = fb->body->GetPosDt2();
        body_states[i].ang_vel = fb->body->GetAngVelLocal();  // body-local frame (sph/ChFsiInterfaceSPH.cpp:46-47)
        body_states[i].ang_acc = fb->body->GetAngAccLocal();
        i++;
    }
}

void ChFsiInterface::LoadSolidForces(const std::vector<FsiBodyForce>& body_forces) {
    if (body_forces.size() != m_fsi_bodies.size()) throw std::runtime_error("ChFsiInterface::LoadSolidForces: incorrect size of force vector");
    size_t i = 0;
    for (auto& fb : m_fsi_bodies) {
        fb->fsi_force = body_forces[i].force;
        fb->fsi_torque = body_forces[i].torque;
        fb->body->EmptyAccumulator();
        fb->body->AccumulateForce(fb->fsi_force);  // at the body position
        fb->body->AccumulateTorque(fb->fsi_torque);
        i++;
    }
}

// ---- ChFsiInterfaceGeneric ---------------------------------------------------------------------------------------------
void ChFsiInterfaceGeneric::Initialize() {
    ChFsiInterface::Initialize();
    AllocateStateVectors(m_body_states);
    AllocateForceVectors(m_body_forces);
}
void ChFsiInterfaceGeneric::ExchangeSolidStates() {
    StoreSolidStates(m_body_states);          // get solid states from the multibody system in cached vectors
    m_sysCFD->LoadSolidStates(m_body_states);  // pass them to the fluid solver
}
void ChFsiInterfaceGeneric::ExchangeSolidForces() {
    m_sysCFD->StoreSolidForces(m_body_forces);  // by value in the reference's interface: a no-op for the caller (ChFsiFluidSystem.h:89)
    if (auto* s = dynamic_cast<sph::ChFsiFluidSystemSPH*>(m_sysCFD)) {
        auto f = s->GetBodyForces();
        for (size_t i = 0; i < m_body_forces.size() && i < f.size(); i++) m_body_forces[i] = f[i];
    }
    LoadSolidForces(m_body_forces);  // apply as external loads
}

// ---- sph::ChFsiInterfaceSPH --------------------------------------------------------------------------------------------
namespace sph {
void ChFsiInterfaceSPH::ExchangeSolidStates() {
    std::vector<FsiBodyState> states(m_fsi_bodies.size());
    StoreSolidStates(states);
    m_sysSPH->LoadSolidStates(states);  // host table -> device (fsiBodyState_H -> fsiBodyState_D, :49-52)
}
void ChFsiInterfaceSPH::ExchangeSolidForces() {
    auto forces = m_sysSPH->GetBodyForces();  // device -> host (rigid_FSI_ForcesD / TorquesD, :112-113)
    LoadSolidForces(forces);
}
}  // namespace sph

// ---- ChFsiSystem -------------------------------------------------------------------------------------------------------
// What the reference's driver guarantees (src/chrono_fsi/ChFsiSystem.cpp:177-303), restated as the contract this file implements:
//   * Initialize: step sizes default from the fluid system / from each other; interface first, then the fluid system with the
//     initial solid states.
//   * DoStepDynamics(step): the multibody phase and the fluid phase both advance by `step` from the SAME exchanged data, concurrently
//     (the multibody phase touches nothing but the multibody system); each phase sub-steps with its own step size and drops a
//     remainder below 1e-6 of it; afterwards, on the caller's thread and in this order: fluid forces computed, forces handed to the
//     solids, new solid states handed to the fluid solver, fluid solver updates its markers.
//   * timers: both phases together, the exchange, each phase alone; RTF and the MBD / CFD ratio from them.
// The reference spawns a std::thread per call (:282-284).  At the 0.5-2 ms a fluid step takes here that is a measurable tax, so the
// multibody phase has ONE worker thread for the lifetime of the system: it sleeps on a condition variable, is handed (step, cut-off)
// pairs, and reports completion -- or the exception it caught -- through the same mutex.

// the sub-stepping both phases share: advance by `span` in pieces of at most size(), stop when what is left is not worth a step
template <class SizeFn, class StepFn>
static void advance_in_pieces(double span, double cutoff, SizeFn size, StepFn step) {
    for (double done = 0; done < span;) {
        const double piece = std::min(size(), span - done);
        if (piece <= cutoff) return;
        step(piece);
        done += piece;
    }
}

struct ChFsiSystem::MbsWorker {
    std::mutex m;
    std::condition_variable wake, finished;
    std::thread th;
    bool has_job = false, quit = false, busy = false;
    double step = 0, cutoff = 0;
    std::exception_ptr error;

    explicit MbsWorker(ChFsiSystem* owner) {
        th = std::thread([this, owner] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                wake.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                has_job = false;
                const double s = step, c = cutoff;
                lk.unlock();
                std::exception_ptr e;
                try {
                    owner->AdvanceMBS(s, c);
                } catch (...) {
                    e = std::current_exception();
                }
                lk.lock();
                error = e;
                busy = false;
                finished.notify_one();
            }
        });
    }
    ~MbsWorker() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        wake.notify_one();
        th.join();
    }
    void post(double s, double c) {
        {
            std::lock_guard<std::mutex> lk(m);
            step = s; cutoff = c; has_job = true; busy = true; error = nullptr;
        }
        wake.notify_one();
    }
    void wait() {  // returns once the posted job is done; rethrows what it threw
        std::unique_lock<std::mutex> lk(m);
        finished.wait(lk, [this] { return !busy; });
        if (error) std::rethrow_exception(error);
    }
};

ChFsiSystem::ChFsiSystem(ChSystemMI* sysMBS, ChFsiFluidSystem* sysCFD, std::shared_ptr<ChFsiInterface> fsi_interface)
    : m_sysMBS(sysMBS), m_sysCFD(sysCFD), m_fsi_interface(fsi_interface) {}
ChFsiSystem::~ChFsiSystem() {}

void ChFsiSystem::SetGravitationalAcceleration(const ChVector3d& gravity) {
    m_sysCFD->SetGravitationalAcceleration(gravity);
    m_sysMBS->SetGravitationalAcceleration(gravity);
}

void ChFsiSystem::Initialize() {
    if (!m_sysCFD || !m_sysMBS || !m_fsi_interface) throw std::runtime_error("ChFsiSystem::Initialize: incomplete system");
    // step sizes: the fluid step from the fluid system unless set here, the multibody step from the fluid step unless set here
    if (m_step_CFD < 0) m_step_CFD = m_sysCFD->GetStepSize();
    if (m_step_CFD < 0) throw std::runtime_error("Integration step size for fluid dynamics not set.");
    if (m_step_MBD < 0) m_step_MBD = m_step_CFD;
    m_fsi_interface->Initialize();
    std::vector<FsiBodyState> initial;
    m_fsi_interface->AllocateStateVectors(initial);
    m_fsi_interface->StoreSolidStates(initial);
    m_sysCFD->SetStepSize(m_step_CFD);
    m_sysCFD->Initialize(initial);  // (sets its own initialized flag; the reference reaches into it as a friend, ChFsiSystem.cpp:215)
    m_worker = std::make_unique<MbsWorker>(this);
    m_is_initialized = true;
}

void ChFsiSystem::AdvanceCFD(double step, double threshold) {
    // a variable-step fluid solver proposes its own piece each time, never below a microsecond
    advance_in_pieces(
        step, threshold, [this] { return std::max(m_sysCFD->GetCurrentStepSize(), 1e-6); }, [this](double h) { m_sysCFD->DoStepDynamics(h); });
    m_timer_CFD = m_sysCFD->GetTimerStep();
}

void ChFsiSystem::AdvanceMBS(double step, double threshold) {
    if (m_MBD_callback) m_MBD_callback->Advance(step, threshold);
    else advance_in_pieces(step, threshold, [this] { return m_step_MBD; }, [this](double h) { m_sysMBS->DoStepDynamics(h); });
    m_timer_MBD = m_sysMBS->GetTimerStep();
}

void ChFsiSystem::DoStepDynamics(double step) {
    if (!m_is_initialized) throw std::runtime_error("FSI system not initialized!\n");
    constexpr double kCutoff = 1e-6;  // of the phase's own step size
    m_timer_step.reset();
    m_timer_FSI.reset();
    m_timer_step.start();
    m_worker->post(step, kCutoff * m_step_MBD);  // multibody phase: the worker thread
    try {
        AdvanceCFD(step, kCutoff * m_step_CFD);  // fluid phase: this thread (the solver context is not thread-safe and needs no locks)
    } catch (...) {
        try { m_worker->wait(); } catch (...) {}  // never leave the worker running behind an exception
        throw;
    }
    m_worker->wait();
    m_timer_step.stop();
    m_timer_FSI.start();
    m_sysCFD->OnExchangeSolidForces();       // fluid -> wrenches on the device
    m_fsi_interface->ExchangeSolidForces();  // wrenches -> external loads of the solids
    m_fsi_interface->ExchangeSolidStates();  // new solid states -> fluid solver
    m_sysCFD->OnExchangeSolidStates();       // fluid solver moves its body markers
    m_timer_FSI.stop();
    m_RTF = m_timer_step() / step;
    m_ratio_MBD = m_timer_CFD > 0 ? m_timer_MBD / m_timer_CFD : 0;
    m_time += step;
}

}  // namespace fsi
}  // namespace chrono_mi

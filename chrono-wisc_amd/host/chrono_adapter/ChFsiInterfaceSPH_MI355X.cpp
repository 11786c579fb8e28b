// ChFsiInterfaceSPH_MI355X.cpp - see the header.  NOT COMPILED IN THIS BUILD IMAGE (needs Chrono core + Eigen).
#include "ChFsiInterfaceSPH_MI355X.h"

#include "chrono/physics/ChBody.h"

namespace chrono {
namespace fsi {
namespace sph {

void ChFsiInterfaceSPH_MI355X::ExchangeSolidStates() {
    std::vector<FsiBodyState> states(m_fsi_bodies.size());
    size_t index = 0;
    for (const auto& fsi_body : m_fsi_bodies) {
        const auto& body = fsi_body->body;
        states[index].pos = body->GetPos();
        states[index].lin_vel = body->GetPosDt();
        states[index].lin_acc = body->GetPosDt2();
        states[index].rot = body->GetRot();
        states[index].ang_vel = body->GetAngVelLocal();
        states[index].ang_acc = body->GetAngAccLocal();
        index++;
    }
    m_sysSPH->LoadSolidStates(states);
}

void ChFsiInterfaceSPH_MI355X::ExchangeSolidForces() {
    const auto wrenches = m_sysSPH->GetBodyForces();
    size_t index = 0;
    for (const auto& fsi_body : m_fsi_bodies) {
        fsi_body->fsi_force = wrenches[index].force;
        fsi_body->fsi_torque = wrenches[index].torque;
        fsi_body->body->EmptyAccumulator(fsi_body->fsi_accumulator);
        fsi_body->body->AccumulateForce(fsi_body->fsi_accumulator, fsi_body->fsi_force, fsi_body->body->GetPos(), false);
        fsi_body->body->AccumulateTorque(fsi_body->fsi_accumulator, fsi_body->fsi_torque, false);
        index++;
    }
}

}  // namespace sph
}  // namespace fsi
}  // namespace chrono

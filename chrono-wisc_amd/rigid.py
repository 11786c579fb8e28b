"""Chrono-free stand-in for the multibody side of ChFsiSystem: free rigid bodies advanced by a
semi-implicit Euler step under gravity and the fluid wrench, and the exchange loop of ChFsiSystem::DoStepDynamics
(src/chrono_fsi/ChFsiSystem.cpp:264-303).  The solver only sees FsiBodyState records (fsi/ChFsiDefinitions.h:41-48):
position/orientation/velocities of the centroidal frame, angular velocity and acceleration in the BODY-LOCAL frame
(sph/ChFsiInterfaceSPH.cpp:46-47).  ChSystemNSC itself is out of scope; this is what the C4 measurement couples to."""
import numpy as np

from . import fsi


def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


class RigidBody:
    """One free body: mass, principal-frame inertia tensor (3x3, body frame), state in double precision."""

    def __init__(self, mass, inertia, pos, rot=fsi.QUNIT, gravity=(0, 0, -9.8), fixed=False):
        self.mass = float(mass)
        self.I = np.asarray(inertia, dtype=np.float64).reshape(3, 3)
        self.Iinv = np.linalg.inv(self.I)
        self.pos = np.asarray(pos, dtype=np.float64).copy()
        self.rot = np.asarray(rot, dtype=np.float64).copy()
        self.lin_vel = np.zeros(3)
        self.ang_vel = np.zeros(3)  # body-local
        self.lin_acc = np.zeros(3)
        self.ang_acc = np.zeros(3)  # body-local
        self.gravity = np.asarray(gravity, dtype=np.float64)
        self.fixed = fixed
        self.force = np.zeros(3)   # fluid force at the COM, global frame (FsiBodyForce, fsi/ChFsiDefinitions.h:51-54)
        self.torque = np.zeros(3)  # fluid torque, global frame

    def state(self):
        return dict(pos=self.pos, rot=self.rot, lin_vel=self.lin_vel, ang_vel=self.ang_vel, lin_acc=self.lin_acc, ang_acc=self.ang_acc)

    def advance(self, h):
        if self.fixed:
            return
        R = fsi.rotmat_from_quat(self.rot)
        self.lin_acc = self.gravity + self.force / self.mass
        tau_loc = R.T @ self.torque
        self.ang_acc = self.Iinv @ (tau_loc - np.cross(self.ang_vel, self.I @ self.ang_vel))
        self.lin_vel = self.lin_vel + h * self.lin_acc
        self.ang_vel = self.ang_vel + h * self.ang_acc
        self.pos = self.pos + h * self.lin_vel
        w = self.ang_vel
        dq = _qmul(self.rot, np.array([0.0, w[0], w[1], w[2]])) * (0.5 * h)
        q = self.rot + dq
        self.rot = q / np.linalg.norm(q)


class FsiSystem:
    """ChFsiSystem for free rigid bodies: DoStepDynamics advances both phases from the state at t, then exchanges."""

    def __init__(self, sysSPH, bodies, advance_cfd=None):
        """advance_cfd: the fluid step; default sysSPH.DoStepDynamics, a slab rank passes SlabRank.step (every rank then
        advances its own copy of the bodies from the same wrenches, so all copies stay identical)."""
        self.sph = sysSPH
        self.bodies = list(bodies)
        self.time = 0.0
        self.advance_cfd = advance_cfd or sysSPH.DoStepDynamics

    def DoStepDynamics(self, step):
        for b in self.bodies:      # AdvanceMBS (concurrent thread in the reference): uses the forces of the last exchange
            b.advance(step)
        self.advance_cfd(step)  # AdvanceCFD
        self.sph.OnExchangeSolidForces()
        for b, (f, t) in zip(self.bodies, self.sph.GetBodyWrenches()):
            b.force, b.torque = np.asarray(f, dtype=np.float64), np.asarray(t, dtype=np.float64)
        self.sph.LoadSolidStates([b.state() for b in self.bodies])
        self.sph.OnExchangeSolidStates()
        self.time += step

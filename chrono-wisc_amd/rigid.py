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


class PrescribedBody:
    """A body whose motion is a given function of time (the wavemaker of ChFsiProblemWavetank: ChLinkMotorLinearPosition along x,
    or ChLinkMotorRotationAngle about the y axis through `pivot`).  `actuation`: f(t), or an object with value(t) / deriv(t) /
    deriv2(t); derivatives default to central differences.  State records like RigidBody (angular quantities body-local)."""

    def __init__(self, pos, actuation, kind="piston", pivot=None):
        self.pos0 = np.asarray(pos, dtype=np.float64).copy()
        self.kind = kind
        self.pivot = None if pivot is None else np.asarray(pivot, dtype=np.float64).copy()
        self.f = actuation
        self.time = 0.0
        self.force = np.zeros(3)
        self.torque = np.zeros(3)
        self._set(0.0)

    def _fd(self, t, order):
        f = self.f
        if hasattr(f, "value"):
            return (f.value(t), f.deriv(t), f.deriv2(t))[order]
        e = 1e-5
        if order == 0:
            return f(t)
        if order == 1:
            return (f(t + e) - f(t - e)) / (2 * e)
        return (f(t + e) - 2 * f(t) + f(t - e)) / (e * e)

    def _set(self, t):
        q, qd, qdd = self._fd(t, 0), self._fd(t, 1), self._fd(t, 2)
        if self.kind == "piston":
            self.pos = self.pos0 + np.array([q, 0.0, 0.0])
            self.rot = np.array(fsi.QUNIT, dtype=np.float64)
            self.lin_vel, self.lin_acc = np.array([qd, 0.0, 0.0]), np.array([qdd, 0.0, 0.0])
            self.ang_vel, self.ang_acc = np.zeros(3), np.zeros(3)
        else:  # flap: rotation by q about the global y axis through the pivot
            self.rot = np.array(fsi.quat_from_angle_axis(q, (0.0, 1.0, 0.0)), dtype=np.float64)
            R = fsi.rotmat_from_quat(self.rot)
            r = R @ (self.pos0 - self.pivot)
            w, al = np.array([0.0, qd, 0.0]), np.array([0.0, qdd, 0.0])
            self.pos = self.pivot + r
            self.lin_vel = np.cross(w, r)
            self.lin_acc = np.cross(al, r) + np.cross(w, np.cross(w, r))
            self.ang_vel, self.ang_acc = w.copy(), al.copy()  # the axis is common to both frames

    def state(self):
        return dict(pos=self.pos, rot=self.rot, lin_vel=self.lin_vel, ang_vel=self.ang_vel, lin_acc=self.lin_acc, ang_acc=self.ang_acc)

    def advance(self, h):
        self.time += h
        self._set(self.time)


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

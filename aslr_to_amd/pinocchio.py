"""Minimal stand-in for the handful of ``pinocchio`` symbols the reference's scripts touch.

The reference builds its problems from ``pinocchio.Model`` objects loaded by example_robot_data
(examples/two_dof_sea.py:18-20) and ``pinocchio.SE3`` targets (examples/two_dof_vsa_boxddp.py:22-23).
Pinocchio is not installable here, and the rigid-body arithmetic of this framework runs inside
the HIP kernels, so this module only carries *descriptions*: a fixed-base serial chain of revolute
joints (``ChainModel``) and a rigid placement (``SE3``).  Pure host logic, numpy only.
"""
import numpy as np

from . import _abi


class SE3(object):
    """pinocchio.SE3(rotation, translation): x_parent = R x_child + p."""

    def __init__(self, rotation=None, translation=None):
        self.rotation = np.eye(3) if rotation is None else np.array(rotation, dtype=float).reshape(3, 3)
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=float).reshape(3)

    @staticmethod
    def Identity():
        return SE3()

    def inverse(self):
        return SE3(self.rotation.T, -self.rotation.T.dot(self.translation))

    def __mul__(self, other):
        return SE3(self.rotation.dot(other.rotation), self.rotation.dot(other.translation) + self.translation)

    def copy(self):
        return SE3(self.rotation.copy(), self.translation.copy())

    def as12(self):
        """row-major R (9) followed by p (3): the layout of aslr_cost_t.ref / frame_ref."""
        return np.concatenate([self.rotation.reshape(9), self.translation])

    def __repr__(self):
        return "SE3(R=%s, p=%s)" % (self.rotation.tolist(), self.translation.tolist())


class _Gravity(object):
    def __init__(self, linear):
        self.linear = np.array(linear, dtype=float)


class Frame(object):
    def __init__(self, name, parent, placement):
        self.name = name
        self.parent = parent  # 0-based joint index
        self.placement = placement


class Joint(object):
    def __init__(self, placement, axis, mass, com, inertia, name=""):
        self.placement = placement
        self.axis = np.array(axis, dtype=float) / np.linalg.norm(axis)
        self.mass = float(mass)
        self.com = np.array(com, dtype=float)
        self.inertia = np.array(inertia, dtype=float).reshape(3, 3)
        self.name = name


class ChainModel(object):
    """Stand-in for pinocchio.Model: fixed base, revolute joints, joint j's parent is j-1."""

    def __init__(self, name, joints, frames, gravity=(0.0, 0.0, -9.81)):
        if not 1 <= len(joints) <= _abi.MAX_NJ:
            raise ValueError("ChainModel supports 1..%d revolute joints" % _abi.MAX_NJ)
        self.name = name
        self.joints = list(joints)
        self.frames = list(frames)
        self.gravity = _Gravity(gravity)
        self.nq = self.nv = self.njoints = len(joints)

    def getFrameId(self, name):
        for i, f in enumerate(self.frames):
            if f.name == name:
                return i
        return len(self.frames)  # pinocchio returns nframes for unknown names

    def existFrame(self, name):
        return self.getFrameId(name) < len(self.frames)

    def to_struct(self):
        c = _abi.Chain()
        c.nj = self.njoints
        for k in range(3):
            c.gravity[k] = float(self.gravity.linear[k])
        for j, jt in enumerate(self.joints):
            R = jt.placement.rotation.reshape(9)
            I = jt.inertia.reshape(9)
            for k in range(9):
                c.joint_R[j][k] = R[k]
                c.inertia[j][k] = I[k]
            for k in range(3):
                c.joint_p[j][k] = jt.placement.translation[k]
                c.axis[j][k] = jt.axis[k]
                c.com[j][k] = jt.com[k]
            c.mass[j] = jt.mass
        return c


class _Utils(object):
    @staticmethod
    def zero(n):
        return np.zeros(n)

    @staticmethod
    def rand(n):
        return np.random.rand(n)


utils = _Utils()


def neutral(model):
    return np.zeros(model.nq)


def randomConfiguration(model):
    # revolute joints without limits in the synthetic tables: uniform in [-pi, pi)
    return np.random.uniform(-np.pi, np.pi, model.nq)

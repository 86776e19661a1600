"""Stand-in for ``example_robot_data.load(name)`` (examples/two_dof_sea.py:18).

THE URDFs ARE UNOBTAINABLE OFFLINE: `asr_twodof` is not an upstream
example-robot-data model at all, and neither `talos_arm` nor `double_pendulum` files exist in
this container.  The tables below are SYNTHETIC, frozen, versioned parameter sets with the same
topology (planar 2R arm / 7R arm / double pendulum).  Model-constant parity with the author's
robots is therefore "unpinned"; every report says so.

TABLE_VERSION is part of the golden fixtures: changing any number below invalidates them.
"""
import numpy as np

from .pinocchio import SE3, ChainModel, Frame, Joint

TABLE_VERSION = 1


def _rod_inertia(mass, length, axis, thin=1e-5):
    """Slender rod of `length` along `axis` (0=x,1=y,2=z): m l^2/12 about the two other axes."""
    d = [mass * length * length / 12.0] * 3
    d[axis] = thin
    return np.diag(d)


def _asr_twodof():
    # planar 2R arm, joint axes +z, base raised to z = 0.18 so the examples' targets (z = 0.18,
    # examples/two_dof_vsa_boxddp.py:23) lie in the arm's plane.  The examples set gravity to
    # (9.81, 0, 0), i.e. +x is "down" in that plane, so q = 0 is taken as the arm hanging along +x.
    # Geometry is anchored on the one number the reference holds: the commented-out target
    # (-2.54999919e-01, 2.03063311e-04, 0.18) at examples/two_dof_sea.py:35 is read as the EE position
    # of the fully stretched, inverted arm q = (pi, 0) (a swing-up target): reach 0.255 m and a 0.2 mm
    # lateral EE offset.  1 mm lateral COM offsets keep q = 0 from being an exact equilibrium (at an
    # exact one the cold-started VSA problem, us = 0 => zero stiffness, has a vanishing gradient).
    l1, l2, m1, m2 = 0.135, 0.12, 0.30, 0.20
    joints = [
        Joint(SE3(np.eye(3), [0.0, 0.0, 0.18]), [0, 0, 1], m1, [l1 / 2, -1e-3, 0], _rod_inertia(m1, l1, 0), "joint1"),
        Joint(SE3(np.eye(3), [l1, 0.0, 0.0]), [0, 0, 1], m2, [l2 / 2, -1e-3, 0], _rod_inertia(m2, l2, 0), "joint2"),
    ]
    frames = [Frame("universe", -1, SE3()), Frame("joint1", 0, SE3()), Frame("joint2", 1, SE3()),
              Frame("EE", 1, SE3(np.eye(3), [l2, -2.03063311e-04, 0.0]))]
    return ChainModel("asr_twodof", joints, frames)


def _double_pendulum():
    # two links of 0.25 m / 0.5 kg along +z at q = 0, axes +y, default gravity (0, 0, -9.81).  Heavier and
    # longer than a desk-top pendulum on purpose: the reference's x0 (examples/double_pendulum.py:52) loads the
    # K = 1 spring with a 3.14 rad deflection, and links much lighter than this make the dt = 1e-2 Euler
    # rollout of that release diverge.
    l, m = 0.25, 0.5
    joints = [
        Joint(SE3(), [0, 1, 0], m, [0, 0, l / 2], _rod_inertia(m, l, 2), "joint1"),
        Joint(SE3(np.eye(3), [0, 0, l]), [0, 1, 0], m, [0, 0, l / 2], _rod_inertia(m, l, 2), "joint2"),
    ]
    frames = [Frame("universe", -1, SE3()), Frame("joint1", 0, SE3()), Frame("joint2", 1, SE3()),
              Frame("tip", 1, SE3(np.eye(3), [0, 0, l]))]
    return ChainModel("double_pendulum", joints, frames)


def _talos_arm():
    # 7R serial arm with alternating axes and arm-like masses (stand-in for talos_arm's left arm:
    # shoulder 3R, elbow 1R, wrist 3R).  Offsets in metres, masses in kg.
    def rotx(a):
        c, s = np.cos(a), np.sin(a)
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])

    spec = [
        # placement translation, placement rotation, axis, mass, com, inertia diag
        ([0.0, 0.1575, 0.232], np.eye(3), [0, 0, 1], 2.71, [-0.01, 0.06, -0.03], [0.012, 0.006, 0.010]),
        ([0.00493, 0.1365, 0.04673], rotx(0.05), [1, 0, 0], 2.43, [0.02, 0.02, -0.05], [0.013, 0.013, 0.004]),
        ([0.0, 0.0, 0.0], np.eye(3), [0, 0, 1], 2.21, [0.007, 0.0, -0.19], [0.007, 0.007, 0.003]),
        ([0.02, 0.0, -0.273], rotx(-0.03), [0, 1, 0], 0.88, [-0.01, 0.0, -0.04], [0.003, 0.003, 0.001]),
        ([-0.02, 0.0, -0.2643], np.eye(3), [0, 0, 1], 1.88, [0.0, 0.006, 0.11], [0.004, 0.004, 0.002]),
        ([0.0, 0.0, 0.0], np.eye(3), [1, 0, 0], 0.41, [0.0, 0.0, 0.0], [0.0001, 0.00015, 0.0001]),
        ([0.0, 0.0, 0.0], np.eye(3), [0, 1, 0], 0.95, [0.005, 0.0, -0.06], [0.0015, 0.0015, 0.0007]),
    ]
    joints = [Joint(SE3(R, p), ax, m, c, np.diag(I), "arm_left_%d_joint" % (i + 1))
              for i, (p, R, ax, m, c, I) in enumerate(spec)]
    frames = [Frame("universe", -1, SE3())]
    frames += [Frame(j.name, i, SE3()) for i, j in enumerate(joints)]
    frames.append(Frame("gripper_left_joint", 6, SE3(np.eye(3), [0.0, 0.0, -0.12])))
    return ChainModel("talos_arm", joints, frames)


_TABLES = {"asr_twodof": _asr_twodof, "double_pendulum": _double_pendulum, "talos_arm": _talos_arm}


class RobotWrapper(object):
    def __init__(self, model):
        self.model = model
        self.q0 = np.zeros(model.nq)


def load(name):
    if name not in _TABLES:
        raise ValueError("unknown synthetic robot table %r (available: %s)" % (name, sorted(_TABLES)))
    return RobotWrapper(_TABLES[name]())

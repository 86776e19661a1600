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


# ---------------------------------------------------------------------------------------------
# URDF subset loader: what `pinocchio.buildModelFromUrdf` gives the reference for a
# fixed-base serial arm (example_robot_data.load(...).model, examples/two_dof_sea.py:18).
# Supported: <link> with <inertial> (origin xyz/rpy, mass, inertia), <joint> of type revolute /
# continuous (one per chain joint) and fixed (its child link is welded onto the parent: masses, centres of
# mass and inertias are composed; the welded link's frame becomes a named Frame).  Anything else
# (prismatic / floating joints, a second moving joint below a link when no tip_link picks the path) raises
# ValueError; with tip_link, moving branches off the path are ignored.
# ---------------------------------------------------------------------------------------------
def _rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz.dot(Ry).dot(Rx)  # URDF: fixed-axis roll, pitch, yaw


def _origin(el):
    if el is None:
        return SE3()
    xyz = [float(v) for v in el.get("xyz", "0 0 0").split()]
    rpy = [float(v) for v in el.get("rpy", "0 0 0").split()]
    return SE3(_rpy(*rpy), xyz)


class _Body(object):
    """mass, centre of mass and inertia about it, all in one frame; `add` welds another body given in a child
    frame placed by M (parallel-axis composition)."""

    def __init__(self, mass=0.0, com=None, inertia=None):
        self.mass = float(mass)
        self.com = np.zeros(3) if com is None else np.array(com, dtype=float)
        self.inertia = np.zeros((3, 3)) if inertia is None else np.array(inertia, dtype=float)

    def add(self, other, M):
        if other.mass == 0.0:
            return
        c2 = M.rotation.dot(other.com) + M.translation
        I2 = M.rotation.dot(other.inertia).dot(M.rotation.T)
        m = self.mass + other.mass
        c = (self.mass * self.com + other.mass * c2) / m

        def shift(I, mass, d):  # inertia about a point displaced by d from the body's own centre of mass
            return I + mass * (d.dot(d) * np.eye(3) - np.outer(d, d))

        self.inertia = shift(self.inertia, self.mass, self.com - c) + shift(I2, other.mass, c2 - c)
        self.mass, self.com = m, c


def buildModelFromUrdf(filename_or_xml, root_link=None, tip_link=None, name=None):
    """Fixed-base serial chain of the URDF (file name or XML text) from `root_link` (default: the link that is
    nobody's child) to `tip_link` (default: follow the only moving path).  Returns a ChainModel whose frames
    are the universe, one per chain joint, and one per welded (fixed-joint) link."""
    import xml.etree.ElementTree as ET
    text = filename_or_xml
    if "<robot" not in text:
        with open(filename_or_xml) as f:
            text = f.read()
    robot = ET.fromstring(text)
    links = {}
    for l in robot.findall("link"):
        ine = l.find("inertial")
        if ine is None:
            links[l.get("name")] = (_Body(), SE3())
            continue
        M = _origin(ine.find("origin"))
        mass = float(ine.find("mass").get("value"))
        i = ine.find("inertia")
        g = lambda k: float(i.get(k, "0"))
        I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        # inertia is given in the inertial frame (origin M): express it about the COM along the link axes
        links[l.get("name")] = (_Body(mass, M.translation, M.rotation.dot(I).dot(M.rotation.T)), SE3())
    joints = []
    for j in robot.findall("joint"):
        ax = j.find("axis")
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
                           child=j.find("child").get("link"), M=_origin(j.find("origin")),
                           axis=[float(v) for v in (ax.get("xyz") if ax is not None else "1 0 0").split()]))
    children = {j["child"] for j in joints}
    if root_link is None:
        roots = [n for n in links if n not in children]
        if len(roots) != 1:
            raise ValueError("URDF: expected exactly one root link, found %s" % roots)
        root_link = roots[0]
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j["parent"], []).append(j)

    def on_path(link):  # does the subtree under `link` contain the tip (or, without a tip, any moving joint)?
        if tip_link is not None and link == tip_link:
            return True
        return any((tip_link is None and c["type"] in ("revolute", "continuous")) or on_path(c["child"])
                   for c in by_parent.get(link, []))

    chain, frames = [], [Frame("universe", -1, SE3())]
    # walk: `cur` is the link the current chain joint carries, `body` its welded composite in the joint frame
    def weld(link, M_joint_link, body, jidx):
        """add `link` (placed by M_joint_link in the current joint frame) and everything fixed to it; returns the
        next moving joint on the path (dict, placement in the current joint frame) or None."""
        body.add(links[link][0], M_joint_link)
        nxt = None
        for c in by_parent.get(link, []):
            Mc = M_joint_link * c["M"]
            if c["type"] == "fixed":
                frames.append(Frame(c["child"], jidx, Mc))
                r = weld(c["child"], Mc, body, jidx)
                if r is not None:
                    if nxt is not None:
                        raise ValueError("URDF: the chain branches below link %r" % link)
                    nxt = r
            elif c["type"] in ("revolute", "continuous"):
                if tip_link is not None and not on_path(c["child"]):
                    continue
                if nxt is not None:
                    raise ValueError("URDF: two moving joints below link %r (give tip_link to pick a path)" % link)
                nxt = dict(c, M=Mc)
            else:
                raise ValueError("URDF: joint %r of type %r is not supported" % (c["name"], c["type"]))
        return nxt

    base = _Body()
    nxt = weld(root_link, SE3(), base, -1)  # the base composite does not move: its inertia is irrelevant
    while nxt is not None:
        jidx = len(chain)
        body = _Body()
        placement, axis, jname, child = nxt["M"], nxt["axis"], nxt["name"], nxt["child"]
        frames.append(Frame(jname, jidx, SE3()))
        nxt = weld(child, SE3(), body, jidx)
        if body.mass <= 0.0:
            raise ValueError("URDF: the links carried by joint %r have no mass" % jname)
        chain.append(Joint(placement, axis, body.mass, body.com, body.inertia, jname))
    if not chain:
        raise ValueError("URDF: no revolute joint between %r and %r" % (root_link, tip_link))
    return ChainModel(name or robot.get("name", "urdf"), chain, frames)


def model_to_urdf(model):
    """URDF text of a ChainModel (one link per joint, welded frames as massless links on fixed joints); the
    inverse of buildModelFromUrdf on its own output up to rounding of the rpy angles."""
    def rpy_of(R):
        p = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
        return np.arctan2(R[2, 1], R[2, 2]), p, np.arctan2(R[1, 0], R[0, 0])

    def origin(M):
        return '<origin xyz="%.17g %.17g %.17g" rpy="%.17g %.17g %.17g"/>' % (tuple(M.translation) + tuple(rpy_of(M.rotation)))

    out = ['<robot name="%s">' % model.name, '  <link name="base_link"/>']
    parent = "base_link"
    for i, jt in enumerate(model.joints):
        I = jt.inertia
        out.append('  <link name="link%d"><inertial><origin xyz="%.17g %.17g %.17g" rpy="0 0 0"/><mass value="%.17g"/>'
                   '<inertia ixx="%.17g" ixy="%.17g" ixz="%.17g" iyy="%.17g" iyz="%.17g" izz="%.17g"/></inertial></link>'
                   % (i, jt.com[0], jt.com[1], jt.com[2], jt.mass, I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]))
        out.append('  <joint name="%s" type="revolute"><parent link="%s"/><child link="link%d"/>%s'
                   '<axis xyz="%.17g %.17g %.17g"/><limit lower="-3.2" upper="3.2" effort="100" velocity="10"/></joint>'
                   % (jt.name or "joint%d" % (i + 1), parent, i, origin(jt.placement), jt.axis[0], jt.axis[1], jt.axis[2]))
        parent = "link%d" % i
    joint_names = {jt.name for jt in model.joints}
    for f in model.frames:
        if f.parent < 0 or f.name in joint_names:
            continue
        out.append('  <link name="%s"/>' % f.name)
        out.append('  <joint name="%s_fixed" type="fixed"><parent link="link%d"/><child link="%s"/>%s</joint>'
                   % (f.name, f.parent, f.name, origin(f.placement)))
    out.append('</robot>')
    return "\n".join(out)

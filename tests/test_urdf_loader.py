"""URDF subset loader (aslr_to_amd.pinocchio.buildModelFromUrdf): host logic only."""
import numpy as np
import pytest

from aslr_to_amd import example_robot_data
from aslr_to_amd import pinocchio as pin


@pytest.mark.parametrize("name", ["asr_twodof", "double_pendulum", "talos_arm"])
def test_round_trip_of_the_synthetic_tables(name):
    m = example_robot_data.load(name).model
    m2 = pin.buildModelFromUrdf(pin.model_to_urdf(m))
    assert m2.njoints == m.njoints
    for a, b in zip(m.joints, m2.joints):
        assert a.name == b.name
        np.testing.assert_allclose(b.placement.rotation, a.placement.rotation, atol=1e-15)
        np.testing.assert_allclose(b.placement.translation, a.placement.translation, atol=0)
        np.testing.assert_allclose(b.axis, a.axis, atol=0)
        np.testing.assert_allclose(b.mass, a.mass, rtol=0)
        np.testing.assert_allclose(b.com, a.com, atol=1e-17)
        np.testing.assert_allclose(b.inertia, a.inertia, atol=1e-18)
    for f in m.frames:   # every frame survives with its parent joint and placement
        g = m2.frames[m2.getFrameId(f.name)]
        assert g.parent == f.parent
        np.testing.assert_allclose(g.placement.translation, f.placement.translation, atol=1e-16)
        np.testing.assert_allclose(g.placement.rotation, f.placement.rotation, atol=1e-15)
    # the chain table handed to the C ABI is the same
    c1, c2 = m.to_struct(), m2.to_struct()
    for j in range(m.njoints):
        np.testing.assert_allclose(list(c2.joint_R[j]), list(c1.joint_R[j]), atol=1e-15)
        np.testing.assert_allclose(list(c2.inertia[j]), list(c1.inertia[j]), atol=1e-18)


URDF = """
<robot name="two_link">
  <link name="world"/>
  <link name="upper"><inertial><origin xyz="0.1 0 0" rpy="0 0 0"/><mass value="2.0"/>
    <inertia ixx="0.01" ixy="0" ixz="0" iyy="0.02" iyz="0" izz="0.03"/></inertial></link>
  <link name="sensor"><inertial><origin xyz="0 0 0.05" rpy="0 0 1.5707963267948966"/><mass value="0.5"/>
    <inertia ixx="0.001" ixy="0" ixz="0" iyy="0.002" iyz="0" izz="0.003"/></inertial></link>
  <link name="lower"><inertial><origin xyz="0 0 -0.1"/><mass value="1.0"/>
    <inertia ixx="0.004" iyy="0.004" izz="0.001"/></inertial></link>
  <link name="tool"/>
  <joint name="shoulder" type="revolute"><parent link="world"/><child link="upper"/>
    <origin xyz="0 0 1" rpy="0 0 0"/><axis xyz="0 2 0"/></joint>
  <joint name="mount" type="fixed"><parent link="upper"/><child link="sensor"/><origin xyz="0.2 0 0" rpy="0 0 0"/></joint>
  <joint name="elbow" type="continuous"><parent link="upper"/><child link="lower"/>
    <origin xyz="0.3 0 0" rpy="1.5707963267948966 0 0"/><axis xyz="0 0 1"/></joint>
  <joint name="flange" type="fixed"><parent link="lower"/><child link="tool"/><origin xyz="0 0 -0.2"/></joint>
</robot>
"""


def test_fixed_joints_are_welded_with_the_parallel_axis_rule():
    m = pin.buildModelFromUrdf(URDF)
    assert [j.name for j in m.joints] == ["shoulder", "elbow"]
    j0, j1 = m.joints
    np.testing.assert_allclose(j0.axis, [0, 1, 0])                      # normalised
    np.testing.assert_allclose(j0.placement.translation, [0, 0, 1])
    # upper (2 kg at x = 0.1) + sensor (0.5 kg at (0.2, 0, 0.05), its inertia rotated by 90 deg about z)
    assert j0.mass == pytest.approx(2.5)
    c = (2.0 * np.array([0.1, 0, 0]) + 0.5 * np.array([0.2, 0, 0.05])) / 2.5
    np.testing.assert_allclose(j0.com, c, atol=1e-15)
    Ia, Ib = np.diag([0.01, 0.02, 0.03]), np.diag([0.002, 0.001, 0.003])  # Rz(90) swaps xx and yy
    def shift(I, mass, d):
        return I + mass * (d.dot(d) * np.eye(3) - np.outer(d, d))
    I = shift(Ia, 2.0, np.array([0.1, 0, 0]) - c) + shift(Ib, 0.5, np.array([0.2, 0, 0.05]) - c)
    np.testing.assert_allclose(j0.inertia, I, atol=1e-15)
    # elbow placement carries the rpy rotation (roll of 90 deg)
    np.testing.assert_allclose(j1.placement.rotation, [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-15)
    np.testing.assert_allclose(j1.placement.translation, [0.3, 0, 0])
    assert j1.mass == pytest.approx(1.0)
    # welded links became frames on their joint
    f = m.frames[m.getFrameId("tool")]
    assert f.parent == 1
    np.testing.assert_allclose(f.placement.translation, [0, 0, -0.2])
    assert m.frames[m.getFrameId("sensor")].parent == 0


def test_unsupported_topologies_are_refused():
    with pytest.raises(ValueError):
        pin.buildModelFromUrdf(URDF.replace('type="continuous"', 'type="prismatic"'))
    branched = URDF.replace('</robot>', '<link name="x"><inertial><mass value="1"/><inertia ixx="1" iyy="1" izz="1"/></inertial></link>'
                            '<joint name="j3" type="revolute"><parent link="upper"/><child link="x"/><axis xyz="1 0 0"/></joint></robot>')
    with pytest.raises(ValueError):
        pin.buildModelFromUrdf(branched)
    m = pin.buildModelFromUrdf(branched, tip_link="tool")   # a tip picks the path
    assert [j.name for j in m.joints] == ["shoulder", "elbow"]

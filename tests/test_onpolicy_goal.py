"""Hindsight goal relabelling of the replay buffer (reference core/replay_memory.py:233-249, core/utils.py:299-306,446-452,
672-676) -- host logic, CPU.  The reference takes mat2quat from transforms3d (not vendored, absent here: parity
unpinned), so the quaternion conversion is checked through its defining properties and the relabelled goals through the
relative poses they must encode."""
import numpy as np

from ga_ddpg_amd.core.replay_memory import BaseMemory
from ga_ddpg_amd.core.utils import mat2quat, pack_pose_rot_first, se3_inverse
from ga_ddpg_amd.experiments.config import load_cfg


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]), (q if w >= 0 else -q)


def _pose(rng):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = _rot(rng)[0]
    T[:3, 3] = rng.normal(size=3)
    return T


def test_mat2quat_inverts_the_rotation_formula():
    rng = np.random.default_rng(0)
    for _ in range(200):
        R, q = _rot(rng)
        got = mat2quat(R)
        assert got[0] >= 0 and abs(np.linalg.norm(got) - 1) < 1e-12
        assert np.abs(got - q).max() < 1e-9
    assert np.allclose(mat2quat(np.eye(3)), [1, 0, 0, 0])
    # rotations by pi (w = 0): either sign of the axis is the same rotation
    R = np.diag([1.0, -1.0, -1.0])
    assert np.allclose(np.abs(mat2quat(R)), [0, 1, 0, 0])


def test_se3_inverse_and_packing():
    rng = np.random.default_rng(1)
    T = _pose(rng)
    assert np.allclose(se3_inverse(T) @ T, np.eye(4), atol=1e-6)
    p = pack_pose_rot_first(T)
    assert p.shape == (7,) and np.allclose(p[4:], T[:3, 3]) and np.allclose(p[:4], mat2quat(T[:3, :3]))


def test_on_policy_rows_get_their_episode_end_as_goal():
    cfg = load_cfg("ddpg_td3_aux.yaml")
    mem = BaseMemory(64, cfg, point_dtype=np.float32)
    mem.name, mem.self_supervision = "online", True              # (RL_TRAIN.self_supervision, reference experiments/config.py:113)
    rng = np.random.default_rng(2)
    n = 24
    for i in range(n):
        mem.state_pose[i] = _pose(rng)
    mem.goal[:n] = rng.normal(size=(n, 7)).astype(np.float32)
    mem.expert_flags[:n] = (np.arange(n) % 3 == 0)               # every third row is an expert row: keeps its stored goal
    mem.episode_map[:n] = np.repeat([7, 15, 23], 8)              # three episodes of eight steps
    mem.timestep[:n] = np.tile(np.arange(8), 3)
    mem.cur_idx = mem.upper_idx = n
    idx = np.array([0, 1, 6, 7, 8, 14, 15, 22, 23])
    data = {"goal_batch": np.float32(mem.goal[idx]), "time_batch": np.float32(mem.timestep[idx])}
    mem.post_process_batch(data, idx)
    end = mem.episode_map[idx]
    nxt = np.minimum(end, idx + 1)
    for j, i in enumerate(idx):
        if mem.expert_flags[i]:
            assert np.array_equal(data["goal_batch"][j], mem.goal[i])
            assert np.array_equal(data["next_goal_batch"][j], mem.goal[nxt[j]])
            continue
        rel = np.linalg.inv(mem.state_pose[i].astype(np.float64)) @ mem.state_pose[end[j]].astype(np.float64)
        assert np.allclose(data["goal_batch"][j][4:], rel[:3, 3], atol=1e-5)
        assert np.allclose(data["goal_batch"][j][:4], mat2quat(rel[:3, :3]), atol=1e-5)
        reln = np.linalg.inv(mem.state_pose[nxt[j]].astype(np.float64)) @ mem.state_pose[end[j]].astype(np.float64)
        assert np.allclose(data["next_goal_batch"][j][4:], reln[:3, 3], atol=1e-5)
    # the last step of an episode looks at itself: identity goal
    j = list(idx).index(7)
    assert np.allclose(data["goal_batch"][j], [1, 0, 0, 0, 0, 0, 0], atol=1e-5)

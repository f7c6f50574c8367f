"""Deterministic synthetic replay-buffer content for tests and bench.py.

No dataset is available offline (reference data/ is absent), so the measured workload is a
seeded synthetic buffer with the reference's storage layout (core/replay_memory.py:359-384,
producer env/panda_scene.py:1178-1206): episodes of 8..20 steps laid out contiguously, clouds of
N points on the surface of a randomly posed box in the end-effector frame, preceded by the 6 fixed
gripper points (core/utils.py:38-40) flagged 1 in the 4th row.
"""
import numpy as np

# gripper key points in the end-effector frame (data; reference core/utils.py:38-40)
HAND_FINGER_POINT = np.array([[0., 0., 0., -0., 0., -0.],
                              [0., 0., 0.053, -0.053, 0.053, -0.053],
                              [0., 0., 0.075, 0.075, 0.105, 0.105]])


class PandaTaskSpace6D(object):
    """6-D end-effector delta action bounds (reference core/utils.py:505-510)."""

    def __init__(self):
        self.high = np.array([0.06, 0.06, 0.06, np.pi / 6, np.pi / 6, np.pi / 6])
        self.low = -self.high
        self.shape = [6]
        self.bounds = np.vstack([self.low, self.high])


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1)
    return R.reshape(n, 3, 3)


def box_surface_cloud(rng, n_points, edges):
    """n_points uniform on the surface of an axis-aligned box with the given edge lengths."""
    ex, ey, ez = edges
    areas = np.array([ey * ez, ey * ez, ex * ez, ex * ez, ex * ey, ex * ey])
    face = rng.choice(6, size=n_points, p=areas / areas.sum())
    uv = rng.uniform(-0.5, 0.5, size=(n_points, 2))
    pts = np.zeros((n_points, 3))
    axis = face // 2
    sign = np.where(face % 2 == 0, 0.5, -0.5)
    for a in range(3):
        sel = axis == a
        others = [i for i in range(3) if i != a]
        pts[sel, a] = sign[sel]
        pts[sel, others[0]] = uv[sel, 0]
        pts[sel, others[1]] = uv[sel, 1]
    return pts * np.array([ex, ey, ez])


def fill_synthetic_buffer(memory, n_transitions, seed=20260928, gamma=None):
    """Fill ``memory`` (a BaseMemory) in place with ``n_transitions`` synthetic transitions."""
    rng = np.random.default_rng(seed)
    n_pts = memory.point_state.shape[2] - 6
    space = PandaTaskSpace6D()
    gamma = memory.gamma if gamma is None else gamma
    n_transitions = min(int(n_transitions), memory.buffer_size)
    memory.point_state[:, :3, :6] = HAND_FINGER_POINT
    memory.point_state[:, 3, :6] = 1.0
    pos = 0
    while pos < n_transitions:
        L = int(rng.integers(8, 21))
        L = min(L, n_transitions - pos)
        sl = slice(pos, pos + L)
        edges = rng.uniform(0.04, 0.16, size=3)
        R = _random_rotations(rng, 1)[0]
        centre = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(0.12, 0.35)])
        base = box_surface_cloud(rng, n_pts, edges) @ R.T
        for t in range(L):
            c = centre - np.array([0.0, 0.0, 0.01 * t])
            memory.point_state[pos + t, :3, 6:] = (base + c).T
        memory.point_state[sl, 3, 6:] = 0.0
        memory.timestep[sl] = np.arange(L)
        memory.terminal[sl] = 0.0
        memory.terminal[pos + L - 1] = 1.0
        memory.episode_map[sl] = pos + L - 1
        memory.action[sl] = rng.uniform(space.low, space.high, size=(L, 6))
        memory.expert_action[sl] = rng.uniform(space.low, space.high, size=(L, 6))
        q = rng.normal(size=(L, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        memory.goal[sl, :4] = q
        memory.goal[sl, 4:] = rng.uniform(-0.15, 0.15, size=(L, 3))
        memory.reward[sl] = 0.0
        if rng.random() < 0.6:
            memory.reward[pos + L - 1] = 1.0
        go = 0.0
        for i in range(L):  # same recursion as add_episode (reference :223-228)
            j = pos + L - 1 - i
            memory.returns[j] = memory.reward[j] + gamma ** i * go
            go = memory.returns[j]
        memory.expert_flags[sl] = 1.0 if rng.random() < 0.5 else 0.0
        memory.perturb_flags[sl] = (rng.random(L) < 0.1).astype(np.float32)
        memory.state_pose[sl] = np.eye(4, dtype=np.float32)
        pos += L
    memory.cur_idx = n_transitions
    memory.total_env_step = n_transitions
    memory.is_full = n_transitions >= memory.buffer_size
    return memory


def sample_valid_batch(memory, batch_size, rng, clouds_out=None, pool=None):
    """A sampled batch that has >=1 expert row and >=1 positive-return row (the reference's
    masked means are NaN otherwise: core/loss.py:23,31).  The validity test reads the flag arrays of the drawn indices, so
    a rejected draw costs no cloud gather; the accepted draw is memory.sample on those indices (same random stream as a
    memory.sample(batch_size, rng) loop with rejection).  clouds_out / pool: BaseMemory.sample's."""
    for _ in range(64):
        idx = memory.draw_indices(batch_size, rng)
        if (memory.expert_flags[idx] >= 1).any() and (memory.returns[idx] > 0).any() and (memory.perturb_flags[idx] < 1).any():
            return memory.sample(batch_size, batch_idx=idx, clouds_out=clouds_out, pool=pool)
    raise RuntimeError("could not draw a batch with expert and positive-return rows")

"""Replay buffer, sampling side.

Mirrors reference core/replay_memory.py: storage layout of ``init_buffer`` (:359-384), and
``sample`` (:166-176) -> ``__getitem__`` (:109-127) -> ``post_process_batch`` (:251-272), i.e. the
22-key batch dict that ``Agent.prepare_data`` consumes.  Differences by design:
  * ``np.int`` (removed from numpy) is not used; indices are int64.
  * ``point_dtype`` selects the cloud storage type (reference: float64 from a bare np.zeros);
    float32 halves host traffic and is what the GPU path uploads anyway.
  * an optional ``rng`` (numpy Generator/RandomState) may replace the global ``np.random`` stream.
Writer side (push / add_episode / save / load) is SURVEY 8f "next" (N2/N4), see bottom of file.
"""
import os
from collections import deque

import numpy as np


def process_image_output(sample):
    """reference core/utils.py:170-178 (image modality is off: arrays with ndim<=2 pass through)."""
    sample = sample.astype(np.float32).copy()
    if sample.ndim <= 2:
        return sample
    sample[:, :3] /= 255.0
    if sample.shape[0] >= 4:
        sample[:, 3] /= 5000
    return sample


class BaseMemory(object):
    """Flat numpy ring buffer of transitions; ``sample(B)`` returns dict[str, np.ndarray]."""

    ATTR_NAMES = ("action", "pose", "point_state", "target_idx", "reward", "terminal", "timestep",
                  "returns", "state_pose", "image_state", "collide", "grasp", "perturb_flags",
                  "goal", "expert_flags", "expert_action")

    def __init__(self, buffer_size, args, name="expert", point_dtype=np.float64):
        self.cur_idx = 0
        self.total_env_step = 0
        self.is_full = False
        self.name = name
        for key, val in args.RL_TRAIN.items():
            setattr(self, key, val)
        self.buffer_size = int(buffer_size)
        self.episode_max_len = args.RL_MAX_STEP
        self.save_data_name = args.RL_SAVE_DATA_NAME
        self.attr_names = list(self.ATTR_NAMES)
        self.point_dtype = point_dtype
        self._REW = deque([0] * 50, maxlen=200)
        self._ONLINE_REW = deque([0] * 50, maxlen=50)
        self._TEST_REW = deque([0] * 50, maxlen=50)
        self._TOTAL_REW, self._TOTAL_CNT = 0, 1
        self.dir = args.RL_DATA_ROOT_DIR
        self.object_performance = {}
        self.init_buffer()

    # ------------------------------------------------------------------ storage (A0)
    def init_buffer(self):
        n = self.buffer_size
        state_size = (5, 112, 112) if self.use_image else (1,)
        f32 = np.float32
        self.image_state = np.zeros((n,) + state_size, dtype=np.uint16)
        self.action = np.zeros((n, 6), dtype=f32)
        self.expert_action = np.zeros((n, 6), dtype=f32)
        self.terminal = np.zeros((n,), dtype=f32)
        self.timestep = np.zeros((n,), dtype=f32)
        self.reward = np.zeros((n,), dtype=f32)
        self.returns = np.zeros((n,), dtype=f32)
        self.pose = np.zeros((n, 64), dtype=f32)
        # rows x,y,z,hand-flag; cols 0-5 gripper points, 6.. the cloud (end-effector frame)
        self.point_state = np.zeros((n, 4, self.uniform_num_pts + 6), dtype=self.point_dtype)
        self.collide = np.zeros((n,), dtype=f32)
        self.grasp = np.zeros((n,), dtype=f32)
        self.state_pose = np.zeros((n, 4, 4), dtype=f32)
        self.target_idx = np.zeros((n,), dtype=f32)
        self.goal = np.zeros((n, 7), dtype=f32)
        self.episode_map = np.zeros((n,), dtype=np.uint32)
        self.expert_flags = np.zeros((n,), dtype=f32)
        self.perturb_flags = np.zeros((n,), dtype=f32)

    def __len__(self):
        return self.upper_idx()

    def upper_idx(self):
        return max(self.cur_idx, 1) if not self.is_full else len(self.point_state)

    def get_cur_idx(self):
        return self.cur_idx

    def get_total_env_step(self):
        return self.total_env_step

    def reset(self):
        self.cur_idx = 0
        self.is_full = False

    # ------------------------------------------------------------------ sampling (A1)
    def draw_indices(self, batch_size, rng=None):
        """uniform indices in [episode_max_len, upper_idx), then shuffled (reference :169-172)."""
        r = np.random if rng is None else rng
        if hasattr(r, "integers"):
            batch_idx = r.integers(self.episode_max_len, self.upper_idx(), batch_size)
        else:
            batch_idx = r.randint(self.episode_max_len, self.upper_idx(), batch_size)
        r.shuffle(batch_idx)
        return batch_idx

    def sample(self, batch_size, rng=None, batch_idx=None):
        if batch_idx is None:
            batch_idx = self.draw_indices(batch_size, rng)
        data = self[batch_idx]
        self.post_process_batch(data, batch_idx)
        return data

    def __getitem__(self, idx):
        f32 = np.float32
        return {
            "image_state_batch": process_image_output(self.image_state[idx]),
            "expert_action_batch": f32(self.expert_action[idx]),
            "action_batch": f32(self.action[idx]),
            "reward_batch": f32(self.reward[idx]),
            "return_batch": f32(self.returns[idx]),
            "next_image_state_batch": None,
            "mask_batch": f32(self.terminal[idx]),
            "time_batch": f32(self.timestep[idx]),
            "point_state_batch": None,
            "next_point_state_batch": None,
            "state_pose_batch": f32(self.state_pose[idx]),
            "collide_batch": f32(self.collide[idx]),
            "grasp_batch": f32(self.grasp[idx]),
            "goal_batch": f32(self.goal[idx]),
        }

    def next_indices(self, batch_idx):
        """index of the successor transition, clamped to the episode's last step (:255)."""
        return np.minimum(self.episode_map[batch_idx], batch_idx + 1).astype(np.int64)

    def post_process_batch(self, data, batch_idx):
        f32 = np.float32
        nxt = self.next_indices(batch_idx)
        data["grasp_sample_batch"] = np.zeros([0, 4, 4])
        data["next_image_state_batch"] = process_image_output(self.image_state[nxt])
        data["next_goal_batch"] = f32(self.goal[nxt])
        data["next_expert_action_batch"] = f32(self.expert_action[nxt])
        data["next_action_batch"] = f32(self.action[nxt])
        data["next_point_state_batch"] = self.point_state[nxt]
        data["next_return_batch"] = self.returns[nxt]
        data["point_state_batch"] = self.point_state[batch_idx]
        # remaining steps to the end of the episode
        data["time_batch"] = f32(self.timestep[self.episode_map[batch_idx]]) + 1 - data["time_batch"]
        data["expert_flag_batch"] = f32(self.expert_flags[batch_idx])
        data["perturb_flag_batch"] = f32(self.perturb_flags[batch_idx])
        data["batch_idx"] = np.uint8(batch_idx)  # truncating cast, as in the reference (:269)
        if self.self_supervision and self.name != "expert":
            raise NotImplementedError("set_onpolicy_goal (reference :233-249) is outside the path")

    # ------------------------------------------------------------------ returns
    def recompute_return_with_gamma(self):
        """discounted return-to-go per episode (reference :152-164)."""
        ends = np.sort(np.unique(self.episode_map))
        out = self.returns.copy()
        for a, b in zip(ends[:-1], ends[1:]):
            a, b = int(a), int(b)
            go = 0.0
            for i in range(b - a):
                j = b - i
                out[j] = self.reward[j] + self.gamma ** i * go
                go = out[j]
        self.returns = out


ReplayMemory = BaseMemory

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


_POOL = []


def _gather_pool():
    """four threads shared by every buffer of the process (created on first use)"""
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        _POOL.append(ThreadPoolExecutor(max_workers=4, thread_name_prefix="gad-sample"))
    return _POOL[0]


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

    def sample(self, batch_size, rng=None, batch_idx=None, clouds_out=None, pool=None):
        """clouds_out = (point, next_point) float32 arrays of shape (B, 4, pts + 6): the two cloud gathers -- 97 % of a
        minibatch's bytes -- go straight into the caller's (pinned) buffers, split over `pool`'s threads, and
        data["point_state_batch"] / ["next_point_state_batch"] are those arrays; everything else as without it"""
        if batch_idx is None:
            batch_idx = self.draw_indices(batch_size, rng)
        data = self[batch_idx]
        self.post_process_batch(data, batch_idx, clouds_out=clouds_out, pool=pool)
        return data

    def gather_clouds(self, batch_idx, nxt, out_point, out_next, pool=None, chunk=32):
        """out_point[i] = point_state[batch_idx[i]], out_next[i] = point_state[nxt[i]] as float32: np.take per chunk of
        rows (it releases the GIL: the chunks run on `pool`'s threads side by side and beside the caller's Python), with a
        per-chunk float64 scratch when the buffer stores float64 clouds (the reference's dtype)"""
        src = self.point_state
        same = src.dtype == np.float32
        jobs = []
        for idx, out in ((np.asarray(batch_idx, dtype=np.int64), out_point), (np.asarray(nxt, dtype=np.int64), out_next)):
            for a in range(0, len(idx), chunk):
                jobs.append((idx[a:a + chunk], out[a:a + chunk]))

        def run(job):
            ix, dst = job
            if same:
                np.take(src, ix, axis=0, out=dst, mode="clip")
            else:
                np.copyto(dst, np.take(src, ix, axis=0, mode="clip"), casting="same_kind")
        if pool is None:
            for j in jobs:
                run(j)
        else:
            list(pool.map(run, jobs))

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

    def post_process_batch(self, data, batch_idx, clouds_out=None, pool=None):
        f32 = np.float32
        nxt = self.next_indices(batch_idx)
        data["grasp_sample_batch"] = np.zeros([0, 4, 4])
        data["next_image_state_batch"] = process_image_output(self.image_state[nxt])
        data["next_goal_batch"] = f32(self.goal[nxt])
        data["next_expert_action_batch"] = f32(self.expert_action[nxt])
        data["next_action_batch"] = f32(self.action[nxt])
        if clouds_out is None and len(batch_idx) >= 64 and self.point_state.dtype == np.float32:
            # the two cloud gathers are 97 % of a minibatch's bytes: chunks of rows on a small shared thread pool (np.take
            # releases the GIL) instead of one fancy index on the calling thread -- same values, same dtype
            shape = (len(batch_idx),) + self.point_state.shape[1:]
            clouds_out = (np.empty(shape, dtype=np.float32), np.empty(shape, dtype=np.float32))
            pool = pool if pool is not None else _gather_pool()
        if clouds_out is None:
            data["next_point_state_batch"] = self.point_state[nxt]
            data["point_state_batch"] = self.point_state[batch_idx]
        else:
            self.gather_clouds(batch_idx, nxt, clouds_out[0], clouds_out[1], pool=pool)
            data["point_state_batch"], data["next_point_state_batch"] = clouds_out
        data["next_return_batch"] = self.returns[nxt]
        # remaining steps to the end of the episode
        data["time_batch"] = f32(self.timestep[self.episode_map[batch_idx]]) + 1 - data["time_batch"]
        data["expert_flag_batch"] = f32(self.expert_flags[batch_idx])
        data["perturb_flag_batch"] = f32(self.perturb_flags[batch_idx])
        data["batch_idx"] = np.uint8(batch_idx)  # truncating cast, as in the reference (:269)
        if self.self_supervision and self.name != "expert":
            self.set_onpolicy_goal(data, batch_idx)

    def onpolicy_goals(self, batch_idx):
        """(mask, goal, next_goal) of the hindsight relabelling: rows whose expert flag is 0 get the pose their own episode
        ended in, seen from the row's pose (goal) / from its successor's (next_goal), packed rotation-first"""
        from .utils import pack_pose_rot_first, se3_inverse
        batch_idx = np.asarray(batch_idx)
        mask = self.expert_flags[batch_idx] == 0.0
        episode_end = self.episode_map[batch_idx]
        increment_idx = np.minimum(episode_end, batch_idx + 1).astype(np.int64)
        n = len(batch_idx)
        goal = np.array([pack_pose_rot_first(se3_inverse(self.state_pose[batch_idx[i]]).dot(self.state_pose[episode_end[i]]))
                         for i in range(n)])
        next_goal = np.array([pack_pose_rot_first(se3_inverse(self.state_pose[increment_idx[i]]).dot(self.state_pose[episode_end[i]]))
                              for i in range(n)])
        return mask, goal, next_goal

    def set_onpolicy_goal(self, data, batch_idx, vis=False):
        """hindsight relabelling of the on-policy transitions (reference core/replay_memory.py:233-249): the goal of a
        non-expert row becomes the pose its own episode ended in, seen from the row's pose (and from its successor's for
        next_goal_batch), packed rotation-first."""
        mask, goal, next_goal = self.onpolicy_goals(batch_idx)
        data["goal_batch"][mask] = goal[mask]
        data["next_goal_batch"][mask] = next_goal[mask]

    # ------------------------------------------------------------------ writer side
    def update_reward(self, reward, test, explore, target_name):
        """success bookkeeping of add_episode (reference :71-80)"""
        self._TOTAL_REW += reward
        self._TOTAL_CNT += 1
        self._REW.append(reward)
        if explore:
            (self._TEST_REW if test else self._ONLINE_REW).append(reward)
        if target_name != "noexists" and target_name not in self.object_performance:
            self.object_performance[target_name] = [0, 0, 0]
        self.object_performance[target_name][0] += 1
        self.object_performance[target_name][1] += reward

    def push(self, step_dict):
        """store one transition at the write cursor (reference :178-207).  Frames whose cloud has fewer than 100
        columns or is all zero are dropped; the cursor wraps to buffer_start_idx."""
        cloud = step_dict["point_state"]
        if cloud.shape[1] < 100 or cloud.sum() == 0:
            return
        slot = self.cur_idx % len(self.point_state)
        for name in self.attr_names:
            if name == "image_state":
                if self.use_image:
                    raise NotImplementedError("image observations are outside the path (SURVEY 2)")
            elif name in step_dict:
                getattr(self, name)[slot] = step_dict[name]
        if self.cur_idx >= len(self.episode_map) - 1:
            self.is_full = True
        self.cur_idx += 1
        self.total_env_step += 1
        if self.cur_idx >= len(self.point_state) or self.cur_idx < self.buffer_start_idx:
            self.cur_idx = self.buffer_start_idx

    def add_episode(self, episode, explore=False, test=False):
        """append a rollout, back-fill its discounted returns-to-go and point every step's episode_map entry at the
        episode's last index (reference :209-231).  Unsuccessful rollouts are skipped outside RL mode."""
        n = len(episode)
        if (not self.RL) and episode[-1]["reward"] < 0.5 and not explore:
            return
        if n > 0:
            self.update_reward(episode[-1]["reward"] > 0.5, test, explore, episode[-1]["target_name"])
        for transition in episode:
            self.push(transition)
        if n > 0 and self.cur_idx - n >= 0:
            go = 0
            for i in range(n):
                j = self.cur_idx - 1 - i
                self.returns[j] = self.reward[j] + self.gamma ** i * go          # gamma**i, as the reference
                go = self.returns[j]
            self.episode_map[self.cur_idx - n:self.cur_idx] = self.cur_idx - 1

    def get_expert_upper_idx(self):
        hi = self.upper_idx()
        if self.expert_flags is not None and np.sum(self.expert_flags[:hi]) > 0:
            return np.where(self.expert_flags[:hi] >= 1)[0][-1]
        return 0

    # ------------------------------------------------------------------ on-disk format
    SAVE_EXTRA = ("episode_map", "is_full", "cur_idx", "total_env_step", "target_idx")

    def save(self, save_dir="."):
        """one .npz named RL_SAVE_DATA_NAME holding every attr_names array plus the cursor fields
        (reference :338-356): loadable by the reference's BaseMemory.load and vice versa."""
        os.makedirs(save_dir, exist_ok=True)
        np.savez(os.path.join(save_dir, self.save_data_name),
                 **{name: getattr(self, name) for name in list(self.attr_names) + list(self.SAVE_EXTRA)})

    def load(self, data_dir, buffer_size=100000, **kwargs):
        """inverse of save (reference :274-336): copies the first max(episode_map) transitions of every array,
        restores the cursor, recomputes the returns.  A missing directory / file leaves the buffer untouched."""
        path = os.path.join(data_dir, self.save_data_name)
        if not os.path.exists(data_dir) or not os.path.exists(path):
            return
        data = np.load(path, allow_pickle=True, mmap_mode="r")
        n = int(np.amax(data["episode_map"]))
        for name in list(self.attr_names) + ["episode_map", "target_idx"]:
            if (name == "image_state" and not self.use_image) or name not in data:
                continue
            arr = data[name]
            if not isinstance(arr, np.ndarray):
                setattr(self, name, arr)
            else:
                getattr(self, name)[:n] = arr[:n]
        self.cur_idx = n
        self.total_env_step = int(data["total_env_step"])
        self.is_full = bool(data["is_full"]) and self.cur_idx >= self.buffer_size - 1
        self.cur_idx = self.upper_idx()
        self.recompute_return_with_gamma()

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

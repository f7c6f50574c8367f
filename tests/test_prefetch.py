"""Sample prefetch (ga_ddpg_amd/core/prefetch.py) on the CPU: the background sampler hands out exactly the minibatches a
synchronous memory.sample loop on the same random stream produces, in the same order, as float32 staging tensors."""
import numpy as np


def _memory(n=300, pts=64, seed=3):
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.experiments.config import load_cfg
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer
    c = load_cfg("ddpg_td3_aux.yaml")
    c.RL_TRAIN.uniform_num_pts = pts
    mem = BaseMemory(n, c, point_dtype=np.float32)
    fill_synthetic_buffer(mem, n, seed=seed)
    return mem


def test_prefetch_matches_synchronous_sampling():
    from ga_ddpg_amd.core.prefetch import KEYS, PrefetchSampler
    mem = _memory()
    want = [mem.sample(16, rng=np.random.default_rng(7)) for _ in range(1)]
    rng_a = np.random.default_rng(11)
    sync = [mem.sample(16, rng=rng_a) for _ in range(9)]
    with PrefetchSampler(mem, 16, depth=2, rng=np.random.default_rng(11), pin=False) as s:
        for i in range(9):
            b = s.next()
            for k in KEYS:
                got = b[k].numpy()
                assert got.dtype == np.float32
                np.testing.assert_array_equal(got, np.asarray(sync[i][k], dtype=np.float32).reshape(got.shape), err_msg="%d %s" % (i, k))
            np.testing.assert_array_equal(b["batch_idx"], sync[i]["batch_idx"])
        assert len(s._sets) <= 3                      # depth + 1 staging sets circulate, however many batches were drawn
    assert want[0]["point_state_batch"].shape[0] == 16


def test_prefetch_surfaces_producer_errors():
    import pytest
    from ga_ddpg_amd.core.prefetch import PrefetchSampler

    def boom(batch_size):
        raise ValueError("empty buffer")
    s = PrefetchSampler(_memory(), 8, sample=boom, pin=False)
    with pytest.raises(RuntimeError, match="prefetch thread failed"):
        s.next()
    s.close()

"""SURVEY 8e on real device code: the fused update step under DataParallelContext with world_size 2, both ranks on
cuda:0 (one GPU is all a test box has; the `gloo` backend carries the CUDA tensors -- RCCL refuses two ranks on one
device).  What bench.py --gpus N exercises, minus the transport:

* identical minibatch on both ranks  ==  the single-process step on that minibatch (same per-replica BatchNorm
  statistics, global mask counts doubled, summed gradients halved by them): losses, actions, Q-values to float32
  rounding, post-Adam parameters to +-lr noise;
* different minibatches: the replicas stay in lock step (bit-equal parameters on both ranks after the step), the
  returned losses are the global ones (equal on both ranks), gradients are finite."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
B = 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _agent(seed=17):
    from ga_ddpg_amd.api import make_agent
    from oracle.detfill import fill_module_
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    for name in ("policy", "policy_target", "critic", "critic_target", "state_feature_extractor"):
        fill_module_(getattr(agent, name), name, seed)
    return agent, cfg


def _batches(cfg, n):
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    mem = BaseMemory(1200, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 1200, seed=9)
    rng = np.random.default_rng(3)
    return [sample_valid_batch(mem, B, rng) for _ in range(n)], rng.random((B, 6)).astype(np.float32)


def _state(agent):
    return {n + "/" + k: p.detach().cpu().clone() for n in ("policy", "critic", "state_feature_extractor")
            for k, p in getattr(agent, n).named_parameters()}


def _worker(rank, world, port, same_batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ga_ddpg_amd.parallel import DataParallelContext
    agent, cfg = _agent()
    batches, u = _batches(cfg, 2)
    batch = batches[0] if same_batch else batches[rank]
    rt = agent.runtime(B, batch["point_state_batch"].shape[2])
    dp = DataParallelContext()
    agent._dp = dp
    dp.attach(rt)
    res = {}
    for s in range(2):                                    # a policy step and a non-policy step
        r = agent.update_parameters(batch, agent.update_step, s, noise_u=u)
        if s == 0:
            res = {"ret": r, "pi": agent.pi.cpu().clone(), "q1": agent.qf1.cpu().clone(), "y": agent.next_q_value.cpu().clone()}
    torch.cuda.synchronize()
    res["ret2"] = r
    res["state"] = _state(agent)
    res["finite"] = all(bool(torch.isfinite(p.grad).all()) for n in ("policy", "critic")
                        for p in getattr(agent, n).parameters() if p.grad is not None)
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run(same_batch, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), same_batch, str(tmp_path)), nprocs=2, join=True)
    return [torch.load(os.path.join(tmp_path, "rank%d.pt" % r), weights_only=False) for r in range(2)]


def test_two_ranks_same_batch_equal_single_process(tmp_path):
    from tests.helpers import assert_close
    r0, r1 = _run(True, tmp_path)
    agent, cfg = _agent()
    batches, u = _batches(cfg, 2)
    ref = agent.update_parameters(batches[0], agent.update_step, 0, noise_u=u)
    for k, v in ref.items():
        tol = 3e-2 if k in ("actor_critic_loss", "critic_grad") else 2e-4       # post-Adam quantities: DESIGN.md 6
        want = 2 * v if k in ("reward_mask_num", "expert_mask_num") else v        # row counts are global sums
        assert_close(r0["ret"][k], want, tol, 1e-6, "rank0 " + k)
        assert_close(r1["ret"][k], want, tol, 1e-6, "rank1 " + k)
    assert_close(r0["pi"].numpy(), agent.pi.cpu().numpy(), 1e-4, 1e-5, "pi")
    assert_close(r0["q1"].numpy(), agent.qf1.cpu().numpy(), 1e-4, 1e-5, "q1")
    assert_close(r0["y"].numpy(), agent.next_q_value.cpu().numpy(), 1e-4, 1e-5, "td target")
    lr = 1e-3
    agent.update_parameters(batches[0], agent.update_step, 1, noise_u=u)
    st = _state(agent)
    for k, v in st.items():                                # two Adam steps: within 2 x 2.2 lr of each other (sign-like first steps)
        assert float((r0["state"][k] - v).abs().max()) <= 4.4 * lr + 1e-6, k


def test_two_ranks_different_batches_stay_in_lock_step(tmp_path):
    r0, r1 = _run(False, tmp_path)
    assert r0["finite"] and r1["finite"]
    for k, v in r0["state"].items():
        if "running" in k:
            continue
        assert torch.equal(v, r1["state"][k]), "replicas diverged at " + k
    for k in r0["ret2"]:
        if k == "critic_grad":
            continue        # the second step is a policy step: critic.grad then also holds the replica's own (unreduced,
                            # discarded) actor-term gradient, exactly like the reference's DataParallel replica
        assert r0["ret2"][k] == r1["ret2"][k] or abs(r0["ret2"][k] - r1["ret2"][k]) <= 1e-6 * abs(r0["ret2"][k]), k


def _worker_nccl1(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from ga_ddpg_amd.parallel import DataParallelContext
    res = {}
    import ga_ddpg_amd.parallel as par
    for use_dp in (False, True, "bucketed"):
        par.BUCKETED = use_dp == "bucketed"       # off by default (parallel.py): the overlapped two-bucket exchange stays covered
        par.PER_LANE_COMMS = use_dp == "bucketed"  # default: ONE communicator; per-lane communicators (GAD_DP_COMMS=lanes) stay covered
        agent, cfg = _agent()
        batches, u = _batches(cfg, 1)
        rt = agent.runtime(B, batches[0]["point_state_batch"].shape[2])
        if use_dp:
            dp = DataParallelContext()
            assert dp.comm is not None                 # backend nccl: the exchanges go through rccl.Communicator on the caller's stream
            agent._dp = dp
            os.environ["GAD_DP_CORUN_CHECK"] = "1"     # (auto = only with more than one rank)
            dp.attach(rt)
            # known-answer all-reduces of a gradient-bucket-sized buffer on every lane while split-bf16 GEMMs run on another lane,
            # and torch's fill / uniform kernels beside the same launches: bit-equal to the closed form / to the launch alone
            tr = dp.transport()
            assert tr["allreduce_corun_checked"] is True and tr["corun_exchange_mismatches"] == 0 and tr["corun_torch_kernel_mismatches"] == 0, tr
            assert tr["corun_exchanges_checked"] >= 12 and tr["corun_aggressor"].count("split") == 2, tr
            # default: one communicator for every lane (collectives serialised in host-issue order); GAD_DP_COMMS=lanes: one per
            # issuing lane (main, A, B, C) -- exchanges of different streams are independent RCCL operations
            want = 4 if par.PER_LANE_COMMS else 1
            assert dp.transport()["rccl_comms"] == want and len({id(c) for c in dp._lane_comm.values()}) == want
            assert rt.bucketed == (use_dp == "bucketed")
        rets = [agent.update_parameters(batches[0], agent.update_step, s, noise_u=u) for s in range(2)]
        torch.cuda.synchronize()
        res[use_dp] = {"rets": rets, "pi": agent.pi.cpu().clone(), "state": _state(agent)}
    torch.save(res, os.path.join(out_dir, "nccl1.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _worker_rccl_direct(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)          # the bootstrap needs a group, not this backend
    from ga_ddpg_amd import rccl
    comm = rccl.Communicator()
    side = torch.cuda.Stream()
    x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    d = torch.arange(1000, dtype=torch.float64, device="cuda")
    ref, refd = x.clone(), d.clone()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x.mul_(2.0)                                    # stream-ordered before ...
        comm.all_reduce_(x)                            # ... the collective on the SAME stream ...
        x.add_(1.0)                                    # ... and after it
        comm.all_reduce_(d[10:20])                     # a slice (contiguous view at an offset)
        comm.broadcast_(d)
    torch.cuda.current_stream().wait_stream(side)
    ok = bool(torch.equal(x, ref * 2.0 + 1.0)) and bool(torch.equal(d, refd))
    comm.destroy()
    torch.save({"ok": ok, "world": comm.world}, os.path.join(out_dir, "rccl_direct.pt"))
    dist.destroy_process_group()


def test_rccl_communicator_runs_on_the_callers_stream(tmp_path):
    """rccl.Communicator (RCCL's C API through ctypes, bootstrapped over a torch.distributed group): one-rank all-reduce /
    broadcast enqueued on a side stream between two kernels of that stream leave the values of a sum over one rank, in
    order -- the property the bucketed gradient exchange relies on (no stream of its own, no fifth hardware queue)."""
    mp.spawn(_worker_rccl_direct, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(str(tmp_path), "rccl_direct.pt"), weights_only=False)
    assert res["ok"] and res["world"] == 1


def test_single_rank_rccl_equals_plain_step(tmp_path):
    """the data-parallel hooks over the REAL transport (backend "nccl" = RCCL) with one rank: the count exchange on its own
    stream, the two gradient all-reduces (one of them issued from the actor stream) and the scalar reduction must leave
    the step unchanged -- this is the stream-ordering contract of RCCL collectives that the gloo tests cannot exercise."""
    mp.spawn(_worker_nccl1, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(str(tmp_path), "nccl1.pt"), weights_only=False)
    for mode in (True, "bucketed"):
        _compare_with_plain(res[False], res[mode])


def _compare_with_plain(a, b):
    for k in a["rets"][0]:
        tol = 2e-2 if k == "actor_critic_loss" else 1e-5
        assert abs(a["rets"][0][k] - b["rets"][0][k]) <= tol * abs(a["rets"][0][k]) + 1e-7, (k, a["rets"][0][k], b["rets"][0][k])
        assert np.isfinite(b["rets"][1][k])
    assert float((a["pi"] - b["pi"]).abs().max()) <= 5e-2 * float(a["pi"].abs().max())   # after two Adam steps: sanity bound
    for n in a["state"]:
        assert bool(torch.isfinite(b["state"][n]).all()), n


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (the driver's invocation form, no launcher around it) starts two ranks itself and rank 0
    prints ONE JSON line with n_gpus == 2.  gloo carries the tensors here: both ranks share the box's single GPU, which
    RCCL refuses -- a plumbing check of the N > 1 path, not a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GAD_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--batch", "32", "--buffer", "1500", "--probe-steps", "2", "--no-cpu-baseline", "--no-host-rate",
                          "--no-sa-kernel"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["scaling"] == "weak"
    assert res["config"]["parallelism"] == "dp2" and res["config"]["global_batch"] == 64
    assert res["value"] > 0 and np.isfinite(res["losses"]["critic_loss"])
    # the line proves its own multi-GPU claim: transport, rank count, replicas still bit-identical after the run
    assert res["config"]["transport"] == "torch.distributed/gloo" and res["config"]["replicas_bit_identical"] is True



def _worker_fallback(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from ga_ddpg_amd import rccl
    from ga_ddpg_amd.parallel import DataParallelContext

    def broken(*a, **k):
        raise RuntimeError("ncclCommInitRank failed: injected by the test")
    rccl.Communicator = broken
    agent, cfg = _agent()
    batches, u = _batches(cfg, 1)
    rt = agent.runtime(B, batches[0]["point_state_batch"].shape[2])
    dp = DataParallelContext()
    agent._dp = dp
    dp.attach(rt)
    flats = [rt.pol.flat, rt.cr.flat, rt.enc.flat, rt.venc.flat]
    dp.broadcast_parameters(flats)
    r = agent.update_parameters(batches[0], agent.update_step, 0, noise_u=u)
    torch.cuda.synchronize()
    res = {"direct": dp._direct, "comm_none": dp.comm is None, "bucketed": rt.bucketed, "transport": dp.transport(),
           "agree": dp.replicas_agree(flats), "finite": all(np.isfinite(v) for v in r.values())}
    torch.save(res, os.path.join(out_dir, "fallback.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_failing_rccl_init_falls_back_to_torch_distributed(tmp_path):
    """a direct-RCCL communicator that cannot be built (here: the constructor raises) must not take the
    job down -- every rank agrees to send its collectives through torch.distributed, the bucketed exchange (which only
    pays on the direct path) stays off, and the step runs."""
    mp.spawn(_worker_fallback, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(str(tmp_path), "fallback.pt"), weights_only=False)
    assert res["direct"] is False and res["comm_none"] and not res["bucketed"]
    assert res["transport"]["transport"] == "torch.distributed/nccl" and res["agree"] and res["finite"]


def _worker_two_gpus(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from ga_ddpg_amd.parallel import DataParallelContext
    agent, cfg = _agent()
    batches, u = _batches(cfg, 2)
    rt = agent.runtime(B, batches[rank]["point_state_batch"].shape[2])
    dp = DataParallelContext()
    agent._dp = dp
    dp.attach(rt)
    flats = [rt.pol.flat, rt.pol_t.flat, rt.cr.flat, rt.cr_t.flat, rt.enc.flat, rt.venc.flat]
    dp.broadcast_parameters(flats)
    for s in range(3):
        r = agent.update_parameters(batches[rank], agent.update_step, s, noise_u=u)
    torch.cuda.synchronize()
    res = {"transport": dp.transport(), "agree": dp.replicas_agree(flats), "bucketed": rt.bucketed, "ret": r,
           "state": _state(agent)}
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    dp.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the direct RCCL path with world_size 2)")
def test_two_gpus_direct_rccl_keeps_replicas_bit_equal(tmp_path):
    """two ranks on two GPUs over the DIRECT RCCL path (ncclCommInitRank with N = 2, bucketed exchange
    on the weight-gradient lanes): after 3 steps on different shards the replicas' parameters are bit-equal, RCCL itself
    reports 2 ranks, and the returned (global) losses agree.  Skipped on the one-GPU test boxes."""
    mp.spawn(_worker_two_gpus, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    for r in (r0, r1):
        assert r["agree"] and r["transport"]["rccl_nranks"] == 2 and r["transport"]["rccl_comms"] in (1, 4)
    for k, v in r0["state"].items():
        assert torch.equal(v, r1["state"][k]), "replicas diverged at " + k
    for k in r0["ret"]:
        if k != "critic_grad":
            assert abs(r0["ret"][k] - r1["ret"][k]) <= 1e-6 * abs(r0["ret"][k]) + 1e-9, k

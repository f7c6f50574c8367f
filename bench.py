"""bench.py -- DDPG gradient steps / second of the fused MI355X path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY 8d "config 2"): DDPG/TD3 update with goal-auxiliary heads
(ddpg_td3_aux.yaml), batch 256 per GPU, 1024-point clouds, synthetic seeded replay buffer, random-init
weights, alternating policy / non-policy steps.  A "step" = one Agent.update_parameters call.
`value` is measured with the minibatches already resident in HBM (a ring of pre-sampled batches);
the host-sampling + PCIe inclusive rate is reported separately as `value_host_inclusive`.
N > 1: one process per GPU, batch rows sharded (256 per rank -> weak scaling), one RCCL all-reduce of
the flat gradients per optimiser phase (ga_ddpg_amd.parallel).  `value` is then the whole-job aggregate: B=256
minibatch-steps per second summed over the ranks (= optimiser iterations/s x N; `config.iterations_per_s` holds the
iteration rate), time = max over ranks between two barrier + synchronize fences.

Extra objects on the JSON line:
  roofline      dominant kernel: algorithmic FLOPs of the layer as the reference computes it (dense
                padded neighbourhoods) / measured launch duration (HIP events on the launch
                stream), against the FP32 MFMA peak; `executed_frac` is the same with the FLOPs the
                de-duplicated kernel really executes.
  cpu_baseline  the CPU oracle (pure-PyTorch port of the reference step) timed on this box's cores,
                rank 0 / N=1 only, on a bounded sample (64 rows of one B=256 step, scaled).
  sa_kernel_hbm BASELINE's second metric: HBM GB/s of the streaming set-abstraction kernels of this workload
                (HIP events; algorithmic bytes and PMC traffic) and of the materialising configs[3] kernel.
  value_host_inclusive / value_device_replay: the same loop fed by host sampling + PCIe upload / by the
                GPU-resident replay mirror (never `value`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK = 157.3e12     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
SEED = 20260928

# dense algorithmic MACs per sample of each shared-MLP layer, as the reference computes them
# (rows = npoint * nsample incl. padded duplicates), keyed by launch tag.
def dense_layer_macs(tag, c_in):
    sa = {"sa1": (32 * 64, [(3 + c_in, 64), (64, 64), (64, 128)]),
          "sa2": (32 * 128, [(131, 128), (128, 128), (128, 256)]),
          "sa3": (32, [(259, 256), (256, 256), (256, 512)])}
    kind, stage, layer = tag.split(".")
    rows, dims = sa[stage]
    k, n = dims[int(layer[1:]) - 1]
    return rows * k * n


def sa_kernel_hbm(iters=50):
    """BASELINE.json's second metric ("SA-kernel HBM GB/s") on configs[3] (B=128, N=4096, npoint 512, radius 0.1,
    nsample 64, C=4): the materialising ball-query + group kernel behind pointnet2_utils.query_and_group -- the
    HBM-bound member of the set-abstraction family (SURVEY 8d "config 4a": 148.9 MB algorithmic bytes per call: points in,
    indices + grouped (B,7,512,64) tensor out).  HIP events on the launching stream."""
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    B, N, M, S, C = 128, 4096, 512, 64, 4
    g = torch.Generator(device="cuda").manual_seed(SEED)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    feats = torch.randn(B, C, N, device="cuda", generator=g)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), pu.furthest_point_sample(xyz, M)).transpose(1, 2).contiguous()
    from ga_ddpg_amd import hip
    idx = torch.empty(B, M, S, dtype=torch.int32, device="cuda")
    out = torch.empty(B, 3 + C, M, S, dtype=torch.float32, device="cuda")

    def launch():                                                  # the C-ABI entry point, caller-allocated outputs
        hip.call("gad_query_and_group", new_xyz, xyz, feats, B, C, N, M, 0.1, S, idx, out)

    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = B * N * (3 + C) * 4 + B * M * S * 4 + B * M * S * (3 + C) * 4
    gbps = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "ball_query_cells_kernel via gad_query_and_group (configs[3]: B=128, N=4096, npoint=512, r=0.1, nsample=64, "
                      "C=4)", "bound": "hbm", "algorithmic_bytes": nbytes, "launch_ms": ms, "achieved": gbps, "peak": 8000.0,
            "unit": "GB/s", "frac": gbps / 8000.0,
            "note": "back-to-back launches between HIP events on the launching stream; the output (134 MB) fits the 256 MB "
                    "Infinity Cache, so part of the write-back to HBM overlaps the next launch"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-sa-kernel", action="store_true", help="skip the configs[3] SA-kernel HBM measurement")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--buffer", type=int, default=20000, help="synthetic replay transitions per rank")
    ap.add_argument("--ring", type=int, default=8, help="pre-sampled minibatches kept resident in HBM")
    ap.add_argument("--roofline-tag", default="fwd.sa1.l3", help="launch tag of the kernel priced against the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-rate", action="store_true")
    return ap.parse_args()


def cpu_baseline(cfg, batch, noise, rows=64):
    """Bounded sample of the same workload on the host cores: ONE DDPG step of the CPU oracle (a port of the
    reference step; the reference itself cannot travel) on the first `rows` rows of a bench minibatch, scaled
    to B=256-step units (the step is linear in rows).  Threads capped at 64: torch's CPU kernels get slower
    beyond that on this many-core host (256 threads: 185 s for a B=256 step)."""
    from ga_ddpg_amd.experiments.config import load_cfg
    from oracle import ref_step
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    oracle = ref_step.OracleAgent(load_cfg("ddpg_td3_aux.yaml").RL_TRAIN)
    B = len(batch["reward_batch"])

    def head(n):
        return {k: (v[:n] if hasattr(v, "shape") and v.ndim > 0 and v.shape[0] == B else v) for k, v in batch.items()}
    oracle.update_parameters(head(8), noise_u=noise[:8])                 # page-in / warm-up on 8 rows
    t0 = time.time()
    oracle.update_parameters(head(rows), noise_u=noise[:rows])
    dt = time.time() - t0
    return {"value": (rows / float(B)) / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": "1 DDPG update step of the CPU oracle on %d of the %d rows (N=1024), %.1f s, scaled by %d/%d"
                      % (rows, B, dt, rows, B)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GAD_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a box with fewer GPUs than ranks (ranks then share
    # devices round-robin; RCCL refuses that) -- a plumbing check, not a measurement
    backend = os.environ.get("GAD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local if backend == "nccl" else local % torch.cuda.device_count())
    dp = None
    # GAD_BENCH_FORCE_DP=1: run the data-parallel hooks (count exchange, gradient all-reduces, scalar reduction) over a
    # ONE-rank RCCL group -- what they cost per step without any transport, the floor under the N > 1 numbers
    force_dp = world == 1 and os.environ.get("GAD_BENCH_FORCE_DP", "0") == "1"
    if world > 1 or force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        dist.init_process_group(backend, rank=rank, world_size=world)
    from ga_ddpg_amd import engine
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.parallel import DataParallelContext, mask_counts
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch

    torch.manual_seed(1234)                      # identical initial weights on every rank
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    B = args.batch
    mem = BaseMemory(args.buffer, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, args.buffer, seed=SEED + rank)
    rng = np.random.default_rng(SEED + 1000 + rank)
    host_batches = [sample_valid_batch(mem, B, rng) for _ in range(args.ring)]
    rt = agent.runtime(B, host_batches[0]["point_state_batch"].shape[2])
    if world > 1 or force_dp:
        dp = DataParallelContext()
        agent._dp = dp
        dp.attach(rt)
        dp.broadcast_parameters([rt.pol.flat, rt.pol_t.flat, rt.cr.flat, rt.cr_t.flat, rt.enc.flat, rt.venc.flat])
    from ga_ddpg_amd.runtime import BATCH_KEYS
    ring = []
    for hb in host_batches:
        d = {k: torch.as_tensor(np.ascontiguousarray(hb[k], dtype=np.float32)).cuda() for k in BATCH_KEYS}
        d["mask_counts"] = mask_counts(hb)
        ring.append(d)

    def step(i):
        out = agent.update_parameters(ring[i % len(ring)], agent.update_step, i)
        agent.step_scheduler(agent.update_step)
        return out

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    engine.TIMING.update(enabled=True, tag=args.roofline_tag, events=[])
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    engine.TIMING["enabled"] = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    ev = engine.TIMING["events"]
    kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) if ev else None
    n_launch = len(ev) / max(args.steps, 1)
    # roofline of the tagged launch: which encoder does it belong to? both encoders run it; use the
    # critic encoder's input width (C=10) for SA1 layer 1, irrelevant for the others.
    macs = dense_layer_macs(args.roofline_tag, 10) * B
    mult = {"fwd": 1}.get(args.roofline_tag.split(".")[0], 1)
    flops = 2.0 * macs * mult
    stage = int(args.roofline_tag.split(".")[1][2:]) - 1
    n_rows = int(rt.geo.rows[stage]["n"].item())
    dense_rows = rt.geo.counts[stage]
    roof = None
    if kernel_ms:
        ach = flops / (kernel_ms * 1e-3)
        # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
        # correction of MI355X_MICROARCH.md + WRITE_SIZE), see profiles/README.md; null when not collected for the tag
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(args.roofline_tag, {}).get("bytes_per_launch")
        kname = "gemm_fwd_stream_kernel" if args.roofline_tag.startswith("fwd.sa1") else "gemm_fwd_kernel"
        roof = {"bound": "mfma", "kernel": "%s (%s: shared-MLP layer, FP32 MFMA)" % (kname, args.roofline_tag),
                "achieved": ach / 1e12, "peak": FP32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK,
                "traffic": traffic, "launch_ms": kernel_ms, "launches_per_step": n_launch,
                "algorithmic_flops_per_launch": flops,
                "executed_frac": ach / FP32_MFMA_PEAK * n_rows / dense_rows,
                "dedup_rows": n_rows, "dense_rows": int(dense_rows)}
    steps_per_s = args.steps * 1.0 / dt
    # whole-job aggregate: every rank processes one B=256 minibatch per optimiser step (weak scaling), so the job does
    # world x (B=256 minibatch-steps) per iteration; at N=1 this is the plain step rate
    res = {"metric": "DDPG grad-steps/sec (B=%d, N=1024 pts)" % B, "value": steps_per_s * world, "unit": "steps/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[1]: DDPG/TD3 offline update (td3_critic_aux_policy_aux), "
                                  "batch=%d per GPU, 1024-pt clouds, synthetic replay buffer" % B,
                      "batch_per_gpu": B, "global_batch": B * world, "points": 1024,
                      "parallelism": "dp%d" % world, "iterations_per_s": steps_per_s,
                      "value_definition": "B=%d minibatch gradient steps per second summed over ranks (= iterations/s x n_gpus)" % B, "inputs": "HBM-resident ring of %d pre-sampled minibatches" % args.ring},
           "losses": {k: out[k] for k in ("critic_loss", "bc_loss", "actor_critic_loss")},
           "roofline": roof}
    # step-level view against the dense-FP32 roofline: mean 5.5078 GFLOP per sample and step
    res["step_dense_tflops"] = steps_per_s * world * B * 5.5078e9 / 1e12
    if world == 1 and not args.no_host_rate:
        rng2 = np.random.default_rng(7)
        n = max(10, args.steps // 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            b = sample_valid_batch(mem, B, rng2)
            agent.update_parameters(b, agent.update_step, i)
        torch.cuda.synchronize()
        res["value_host_inclusive"] = n / (time.perf_counter() - t0)
        # same loop fed by the GPU-resident replay mirror: indices drawn on the host with the
        # reference's arithmetic, gather in HBM -- the rate a training loop sees without the 17 MB/step host gather
        from ga_ddpg_amd.core.device_replay import DeviceReplay
        dmem = DeviceReplay(mem)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            agent.update_parameters(dmem.sample_lazy(B, rng2), agent.update_step, i)
        torch.cuda.synchronize()
        res["value_device_replay"] = n / (time.perf_counter() - t0)
    if world == 1 and not args.no_sa_kernel:
        # BASELINE.json's second metric, "SA-kernel HBM GB/s": the streaming set-abstraction forward kernels of THIS
        # workload (HBM-bound: 0.5-0.75 KB moved per row against 8-16 kFLOP), HIP-event duration from a short extra
        # pass that brackets every tagged launch, bytes = the committed PMC traffic (profiles/r01_traffic.json) and,
        # next to it, the algorithmic bytes (rows x (c_in + c_out) x 4); plus the materialising configs[3] kernel.
        engine.TIMING.update(enabled=True, tag="*", events=[])
        for i in range(10):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        engine.TIMING["enabled"] = False
        by_tag = {}
        for e0, e1, tag in engine.TIMING["events"]:
            by_tag.setdefault(tag, []).append(e0.elapsed_time(e1))
        # the same launches with the whole step on ONE stream: the kernel by itself, without the other passes of the
        # step competing for CUs and HBM (in the overlapped step two or three encoder passes run side by side)
        engine.SERIAL = True
        engine.TIMING.update(enabled=True, tag="*", events=[])
        for i in range(6):
            step(args.warmup + args.steps + 10 + i)
        torch.cuda.synchronize()
        engine.TIMING["enabled"] = False
        engine.SERIAL = False
        alone = {}
        for e0, e1, tag in engine.TIMING["events"]:
            alone.setdefault(tag, []).append(e0.elapsed_time(e1))
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        n1 = int(rt.geo.rows[0]["n"].item())
        sa = {}
        for tag, cin, cout in (("fwd.sa1.l2", 64, 64), ("fwd.sa1.l3", 64, 128)):
            if tag in by_tag:
                ms = float(np.mean(by_tag[tag]))
                ms1 = float(np.mean(alone[tag])) if tag in alone else None
                alg = n1 * (cin + cout) * 4.0
                pmc = tj.get(tag, {}).get("bytes_per_launch")
                sa[tag] = {"kernel": "gemm_fwd_stream_kernel", "bound": "hbm", "launch_ms": ms, "rows": n1,
                           "algorithmic_bytes": alg, "traffic": pmc, "achieved": alg / (ms * 1e-3) / 1e9, "peak": 8000.0,
                           "unit": "GB/s", "frac": alg / (ms * 1e-3) / 8e12,
                           "launch_ms_alone": ms1, "frac_alone": None if ms1 is None else alg / (ms1 * 1e-3) / 8e12,
                           "note": "launch_ms / frac: inside the overlapped step (other encoder passes run beside it); "
                                   "*_alone: the same launch with the step serialised on one stream"}
        if args.roofline_tag in alone and res.get("roofline"):
            ms1 = float(np.mean(alone[args.roofline_tag]))
            res["roofline"]["launch_ms_alone"] = ms1
            res["roofline"]["frac_alone"] = res["roofline"]["algorithmic_flops_per_launch"] / (ms1 * 1e-3) / 1e12 / res["roofline"]["peak"]
        sa["query_and_group"] = sa_kernel_hbm()
        sa["query_and_group"]["traffic"] = (json.load(open(tpath)).get("query_and_group", {}).get("bytes_per_launch")
                                            if os.path.exists(tpath) else None)
        res["sa_kernel_hbm"] = sa
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg, host_batches[0], np.random.default_rng(3).random((B, 6)).astype(np.float32))
    else:
        res["cpu_baseline"] = None
    print(json.dumps(res))


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()

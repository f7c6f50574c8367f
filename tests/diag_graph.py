"""Diagnostic: which part of the update step survives HIP-graph capture on this ROCm build.
    python tests/diag_graph.py            # runs every case in its own process and prints OK / the failure
    python tests/diag_graph.py <case>     # one case in this process"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = ["full", "full_np", "full_host", "steps"]


def run_case(name):
    import numpy as np
    import torch
    from ga_ddpg_amd import engine, hip, runtime
    from ga_ddpg_amd.api import make_agent
    from ga_ddpg_amd.core.replay_memory import BaseMemory
    from ga_ddpg_amd.synth_data import fill_synthetic_buffer, sample_valid_batch
    runtime.GRAPHS = False
    if "nodw" in name:
        engine.CONCURRENT_DW = False
    agent, cfg = make_agent("ddpg_td3_aux.yaml")
    B = 64
    mem = BaseMemory(800, cfg, point_dtype=np.float32)
    fill_synthetic_buffer(mem, 800, seed=1)
    rng = np.random.default_rng(1)
    b = sample_valid_batch(mem, B, rng)
    for i in range(3):
        agent.update_parameters(b, agent.update_step, i)
    rt = agent._rt
    d, P = rt.dbuf, rt.plans
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    ev, ev2 = torch.cuda.Event(), torch.cuda.Event()
    s1 = engine.side_stream(which=1)

    def body():
        main = torch.cuda.current_stream()
        if name == "geo":
            rt.geo_next.run(d["next_point_state_batch"])
        elif name == "t1":
            P["t1"].run()
        elif name == "fork_geo":
            ev.record(main); s1.wait_event(ev)
            with torch.cuda.stream(s1):
                rt.geo.run(d["point_state_batch"])
                ev2.record(s1)
            main.wait_event(ev2)
        elif name == "stats":
            rt._stats()
        elif name == "hyper":
            rt.pol.flat.hyper.copy_(rt.pol.flat.hyper_host, non_blocking=True)
            rt.dbuf["time_batch"].copy_(rt.hbuf["time_batch"], non_blocking=True)
            rt.scal_host.copy_(rt.scal, non_blocking=True)
        elif name == "rng":
            rt.noise_u.uniform_(0.0, 1.0)
        elif name == "zero":
            rt.scal.zero_(); rt.clip_sumsq.zero_(); rt.enc.bump_batches_tracked(2)
        elif name == "c_fwd_side":
            ev.record(main); s1.wait_event(ev)
            with torch.cuda.stream(s1):
                P["c_fwd"].run()
                ev2.record(s1)
            main.wait_event(ev2)
        elif name == "c_bwd":
            P["c_bwd"].run()
        elif name == "p_bwd":
            P["p_bwd"].run()
        elif name == "adam":
            rt._prestaged = "dev"
            rt._adam(rt.pol.flat, agent.policy_optim)
            rt._target_updates()
            rt._prestaged = None
        elif name in ("full", "full_wait", "full_tl"):
            rt._prestaged = "dev"
            rt._ddpg_enqueue(b, None, True)
            rt._prestaged = None
        elif name in ("np_prejoin", "np_prejoin_lanes"):
            which = (10 + 1, 10 + 2) if name.endswith("lanes") else (1, 2, 3, 10 + 1, 10 + 2)
            sts = [engine.side_stream(which=w) for w in which]
            e0 = torch.cuda.Event()
            e0.record(main)
            for st_ in sts:
                st_.wait_event(e0)
            rt._prestaged = "dev"
            rt._ddpg_enqueue(b, None, False)
            rt._prestaged = None
            for st_ in sts:
                e = torch.cuda.Event()
                e.record(st_)
                main.wait_event(e)
        elif name in ("full_np", "np_nodw", "np_noh2d", "np_nodw_noh2d"):
            if "noh2d" in name:
                class _NoCopy(object):
                    def __init__(self, t): self.t = t
                    def copy_(self, *a, **k): return self.t
                    def __getattr__(self, k): return getattr(self.t, k)
                for f in (rt.pol.flat, rt.enc.flat):
                    f.hyper_dev = f.hyper
                import types
                def _adam(self, flat, optim, clip=None):
                    hip.call("gad_adam_step", flat.master, flat.grad, flat.exp_avg, flat.exp_avg_sq, flat.active, flat.m2p,
                             flat.packed, flat.n, flat.hyper, clip, float(self.agent.clip_grad) if clip is not None else 0.0)
                rt._adam = types.MethodType(_adam, rt)
            rt._prestaged = "dev"
            rt._ddpg_enqueue(b, None, False)
            rt._prestaged = None
        elif name == "full_host":
            rt._stage_inputs(b)
            rt._prestaged = "host"
            rt._ddpg_enqueue(b, np.zeros((B, 6), np.float32), True)
            rt._prestaged = None
    if name == "steps":
        runtime.GRAPHS = True
        import faulthandler; faulthandler.enable()
        for i in range(6):
            agent.update_parameters(b, agent.update_step, i)
            print("step", i, "done", flush=True)
        print("replayed steps", rt.graph_replays, flush=True)
        return
    if name == "full_wait":
        cap = engine.side_stream(which=9)
        cur = torch.cuda.current_stream()
        rt.upload({k: torch.as_tensor(np.ascontiguousarray(b[k], dtype=np.float32)).cuda() for k in runtime.BATCH_KEYS})
        cap.wait_stream(cur)
    kw = dict(capture_error_mode="thread_local") if name == "full_tl" else {}
    with torch.cuda.graph(g, stream=cap, **kw):
        body()
    print("captured", name, flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", name, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True)
            last = [l for l in r.stdout.splitlines() if l.startswith(("captured", "replayed"))]
            err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Fatal" in l]
            print("%-12s rc=%4d  %s  %s" % (c, r.returncode, last[-1] if last else "-", err[-1][:160] if err else ""), flush=True)

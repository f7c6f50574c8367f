"""FPS launch time at the step's shape (B=256, N=1024 -> 32) and at configs[3] (B=128, N=4096 -> 512), HIP events.
    python tools/ubench_fps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_ddpg_amd import hip
cfgs = [int(c) for c in os.environ.get("FPS_CFGS", "0").split(",")]
for cfg, (B, N, M) in [(c, shp) for shp in ((256, 1024, 32), (128, 4096, 512), (128, 4096, 128)) for c in cfgs]:
    try:
        hip.set_option("fps_cfg", cfg)
    except RuntimeError:
        pass
    g = torch.Generator(device="cuda").manual_seed(1)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    idx = torch.empty(B, M, dtype=torch.int32, device="cuda")
    nx = torch.empty(B, M, 3, device="cuda")
    for _ in range(3):
        hip.call("gad_furthest_point_sampling", xyz, B, N, M, idx, nx)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        hip.call("gad_furthest_point_sampling", xyz, B, N, M, idx, nx)
    e1.record(); torch.cuda.synchronize()
    print("cfg %d " % cfg + "fps B=%d N=%d M=%d: %.1f us per launch (%.3f us per round)" % (B, N, M, e0.elapsed_time(e1) * 50, e0.elapsed_time(e1) * 50 / max(1, M - 1)))

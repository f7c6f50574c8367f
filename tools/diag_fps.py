"""diagnostic: FPS duration for configs[3] (B=128, N=4096 -> 512) under the thread-layout option, results must agree"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_ddpg_amd import hip
from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu

B, N, M = 128, 4096, 512
g = torch.Generator(device="cuda").manual_seed(1)
xyz = torch.rand(B, N, 3, device="cuda", generator=g)
ref = None
for cfg in (0, 1, 2):
    hip.set_option("fps_cfg", cfg)
    idx = pu.furthest_point_sample(xyz, M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        idx = pu.furthest_point_sample(xyz, M)
    e1.record(); torch.cuda.synchronize()
    if ref is None:
        ref = idx.clone()
    print("fps_cfg=%d: %.1f us   identical to cfg 0: %s" % (cfg, e0.elapsed_time(e1) / 10 * 1e3, bool((idx == ref).all())))
hip.set_option("fps_cfg", 0)

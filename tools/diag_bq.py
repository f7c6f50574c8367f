"""Diagnostic: configs[3] radius search / query_and_group timings, cell-list vs brute-force tile scan.
    python tools/diag_bq.py [modes ...] [qg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ga_ddpg_amd import hip
from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


B, N, M, S, C = 128, 4096, 512, 64, 4
g = torch.Generator(device="cuda").manual_seed(1)
xyz = torch.rand(B, N, 3, device="cuda", generator=g)
feats = torch.randn(B, C, N, device="cuda", generator=g)
new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), pu.furthest_point_sample(xyz, M)).transpose(1, 2).contiguous()
idx = torch.empty(B, M, S, dtype=torch.int32, device="cuda")
out = torch.empty(B, 3 + C, M, S, device="cuda")
nbytes = B * N * (3 + C) * 4 + B * M * S * 4 + B * M * S * (3 + C) * 4
QG_ONLY = "qg" in sys.argv[1:]                                    # profile runs: only the query_and_group launches
modes = [int(a) for a in sys.argv[1:] if a != "qg"] or (0, 1, 1 | 8, 1 | 2 | 4, 1 | 4, 1 | 2)   # debug bits: 8 empty, 2 no scan, 4 no write-out
for cells in modes:
    hip.set_option("bq_cells", cells)
    t_bq = 0.0 if QG_ONLY else timeit(lambda: hip.call("gad_ball_query", new_xyz, xyz, B, N, M, 0.1, S, idx, None))
    t_qg = timeit(lambda: hip.call("gad_query_and_group", new_xyz, xyz, feats, B, C, N, M, 0.1, S, idx, out))
    print("bq_cells=%d: ball_query %.1f us   query_and_group %.1f us (%.0f GB/s, %.1f%% of 8 TB/s)" %
          (cells, t_bq, t_qg, nbytes / t_qg / 1e3, nbytes / t_qg / 1e3 / 80))

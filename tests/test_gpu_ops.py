"""HIP operators (include/gaddpg.h section A/B) against the CPU oracle.  Index outputs must be
bit-exact; gathered values bit-exact; scatter-add gradients within float summation-order noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(B, N, seed, scale=0.1, offset=0.2):
    rng = np.random.default_rng(seed)
    return (rng.random((B, N, 3)) * scale + offset).astype(np.float32)


@pytest.mark.parametrize("B,N,M", [(3, 1024, 32), (2, 32, 32), (2, 100, 17), (1, 4096, 512), (2, 64, 64), (2, 5000, 40)])
def test_fps_matches_oracle(B, N, M):
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    from oracle import cref
    xyz = _clouds(B, N, 10 + N)
    xyz[0, : N // 7] *= 0.05           # some points inside the |p|^2 <= 1e-3 skip ball
    if N >= 100:
        xyz[-1, N // 2:N // 2 + 20] = xyz[-1, 5]   # exact duplicates -> exact ties
    want = cref.fps(xyz, M)
    got = pu.furthest_point_sample(torch.from_numpy(xyz).cuda(), M).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_fps_all_points_skipped():
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    xyz = torch.full((2, 64, 3), 0.001).cuda()
    assert (pu.furthest_point_sample(xyz, 8).cpu().numpy() == 0).all()


@pytest.mark.parametrize("B,N,M,r,S", [(3, 1024, 32, 0.02, 64), (2, 32, 32, 0.04, 128), (2, 777, 19, 0.05, 16),
                                        (1, 4096, 128, 0.1, 64), (2, 64, 8, 1e-4, 8)])
def test_ball_query_matches_oracle(B, N, M, r, S):
    from ga_ddpg_amd import hip
    from oracle import cref
    xyz = _clouds(B, N, 20 + N)
    fps = cref.fps(xyz, M)
    new_xyz = np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1)
    new_xyz[0, 0] += 10.0              # a centre with an empty ball -> all zeros
    want, wcnt = cref.ball_query(new_xyz, xyz, r, S, return_count=True)
    d_xyz, d_new = torch.from_numpy(xyz).cuda(), torch.from_numpy(new_xyz).cuda()
    idx = torch.full((B, M, S), -1, dtype=torch.int32, device="cuda")
    cnt = torch.full((B, M), -1, dtype=torch.int32, device="cuda")
    hip.call("gad_ball_query", d_new, d_xyz, B, N, M, float(r), S, idx, cnt)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    assert (want[0, 0] == 0).all()


def test_group_gather_and_grads():
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    from oracle import cref
    rng = np.random.default_rng(3)
    B, C, N, M, S = 3, 7, 257, 19, 12
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, M, S)).astype(np.int32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    out = pu.grouping_operation(f, torch.from_numpy(idx).cuda())
    np.testing.assert_array_equal(out.detach().cpu().numpy(), cref.group_points(feats, idx))
    go = rng.normal(size=out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).cuda())
    np.testing.assert_allclose(f.grad.cpu().numpy(), cref.group_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)
    # scalar-store path (M*S not a multiple of 4) and gather
    idx2 = rng.integers(0, N, size=(B, 5, 3)).astype(np.int32)
    out2 = pu.grouping_operation(torch.from_numpy(feats).cuda(), torch.from_numpy(idx2).cuda())
    np.testing.assert_array_equal(out2.cpu().numpy(), cref.group_points(feats, idx2))
    gi = rng.integers(0, N, size=(B, M)).astype(np.int32)
    f2 = torch.from_numpy(feats).cuda().requires_grad_(True)
    g = pu.gather_operation(f2, torch.from_numpy(gi).cuda())
    np.testing.assert_array_equal(g.detach().cpu().numpy(), cref.gather_points(feats, gi))
    gg = rng.normal(size=g.shape).astype(np.float32)
    g.backward(torch.from_numpy(gg).cuda())
    np.testing.assert_allclose(f2.grad.cpu().numpy(), cref.gather_points_grad(gg, gi, N), rtol=1e-5, atol=1e-5)


def test_query_and_group_matches_unfused():
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    from oracle.pointnet2_ops import pointnet2_utils as opu
    B, N, M, S = 2, 1024, 32, 64
    xyz = _clouds(B, N, 5)
    txyz = torch.from_numpy(xyz)
    feats = torch.cat([txyz.transpose(1, 2), torch.zeros(B, 1, N)], 1).contiguous()
    fps = opu.furthest_point_sample(txyz, M)
    new_xyz = opu.gather_operation(txyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    want = opu.QueryAndGroup(0.02, S)(txyz, new_xyz, feats).numpy()
    idx, out = pu.query_and_group(0.02, S, txyz.cuda(), new_xyz.cuda(), feats.cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    np.testing.assert_array_equal(idx.cpu().numpy(), opu.ball_query(0.02, S, txyz, new_xyz).numpy())
    # the unfused HIP module composition gives the same tensor
    got2 = pu.QueryAndGroup(0.02, S)(txyz.cuda(), new_xyz.cuda(), feats.cuda())
    np.testing.assert_array_equal(got2.cpu().numpy(), want)


def test_cpu_tensors_rejected():
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(torch.zeros(1, 8, 3), 2)


def test_rows_compaction():
    from ga_ddpg_amd import hip
    from oracle import cref
    B, N, M, S = 3, 512, 16, 32
    xyz = _clouds(B, N, 8)
    fps = cref.fps(xyz, M)
    new_xyz = np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1)
    new_xyz[1, 3] += 5.0
    idx, cnt = cref.ball_query(new_xyz, xyz, 0.015, S, return_count=True)
    G = B * M

    def d(a):
        return torch.from_numpy(a).cuda()
    off = torch.zeros(G + 1, dtype=torch.int32, device="cuda")
    cap = G * S
    pt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    grp = torch.zeros(cap, dtype=torch.int32, device="cuda")
    w = torch.zeros(cap, dtype=torch.float32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.call("gad_rows_from_ball_query", d(idx), d(cnt), G, M, N, S, off, pt, grp, w, n)
    c = np.maximum(cnt.reshape(-1), 1)
    want_off = np.concatenate([[0], np.cumsum(c)])
    np.testing.assert_array_equal(off.cpu().numpy(), want_off)
    assert int(n.item()) == want_off[-1]
    pt, grp, w = pt.cpu().numpy(), grp.cpu().numpy(), w.cpu().numpy()
    flat_idx = idx.reshape(G, S)
    for g in range(G):
        lo, hi = want_off[g], want_off[g + 1]
        np.testing.assert_array_equal(pt[lo:hi], (g // M) * N + flat_idx[g, :hi - lo])
        assert (grp[lo:hi] == g).all()
        assert w[lo] == S - (hi - lo) + 1 and (w[lo + 1:hi] == 1).all()
        assert w[lo:hi].sum() == S           # multiplicities reproduce the padded neighbourhood

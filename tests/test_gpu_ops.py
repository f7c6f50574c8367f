"""HIP operators (include/gaddpg.h section A/B) against the CPU oracle.  Index outputs must be
bit-exact; gathered values bit-exact; scatter-add gradients within float summation-order noise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _clouds(B, N, seed, scale=0.1, offset=0.2):
    rng = np.random.default_rng(seed)
    return (rng.random((B, N, 3)) * scale + offset).astype(np.float32)


@pytest.mark.parametrize("B,N,M", [(3, 1024, 32), (2, 32, 32), (2, 100, 17), (1, 4096, 512), (2, 64, 64), (2, 5000, 40), (1, 8000, 3000)])
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


@pytest.mark.parametrize("N,M", [(4096, 512), (1024, 32), (512, 64), (300, 50)])
def test_fps_exact_ties_on_a_lattice(N, M):
    """points on a coarse lattice (and whole duplicated blocks): almost every round has exact ties, resolved by upstream's
    block-reduction rule (smallest bit-reversed thread id, then smallest index) -- the kernel's float arg-max in key order,
    its tied-lane fallback and the cross-wavefront exchange must all reproduce the oracle's indices"""
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    from oracle import cref
    rng = np.random.default_rng(N)
    xyz = (rng.integers(0, 6, size=(3, N, 3)).astype(np.float32) * 0.125 + 0.25)
    xyz[1, N // 2:] = xyz[1, : N - N // 2]                      # every point twice
    xyz[2, ::7] = 0.0                                           # a seventh of the cloud inside the skip ball
    got = pu.furthest_point_sample(torch.from_numpy(xyz).cuda(), M).cpu().numpy()
    np.testing.assert_array_equal(got, cref.fps(xyz, M))


def test_fps_skip_rule_edge_point():
    """|p|^2 == 0x3A83126F (== 1e-3f, > the double 1e-3) is kept, the float below it is skipped: kernel == oracle == upstream"""
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    from oracle import cref
    from tests.test_oracle_ops import edge_points
    edge, below = edge_points()
    xyz = np.zeros((2, 64, 3), np.float32)
    xyz[:, :, 0] = np.linspace(0.5, 0.9, 64, dtype=np.float32)
    xyz[:, 1, :2], xyz[:, 2, :2] = edge, below
    xyz[1, 7, :2] = edge
    got = pu.furthest_point_sample(torch.from_numpy(xyz).cuda(), 8).cpu().numpy()
    np.testing.assert_array_equal(got, cref.fps(xyz, 8))
    assert got[0, 1] == 1 and 2 not in got[0]                  # the edge point is the farthest from point 0 (x = 0.5 .. 0.9 otherwise)


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


def _bq(new_xyz, xyz, r, S):
    from ga_ddpg_amd import hip
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.full((B, M, S), -1, dtype=torch.int32, device="cuda")
    cnt = torch.full((B, M), -1, dtype=torch.int32, device="cuda")
    hip.call("gad_ball_query", new_xyz, xyz, B, N, M, float(r), S, idx, cnt)
    return idx, cnt


@pytest.mark.parametrize("case", ["uniform", "ragged_n", "clustered", "tiny_radius", "huge_radius", "degenerate", "outside",
                                  "planar", "one_centroid"])
def test_cell_list_ball_query_matches_oracle(case):
    """the large-cloud radius search (1024 < N <= 4096: uniform grid in LDS + per-centroid bitmap, geometry.hip
    ball_query_cells_kernel) against the C oracle on the cases that stress the grid: ragged N, clusters that overflow
    single cells, a radius so small the grid resolution is capped, a radius larger than the cloud (every point a hit,
    truncation to the first nsample by index), zero-extent clouds, centroids outside the bounding box, M % 4 != 0."""
    from ga_ddpg_amd import hip
    from oracle import cref
    rng = np.random.default_rng(sum(map(ord, case)))
    B, N, M, r, S = 2, 4096, 67, 0.1, 64
    xyz = rng.random((B, N, 3)).astype(np.float32)
    if case == "ragged_n":
        N = 2501; xyz = xyz[:, :N].copy(); S = 17
    elif case == "clustered":
        xyz[:, : N // 2] = (0.5 + 0.01 * rng.normal(size=(B, N // 2, 3))).astype(np.float32); S = 128
    elif case == "tiny_radius":
        r = 1e-3; xyz[:, 1::2] = xyz[:, ::2] + np.float32(4e-4)
    elif case == "huge_radius":
        r = 5.0; S = 32
    elif case == "degenerate":
        xyz[:] = np.float32(0.25); xyz[1, 7] = 0.2501
    elif case == "planar":
        xyz[..., 2] = np.float32(-3.0); r = 0.05
    elif case == "one_centroid":
        M = 1
    pick = rng.integers(0, N, size=(B, M))
    new_xyz = np.take_along_axis(xyz, pick[..., None], 1).copy()
    if case == "outside":
        new_xyz[:, ::3] += np.float32(0.07) * rng.normal(size=new_xyz[:, ::3].shape).astype(np.float32)
        new_xyz[0, 0] = (-0.05, 0.5, 0.5); new_xyz[0, 1] = (1.5, 1.5, 1.5); new_xyz[1, 2] = (0.5, 1.09, -0.09)
    want, wcnt = cref.ball_query(new_xyz, xyz, r, S, return_count=True)
    idx, cnt = _bq(torch.from_numpy(new_xyz).cuda(), torch.from_numpy(xyz).cuda(), r, S)
    np.testing.assert_array_equal(idx.cpu().numpy(), want)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)


def test_cell_list_equals_brute_force_at_full_size():
    """configs[3] size: the cell-list kernel and the brute-force tile scan (option bq_cells = 0) agree bit for bit on
    indices, counts and the grouped tensor."""
    from ga_ddpg_amd import hip
    from ga_ddpg_amd.pointnet2_ops import pointnet2_utils as pu
    B, N, M, S = 128, 4096, 512, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    xyz = torch.rand(B, N, 3, device="cuda", generator=g)
    feats = torch.randn(B, 4, N, device="cuda", generator=g)
    new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), pu.furthest_point_sample(xyz, M)).transpose(1, 2).contiguous()
    try:
        hip.set_option("bq_cells", 0)
        i0, c0 = _bq(new_xyz, xyz, 0.1, S)
        q0, o0 = pu.query_and_group(0.1, S, xyz, new_xyz, feats)
    finally:
        hip.set_option("bq_cells", 1)
    i1, c1 = _bq(new_xyz, xyz, 0.1, S)
    q1, o1 = pu.query_and_group(0.1, S, xyz, new_xyz, feats)
    assert torch.equal(i0, i1) and torch.equal(c0, c1) and torch.equal(q0, q1) and torch.equal(q1, i1)
    assert torch.equal(o0, o1)
    assert 5 < float(c1.float().mean()) < 40                      # ~17 neighbours expected: the test is not vacuous


def _contraction_sensitive_case(rng, N):
    """a cloud, one centroid and a radius for which the squared distance of one point lies on different sides of r^2
    depending on whether dy*dy + dx*dx is fused into an FMA or rounded twice (the oracle: every operation rounded)"""
    f32 = np.float32
    while True:
        c = rng.random(3).astype(f32)
        pts = (c + (rng.random((4096, 3)).astype(f32) - f32(0.5)) * f32(0.2)).astype(f32)
        d = (c[None] - pts).astype(f32)
        xx, yy, zz = (d[:, 0] * d[:, 0]).astype(f32), (d[:, 1] * d[:, 1]).astype(f32), (d[:, 2] * d[:, 2]).astype(f32)
        pinned = ((xx + yy).astype(f32) + zz).astype(f32)
        fused = ((d[:, 1].astype(np.float64) ** 2 + xx.astype(np.float64)).astype(f32) + zz).astype(f32)
        for i in np.nonzero(pinned != fused)[0]:
            lo, hi = min(pinned[i], fused[i]), max(pinned[i], fused[i])
            r = f32(np.sqrt(np.float64(hi)))
            for _ in range(8):                                     # a float32 radius whose float32 square is in (lo, hi]
                r2 = f32(r * r)
                if lo < r2 <= hi:
                    xyz = (rng.random((1, N, 3)) * 2.0 - 0.5).astype(f32)          # mostly far away
                    xyz[0, :32] = pts[:32]
                    xyz[0, 5] = pts[i]
                    return xyz, c.reshape(1, 1, 3), float(r)
                r = np.nextafter(r, f32(0) if r2 > hi else f32(4), dtype=f32)


@pytest.mark.parametrize("N,cells", [(512, 1), (2048, 1), (2048, 0)])
def test_ball_query_distance_is_not_fma_contracted(N, cells):
    """d^2 < r^2 is evaluated as ((dx*dx) + (dy*dy)) + (dz*dz) with every operation rounded (oracle/pn2_ref.c sqdist);
    radii chosen so that a fused multiply-add anywhere in that expression flips a neighbour"""
    from ga_ddpg_amd import hip
    from oracle import cref
    rng = np.random.default_rng(77 + N + cells)
    try:
        hip.set_option("bq_cells", cells)
        for _ in range(6):
            xyz, ctr, r = _contraction_sensitive_case(rng, N)
            want, wcnt = cref.ball_query(ctr, xyz, r, 64, return_count=True)
            idx, cnt = _bq(torch.from_numpy(ctr).cuda(), torch.from_numpy(xyz).cuda(), r, 64)
            np.testing.assert_array_equal(idx.cpu().numpy(), want)
            np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    finally:
        hip.set_option("bq_cells", 1)


@pytest.mark.parametrize("B,R,C_,sp,dp", [(3, 4, 1024, 1024, 4), (2, 128, 512, 512, 128), (5, 37, 45, 48, 40), (1, 1, 1, 1, 1), (2, 1000, 7, 8, 1000)])
def test_transpose_batched(B, R, C_, sp, dp):
    """gad_transpose_batched == tensor.transpose(1, 2) on the leading R x C block of every batch, pitches honoured, padding untouched"""
    import ctypes as C
    from ga_ddpg_amd import hip
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + R)
    src = torch.randn(B, R, sp, device="cuda", generator=g)
    dst = torch.full((B, C_, dp), -7.0, device="cuda")
    hip.call("gad_transpose_batched", src, dst, B, R, C_, sp, C.c_longlong(R * sp), dp, C.c_longlong(C_ * dp))
    assert torch.equal(dst[:, :, :R], src[:, :, :C_].transpose(1, 2))
    assert bool((dst[:, :, R:] == -7.0).all())


@pytest.mark.parametrize("G,unaligned", [(65536, False), (4099, False), (8192, True), (5, False), (4096, False), (12288, False)])
def test_rows_scan_offsets(G, unaligned):
    """the group-offset scan of gad_rows_from_ball_query (one workgroup, 4096 groups per pass, 16-byte loads when the pointers
    allow): exclusive sums of max(cnt, 1), counts of 0 included, several passes, a ragged last pass, unaligned buffers"""
    from ga_ddpg_amd import hip
    S, M, N = 4, 1, 64
    rng = np.random.default_rng(G)
    cnt_h = rng.integers(0, S + 1, size=G).astype(np.int32)
    pad = 1 if unaligned else 0
    cnt = torch.zeros(G + pad, dtype=torch.int32, device="cuda")
    cnt[pad:] = torch.from_numpy(cnt_h).cuda()
    idx = torch.zeros(G, S, dtype=torch.int32, device="cuda")
    off = torch.zeros(G + 1 + pad, dtype=torch.int32, device="cuda")
    cap = G * S
    pt = torch.zeros(cap, dtype=torch.int32, device="cuda")
    grp = torch.zeros(cap, dtype=torch.int32, device="cuda")
    w = torch.zeros(cap, dtype=torch.float32, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    hip.call("gad_rows_from_ball_query", idx, cnt[pad:], G, M, N, S, off[pad:], pt, grp, w, n)
    want = np.concatenate([[0], np.cumsum(np.maximum(cnt_h, 1))])
    np.testing.assert_array_equal(off[pad:].cpu().numpy(), want)
    assert int(n.item()) == want[-1]

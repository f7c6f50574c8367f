"""Known-answer and property tests pinning the CPU oracle's PointNet++ operator layer
(oracle/pn2_ref.c, oracle/pointnet2_ops).  The reference holds no test for these operators (they
live in the un-vendored pointnet2_ops extension), so they are pinned by hand-computed cases,
brute-force numpy restatements and invariants."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import cref


def test_fps_hand_computed():
    # points on a line x = [1,2,3,4,9,6]: start at 0, farthest is 4 (d2=64).  Then temp = [0,1,4,9,0,9]:
    # a tie between k=3 and k=5.  Upstream block of 4 threads: thread t = k mod 4 -> (9,k=5) sits in
    # thread 1, (9,k=3) in thread 3; the tree's first level merges (t1,t3) with `v2 > v1 ? i2 : i1`
    # -> thread 1's candidate k=5 survives.  Next temp = [0,1,4,4,0,0]: tie k=2 (thread 2) vs k=3
    # (thread 3); level one keeps both (vs t0, t1), level two merges (t0<-k=2, t1<-k=3) -> k=2.
    x = np.array([1.0, 2.0, 3.0, 4.0, 9.0, 6.0], dtype=np.float32)
    xyz = np.stack([x, np.zeros(6, np.float32), np.zeros(6, np.float32)], 1)[None]
    np.testing.assert_array_equal(cref.fps(xyz, 4)[0], [0, 4, 5, 2])


def test_fps_skips_points_near_origin():
    xyz = np.array([[[0.5, 0, 0], [0.01, 0.0, 0.0], [0.6, 0, 0], [-3.0, 0, 0], [0.02, 0.01, 0.0]]], dtype=np.float32)
    idx = cref.fps(xyz, 5)[0]
    assert idx[0] == 0 and idx[1] == 3 and idx[2] == 2
    # |p|^2 <= 1e-3 points (1 and 4) are never selected; once every valid point has distance 0 the
    # emulated upstream tree reduction falls back to a valid index again, never to 1 or 4
    assert 1 not in idx and 4 not in idx


def edge_points():
    """points (x, y, 0) whose |p|^2 = (x*x + y*y) + 0, rounded as the kernels round it, is exactly 0x3A83126F (the float nearest
    to 0.001, slightly ABOVE it) resp. 0x3A83126E (the float below)"""
    want = {0x3A83126F: None, 0x3A83126E: None}
    x = np.float32(0.02)
    xx = np.float32(x * x)
    y = np.float32(np.sqrt(1e-3 - float(xx)))
    for _ in range(4000):
        y = np.nextafter(y, np.float32(0), dtype=np.float32)
    for _ in range(8000):
        bits = int(np.float32(xx + np.float32(y * y)).view(np.uint32))
        if bits in want and want[bits] is None:
            want[bits] = (x, y)
        y = np.nextafter(y, np.float32(1), dtype=np.float32)
    assert all(v is not None for v in want.values())
    return want[0x3A83126F], want[0x3A83126E]


def test_fps_skip_rule_is_the_float_vs_double_literal_compare():
    """upstream: `if (mag <= 1e-3) continue;` with a float mag and a DOUBLE literal: |p|^2 = 0x3A83126F (0.00100000005) is NOT
    skipped although it equals 1e-3f; the next float below is"""
    (xe, ye), (xb, yb) = edge_points()
    xyz = np.array([[[0.5, 0, 0], [xe, ye, 0], [xb, yb, 0], [0.6, 0, 0]]], dtype=np.float32)
    idx = cref.fps(xyz, 3)[0]
    np.testing.assert_array_equal(idx, [0, 1, 3])              # the edge point is a candidate (the farthest from 0.5); x_below never is


def _brute_fps(xyz, m):
    n = xyz.shape[0]
    temp = np.full(n, 1e10, np.float32)
    valid = (xyz.astype(np.float32) ** 2).sum(1).astype(np.float64) > 1e-3
    out = [0]
    old = 0
    for _ in range(1, m):
        d = ((xyz - xyz[old]) ** 2)
        d = (d[:, 0] + d[:, 1]) + d[:, 2]
        temp = np.where(valid, np.minimum(temp, d.astype(np.float32)), temp)
        cand = np.where(valid, temp, -1.0)
        old = int(np.argmax(cand))            # first arg-max: equals upstream only when there is no tie
        out.append(old)
    return np.array(out)


def test_fps_matches_bruteforce_without_ties():
    rng = np.random.default_rng(0)
    xyz = (rng.random((3, 300, 3)) + 0.5).astype(np.float32)
    got = cref.fps(xyz, 40)
    for b in range(3):
        np.testing.assert_array_equal(got[b], _brute_fps(xyz[b], 40))


def test_fps_tie_rule_is_bit_reversed_thread_order():
    # 8 points, identical distances after the first pick -> upstream's tree reduction (block of 8) picks
    # the candidate with the smallest bit-reversed thread id among the maxima: 4 (100b -> 001b)
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, :, 0] = 1.0
    xyz[0, 0, 0] = 2.0
    assert cref.fps(xyz, 2)[0, 1] == 4


def test_ball_query_hand_computed():
    xyz = np.array([[[0, 0, 1.0], [0.05, 0, 1.0], [0.2, 0, 1.0], [0.09, 0, 1.0], [0.0, 0.099, 1.0]]], np.float32)
    new_xyz = np.array([[[0, 0, 1.0], [5, 5, 5]]], np.float32)
    idx, cnt = cref.ball_query(new_xyz, xyz, 0.1, 6, return_count=True)
    np.testing.assert_array_equal(idx[0, 0], [0, 1, 3, 4, 0, 0])      # ascending hits, padded with the first
    np.testing.assert_array_equal(idx[0, 1], [0] * 6)                 # empty ball -> zeros
    np.testing.assert_array_equal(cnt[0], [4, 0])
    idx2 = cref.ball_query(new_xyz, xyz, 0.1, 2)
    np.testing.assert_array_equal(idx2[0, 0], [0, 1])                 # truncated at nsample
    # strict '<' on the squared distance in float32
    edge = np.array([[[0.1, 0, 1.0]]], np.float32)
    r = np.float32(0.1)
    d2 = np.float32(edge[0, 0, 0]) * np.float32(edge[0, 0, 0])
    assert (cref.ball_query(new_xyz[:, :1], edge, 0.1, 1, True)[1][0, 0] == 1) == bool(d2 < r * r)


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 3), st.integers(5, 90), st.integers(1, 9), st.integers(1, 12), st.floats(0.05, 0.6),
       st.integers(0, 10 ** 6))
def test_ball_query_properties(B, N, M, S, r, seed):
    rng = np.random.default_rng(seed)
    xyz = rng.random((B, N, 3)).astype(np.float32)
    new_xyz = xyz[:, rng.integers(0, N, M)]
    idx, cnt = cref.ball_query(new_xyz, xyz, r, S, return_count=True)
    r2 = np.float32(r) * np.float32(r)
    for b in range(B):
        for m in range(M):
            d = xyz[b] - new_xyz[b, m]
            d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32)
            hits = np.nonzero(d2 < r2)[0]
            k = min(len(hits), S)
            assert cnt[b, m] == k
            np.testing.assert_array_equal(idx[b, m, :k], hits[:k])
            assert (idx[b, m, k:] == (hits[0] if k else 0)).all()


def test_group_gather_against_numpy():
    rng = np.random.default_rng(1)
    B, C, N, M, S = 2, 5, 33, 7, 4
    f = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, M, S)).astype(np.int32)
    want = np.stack([f[b][:, idx[b]] for b in range(B)])
    np.testing.assert_array_equal(cref.group_points(f, idx), want)
    go = rng.normal(size=want.shape).astype(np.float32)
    g = np.zeros_like(f)
    for b in range(B):
        for m in range(M):
            for s in range(S):
                g[b, :, idx[b, m, s]] += go[b, :, m, s]
    np.testing.assert_allclose(cref.group_points_grad(go, idx, N), g, rtol=1e-6, atol=1e-6)
    gi = rng.integers(0, N, (B, M)).astype(np.int32)
    np.testing.assert_array_equal(cref.gather_points(f, gi), np.stack([f[b][:, gi[b]] for b in range(B)]))


def test_sa_module_pool_is_invariant_to_duplicate_padding():
    """The premise of the de-duplicated GPU path: the padded copies of the first neighbour change the
    BatchNorm statistics only through their multiplicity, never the max-pool."""
    from oracle.pointnet2_ops import pointnet2_modules as pm
    torch.manual_seed(0)
    m = pm.PointnetSAModule(npoint=8, radius=0.15, nsample=16, mlp=[3, 8, 8]).eval()   # eval: BN affine only
    xyz = torch.rand(2, 64, 3) + 0.3
    f = xyz.transpose(1, 2).contiguous()
    _, out16 = m(xyz, f)
    m.groupers[0].nsample = 64
    _, out64 = m(xyz, f)
    cnt = cref.ball_query(cref_centres(xyz, 8), xyz.numpy(), 0.15, 64, True)[1]
    if cnt.max() <= 16:                                   # same unique neighbours -> identical pooled output
        torch.testing.assert_close(out16, out64)


def cref_centres(xyz, m):
    idx = cref.fps(xyz.numpy(), m)
    return np.take_along_axis(xyz.numpy(), idx[..., None].astype(np.int64), 1)

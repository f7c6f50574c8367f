"""Shared comparison helpers for the parity tests."""
import numpy as np
import torch

from oracle.detfill import summarize


def assert_close(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not np.all(err <= tol):
        i = int(np.argmax(err - tol))
        raise AssertionError("%s: max violation at flat index %d: got %.9g expected %.9g (|err| %.3g > tol %.3g); "
                             "max abs err %.3g" % (what, i, a.ravel()[i], b.ravel()[i], err.ravel()[i],
                                                   tol.ravel()[i], err.max()))


def assert_close_rows(a, b, rtol, atol, what="", stray_rows=1, stray_factor=20.0):
    """Entry-wise rtol / atol for per-sample tensors (one row per sample), except that `stray_rows` rows may be off by
    up to `stray_factor` x the tolerance: two float32 evaluations of the step resolve the odd ReLU / max-pool tie
    differently (tests/test_gpu_forced_decisions.py counts them: 0-4 decisions in 8e5), which moves the outputs of the
    sample it sits in by ~1e-5 and nothing else."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if a.ndim < 1 or a.shape[0] < 4:
        return assert_close(a, b, rtol, atol, what)
    err = np.abs(a - b).reshape(a.shape[0], -1)
    tol = (atol + rtol * np.abs(b)).reshape(a.shape[0], -1)
    over = (err / tol).max(axis=1)
    stray = np.nonzero(over > 1.0)[0]
    if len(stray) > stray_rows or (len(stray) and over.max() > stray_factor):
        i = int(np.argmax(over))
        raise AssertionError("%s: %d rows beyond tolerance (allowed: %d within %gx); worst row %d at %.3g x tol, max abs err %.3g"
                             % (what, len(stray), stray_rows, stray_factor, i, over[i], err.max()))


def check_summaries(golden, prefix, named_tensors, rtol, atol, skip=(), normwise=False, l2_rtol=2e-2, max_rtol=5e-2):
    """Compare tensors against fingerprints written by oracle.detfill.summarize_named.
    The tolerance on the aggregate stats is scaled by the tensor's abs-sum / l2.
    normwise=True: entries are compared with tolerance rtol * max|tensor| (+atol) instead of
    rtol * |entry| -- the right yardstick for gradients, whose small entries are differences of
    large terms (a weight-gradient tensor here spans 1e0 .. 7e2)."""
    n = 0
    for name, t in named_tensors:
        if t is None or any(s in name for s in skip):
            continue
        ks, kv = prefix + name + "#stats", prefix + name + "#vals"
        assert ks in golden, "missing golden entry " + ks
        stats, vals = summarize(t)
        g_stats, g_vals = golden[ks], golden[kv]
        if normwise:
            # Gradients of this network are piecewise-smooth: a ReLU / max-pool / L1 / min() kink that two
            # float32 evaluations resolve differently reroutes one row's contribution (expected ~10 such
            # flips per pass among SA1's 1.4e7 pre-activations, ~1 in SA3 where a row is 1/1024 of the
            # batch; measured between torch-float32 and torch-float64 as well).  Hence:
            # median entry error within rtol*max|tensor|, every entry within max_rtol*max|tensor|.
            scale = float(g_stats[3])
            err = np.abs(np.asarray(vals, np.float64) - g_vals)
            assert np.median(err) <= atol + rtol * scale, "%s: median err %.3g > %.3g" % (kv, np.median(err), rtol * scale)
            assert err.max() <= atol + max_rtol * scale, "%s: max err %.3g (scale %.3g)" % (kv, err.max(), scale)
            assert abs(stats[2] - g_stats[2]) <= l2_rtol * g_stats[2] + atol, ks + " l2"
            n += 1
            continue
        else:
            assert_close(vals, g_vals, rtol, atol, what=kv)
        abs_sum = max(g_stats[1], 1e-30)
        assert abs(stats[0] - g_stats[0]) <= rtol * abs_sum + atol * max(1.0, vals.size), ks + " sum"
        assert abs(stats[1] - g_stats[1]) <= rtol * abs_sum + atol * max(1.0, vals.size), ks + " abs-sum"
        assert abs(stats[2] - g_stats[2]) <= rtol * g_stats[2] + atol, ks + " l2"
        assert abs(stats[3] - g_stats[3]) <= rtol * g_stats[3] + atol, ks + " abs-max"
        n += 1
    assert n > 0, "nothing compared under " + prefix
    return n


def golden_batch(golden, prefix):
    p = prefix + "batch/"
    return {k[len(p):]: golden[k] for k in golden.files if k.startswith(p)}


def grad_accuracy_rows(g32, g64, prefix, named, skip=()):
    """Per-tensor accuracy of gradients against the REFERENCE's float64 evaluation (tests/golden/ddpg_steps_B32_f64.npz),
    with the reference's own float32 run (ddpg_steps_B32.npz) as the yardstick.  Both goldens hold strided samples
    (oracle.detfill.summarize).  -> rows (name, scale, e_hip_median, e_hip_max, e_ref32_median, e_ref32_max), errors
    relative to the tensor's max |entry| in float64."""
    rows = []
    for name, t in named:
        kv, ks = prefix + name + "#vals", prefix + name + "#stats"
        if t is None or kv not in g64.files or any(s in name for s in skip):
            continue
        _, vals = summarize(t)
        ref = g64[kv].astype(np.float64)
        scale = float(g64[ks][3]) + 1e-300
        e_hip = np.abs(np.asarray(vals, np.float64) - ref) / scale
        e_r32 = np.abs(g32[kv].astype(np.float64) - ref) / scale
        rows.append((name, scale, float(np.median(e_hip)), float(e_hip.max()), float(np.median(e_r32)), float(e_r32.max())))
    return rows

"""ctypes binding of oracle/pn2_ref.c (built on demand with oracle/Makefile).  Oracle only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "_build", "libpn2ref.so")
    src = os.path.join(_HERE, "pn2_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        ci = ctypes.c_int
        L.pn2ref_fps.argtypes = [fp, ci, ci, ci, ip]
        L.pn2ref_ball_query.argtypes = [fp, fp, ci, ci, ci, ctypes.c_float, ci, ip, ip]
        L.pn2ref_group_points.argtypes = [fp, ip, ci, ci, ci, ci, ci, fp]
        L.pn2ref_group_points_grad.argtypes = [fp, ip, ci, ci, ci, ci, ci, fp]
        L.pn2ref_gather_points.argtypes = [fp, ip, ci, ci, ci, ci, fp]
        L.pn2ref_gather_points_grad.argtypes = [fp, ip, ci, ci, ci, ci, fp]
        for f in ("pn2ref_fps", "pn2ref_ball_query", "pn2ref_group_points", "pn2ref_group_points_grad",
                  "pn2ref_gather_points", "pn2ref_gather_points_grad"):
            getattr(L, f).restype = None
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def fps(xyz, npoint):
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    idx, pi = _i(np.zeros((B, npoint), np.int32))
    lib().pn2ref_fps(px, B, N, npoint, pi)
    return idx


def ball_query(new_xyz, xyz, radius, nsample, return_count=False):
    new_xyz, pn = _f(new_xyz)
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx, pi = _i(np.zeros((B, M, nsample), np.int32))
    cnt, pc = _i(np.zeros((B, M), np.int32))
    lib().pn2ref_ball_query(pn, px, B, N, M, float(radius), nsample, pi, pc)
    return (idx, cnt) if return_count else idx


def group_points(pts, idx):
    pts, pp = _f(pts)
    idx, pi = _i(idx)
    B, C, N = pts.shape
    _, M, S = idx.shape
    out, po = _f(np.zeros((B, C, M, S), np.float32))
    lib().pn2ref_group_points(pp, pi, B, C, N, M, S, po)
    return out


def group_points_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M, S = grad_out.shape
    out, po = _f(np.zeros((B, C, N), np.float32))
    lib().pn2ref_group_points_grad(pg, pi, B, C, N, M, S, po)
    return out


def gather_points(pts, idx):
    pts, pp = _f(pts)
    idx, pi = _i(idx)
    B, C, N = pts.shape
    M = idx.shape[1]
    out, po = _f(np.zeros((B, C, M), np.float32))
    lib().pn2ref_gather_points(pp, pi, B, C, N, M, po)
    return out


def gather_points_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    out, po = _f(np.zeros((B, C, N), np.float32))
    lib().pn2ref_gather_points_grad(pg, pi, B, C, N, M, po)
    return out

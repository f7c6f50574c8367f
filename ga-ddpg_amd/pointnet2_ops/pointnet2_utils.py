"""pointnet2_ops.pointnet2_utils surface over libgaddpg (include/gaddpg.h section A).

Each function mirrors upstream's autograd.Function of the same name: same argument order, shapes,
dtypes (indices are int32), contiguity / device checks raising RuntimeError, gradients only where
upstream defines them (features of grouping_operation / gather_operation)."""
import torch
import torch.nn as nn

from .. import hip


def _check(*tensors):
    hip.require_cuda(*tensors)


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) float32 CUDA contiguous -> (B,npoint) int32."""
    _check(xyz)
    if xyz.dtype != torch.float32:
        raise RuntimeError("xyz must be a float tensor")
    B, N, _ = xyz.shape
    idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    hip.call("gad_furthest_point_sampling", xyz, B, N, int(npoint), idx, None)
    return idx


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        _check(features, idx)
        B, C, N = features.shape
        M = idx.shape[1]
        out = torch.empty(B, C, M, dtype=torch.float32, device=features.device)
        hip.call("gad_gather_points", features, idx, B, C, N, M, out)
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, M = grad_out.shape
        g = torch.empty(B, C, ctx.N, dtype=torch.float32, device=grad_out.device)
        hip.call("gad_gather_points_grad", grad_out, idx, B, C, ctx.N, M, g)
        return g, None


def gather_operation(features, idx):
    return _Gather.apply(features, idx)


def ball_query(radius, nsample, xyz, new_xyz):
    """-> (B,M,nsample) int32."""
    _check(xyz, new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty(B, M, nsample, dtype=torch.int32, device=xyz.device)
    hip.call("gad_ball_query", new_xyz, xyz, B, N, M, float(radius), int(nsample), idx, None)
    return idx


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        _check(features, idx)
        B, C, N = features.shape
        _, M, S = idx.shape
        out = torch.empty(B, C, M, S, dtype=torch.float32, device=features.device)
        hip.call("gad_group_points", features, idx, B, C, N, M, S, out)
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, M, S = grad_out.shape
        g = torch.empty(B, C, ctx.N, dtype=torch.float32, device=grad_out.device)
        hip.call("gad_group_points_grad", grad_out, idx, B, C, ctx.N, M, S, g)
        return g, None


def grouping_operation(features, idx):
    return _Group.apply(features, idx)


def query_and_group(radius, nsample, xyz, new_xyz, features):
    """fused QueryAndGroup(use_xyz=True) forward (no autograd): -> (idx, (B,3+C,M,S))."""
    _check(xyz, new_xyz, features)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    C = features.shape[1]
    idx = torch.empty(B, M, nsample, dtype=torch.int32, device=xyz.device)
    out = torch.empty(B, 3 + C, M, nsample, dtype=torch.float32, device=xyz.device)
    hip.call("gad_query_and_group", new_xyz, xyz, features, B, C, N, M, float(radius), int(nsample), idx, out)
    return idx, out


class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped

"""Oracle restatement of pointnet2_ops.pointnet2_utils (CPU, float32).  Oracle only.

Index-producing ops (FPS, ball query) run in oracle/pn2_ref.c with a pinned evaluation order;
gathers are plain torch indexing so autograd provides the scatter-add gradients that upstream
implements with atomicAdd (group_points_grad / gather_points_grad).
"""
import torch
import torch.nn as nn

from oracle import cref


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) f32 -> (B,npoint) int32 indices (upstream furthest_point_sampling)."""
    idx = cref.fps(xyz.detach().cpu().numpy(), int(npoint))
    return torch.from_numpy(idx).to(xyz.device)


def gather_operation(features, idx):
    """features (B,C,N), idx (B,M) -> (B,C,M);  out[b,c,m] = features[b,c,idx[b,m]]."""
    B, C, _ = features.shape
    return features.gather(2, idx.long().unsqueeze(1).expand(B, C, idx.shape[1]))


def ball_query(radius, nsample, xyz, new_xyz):
    """-> (B,M,nsample) int32: first nsample in-radius indices, padded with the first hit."""
    idx = cref.ball_query(new_xyz.detach().cpu().numpy(), xyz.detach().cpu().numpy(),
                          float(radius), int(nsample))
    return torch.from_numpy(idx).to(xyz.device)


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,M,S) -> (B,C,M,S);  out[b,c,m,s] = features[b,c,idx[b,m,s]]."""
    B, C, _ = features.shape
    _, M, S = idx.shape
    flat = idx.long().reshape(B, 1, M * S).expand(B, C, M * S)
    return features.gather(2, flat).reshape(B, C, M, S)


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

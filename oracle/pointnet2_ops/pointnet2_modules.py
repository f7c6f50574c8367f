"""Oracle restatement of pointnet2_ops.pointnet2_modules (CPU).  Oracle only.

Module tree and state-dict key layout follow upstream: ``groupers`` / ``mlps`` ModuleLists, each
shared MLP an ``nn.Sequential`` of [Conv2d 1x1 (bias = not bn), BatchNorm2d, ReLU] triples, so the
keys are ``mlps.0.{0,3,6}.weight`` and ``mlps.0.{1,4,7}.{weight,bias,running_*}``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def build_shared_mlp(mlp_spec, bn=True):
    layers = []
    for cin, cout in zip(mlp_spec[:-1], mlp_spec[1:]):
        layers.append(nn.Conv2d(cin, cout, kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class PointnetSAModuleMSG(nn.Module):
    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            spec = list(spec)
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(build_shared_mlp(spec, bn))

    def forward(self, xyz, features):
        new_xyz = None
        if self.npoint is not None:
            fps_idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            x = mlp(grouper(xyz, new_xyz, features))
            outs.append(F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn,
                         use_xyz=use_xyz)

"""pointnet2_ops.pointnet2_modules surface: PointnetSAModule / PointnetSAModuleMSG with upstream's
module tree (`groupers`, `mlps`; shared MLP = Sequential of [Conv2d 1x1 no-bias, BatchNorm2d, ReLU]
triples) so reference checkpoints' keys `mlps.0.{0,1,3,4,6,7}.*` load unchanged.

forward(xyz (B,N,3), features (B,C,N)) -> (new_xyz (B,npoint,3) | None, (B,C',npoint)) runs the fused
set-abstraction path of libgaddpg (FPS, ball query, de-duplicated rows, FP32-MFMA shared MLP with
train-mode BatchNorm statistics, segment max-pool); see ga_ddpg_amd.sa_function for the autograd
wrapper.  Modules hold ordinary nn.Parameters; the fused update step (core.agent) re-homes them into
flat buffers."""
import torch.nn as nn

from . import pointnet2_utils


def build_shared_mlp(mlp_spec, bn=True):
    layers = []
    for cin, cout in zip(mlp_spec[:-1], mlp_spec[1:]):
        layers.append(nn.Conv2d(cin, cout, kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


class _Scale(nn.Module):
    """one (radius, nsample, shared MLP) scale of a multi-scale module, shaped like a single-scale module for the fused path; it
    shares the parent's grouper and MLP objects (the same nn.Parameters) and is NOT registered in the parent's module tree, so
    the parent's state_dict keeps upstream's keys"""

    def __init__(self, parent, i):
        super().__init__()
        self.npoint, self.radius, self.nsample = parent.npoint, parent.radii[i], parent.nsamples[i]
        self.groupers = nn.ModuleList([parent.groupers[i]])
        self.mlps = nn.ModuleList([parent.mlps[i]])


class PointnetSAModuleMSG(nn.Module):
    """bn=True, use_xyz=True with point features (every module GA-DDPG builds, reference core/networks.py:66-81): the fused
    de-duplicated-row path of libgaddpg.  The other forms upstream's constructor accepts -- bn=False (biased convolutions, no
    BatchNorm), use_xyz=False (the recentred coordinates are not concatenated), features=None -- take upstream's own composition
    (_PointnetSAModuleBase.forward) over this package's operators: furthest_point_sample -> gather_operation -> QueryAndGroup /
    GroupAll (libgaddpg section A kernels, differentiable through their _grad counterparts) -> the shared MLP as torch modules ->
    max-pool over the neighbourhood.  Same results as upstream's module; the padded (B, C, npoint, nsample) tensor is materialised,
    as it is there (no shipped GA-DDPG configuration builds these forms)."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        self.bn, self.use_xyz = bool(bn), bool(use_xyz)
        self.npoint = npoint
        self.radii, self.nsamples = list(radii), list(nsamples)
        self.radius, self.nsample = radii[0], nsamples[0]
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            spec = list(spec)
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                spec[0] += 3
            self.mlps.append(build_shared_mlp(spec, bn))
        if len(self.radii) > 1:           # (plain attribute: kept out of _modules / state_dict)
            object.__setattr__(self, "_scales", [_Scale(self, i) for i in range(len(self.radii))])

    def _generic_forward(self, xyz, features):
        """upstream _PointnetSAModuleBase.forward over this package's operators (see the class docstring)"""
        import torch
        import torch.nn.functional as F
        new_xyz = None
        if self.npoint is not None:
            idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            x = mlp(grouper(xyz, new_xyz, features))
            outs.append(F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)

    def forward(self, xyz, features):
        from ..sa_function import sa_module_forward
        if not (self.bn and self.use_xyz) or features is None:
            return self._generic_forward(xyz, features)
        if len(self.radii) == 1:
            return sa_module_forward(self, xyz, features)
        # multi-scale grouping (upstream PointnetSAModuleMSG.forward): every scale groups around the SAME centroids -- furthest
        # point sampling is deterministic, so each scale's fused pass re-derives them -- and the pooled features are
        # concatenated along the channels in scale order
        import torch
        new_xyz, outs = None, []
        for sc in self._scales:
            sc.train(self.training)
            new_xyz, f = sa_module_forward(sc, xyz, features)
            outs.append(f)
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn, use_xyz=use_xyz)

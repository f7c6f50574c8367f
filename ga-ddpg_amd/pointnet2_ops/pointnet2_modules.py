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


class PointnetSAModuleMSG(nn.Module):
    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        if len(radii) != 1 or not bn or not use_xyz:
            raise NotImplementedError("GA-DDPG uses single-scale SA modules with bn=True, use_xyz=True")
        self.npoint = npoint
        self.radius, self.nsample = radii[0], nsamples[0]
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            spec = list(spec)
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            spec[0] += 3
            self.mlps.append(build_shared_mlp(spec, bn))

    def forward(self, xyz, features):
        from ..sa_function import sa_module_forward
        return sa_module_forward(self, xyz, features)


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(npoint=npoint, radii=[radius], nsamples=[nsample], mlps=[mlp], bn=bn, use_xyz=use_xyz)

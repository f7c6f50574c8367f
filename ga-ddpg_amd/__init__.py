"""ga-ddpg_amd: MI355X-native (gfx950) implementation of GA-DDPG's PointNet++-encoded
actor-critic update step.  Import as ``ga_ddpg_amd`` (alias package at the repo root).

Layout
  csrc/         hand-written HIP kernels + the C-ABI shared library (libgaddpg.so, include/gaddpg.h)
  hip.py        ctypes binding of the C-ABI (raises if the library is missing: no CPU fallback)
  pointnet2_ops/  drop-in for the reference's external `pointnet2_ops` package (HIP-backed)
  core/         host-side mirror of the reference's core/{networks,agent,ddpg,bc,loss,utils,
                replay_memory}.py surface for the update-step path
  experiments/  config defaults + yaml merge with the reference's semantics
"""
__version__ = "0.1.0"

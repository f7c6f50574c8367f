"""HIP-backed drop-in for the reference's external `pointnet2_ops` package (imported at reference
core/networks.py:10 and core/utils.py:32).  Same Python surface, same state-dict keys; the
operators run in libgaddpg.so (gfx950) and reject CPU tensors like the upstream extension does."""

"""CPU oracle stand-in for the un-vendored `pointnet2_ops` package (TEST INFRASTRUCTURE ONLY).

Restates the Python surface GA-DDPG imports (reference core/networks.py:10, core/utils.py:32):
``pointnet2_modules.PointnetSAModule`` and ``pointnet2_utils.{furthest_point_sample,
gather_operation, ball_query, grouping_operation, QueryAndGroup, GroupAll}`` following the
published erikwijmans/Pointnet2_PyTorch v3.x semantics.  It is used
(a) as the oracle for the HIP operators and (b) as the ``pointnet2_ops`` shim when
oracle/make_golden.py imports the reference's own Python to generate tests/golden.
"""

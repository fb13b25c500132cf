"""pointnet2_ops.pointnet2_utils surface used by the reference (furthest_point_sample, gather_operation)."""
from gm3d_amd.ops import furthest_point_sample, gather_operation  # noqa: F401

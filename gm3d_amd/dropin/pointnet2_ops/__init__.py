"""Import-name shim: `from pointnet2_ops import pointnet2_utils` (Point-MAE_SA3D/models_mae_learn_loss.py:25)."""
from . import pointnet2_utils  # noqa: F401

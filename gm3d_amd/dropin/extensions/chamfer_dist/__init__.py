"""`from extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2` (models_mae_learn_loss.py:26)."""
from gm3d_amd.ops import ChamferDistanceL1, ChamferDistanceL2  # noqa: F401

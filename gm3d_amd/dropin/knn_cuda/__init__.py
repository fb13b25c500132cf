"""`from knn_cuda import KNN` (Point-MAE_SA3D/models_mae_learn_loss.py:24)."""
from gm3d_amd.ops import KNN  # noqa: F401

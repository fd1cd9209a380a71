"""Import path of the reference (clair_torch/inference/measure_linearity.py); the implementation lives with the
pair-residual kernels' host code in training/linearity.py."""
from ..training.linearity import measure_linearity  # noqa: F401

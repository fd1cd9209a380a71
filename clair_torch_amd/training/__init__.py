from .losses import (pixelwise_linearity_loss, compute_spatial_linearity_loss, gaussian_value_weights,
                     combined_gaussian_pair_weights, compute_monotonicity_penalty,
                     compute_smoothness_penalty, compute_range_penalty, compute_endpoint_penalty)
from .linearity import linearity_loss, measure_linearity, spatial_mean, spatial_statistics
from .icrf_training import train_icrf

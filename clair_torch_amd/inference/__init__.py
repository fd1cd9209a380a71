from .hdr_merge import compute_hdr_image
from .linearization import linearize_dataset_generator
from .inferential_statistics import compute_video_mean_and_std


def __getattr__(name):  # measure_linearity lives with the training kernels; import lazily to avoid a cycle
    if name == "measure_linearity":
        from ..training.linearity import measure_linearity
        return measure_linearity
    raise AttributeError(name)

"""Host-side helpers that sit on the hot path's boundary (enums, transforms, pair selection)."""
from .enums import (InterpMode, MissingStdMode, VarianceMode, ChannelOrder, DimensionOrder, DTYPE_MAP,
                    REVERSE_DTYPE_MAP)
from .typecheck import TypeCheckError
from .statistics import WBOMean, WBOMeanVar
from .data_io import load_icrf_txt, save_icrf_txt
from .general_functions import (get_valid_exposure_pairs, get_pairwise_valid_pixel_mask, weighted_mean_and_std,
                                flat_field_mean, flatfield_correction)

"""Host-side helpers that sit on the hot path's boundary (enums, transforms, pair selection)."""
from .enums import (InterpMode, MissingStdMode, VarianceMode, ChannelOrder, DimensionOrder, DTYPE_MAP,
                    REVERSE_DTYPE_MAP)
from .typecheck import TypeCheckError
from .statistics import WBOMean, WBOMeanVar
from .data_io import load_icrf_txt, save_icrf_txt

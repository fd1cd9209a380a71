"""Host-side helpers that sit on the hot path's boundary (enums, transforms, pair selection)."""
from .enums import InterpMode, MissingStdMode, VarianceMode, DTYPE_MAP, REVERSE_DTYPE_MAP
from .typecheck import TypeCheckError
from .statistics import WBOMean, WBOMeanVar

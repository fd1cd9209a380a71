from .base import ICRFModelBase
from .icrf_model import ICRFModelDirect

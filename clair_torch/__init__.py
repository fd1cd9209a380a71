"""Import-path compatibility: ``clair_torch.<subpackage>.<module>`` names used by the reference's scripts
(scripts/run_hdr_merging.py:5-14, run_icrf_model_training.py, run_image_linearization.py, run_linearity_measurement.py)
resolve to the MI355X implementation in ``clair_torch_amd``.  Only the hot-path modules exist; the
reference's file I/O, metadata and plotting packages are out of scope and importing them raises ImportError."""
import importlib
import sys

_ALIASES = {
    "clair_torch.common": "clair_torch_amd.common",
    "clair_torch.common.enums": "clair_torch_amd.common.enums",
    "clair_torch.common.transforms": "clair_torch_amd.common.transforms",
    "clair_torch.common.general_functions": "clair_torch_amd.common.general_functions",
    "clair_torch.common.statistics": "clair_torch_amd.common.statistics",
    "clair_torch.common.data_io": "clair_torch_amd.common.data_io",
    "clair_torch.inference.inferential_statistics": "clair_torch_amd.inference.inferential_statistics",
    "clair_torch.datasets": "clair_torch_amd.datasets",
    "clair_torch.datasets.collate": "clair_torch_amd.datasets.collate",
    "clair_torch.models": "clair_torch_amd.models",
    "clair_torch.models.base": "clair_torch_amd.models.base",
    "clair_torch.models.icrf_model": "clair_torch_amd.models.icrf_model",
    "clair_torch.inference": "clair_torch_amd.inference",
    "clair_torch.inference.hdr_merge": "clair_torch_amd.inference.hdr_merge",
    "clair_torch.inference.linearization": "clair_torch_amd.inference.linearization",
    "clair_torch.inference.measure_linearity": "clair_torch_amd.inference.measure_linearity",
    "clair_torch.training": "clair_torch_amd.training",
    "clair_torch.training.losses": "clair_torch_amd.training.losses",
    "clair_torch.training.icrf_training": "clair_torch_amd.training.icrf_training",
}
for _alias, _target in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_target)

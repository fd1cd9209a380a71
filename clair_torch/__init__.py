"""Import-path compatibility: ``clair_torch.<subpackage>.<module>`` names used by the reference's scripts
(scripts/run_hdr_merging.py:3-12, run_icrf_model_training.py:10-18, run_image_linearization.py,
run_linearity_measurement.py) resolve to the MI355X implementation in ``clair_torch_amd``.

Only the hot-path modules exist here (the table below).  Everything else the scripts import --
``clair_torch.common.parameters``, ``.file_settings``, ``clair_torch.datasets.image_dataset``, ``clair_torch.metadata``,
``save_image`` / ``load_image`` of ``clair_torch.common.data_io`` -- is file I/O, configuration and filename parsing that
this build does not re-implement.  Those names resolve in exactly one situation: a reference install is importable
further down ``sys.path`` (or named by ``CLAIR_TORCH_REFERENCE``).  Then this package acts as an OVERLAY:
  * a module this package does not provide is loaded from the reference's file of the same dotted name;
  * an attribute missing from a module this package does provide (``load_image`` in ``common.data_io``) is taken from
    the reference's module of the same name, loaded privately.
Inside the reference's modules ``import clair_torch.<hot path>`` lands here, so its file handling feeds the HIP
kernels.  Without a reference install those imports raise ImportError / AttributeError naming this docstring.
"""
import importlib
import importlib.abc
import importlib.util
import os
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
_HERE = os.path.dirname(os.path.realpath(__file__))
_PRIVATE = "_clair_torch_reference"  # reference modules shadowed by an alias are loaded under this prefix


def _reference_root():
    """Directory of a reference ``clair_torch`` package other than this one, or None."""
    candidates = [os.environ.get("CLAIR_TORCH_REFERENCE")] + [os.path.join(p or ".", "clair_torch") for p in sys.path]
    for cand in candidates:
        if cand and os.path.isfile(os.path.join(cand, "__init__.py")) and os.path.realpath(cand) != _HERE:
            return os.path.realpath(cand)
    return None


def _spec_in_reference(fullname, parts):
    root = _reference_root()
    if root is None:
        return None
    base = os.path.join(root, *parts)
    if os.path.isfile(os.path.join(base, "__init__.py")):
        return importlib.util.spec_from_file_location(fullname, os.path.join(base, "__init__.py"),
                                                      submodule_search_locations=[base])
    if os.path.isfile(base + ".py"):
        return importlib.util.spec_from_file_location(fullname, base + ".py")
    return None


class _ReferenceOverlay(importlib.abc.MetaPathFinder):
    """Last-resort finder: ``clair_torch.x.y`` that no earlier finder located -> the reference install's x/y.py."""

    def find_spec(self, fullname, path, target=None):
        if fullname.startswith("clair_torch.") and fullname not in _ALIASES:
            return _spec_in_reference(fullname, fullname.split(".")[1:])
        if fullname.startswith(_PRIVATE + "."):
            return _spec_in_reference(fullname, fullname.split(".")[1:])
        return None


def _fallback_getattr(alias):
    """Module-level __getattr__ (PEP 562) for an aliased module: names it lacks come from the reference's module."""
    parts = alias.split(".")[1:]

    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        private = ".".join([_PRIVATE] + parts)
        if private not in sys.modules and _spec_in_reference(private, parts) is None:
            raise AttributeError(f"{alias} (MI355X hot-path build) has no attribute {name!r} and no reference install of "
                                 "clair_torch is importable to supply it -- see clair_torch/__init__.py")
        if _PRIVATE not in sys.modules:  # parent package shell so that the dotted private name can be imported
            shell = importlib.util.module_from_spec(importlib.util.spec_from_loader(_PRIVATE, loader=None, is_package=True))
            shell.__path__ = []
            sys.modules[_PRIVATE] = shell
        for k in range(1, len(parts)):
            pkg = ".".join([_PRIVATE] + parts[:k])
            if pkg not in sys.modules:
                sub = importlib.util.module_from_spec(importlib.util.spec_from_loader(pkg, loader=None, is_package=True))
                sub.__path__ = []
                sys.modules[pkg] = sub
        return getattr(importlib.import_module(private), name)

    return __getattr__


for _alias, _target in _ALIASES.items():
    _module = importlib.import_module(_target)
    sys.modules[_alias] = _module
    if "__getattr__" not in vars(_module):
        _module.__getattr__ = _fallback_getattr(_alias)
if not any(isinstance(f, _ReferenceOverlay) for f in sys.meta_path):
    sys.meta_path.append(_ReferenceOverlay())

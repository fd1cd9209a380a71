"""Argument checking at the API boundary.

The reference decorates every entry point with typeguard's ``@typechecked`` and so raises
``typeguard.TypeCheckError`` on a wrong argument type (e.g. clair_torch/inference/hdr_merge.py:18).
typeguard is an optional dependency here: when it is importable its exception class is re-used, otherwise a
local ``TypeCheckError`` (a TypeError) stands in so callers can catch the same name.
"""
try:  # pragma: no cover - depends on the environment
    from typeguard import TypeCheckError
except Exception:  # typeguard absent
    class TypeCheckError(TypeError):
        pass


def expect(value, types, name, allow_none=False):
    if value is None and allow_none:
        return
    if not isinstance(value, types):
        raise TypeCheckError(f"argument \"{name}\" ({type(value).__qualname__}) is not an instance of {types}")

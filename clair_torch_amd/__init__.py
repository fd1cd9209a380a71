"""clair_torch_amd -- MI355X (gfx950) implementation of clair-torch's per-pixel hot path.

ICRF linearization, exposure-weighted HDR merge with propagated uncertainty, and the per-pixel
linearity residual used in ICRF training, as hand-written HIP kernels behind the reference's
``clair_torch.inference`` / ``clair_torch.models`` / ``clair_torch.training`` API.
"""
__version__ = "0.1.0"

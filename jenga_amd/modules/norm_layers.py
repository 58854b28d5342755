"""Per-head RMSNorm, drop-in for hyvideo/modules/norm_layers.py:5-59 (same constructor, same `.weight` state-dict
key).  forward() runs the HIP kernel jenga_rmsnorm_rope (norm only) for [..., H, 128] inputs on the GPU."""
import torch
import torch.nn as nn

from .. import _capi


class RMSNorm(nn.Module):
    def __init__(self, dim: int, elementwise_affine=True, eps: float = 1e-6, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.eps = eps
        self.dim = dim
        if elementwise_affine:
            self.weight = nn.Parameter(torch.ones(dim, **factory_kwargs))

    def forward(self, x):
        if x.shape[-1] != 128 or self.dim != 128:
            raise ValueError("jenga_amd.RMSNorm implements the per-head (dim=128) QK-norm of the Jenga DiT blocks")
        w = getattr(self, "weight", None)
        x4 = x if x.dim() == 4 else x.reshape(1, -1, 1, 128)
        y = _capi.rmsnorm_rope(x4, w, None, None, eps=self.eps)
        return y if x.dim() == 4 else y.reshape(x.shape)


def get_norm_layer(norm_layer):
    if norm_layer == "layer":
        return nn.LayerNorm
    if norm_layer == "rms":
        return RMSNorm
    raise NotImplementedError(f"Norm layer {norm_layer} is not implemented")

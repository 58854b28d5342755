"""RMSNorm, drop-in for hyvideo/modules/norm_layers.py:5-59 (same constructor, same `.weight` state-dict key).  forward() runs
the HIP kernel jenga_rmsnorm_rope (norm only) for the per-head [..., H, 128] QK-norm of the Jenga blocks, and jenga_rmsnorm_rows
for any other width (a multiple of 8, <= 8192: the same formula, `(x.float() * rsqrt(mean(x^2) + eps)).type_as(x) * weight`)."""
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
        w = getattr(self, "weight", None)
        if x.shape[-1] != self.dim:
            raise ValueError(f"jenga_amd.RMSNorm({self.dim}): the input's last dimension is {x.shape[-1]}")
        if self.dim != 128:
            # any other width (no Jenga entry script uses one): the full-row kernel of the Wan flavour, weight in x's dtype
            if self.dim % 8 or self.dim > 8192:
                raise ValueError("jenga_amd.RMSNorm: the width must be a multiple of 8 and at most 8192")
            return _capi.rmsnorm_rows(x, w.to(x.dtype) if w is not None else torch.ones(self.dim, dtype=x.dtype, device=x.device),
                                      self.eps)
        x4 = x if x.dim() == 4 else x.reshape(1, -1, 1, 128)
        y = _capi.rmsnorm_rope(x4, w, None, None, eps=self.eps)
        return y if x.dim() == 4 else y.reshape(x.shape)


def get_norm_layer(norm_layer):
    if norm_layer == "layer":
        return nn.LayerNorm
    if norm_layer == "rms":
        return RMSNorm
    raise NotImplementedError(f"Norm layer {norm_layer} is not implemented")

"""xfuser.core.long_ctx_attention (hyvideo/inference.py:84; the driver then takes hyvideo.modules.xdit_ring_atten's class of
the same name) -> jenga_amd's Ulysses module.  Only served when xfuser itself is not installed."""
from jenga_amd.modules.ulysses import xFuserLongContextAttention  # noqa: F401

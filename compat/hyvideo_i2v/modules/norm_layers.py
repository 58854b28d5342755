"""hyvideo_i2v.modules.norm_layers -> jenga_amd."""
from jenga_amd.modules.norm_layers import RMSNorm, get_norm_layer  # noqa: F401

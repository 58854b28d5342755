"""hyvideo.modules.norm_layers (models_mul_block_gc_ha_multigpu.py:15) -> jenga_amd."""
from jenga_amd.modules.norm_layers import RMSNorm, get_norm_layer  # noqa: F401

"""hyvideo.modules.attenion (jenga_hyvideo.py:19, models_mul_block_gc_ha_multigpu.py:19) -> jenga_amd."""
from jenga_amd.modules.attention import attention, get_cu_seqlens, my_parallel_attention, parallel_attention  # noqa: F401

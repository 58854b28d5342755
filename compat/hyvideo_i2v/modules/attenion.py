"""hyvideo_i2v.modules.attenion (jenga_hyi2v.py:19) -> jenga_amd."""
from jenga_amd.modules.attention import attention, get_cu_seqlens, my_parallel_attention, parallel_attention  # noqa: F401

"""hyvideo.modules.xdit_ring_atten (jenga_hyvideo_multigpu.py:181) -> jenga_amd (RCCL Ulysses exchange)."""
from jenga_amd.modules.ulysses import xFuserLongContextAttention  # noqa: F401

"""hyvideo.modules.attention_block_triton_diffres (models_mul_block_gc_ha_multigpu.py:27) -> jenga_amd."""
from jenga_amd.modules.attention_block_sparse import block_sparse_attention  # noqa: F401

"""wan.modules.attention_block_triton_diffres (wan/modules/model_mul.py:9) -> the Wan flavour of jenga_amd."""
from jenga_amd.modules.attention_block_sparse import block_sparse_attention_wan as block_sparse_attention  # noqa: F401

"""hyvideo_i2v.modules.attention_block_triton_diffres -> the padded, four-text-block flavour of jenga_amd."""
from jenga_amd.modules.attention_block_sparse import block_sparse_attention_i2v as block_sparse_attention  # noqa: F401

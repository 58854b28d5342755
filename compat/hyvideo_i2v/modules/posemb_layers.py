"""hyvideo_i2v.modules.posemb_layers -> jenga_amd."""
from jenga_amd.modules.posemb_layers import (apply_rotary_emb, apply_rotary_emb_single, get_1d_rotary_pos_embed,  # noqa: F401
                                             get_meshgrid_nd, get_nd_rotary_pos_embed)

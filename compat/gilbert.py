"""`gilbert` as the reference drivers import it (jenga_hyvideo.py:24, jenga_wan.py:34) -> jenga_amd.gilbert (HIP kernels)."""
from jenga_amd.gilbert import (gilbert_block_neighbor_mapping, gilbert_mapping,  # noqa: F401
                               sliced_gilbert_block_neighbor_mapping, sliced_gilbert_mapping, transpose_gilbert_mapping)

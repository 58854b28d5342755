"""xfuser.core.distributed accessors the Jenga drivers use (jenga_hyvideo_multigpu.py:28-32, models_mul...:28-32)
-> jenga_amd.modules.ulysses (torch.distributed on RCCL).  Only served when xfuser itself is not installed."""
from jenga_amd.modules.ulysses import (get_sequence_parallel_rank, get_sequence_parallel_world_size,  # noqa: F401
                                       get_sp_group, init_sequence_parallel)

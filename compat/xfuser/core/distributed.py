"""xfuser.core.distributed as the Jenga drivers use it (hyvideo/inference.py:22-30, 171-182; jenga_hyvideo_multigpu.py:28-32;
models_mul_block_gc_ha_multigpu.py:28-32) -> jenga_amd.modules.ulysses (torch.distributed on RCCL).  Only served when xfuser
itself is not installed."""
from jenga_amd.modules.ulysses import (get_sequence_parallel_rank, get_sequence_parallel_world_size,  # noqa: F401
                                       get_sp_group, init_distributed_environment, init_sequence_parallel,
                                       initialize_model_parallel)

from .data_loader import SyntheticSmokeDataset, create_data_loaders
from .distributed import init_distributed, shard_range, wrap_ddp

__all__ = ["SyntheticSmokeDataset", "create_data_loaders", "init_distributed", "shard_range", "wrap_ddp"]

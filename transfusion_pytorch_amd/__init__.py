"""transfusion_pytorch_amd - MI355X-native Transfusion training / sampling hot path.

Mirrors the reference's public surface (transfusion_pytorch/__init__.py:1-6)."""
from .transfusion import (
    Transfusion,
    Transformer,
    LossBreakdown,
    print_modality_sample,
    create_dataloader,
)

from .ema import EMA

__all__ = ['Transfusion', 'Transformer', 'LossBreakdown', 'print_modality_sample', 'create_dataloader', 'EMA']

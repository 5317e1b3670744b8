"""transfusion_pytorch_amd - MI355X-native Transfusion training / sampling hot path.

Mirrors the reference's public surface (transfusion_pytorch/__init__.py:1-6)."""
from .transfusion import (
    Transfusion,
    Transformer,
    LossBreakdown,
    print_modality_sample,
    create_dataloader,
    exists,
    apply_fn_modality_type,
    stack_same_shape_tensors_with_inverse,
    filter_with_inverse,
    random_modality_length_to_time_fn,
    default_modality_length_to_time_fn,
)

from .ema import EMA

__all__ = ['Transfusion', 'Transformer', 'LossBreakdown', 'print_modality_sample', 'create_dataloader', 'EMA']

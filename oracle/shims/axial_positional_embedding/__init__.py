"""Import-only stub for `axial_positional_embedding` (off by default, out of scope)."""
from torch import nn
class ContinuousAxialPositionalEmbedding(nn.Module):
    def __init__(self, dim, num_axial_dims, **kw):
        super().__init__()
        raise NotImplementedError('axial positional embedding is out of scope for the oracle shims')

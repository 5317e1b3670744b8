"""Stand-in for `torch_einops_utils` (>=0.1.12): layout helpers only
(oracle test infrastructure; semantics per SURVEY.md Appendix D)."""
from functools import wraps
import torch
import torch.nn.functional as F
from torch.utils._pytree import tree_map
from einops import pack, unpack

def pack_with_inverse(t, pattern):
    is_list = isinstance(t, (list, tuple))
    ts = t if is_list else [t]
    packed, ps = pack(ts, pattern)
    def inverse(out, inv_pattern=None):
        outs = unpack(out, ps, inv_pattern if inv_pattern is not None else pattern)
        return outs if is_list else outs[0]
    return packed, inverse

def tree_map_tensor(fn, tree):
    return tree_map(lambda t: fn(t) if torch.is_tensor(t) else t, tree)

def tree_map_tensor_to_device(tree, device):
    return tree_map_tensor(lambda t: t.to(device), tree)

def temp_eval(fn):
    @wraps(fn)
    def inner(self, *args, **kwargs):
        was_training = self.training
        self.eval()
        try:
            return fn(self, *args, **kwargs)
        finally:
            self.train(was_training)
    return inner

def reverse_cumsum(t, dim=-1):
    return t.flip(dims=(dim,)).cumsum(dim=dim).flip(dims=(dim,))

def pad_at_dim(t, pad, dim=-1, value=0.):
    dims_from_right = (-dim - 1) if dim < 0 else (t.ndim - dim - 1)
    zeros = (0, 0) * dims_from_right
    return F.pad(t, (*zeros, *pad), value=value)

def pad_left_at_dim(t, pad, dim=-1, value=0.):
    return pad_at_dim(t, (pad, 0), dim=dim, value=value)

def pad_right_at_dim(t, pad, dim=-1, value=0.):
    return pad_at_dim(t, (0, pad), dim=dim, value=value)

def pad_sequence(tensors, dim=-1, value=0., **kw):
    max_len = max(t.shape[dim] for t in tensors)
    padded = [pad_right_at_dim(t, max_len - t.shape[dim], dim=dim, value=value) for t in tensors]
    return torch.stack(padded)

def batched_index_select(t, idx):
    # t (b, m, ...), idx (b, m) -> gather along dim 1
    extra = t.ndim - idx.ndim
    idx_e = idx.reshape(*idx.shape, *((1,) * extra)).expand(*idx.shape, *t.shape[idx.ndim:])
    return t.gather(1, idx_e)

"""Stand-in for `rotary_embedding_torch` (>=0.8.4), public semantics only
(SURVEY.md Appendix D): freqs = 1/theta^(arange(0,dim,2)/dim) as a frozen
Parameter named `freqs`; forward(t) = t[..., None] * freqs, each frequency
duplicated into ADJACENT slots; apply = t*cos + rotate_half(t)*sin with
interleaved pairs (x1, x2) -> (-x2, x1).  PARITY UNPINNED by any reference test.
"""
import torch
from torch import nn
from einops import rearrange, repeat

def rotate_half(x):
    x = rearrange(x, '... (d r) -> ... d r', r=2)
    x1, x2 = x.unbind(dim=-1)
    x = torch.stack((-x2, x1), dim=-1)
    return rearrange(x, '... d r -> ... (d r)')

def apply_rotary_emb(freqs, t, start_index=0, scale=1., seq_dim=-2, freqs_seq_dim=None):
    dtype = t.dtype
    if freqs_seq_dim is None:
        if freqs.ndim == 2 or t.ndim == 3:
            freqs_seq_dim = 0
    if t.ndim == 3 or freqs_seq_dim is not None:
        seq_len = t.shape[seq_dim]
        fdim = freqs_seq_dim if freqs_seq_dim is not None else 0
        freqs = freqs.narrow(fdim, freqs.shape[fdim] - seq_len, seq_len)
    rot_dim = freqs.shape[-1]
    end_index = start_index + rot_dim
    t_left, t_mid, t_right = t[..., :start_index], t[..., start_index:end_index], t[..., end_index:]
    t_mid = (t_mid * freqs.cos() * scale) + (rotate_half(t_mid) * freqs.sin() * scale)
    return torch.cat((t_left, t_mid, t_right), dim=-1).type(dtype)

class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1. / (theta ** (torch.arange(0, dim, 2)[:(dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def forward(self, t, seq_len=None, offset=0):
        freqs = self.freqs
        freqs = torch.einsum('..., f -> ... f', t.type(freqs.dtype), freqs)
        freqs = repeat(freqs, '... n -> ... (n r)', r=2)
        return freqs

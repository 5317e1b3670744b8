"""Continuous axial positional embedding for modality tokens (`add_pos_emb`, reference T:1384-1403; added to the projected modality
tokens before the transformer, T:2795-2796 / T:3173-3176 via MP:1003-1045).

The reference takes this module from the third-party package `axial_positional_embedding` (pyproject dependency, absent from
/root/reference): one small MLP per axial dimension maps the integer coordinate to a `dim`-vector and the embedding of a position is the
SUM of its per-axis vectors.  This is a restatement of that published behaviour with the call surface the reference uses:

    pos_emb(axial_dims, flatten=True)                 -> (prod(axial_dims), dim)                          T:2795
    pos_emb(max_dims, return_factorized=True)         -> [(max_dim_i, dim)] per axis                       MP:1016
    pos_emb.combine_factorized(factors, dims, flatten=True)                                               MP:1039

PARITY UNPINNED for the MLP's own arithmetic (width, depth, activation, parameter names): the package is not available here, so a
reference checkpoint's `pos_emb_mlp.*` entries are not guaranteed to load.  Everything downstream of the rows it produces (the add into the
token stream, the gradient back into the rows) is the native path and is tested against PyTorch.

The rows are (L, dim) for L = a few hundred at most: the MLP runs in PyTorch (autograd gives its parameter gradients from the row gradients
the engine hands back); the add into the packed token stream and its backward are HIP (engine.Plan `ext_add`).
"""
from __future__ import annotations

import torch
from torch import nn


def _mlp(dim_in: int, dim_out: int, depth: int, expansion: float) -> nn.Sequential:
    hidden = int(expansion * max(dim_in, dim_out))
    layers, cur = [], dim_in
    for _ in range(depth):
        layers += [nn.Linear(cur, hidden), nn.SiLU()]
        cur = hidden
    layers.append(nn.Linear(cur, dim_out))
    return nn.Sequential(*layers)


class ContinuousAxialPositionalEmbedding(nn.Module):
    def __init__(self, dim: int, num_axial_dims: int, mlp_depth: int = 2, mlp_expansion: float = 2.):
        super().__init__()
        self.dim, self.num_axial_dims = dim, num_axial_dims
        self.mlps = nn.ModuleList([_mlp(1, dim, mlp_depth, mlp_expansion) for _ in range(num_axial_dims)])

    @property
    def device(self):
        return next(self.parameters()).device

    def combine_factorized(self, axial_embeds, axial_dims=None, flatten: bool = False):
        if axial_dims is None:
            axial_dims = tuple(e.shape[0] for e in axial_embeds)
        axial_dims = tuple(int(a) for a in axial_dims)
        assert len(axial_dims) == len(axial_embeds)
        out = None
        for e, n in zip(axial_embeds, axial_dims):
            e = e[:n]
            out = e if out is None else out[..., None, :] + e          # (..., n_prev, 1, d) + (n, d): row-major over the axes
        assert tuple(out.shape[:-1]) == axial_dims
        return out.reshape(-1, out.shape[-1]) if flatten else out

    def forward(self, axial_dims, return_factorized: bool = False, flatten: bool = False):
        dims = [int(a) for a in (axial_dims.tolist() if torch.is_tensor(axial_dims) else axial_dims)]
        assert len(dims) == self.num_axial_dims, f'received {len(dims)} axial dimensions, expected {self.num_axial_dims}'
        dev = self.device
        embeds = [mlp(torch.arange(n, device=dev, dtype=torch.float32)[:, None]) for mlp, n in zip(self.mlps, dims)]
        if return_factorized:
            return embeds
        return self.combine_factorized(embeds, flatten=flatten)

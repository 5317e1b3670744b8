"""The reference's smallest training loop (`train_toy.py`: a one-token text prefix + a (2, 16) latent per sample, torch Adam + global-norm clip,
a sample every so often) on the native MI355X path - the only change against the reference script is the import.

    python examples/toy_text_latent.py --steps 200                # torch.optim.Adam over model.parameters(), as in the reference
    python examples/toy_text_latent.py --steps 200 --fused        # transfusion_pytorch_amd.optim.FusedAdam: one clip + Adam launch over the flat buffer
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
from torch.utils.data import DataLoader, Dataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import Transfusion, print_modality_sample          # noqa: E402
from transfusion_pytorch_amd.optim import FusedAdam                             # noqa: E402


class ToyPairs(Dataset):
    """every item: the text token 1, then a latent whose rows are a fixed pattern plus a little noise - something a depth-1 model can fit"""

    def __init__(self, n=128, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.base = torch.randn(2, 16, generator=g)
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        return torch.ones((1,)).long(), self.base + 0.05 * torch.randn(2, 16)


def main(steps=200, batch_size=4, fused=False, sample_every=100, log=print):
    torch.manual_seed(0)
    model = Transfusion(num_text_tokens=8, dim_latent=16, modality_default_shape=(2,),
                        transformer=dict(dim=64, depth=1, dim_head=8, heads=2)).cuda()
    loader = DataLoader(ToyPairs(), batch_size=batch_size, shuffle=True, collate_fn=lambda items: [list(it) for it in items])
    if fused:
        opt = FusedAdam(model, lr=3e-4, max_grad_norm=0.5)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    losses, step = [], 0
    while step < steps:
        for batch in loader:
            step += 1
            loss = model(batch)
            loss.backward()
            if not fused:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            opt.step()
            opt.zero_grad()
            losses.append(float(loss.detach()))
            if step % 20 == 0:
                log(f'{step}: {losses[-1]:.3f}')
            if step % sample_every == 0:
                print_modality_sample(model.sample(max_length=8))
            if step >= steps:
                break
    return losses


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--fused', action='store_true')
    a = ap.parse_args()
    main(steps=a.steps, fused=a.fused)

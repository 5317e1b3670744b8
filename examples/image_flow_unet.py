"""Image-only flow matching with the paper's U-Net style down / up sampling around the transformer - the configuration of the reference's
`train_image_only_with_unet.py`: frozen patchify encoder / decoder (1 x 28 x 28 -> 4 x 14 x 14 channel-first latents), a learnable
Conv2d / ConvTranspose2d pair in place of latent_to_model / model_to_latent, axial positional embedding on the down-sampled grid, an EMA teacher
for the velocity-consistency term, `generate_modality_only` for samples.  There is no network here, so the digits are synthetic (a few Gaussian
strokes per image); everything else is the reference script with the import changed.

    python examples/image_flow_unet.py --steps 300
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transfusion_pytorch_amd import Transfusion          # noqa: E402


class Patchify(nn.Module):                                # (..., 1, 28, 28) in [0, 1] -> (..., 4, 14, 14) in [-1, 1]
    def forward(self, x):
        *lead, _, H, W = x.shape
        x = x.reshape(*lead, H // 2, 2, W // 2, 2).movedim(-3, -4).movedim(-1, -3)         # (..., p1, p2, h, w)
        return x.reshape(*lead, 4, H // 2, W // 2) * 2 - 1


class Unpatchify(nn.Module):
    def forward(self, x):
        *lead, _, h, w = x.shape
        x = x.reshape(*lead, 2, 2, h, w).movedim(-4, -2).movedim(-4, -1)                    # (..., h, p1, w, p2)
        return ((x.reshape(*lead, 1, 2 * h, 2 * w) + 1) * 0.5).clamp(0., 1.)


def synthetic_digits(n, seed=0):
    """n images (n, 1, 28, 28): three soft strokes each, positions drawn from a handful of templates"""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(28.), torch.arange(28.), indexing='ij')
    templates = torch.rand(10, 3, 2, generator=g) * 16 + 6
    which = torch.randint(0, 10, (n,), generator=g)
    centers = templates[which] + torch.randn(n, 3, 2, generator=g) * 0.7
    img = torch.zeros(n, 28, 28)
    for k in range(3):
        cy, cx = centers[:, k, 0, None, None], centers[:, k, 1, None, None]
        img = torch.maximum(img, torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 8.))
    return img[:, None]


def main(steps=300, batch_size=32, log=print):
    torch.manual_seed(0)
    model = Transfusion(
        num_text_tokens=10, dim_latent=4, channel_first_latent=True, modality_default_shape=(14, 14),
        modality_encoder=Patchify(), modality_decoder=Unpatchify(),
        pre_post_transformer_enc_dec=(nn.Conv2d(4, 64, 3, 2, 1), nn.ConvTranspose2d(64, 4, 3, 2, 1, output_padding=1)),
        add_pos_emb=True, modality_num_dim=2, velocity_consistency_loss_weight=0.1,
        transformer=dict(dim=64, depth=4, dim_head=32, heads=8)).cuda()
    ema_model = model.create_ema()
    data = synthetic_digits(2048).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=8e-4)
    losses = []
    for step in range(1, steps + 1):
        batch = data[torch.randint(0, data.shape[0], (batch_size,), device=data.device)]
        loss = model(batch, velocity_consistency_ema_model=ema_model)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        opt.zero_grad()
        ema_model.update()
        losses.append(float(loss.detach()))
        if step % 50 == 0:
            log(f'{step}: {losses[-1]:.3f}')
    images = ema_model.generate_modality_only(batch_size=4)
    return losses, images


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    a = ap.parse_args()
    losses, images = main(steps=a.steps)
    print('generated', tuple(images.shape), 'range', float(images.min()), float(images.max()))

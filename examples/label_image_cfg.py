"""Interleaved text + image training and guided sampling - the configuration of the reference's `train_mnist.py`: a class-label token followed
by (or following) a 28 x 28 image, frozen patchify encoder / decoder to 14 x 14 x 4 latents (channel-first), axial positional embedding,
classifier-free-guidance text drop during training (`prob_uncond`), an EMA copy that does the sampling (`ema_model.sample(prompt=..., cfg_scale=3)`).
No network here: the "digits" are ten synthetic stroke templates (examples/image_flow_unet.py), the label is the template index.

    python examples/label_image_cfg.py --steps 300
"""
from __future__ import annotations

import argparse
import os
import sys

import torch
from torch.utils.data import Dataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from transfusion_pytorch_amd import Transfusion, print_modality_sample          # noqa: E402
from image_flow_unet import Patchify, Unpatchify                          # noqa: E402


class LabelledStrokes(Dataset):
    def __init__(self, n=2048, seed=0, image_after_text=True):
        g = torch.Generator().manual_seed(seed)
        ys, xs = torch.meshgrid(torch.arange(28.), torch.arange(28.), indexing='ij')
        templates = torch.rand(10, 3, 2, generator=g) * 16 + 6
        self.labels = torch.randint(0, 10, (n,), generator=g)
        centers = templates[self.labels] + torch.randn(n, 3, 2, generator=g) * 0.7
        img = torch.zeros(n, 28, 28)
        for k in range(3):
            cy, cx = centers[:, k, 0, None, None], centers[:, k, 1, None, None]
            img = torch.maximum(img, torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 8.))
        self.images, self.image_after_text = img[:, None], image_after_text

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, idx):
        pair = (self.labels[idx].clone(), self.images[idx])          # a 0-dim label tensor, as torchvision's MNIST gives
        return pair if self.image_after_text else pair[::-1]


def main(steps=300, batch_size=16, sample=True, log=print, fallback_shape=False, unet=False):
    """fallback_shape: an under-trained model may spell a malformed shape string; True falls back to `modality_default_shape` (T:1636-1640).
    unet: the configuration of the reference's `train_mnist_with_unet.py` - a stride-2 conv / transposed-conv pair around the transformer (49 tokens
    per 14 x 14 latent image); `sample()` then decodes through the un-cached loop, guidance included."""
    torch.manual_seed(0)
    extra = dict(pre_post_transformer_enc_dec=(torch.nn.Conv2d(4, 64, 3, 2, 1), torch.nn.ConvTranspose2d(64, 4, 3, 2, 1, output_padding=1))) if unet else {}
    model = Transfusion(num_text_tokens=10, dim_latent=4, modality_default_shape=(14, 14), modality_encoder=Patchify(), modality_decoder=Unpatchify(),
                        add_pos_emb=True, modality_num_dim=2, prob_uncond=0.1, channel_first_latent=True, fallback_to_default_shape_if_invalid=fallback_shape,
                        transformer=dict(dim=64, depth=4, dim_head=32, heads=8), **extra).cuda()
    ema_model = model.create_ema()
    loader = model.create_dataloader(LabelledStrokes(), batch_size=batch_size, shuffle=True)
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    losses, step = [], 0
    while step < steps:
        for batch in loader:
            step += 1
            model.train()
            loss = model(batch)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
            opt.step()
            opt.zero_grad()
            ema_model.update()
            losses.append(float(loss.detach()))
            if step % 50 == 0:
                log(f'{step}: {losses[-1]:.3f}')
            if step >= steps:
                break
    out = None
    if sample:
        label = torch.randint(0, 10, ()).cuda()
        out = ema_model.sample(prompt=label, max_length=384, cfg_scale=3.0)
        print_modality_sample(out)
    return losses, out


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--unet', action='store_true')
    a = ap.parse_args()
    main(steps=a.steps, unet=a.unet)

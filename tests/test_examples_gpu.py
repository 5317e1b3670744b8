"""The example scripts under examples/ are the reference's training scripts (train_toy.py, train_image_only_with_unet.py, train_mnist.py) with the
import changed and synthetic data: torch.optim.Adam over `model.parameters()` + `clip_grad_norm_`, EMA, velocity consistency, guided sampling -
the CALLERS of the hot path.  A few dozen steps each: the loss must fall and the samplers must return the reference's output structure."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples'))


def falling(losses, k=10):
    first, last = sum(losses[:k]) / k, sum(losses[-k:]) / k
    print(f'  loss: first {k} steps {first:.3f} -> last {k} steps {last:.3f}')
    return last < 0.9 * first


@pytest.mark.parametrize('fused', [False, True])
def test_train_toy_loop(fused):
    import toy_text_latent as train_toy
    losses = train_toy.main(steps=120, fused=fused, sample_every=60, log=lambda *a: None)
    assert len(losses) == 120 and all(l == l for l in losses)
    assert falling(losses)


def test_image_flow_with_unet_ema_teacher_and_generation():
    import image_flow_unet as ex
    losses, images = ex.main(steps=80, log=lambda *a: None)
    assert falling(losses)
    assert images.shape == (4, 1, 28, 28) and float(images.min()) >= 0. and float(images.max()) <= 1.


@pytest.mark.parametrize('unet', [False, True])       # True: train_mnist_with_unet.py (conv pair around the transformer; sample() through the un-cached loop)
def test_text_image_interleaved_with_guided_sampling(unet):
    import label_image_cfg as ex
    losses, out = ex.main(steps=80, log=lambda *a: None, fallback_shape=True, unet=unet)
    assert falling(losses)
    # the prompt label, then - if the model opened a modality - a decoded (1, 28, 28) image (T:2581-2583 decodes unless asked not to)
    assert torch.is_tensor(out[0]) and out[0].dtype == torch.long
    for part in out:
        if isinstance(part, tuple):
            assert part[1].shape == (1, 28, 28) and 0. <= float(part[1].min()) and float(part[1].max()) <= 1.

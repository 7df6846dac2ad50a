"""GPU input pipeline (resize -> ToTensor -> *2-1 -> channel replication, optional flip) against the reference's CPU
transforms restated with PIL: bit-exact (integer resampling arithmetic, one float division and one fused multiply-add-free
rescale per pixel)."""
import numpy as np
import pytest
import torch

from dm_amd.datasets import GpuImageTransform
from oracle.input_pipeline import reference_transform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h0,w0,c0,size,channels", [(28, 28, 1, 32, 3),          # MNIST / FashionMNIST
                                                     (48, 48, 1, 32, 3),          # FER: downscale (wider filter support)
                                                     (500, 375, 3, 64, 3),        # Flowers102-shaped: non-square RGB
                                                     (64, 64, 3, 64, 3),          # already the right size: no resampling
                                                     (28, 28, 1, 32, 1),          # single-channel model input
                                                     (40, 56, 3, (32, 48), 3)])   # non-square target
def test_matches_reference_transforms(dev, h0, w0, c0, size, channels):
    rng = np.random.default_rng(h0 * 1000 + w0)
    n = 5
    imgs = rng.integers(0, 256, (n, h0, w0, c0), dtype=np.uint8)
    flips = [False, True, False, True, True]
    ref = torch.stack([reference_transform(imgs[i, :, :, 0] if c0 == 1 else imgs[i], size, channels, flips[i]) for i in range(n)])
    tf = GpuImageTransform(size, channels)
    x = torch.from_numpy(imgs if c0 > 1 else imgs[..., 0]).to(dev)
    got = tf(x, flip=torch.tensor(flips, device=dev))
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert torch.equal(got.cpu(), ref)
    # no flips requested: equals the un-flipped reference
    got0 = tf(x)
    ref0 = torch.stack([reference_transform(imgs[i, :, :, 0] if c0 == 1 else imgs[i], size, channels) for i in range(n)])
    assert torch.equal(got0.cpu(), ref0)


def test_random_flip_rate_and_cpu_rejection(dev):
    tf = GpuImageTransform(32, 3, flip_p=0.5)
    x = torch.arange(32, dtype=torch.uint8).repeat(4000, 32, 1).to(dev)      # every row is a ramp: a flip reverses it
    out = tf(x)
    flipped = (out[:, 0, 0, 0] > out[:, 0, 0, -1]).float().mean().item()
    assert 0.45 < flipped < 0.55
    with pytest.raises(RuntimeError):
        tf(torch.zeros(2, 28, 28, dtype=torch.uint8))

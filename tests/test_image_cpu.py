"""On-device image transform (visualrwkv_amd/image.py) against PIL's bicubic resize + the reference's normalisation."""
import numpy as np
import pytest
import torch
from PIL import Image


@pytest.mark.parametrize("hw", [(600, 800), (300, 210), (1400, 1000)])
def test_resize_normalize_matches_pil(hw):
    from visualrwkv_amd import image
    rng = np.random.default_rng(hw[0])
    # a smooth image plus noise: pure noise would make every resampling kernel difference look large
    yy, xx = np.mgrid[0:hw[0], 0:hw[1]]
    base = np.stack([127 + 100 * np.sin(xx / 37.0 + c) * np.cos(yy / 53.0) for c in range(3)], axis=-1)
    img = np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)
    for tower, (size, mean, std) in image.TOWER_SPECS.items():
        ref = np.asarray(Image.fromarray(img).resize((size, size), Image.BICUBIC)).astype(np.float32) / 255.0
        ref = (ref - np.array(mean, dtype=np.float32)) / np.array(std, dtype=np.float32)
        got = image.resize_normalize(torch.from_numpy(img), size, mean, std)[0].permute(1, 2, 0).numpy()
        err = np.abs(got - ref)
        # PIL rounds its output to 8 bits (half a level = 0.002 / std) and uses fixed-point filter weights
        assert err.mean() < 0.01 / min(std) and err.max() < 0.06 / min(std), (tower, err.mean(), err.max())


def test_process_images_shapes_and_dtype():
    from visualrwkv_amd import image
    imgs = [torch.randint(0, 256, (50, 70, 3), dtype=torch.uint8), torch.randint(0, 256, (90, 40, 3), dtype=torch.uint8)]
    out = image.process_images(imgs, towers=("dino", "siglip", "sam"))
    assert out["dino"].shape == (2, 3, 448, 448) and out["siglip"].shape == (2, 3, 448, 448) and out["sam"].shape == (2, 3, 1024, 1024)
    assert all(v.dtype == torch.bfloat16 for v in out.values())
    assert float(out["siglip"].float().abs().max()) <= 1.0 + 1e-2

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


@pytest.mark.parametrize("hw,size", [((37, 53), 64), ((300, 200), 64), ((64, 64), 64), ((20, 31), 96)])
def test_hip_resize_kernel_on_the_emulator(hw, size):
    """csrc/image_kernels.h (window / weight tables / accumulation) run on the host emulator against the torch statement
    (F.interpolate bicubic antialias + clip + normalise): down-sampling, up-sampling, identity, mixed."""
    import ctypes

    from tests.emu.build import build_emu
    from visualrwkv_amd import image
    lib = ctypes.CDLL(build_emu())
    g = torch.Generator().manual_seed(hw[0])
    img = torch.randint(0, 256, (*hw, 3), dtype=torch.uint8, generator=g)
    out = torch.zeros(3, size, size)
    m3, s3 = (ctypes.c_float * 3)(*image.IMAGENET_MEAN), (ctypes.c_float * 3)(*image.IMAGENET_STD)
    assert lib.emu_resize_normalize_u8(hw[0], hw[1], ctypes.c_void_p(img.data_ptr()), size, m3, s3, ctypes.c_void_p(out.data_ptr()), 1) == 0
    ref = image.resize_normalize(img, size, image.IMAGENET_MEAN, image.IMAGENET_STD)[0]
    assert (out - ref).abs().max() < 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(480, 640), (1365, 2048), (100, 75), (448, 448)])
def test_hip_resize_kernel_on_the_gpu(hw):
    """The HIP transform against the torch statement evaluated on the CPU, for the three towers, fp32 and bf16 outputs."""
    from visualrwkv_amd import image
    g = torch.Generator().manual_seed(hw[1])
    img = torch.randint(0, 256, (*hw, 3), dtype=torch.uint8, generator=g)
    for t, (size, mean, std) in image.TOWER_SPECS.items():
        ref = image.resize_normalize(img, size, mean, std)                  # CPU: F.interpolate path
        out = image.resize_normalize(img.cuda(), size, mean, std)           # GPU: HIP kernel
        assert out.is_cuda and out.shape == ref.shape
        assert (out.cpu() - ref).abs().max() < 5e-4, t
        ob = image.resize_normalize(img.cuda(), size, mean, std, dtype=torch.bfloat16)
        assert ob.dtype == torch.bfloat16 and (ob.float().cpu() - ref).abs().max() < 2e-2

"""On-device image transform for the three towers: what the reference does per image on the CPU
with PIL / torchvision in its single dataloader worker (VisualRWKV-v7/v7.00/src/vision.py:96-121: `Resize((S,S),
bicubic)`, `ToTensor`, `Normalize(mean, std)` with the timm data configs -- DINOv2 and SAM: ImageNet mean/std, SigLIP:
0.5/0.5; S = 448, 448, 1024), done on the GPU on a batch of decoded uint8 images: one antialiased bicubic resample per
target size and a fused scale/shift.  Resampling follows PIL's convention (antialias on down-sampling, support scaled
with the ratio); it matches `PIL.Image.resize(..., BICUBIC)` to within the 8-bit rounding PIL applies to its output
(tests/test_image_cpu.py), it is not bit-identical.  timm / torchvision are not vendored by the reference and not
installed here, so the transform's constants are restated from the timm model configs named in src/vision.py:52-53."""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)     # vit_large_patch14_reg4_dinov2.lvd142m
SIGLIP_MEAN, SIGLIP_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)                      # vit_so400m_patch14_siglip_384
TOWER_SPECS = {"dino": (448, IMAGENET_MEAN, IMAGENET_STD), "siglip": (448, SIGLIP_MEAN, SIGLIP_STD),
               "sam": (1024, IMAGENET_MEAN, IMAGENET_STD)}                       # SAM reuses the DINOv2 transform (vision.py:114-119)


def _hip_resize_normalize(img_u8: torch.Tensor, size: int, mean, std, dtype) -> torch.Tensor:
    """One (H,W,3) uint8 CUDA image through vrwkv_resize_normalize_u8 (csrc/image_kernels.h)."""
    import ctypes
    from . import hip_lib
    img = img_u8.contiguous()
    H, W, _ = img.shape
    out = torch.empty(1, 3, size, size, dtype=dtype, device=img.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    rc = hip_lib.load().vrwkv_resize_normalize_u8(H, W, img.data_ptr(), size, m3, s3, out.data_ptr(), int(dtype == torch.float32),
                                                  hip_lib.launch_stream(img.device))
    hip_lib.check(rc, "vrwkv_resize_normalize_u8")
    return out


def resize_normalize(img_u8: torch.Tensor, size: int, mean: Sequence[float], std: Sequence[float],
                     dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """img_u8: (H,W,3) or (N,H,W,3) uint8 -> (N,3,size,size) normalised.  The aspect ratio is not kept (Resize((S,S))).
    On the GPU: one HIP kernel per image (resample + clip + normalise + layout); elsewhere the same arithmetic in torch."""
    if img_u8.is_cuda and img_u8.dtype == torch.uint8 and dtype in (torch.float32, torch.bfloat16):
        imgs = [img_u8] if img_u8.dim() == 3 else list(img_u8)
        return torch.cat([_hip_resize_normalize(im, size, mean, std, dtype) for im in imgs], dim=0)
    x = img_u8 if img_u8.dim() == 4 else img_u8.unsqueeze(0)
    x = x.permute(0, 3, 1, 2).float()
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=False, antialias=True)
    x = x.clamp_(0.0, 255.0)                                    # bicubic overshoot; PIL clips to the 8-bit range
    m = torch.tensor(mean, device=x.device).view(1, 3, 1, 1) * 255.0
    s = torch.tensor(std, device=x.device).view(1, 3, 1, 1) * 255.0
    return ((x - m) / s).to(dtype)


def process_images(imgs_u8: Sequence[torch.Tensor], towers: Sequence[str] = ("dino", "siglip", "sam"),
                   dtype: torch.dtype = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """One sample's images (decoded, uint8 HWC, any sizes, already on the target device) -> the `images` dict of a batch
    item: tower -> (n_images,3,S,S), as `SamDinoSigLIPImageTransform.__call__` + the stacking of dataset.py:207-217."""
    out = {}
    for t in towers:
        size, mean, std = TOWER_SPECS[t]
        out[t] = torch.cat([resize_normalize(im, size, mean, std, dtype) for im in imgs_u8], dim=0)
    return out

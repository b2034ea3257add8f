"""Tower image transform (three resizes: 448, 448, 1024 + normalise) of one decoded photo: HIP kernel vs the torch-ops
statement on the GPU vs PIL on one CPU core (what the reference's single DataLoader worker does).  One JSON line per size."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualrwkv_amd import image

def gpu_time(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def torch_ops(img):
    out = {}
    for t, (size, mean, std) in image.TOWER_SPECS.items():
        x = img.unsqueeze(0).permute(0, 3, 1, 2).float()
        x = torch.nn.functional.interpolate(x, size=(size, size), mode="bicubic", align_corners=False, antialias=True).clamp_(0, 255)
        m = torch.tensor(mean, device=x.device).view(1, 3, 1, 1) * 255.0
        s = torch.tensor(std, device=x.device).view(1, 3, 1, 1) * 255.0
        out[t] = ((x - m) / s).bfloat16()
    return out

def pil_cpu(arr):
    from PIL import Image
    im = Image.fromarray(arr)
    out = {}
    for t, (size, mean, std) in image.TOWER_SPECS.items():
        r = np.asarray(im.resize((size, size), Image.BICUBIC), dtype=np.float32) / 255.0
        out[t] = torch.from_numpy(((r - np.array(mean, dtype=np.float32)) / np.array(std, dtype=np.float32)).transpose(2, 0, 1).copy())
    return out

torch.set_num_threads(1)
for hw in [(480, 640), (1365, 2048), (3000, 4000)]:
    arr = np.random.default_rng(0).integers(0, 256, (*hw, 3), dtype=np.uint8)
    img = torch.from_numpy(arr).cuda()
    t_hip = gpu_time(lambda: image.process_images([img], ("dino", "siglip", "sam"), torch.bfloat16))
    t_ops = gpu_time(lambda: torch_ops(img))
    t0 = time.perf_counter(); n = 3
    for _ in range(n): pil_cpu(arr)
    t_pil = (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"image_hw": hw, "hip_ms": round(t_hip, 4), "torch_ops_gpu_ms": round(t_ops, 4), "pil_one_core_ms": round(t_pil, 2),
                      "hip_images_per_s": round(1e3 / t_hip)}), flush=True)

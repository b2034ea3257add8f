"""Re-tune the step's LARGE library GEMMs under sustained load: PyTorch TunableOp with long per-kernel durations (default 150 ms warm-up + 250 ms timing per
candidate instead of tune_gemms.sh's 5 + 20 ms), so that every candidate is timed at the clock the power-limited chip actually sustains, not at the boost
clock of a 20 ms burst.  Writes a TunableOp result file with the entries of these shapes only; merge by hand.
    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/retune.csv \
    PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=250 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=150 python benchmarks/retune_big_gemms.py"""
import torch
import torch.nn.functional as F

M, C = 16 * 2624, 2048
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.3).bfloat16()
x, x4 = r(M, C), r(M, 4 * C)
w_cc, w_c4, w_4c = r(C, C), r(4 * C, C), r(C, 4 * C)
for _ in range(2):
    F.linear(x, w_cc)            # tn_2048_41984_2048   r k v o forward
    x @ w_cc                     # nn_2048_41984_2048   their input gradients (T,N layout on a transposed weight: fused._LinearTN issues F.linear; this is autograd's form)
    F.linear(x, w_c4)            # tn_8192_41984_2048   channel-mix key
    F.linear(x4, w_4c)           # tn_2048_41984_8192   channel-mix value
    x4 @ w_c4                    # nn_2048_41984_8192
    x @ w_4c                     # nn_8192_41984_2048
torch.cuda.synchronize()
print("done")

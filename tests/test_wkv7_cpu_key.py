"""The CPU dispatch key of torch.ops.wind_backstepping (csrc/wkv7_host.hip; BASELINE config 1 "fp32 CPU WKV path") against
the oracle: the C restatement of the reference kernels for the bf16 contract, fp64 autograd through the naive recurrence
for float32 -- on the config-1 shape (1, 384, 12, 64) and a ragged batch."""
import pytest
import torch

from oracle import wkv7_c
from oracle.wkv7_oracle import bf16_round, make_inputs, rel_rms, wkv7_autograd
from visualrwkv_amd.wkv7 import RUN_CUDA_RWKV7g


def _run(leaves_btHC, dy):
    y = RUN_CUDA_RWKV7g(*leaves_btHC)
    y.backward(dy)
    return y.detach(), [l.grad for l in leaves_btHC]


@pytest.mark.parametrize("B,T,H", [(1, 384, 12), (3, 48, 2)])
def test_bf16_contract_matches_the_c_oracle(B, T, H):
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=5 + T)
    yr, sr, sar = wkv7_c.forward(w, q, k, v, z, a)
    ref = wkv7_c.backward(w, q, k, v, z, a, dy, sr, sar)
    leaves = [x.clone().view(B, T, H * 64).requires_grad_(True) for x in (q, w, k, v, z, a)]      # RUN_CUDA_RWKV7g(q,w,k,v,a,b)
    y, (dq, dw, dk, dv, dz, da) = _run(leaves, dy.view(B, T, H * 64))
    assert y.dtype == torch.bfloat16
    assert rel_rms(y.float().view(B, T, H, 64), yr.float()) < 1e-3
    for n, o, r in zip(["dw", "dq", "dk", "dv", "dz", "da"], (dw, dq, dk, dv, dz, da), ref):
        assert rel_rms(o.float().view(B, T, H, 64), r.float()) < 1e-3, n
    # the saved by-products are the op's: S^T at the chunk ends and sa, fp32
    s = torch.empty(B, H, T // 16, 64, 64)
    sa = torch.empty(B, T, H, 64)
    torch.ops.wind_backstepping.forward(w, q, k, v, z, a, torch.empty_like(v), s, sa)
    assert rel_rms(s, sr) < 2e-5 and rel_rms(sa, sar) < 2e-5


def test_float32_mode_matches_fp64_autograd_on_the_config1_shape():
    B, T, H = 1, 384, 12
    w, q, k, v, z, a, dy = [x.float() for x in make_inputs(B, T, H, seed=42)]
    yt, gt = wkv7_autograd(w, q, k, v, z, a, dy)                                   # fp64 truth, (w,q,k,v,z,a) order
    leaves = [x.clone().view(B, T, H * 64).requires_grad_(True) for x in (q, w, k, v, z, a)]
    y, (dq, dw, dk, dv, dz, da) = _run(leaves, dy.view(B, T, H * 64))
    assert y.dtype == torch.float32
    assert rel_rms(y.view(B, T, H, 64), yt.float()) < 1e-5
    for n, o, r in zip(["dw", "dq", "dk", "dv", "dz", "da"], (dw, dq, dk, dv, dz, da), gt):
        assert rel_rms(o.view(B, T, H, 64), r.float()) < 2e-5, n


def test_threads_do_not_change_the_result(monkeypatch):
    import visualrwkv_amd.wkv7 as wk
    B, T, H = 2, 32, 3
    w, q, k, v, z, a, dy = make_inputs(B, T, H, seed=9)
    outs = []
    for nt in (1, 4):
        monkeypatch.setattr(wk, "HOST_THREADS", nt)
        leaves = [x.clone().view(B, T, H * 64).requires_grad_(True) for x in (q, w, k, v, z, a)]
        y, grads = _run(leaves, dy.view(B, T, H * 64))
        outs.append([y] + grads)
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_the_host_pool_survives_a_fork():
    """The persistent host thread pool of csrc/wkv7_host.hip after os.fork(): the child inherits a pool whose worker threads do not
    exist there (the call blocked forever).  A pool abandons the parent's state when the pid changes."""
    import os
    import signal
    B, T, H = 2, 32, 4
    w, q, k, v, z, a, _ = make_inputs(B, T, H, seed=3)

    def fwd():
        y, s, sa = torch.empty_like(v), torch.empty(B, H, T // 16, 64, 64), torch.empty(B, T, H, 64)
        torch.ops.wind_backstepping.forward(w, q, k, v, z, a, y, s, sa)
        return y

    y0 = fwd()                                       # the parent's pool exists now (B * H = 8 tasks: more than one thread)
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:                                     # child: must neither hang nor differ
        try:
            signal.alarm(20)
            ok = torch.equal(fwd(), y0) and torch.equal(fwd(), y0)
            os.write(wr, b"1" if ok else b"0")
        finally:
            os._exit(0)
    os.close(wr)
    _, status = os.waitpid(pid, 0)
    assert os.read(rd, 1) == b"1", f"forked child did not finish the CPU op (wait status {status})"
    assert torch.equal(fwd(), y0)                    # the parent's pool is untouched

"""Host logic of stateful generation (RWKV7State / forward_stateful) on CPU: the WKV kernels are replaced by the
oracle's plain recurrence (tests may use the oracle; the product path has no CPU implementation)."""
from types import SimpleNamespace

import pytest
import torch

from oracle.wkv7_oracle import rel_rms, wkv7_naive


def lm_args(**kw):
    d = dict(n_embd=128, n_layer=3, dim_att=128, head_size_a=64, head_size_divisor=8, vocab_size=300, dropout=0,
             grad_cp=0, ctx_len=64, load_model="", num_token_per_image=16, proj_type="mlp")
    d.update(kw)
    return SimpleNamespace(**d)


@pytest.fixture()
def oracle_kernels(monkeypatch):
    from visualrwkv_amd import rwkv7, wkv7
    calls = {"prefill": 0, "step": 0}

    def full(q, w, k, v, a, b):                                  # RUN_CUDA_RWKV7g's argument order
        B, T, HC = q.shape
        ops = [i.view(B, T, HC // 64, 64) for i in (w, q, k, v, a, b)]
        return wkv7_naive(*ops)[0].reshape(B, T, HC)

    def prefill(w, q, k, v, z, a, state0=None, segments=None):
        calls["prefill"] += 1
        assert w.shape[1] % 16 == 0
        return wkv7_naive(w, q, k, v, z, a, state0=state0)

    def step(w, q, k, v, z, a, state):
        calls["step"] += 1
        y, s = wkv7_naive(*[i.unsqueeze(1) for i in (w, q, k, v, z, a)], state0=state)
        state.copy_(s)
        return y[:, 0]

    class F64State(rwkv7.RWKV7State):            # the product keeps S in fp32 (kernel contract); exact check in fp64
        def __init__(self, *a, **kw):
            super().__init__(*a, **kw)
            self.S = [s.double() for s in self.S]

    monkeypatch.setattr(rwkv7, "RWKV7State", F64State)
    monkeypatch.setattr(rwkv7, "RUN_CUDA_RWKV7g", full)
    monkeypatch.setattr(wkv7, "wkv7_forward_tparallel", prefill)
    monkeypatch.setattr(wkv7, "wkv7_step", step)
    return calls


def _model():
    from visualrwkv_amd.rwkv7 import RWKV
    torch.manual_seed(3)
    m = RWKV(lm_args()).double()
    with torch.no_grad():                        # zero-initialised projections would hide the state's effect
        for b in m.blocks:
            b.att.output.weight.normal_(0, 0.05)
            b.ffn.value.weight.normal_(0, 0.05)
    return m.eval()


@pytest.mark.parametrize("splits", [[48], [21, 1, 1, 9, 16], [5, 43], [16, 32], [1] * 20 + [28]])
def test_stateful_equals_full_forward(oracle_kernels, splits):
    m = _model()
    x = torch.randn(2, 48, 128, dtype=torch.float64)
    with torch.no_grad():
        full = m(x)
    state, outs = None, []
    for n in splits:
        p0 = sum(o.size(1) for o in outs)
        o, state = m.forward_stateful(x[:, p0:p0 + n], state)
        outs.append(o)
    got = torch.cat(outs, dim=1)
    assert state.n_tokens == 48
    assert rel_rms(got, full) < 1e-12
    # whole chunks of every call go through the chunked kernel (continuing from the carried state), the rest is stepped
    assert oracle_kernels["prefill"] == 3 * sum(1 for n in splits if n >= 16)
    assert oracle_kernels["step"] == 3 * sum(n % 16 for n in splits)


def test_last_only_and_state_isolation(oracle_kernels):
    m = _model()
    x = torch.randn(1, 20, 128, dtype=torch.float64)
    lo, st = m.forward_stateful(x, None, last_only=True)
    full, _ = m.forward_stateful(x, None)
    assert lo.shape == (1, 300) and rel_rms(lo, full[:, -1]) < 1e-13
    # continuing from a state must not depend on tokens fed to another state
    a, _ = m.forward_stateful(x[:, :1], st)
    _, st2 = m.forward_stateful(x, None)
    b, _ = m.forward_stateful(x[:, :1], st2)
    assert torch.equal(a, b)


def test_generate_stateful_vs_generate(oracle_kernels):
    """Same left-padded prompt => the first generated token / logit / prob equal `generate`'s.  With a prompt
    of 16k+15 tokens the second step of `generate` needs no padding at all while the stateful path keeps its one
    pad token, so later tokens are only required to be a valid greedy continuation of the stateful prefix."""
    from visualrwkv_amd.visual import VisualRWKV, IMAGE_TOKEN_INDEX
    args = lm_args(vocab_size=65536, ctx_len=128, vision_towers=("dino",), vision_image_size=56,
                   vision_tower_kwargs={"dino": dict(depth=1, dim=64, heads=1)})
    torch.manual_seed(0)
    m = VisualRWKV(args).double().eval()
    with torch.no_grad():
        for b in m.rwkv.blocks:
            b.att.output.weight.normal_(0, 0.05)
            b.ffn.value.weight.normal_(0, 0.05)
    ids = torch.randint(0, 256, (1, 27))
    ids[0, 3:19] = IMAGE_TOKEN_INDEX
    images = {"dino": torch.randn(1, 3, 56, 56, dtype=torch.float64)}
    ref = m.generate(ids, images, False, 1.0, 1.0, 1, stop_token_idx=-7)
    got = m.generate_stateful(ids, images, False, 1.0, 1.0, 6, stop_token_idx=-7)
    assert got[0][0] == ref[0][0]
    assert got[1][0] == pytest.approx(ref[1][0], rel=1e-9) and got[2][0] == pytest.approx(ref[2][0], rel=1e-9)
    assert len(got[0]) == 6
    # the stateful tokens are the greedy continuation of the fixed left-padded prefix
    samples = {"input_ids": ids, "images": images, "labels": ids}
    x, _ = m.preparing_embedding(samples)
    x = m.rwkv.pad_left(x, 5)
    for t in got[0][:-1]:
        x = torch.cat((x, m.rwkv.emb(torch.tensor([[t]]))), dim=1)
    logits, _ = m.rwkv.forward_stateful(x, None)
    greedy = logits[0, 31:].argmax(-1).tolist()
    assert greedy == got[0]

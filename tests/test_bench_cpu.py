"""bench.py pieces that do not need a GPU: argument object, synthetic batch schema, and that importing the
benchmark does not touch the device or the oracle."""
import importlib
import sys

import torch


def test_bench_imports_without_gpu_and_without_oracle():
    for k in [k for k in sys.modules if k == "oracle" or k.startswith("oracle.")]:
        del sys.modules[k]
    bench = importlib.import_module("bench")
    assert not any(k == "oracle" or k.startswith("oracle.") for k in sys.modules), "bench must only use the oracle in cpu_baseline()"
    args = bench.build_args("1b5", 2624, 576, ("dino", "siglip"), 0, True)
    assert (args.n_layer, args.n_embd, args.vocab_size, args.ctx_len) == (24, 2048, 65536, 2624)
    assert bench.FWD_B == 34 and bench.BWD_B == 46 and bench.HBM_PEAK_GBPS == 8000.0


def test_synthetic_batch_schema():
    import bench
    b = bench.synthetic_batch(2, 64, 16, ("dino", "siglip", "sam"), torch.device("cpu"), seed=3)
    ids, lab = b["input_ids"], b["labels"]
    assert ids.shape == (2, 64) and lab.shape == (2, 64)
    assert bool((ids[:, 4:20] == 65535).all()) and int((ids == 65535).sum()) == 32
    assert bool((lab[:, :38] == -100).all()) and bool((lab[ids == 65535] == -100).all())
    assert torch.equal(lab[:, 38:], ids[:, 38:])
    assert b["images"]["dino"].shape == (2, 3, 448, 448) and b["images"]["sam"].shape == (2, 3, 1024, 1024)
    assert b["images"]["siglip"].dtype == torch.bfloat16 and len(b["sample_id"]) == 2

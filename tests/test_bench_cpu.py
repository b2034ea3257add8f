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


def test_bench_starts_its_own_ranks_two_gloo_ranks_on_host_cores():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py re-executes itself under
    torch.distributed.run, both ranks join the collective, rank 0 prints the one JSON line (the tiny fp32 model through the
    op's CPU key and the ZeRO-1 engine on gloo -- the launch path of the 8-GPU run, train.py:75-76,98)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--model", "tiny",
                          "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen_by_collective"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["global_batch"] == 4 and rec["value"] > 0 and 5.0 < rec["config"]["loss"] < 12.5
    # the N > 1 self-description of the line: what one rank processed, which collectives ran, and that `value` is the whole job's
    pr = rec["per_rank"]
    assert pr["ranks"] == 2 and pr["micro_bsz"] == 2 and pr["tokens_per_step"] == 2 * rec["config"]["seq_len"]
    assert abs(rec["value"] - 2 * pr["tokens_per_step"] * rec["steps"] / (rec["ms_per_step"] * 1e-3 * rec["steps"])) < 1e-6 * rec["value"]
    assert "reduce-scatter" in pr["data_path_collectives"] and "comm" in rec          # comm: stream-event timeline, GPU runs only
    assert "1 = the reference's --grad_cp 1" in rec["config"]["grad_cp_meaning"]


def test_baseline_config_1_end_to_end_on_host_cores():
    """BASELINE.json configs[0]: RWKV-x070 0.1B, fp32 CPU WKV path, one 224 x 224 dummy image (SigLIP-so400m/14: 256 patch tokens) +
    128 text tokens, batch 1 -- one full training step (tower, projector, 12 blocks through the wind_backstepping op's CPU key, loss,
    backward, ZeRO-1 AdamW) through the benchmark's own entry point.  Random-init weights: the first loss is ln 65536."""
    import json
    import math
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--backend", "gloo", "--model", "0b1", "--ctx-len", "384",
                          "--img-tokens", "256", "--towers", "siglip", "--image-size", "224", "--micro-bsz", "1", "--steps", "1",
                          "--warmup", "0", "--fast-init"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["dtype"] == "fp32" and line["config"]["seq_len"] == 384 and line["config"]["global_batch"] == 1
    assert "256 img + 128 text" in line["config"]["workload"] and "siglip" in line["config"]["workload"]
    assert abs(line["config"]["loss"] - math.log(65536)) < 0.5 and line["value"] > 0      # 11.09 +- the spread of a random head

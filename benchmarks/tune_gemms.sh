#!/bin/bash
# Produce visualrwkv_amd/tuning/tunableop_gfx950_1b5_mb16.csv on an MI355X box (about 9 minutes):
# PyTorch TunableOp times every hipBLASLt / rocBLAS kernel for each GEMM shape of one benchmark step.
#   bash benchmarks/tune_gemms.sh [micro_bsz]
MB=${1:-16}; OUT=$PWD/gpurun_out/tunableop_mb${MB}
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$OUT.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --micro-bsz $MB --gemm-tuning 0
ls -la ${OUT}*.csv    # PyTorch appends the device ordinal: copy ${OUT}0.csv to visualrwkv_amd/tuning/

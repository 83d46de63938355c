#!/bin/bash
# Round profiles of MW_PRECISION_FP8_MX (run on the GPU box from the repository root; tools/collect_profiles.sh holds the
# bf16 / e4m3 ones): encoder per window and per kernel, its matrix-core counters at 16 windows, the bench lines (headline
# workload, 64 and 8 turbo streams), the streaming latency, the MX GEMM microbenchmark and the operand-map probe.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof_mx
mkdir -p $OUT
timeout -k 10 60 tools/microbench/mb_mxfp8_probe > $OUT/mxfp8_probe.txt 2>&1
timeout -k 10 240 tools/microbench/mb_gemm_mx 1 2 8 16 2>&1 | grep -v "^   " > $OUT/gemm_mx.txt || true
for W in 1 2 4 8 16; do timeout -k 10 100 python3 tools/encoder_profile.py large-v3 5 $W fp8mx 2>&1 | tail -1; done > $OUT/encoder_per_window_fp8mx.txt
for W in 1 16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_k$W -o k -- python3 tools/encoder_profile.py large-v3 5 $W fp8mx > /dev/null 2>&1
  python3 tools/encoder_profile.py --summarise $(find /tmp/pm_k$W -name "*kernel_trace.csv") > $OUT/encoder_kernels_w${W}_fp8mx.txt
done
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pm_mfma -o m -- python3 tools/encoder_profile.py large-v3 2 16 fp8mx > /dev/null 2>&1
python3 tools/pmc_by_grid.py $(find /tmp/pm_mfma -name "*counter_collection.csv") > $OUT/pmc_encoder_mfma_w16_fp8mx.txt
timeout -k 10 300 python3 bench.py --precision fp8mx --no-streams > $OUT/bench_line_fp8mx.json 2> $OUT/bench_line_fp8mx.err
timeout -k 10 300 python3 bench.py --workload streams --precision fp8mx --steps 3 --warmup 1 > $OUT/bench_line_streams64_fp8mx.json 2> /dev/null
timeout -k 10 300 python3 bench.py --workload streams --streams 8 --precision fp8mx --steps 3 --warmup 1 > $OUT/bench_line_streams8_fp8mx.json 2> /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_streams -o s -- python3 bench.py --workload streams --precision fp8mx --steps 2 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/pm_streams -name "*kernel_stats.csv") $OUT/kernel_stats_streams64_turbo_fp8mx.csv
timeout -k 10 200 python3 tools/streaming_latency.py --precision fp8mx > $OUT/streaming_latency_fp8mx.json 2> /dev/null
echo done

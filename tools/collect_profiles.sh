#!/bin/bash
# Round profiles (run on the GPU box from the repository root): kernel statistics of the bench command, the two HBM-traffic
# counter passes, the encoder's matrix-core counters for 1, 8 and 16 windows per pass, the 64-stream kernel statistics, the
# bench lines (bf16, e4m3, 64 streams), the streaming latencies and the Silero VAD timing.  The profiled bench commands run
# the headline workload alone (--no-streams): the default bench line also carries the 64-stream object, whose contended
# launches of the same kernels would be averaged into the headline's kernel statistics.  Each profiler pass is a run of
# its own (counters never share a run with trace domains other than --kernel-trace); raw traces stay in /tmp, only the
# summaries are copied to gpurun_out/ (the caller moves them to profiles/).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_single -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streams > $OUT/bench_under_profiler.log 2>&1
cp $(find /tmp/p_single -name "*kernel_stats.csv") $OUT/kernel_stats_single.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-streams --roofline-others "" > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-streams --roofline-others "" > /dev/null 2>&1
python3 tools/pmc_summary.py $(find /tmp/p_fetch -name "*counter_collection.csv") $(find /tmp/p_write -name "*counter_collection.csv") $OUT/pmc_traffic.json
for W in 1 8 16; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma$W -o m -- python3 tools/encoder_profile.py large-v3 2 $W > $OUT/encoder_w$W.log 2>&1
  python3 tools/pmc_by_grid.py $(find /tmp/p_mfma$W -name "*counter_collection.csv") > $OUT/pmc_encoder_mfma_w$W.txt
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_streams -o s -- python3 bench.py --workload streams --steps 2 --warmup 1 > $OUT/streams_under_profiler.log 2>&1
cp $(find /tmp/p_streams -name "*kernel_stats.csv") $OUT/kernel_stats_streams64_turbo.csv
timeout -k 10 300 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
timeout -k 10 300 python3 bench.py --precision fp8 > $OUT/bench_line_fp8.json 2> $OUT/bench_line_fp8.err
timeout -k 10 300 python3 bench.py --workload streams --steps 3 --warmup 1 > $OUT/bench_line_streams64.json 2> $OUT/bench_line_streams64.err
# the single-GPU proxy of the 8-GPU strong-scaling case: 64 streams / 8 GPUs = 8 streams per GPU
timeout -k 10 300 python3 bench.py --workload streams --streams 8 --steps 3 --warmup 1 > $OUT/bench_line_streams8.json 2> $OUT/bench_line_streams8.err
timeout -k 10 200 python3 tools/streaming_latency.py > $OUT/streaming_latency_bf16.json 2> /dev/null
timeout -k 10 200 python3 tools/streaming_latency.py --precision fp8 > $OUT/streaming_latency_fp8.json 2> /dev/null
timeout -k 10 100 python3 tools/silero_bench.py > $OUT/silero_vad_timing.txt 2>&1
for W in 1 8 16; do timeout -k 10 100 python3 tools/encoder_profile.py large-v3 5 $W 2>&1 | tail -1; done > $OUT/encoder_per_window.txt
echo done

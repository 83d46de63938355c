#!/bin/bash
# Round profiles (run on the GPU box from the repository root): kernel statistics of the bench command, the two HBM-traffic
# counter passes, and the encoder's matrix-core counters for 1, 4 and 8 windows per pass.  Each profiler pass is a run of
# its own (counters never share a run with trace domains other than --kernel-trace); raw traces stay in /tmp, only the
# summaries are copied to gpurun_out/ (the caller moves them to profiles/).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_single -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_profiler.log 2>&1
cp $(find /tmp/p_single -name "*kernel_stats.csv") $OUT/kernel_stats_single.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --roofline-others "" > /dev/null 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --roofline-others "" > /dev/null 2>&1
python3 tools/pmc_summary.py $(find /tmp/p_fetch -name "*counter_collection.csv") $(find /tmp/p_write -name "*counter_collection.csv") $OUT/pmc_traffic.json
for W in 1 4 8; do
  timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma$W -o m -- python3 tools/encoder_profile.py large-v3 2 $W > $OUT/encoder_w$W.log 2>&1
  python3 tools/pmc_by_grid.py $(find /tmp/p_mfma$W -name "*counter_collection.csv") > $OUT/pmc_encoder_mfma_w$W.txt
done
echo done

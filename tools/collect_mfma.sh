#!/bin/bash
# Encoder matrix-core counters for 1, 4 and 8 windows per pass (see tools/collect_profiles.sh).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
for W in 1 4 8; do
  rm -rf /tmp/p_mfma$W
  timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma$W -o m -- python3 tools/encoder_profile.py large-v3 2 $W > $OUT/encoder_w$W.log 2>&1
  python3 tools/pmc_by_grid.py $(find /tmp/p_mfma$W -name "*counter_collection.csv") > $OUT/pmc_encoder_mfma_w$W.txt
  head -1 $(find /tmp/p_mfma$W -name "*counter_collection.csv") > $OUT/cc_header.txt
done
echo done

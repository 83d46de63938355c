#!/bin/bash
# Development aid: the headline bench under HIP runtime knobs that touch the kernel boundary (measurement only).
set -e
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-140; }
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run AMD_OPT_FLUSH=0

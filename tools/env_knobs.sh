#!/bin/bash
# Development aid: the headline bench under HIP runtime knobs that touch the kernel boundary / the host wait (measurement only).
set -e
run() { echo "== $*"; env "$@" timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --roofline-others "" 2>&1 | tail -1 | cut -c1-140; }
for kv in "$@"; do run $kv; done

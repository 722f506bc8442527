#!/bin/bash
# tools/plain_probe.sh OUT -- plain (reference-made) containers of several bands decoded from the stream alone: kernel times, wall, status
out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" >> $out/plain.log; PROBE_PLAIN=1 QB3_DEBUG_DEC=1 timeout -k 10 200 python tools/kernel_probe.py "$@" 1 2>&1 | grep -E "^decode_plain|decode turn|rror" | tail -4 >> $out/plain.log; }
run 2048 2048 4 0 NOISY3 7
run 2048 2048 4 0 NOISY3 8
run 2048 2048 2 0 NOISY3 8
run 2048 2048 5 0 NOISY3 8
run 2048 2048 5 0 NOISY3 5
run 2048 2048 8 2 LANDSAT16 4
run 1024 1024 8 2 LANDSAT16 5
run 2048 2048 8 2 LANDSAT16 5
run 2048 2048 7 2 LANDSAT16 4
run 2048 2048 3 2 LANDSAT16 5
run 2048 2048 2 5 DEM 4
run 2048 2048 2 5 DEM 5
cat $out/plain.log

#!/bin/bash
# kernel times across raster shapes: which fall on slow generic kernels
out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" >> $out/survey.log; timeout -k 10 120 python tools/kernel_probe.py "$@" >> $out/survey.log 2>&1 || echo "FAILED rc $?" >> $out/survey.log; }
# 8-bit
run 4096 4096 2 0 NOISY3 8
run 4096 4096 5 0 NOISY3 8
run 4096 4096 2 0 NOISY3 7
run 4096 4096 5 0 NOISY3 7
# 16-bit
run 4096 4096 3 2 LANDSAT16 4
run 4096 4096 5 2 LANDSAT16 4
run 4096 4096 7 2 LANDSAT16 4
run 8192 8192 8 2 LANDSAT16 7
run 8192 8192 8 2 LANDSAT16 5
run 4096 4096 3 2 LANDSAT16 7
run 4096 4096 4 2 LANDSAT16 7
# 32/64-bit several bands
run 4096 4096 2 5 DEM 8
run 4096 4096 3 5 DEM 8
run 4096 4096 2 7 DEM 8
run 4096 4096 2 5 DEM 7
run 4096 4096 3 4 NOISY3 8
run 4096 4096 3 4 NOISY3 7

#!/bin/bash
# tools/small_plain.sh OUT -- plain decode wall times of small rasters: exits against the chain (QB3_WIDE_BAND=17)
out=gpurun_out/$1; mkdir -p $out; rm -f $out/small.log
for a in "128 128 3 0 NOISY3 8" "256 256 3 0 NOISY3 8" "512 512 3 0 NOISY3 8" "1024 1024 3 0 NOISY3 8" "256 256 2 2 LANDSAT16 4" "512 512 2 2 LANDSAT16 4" "1024 1024 2 2 LANDSAT16 4" "256 256 1 2 DEM 4" "512 512 1 5 DEM 8" "256 256 2 0 NOISY3 8" "512 512 2 0 NOISY3 8"; do
  for wb in 16 17; do
    echo "== $a wide_band $wb" >> $out/small.log
    QB3_WIDE_BAND=$wb PROBE_PLAIN=1 timeout -k 10 100 python tools/kernel_probe.py $a 1 2>&1 | grep -E "^decode_plain" | sed 's/.*wall ms/wall ms/' >> $out/small.log
  done
done
paste - - - - < $out/small.log

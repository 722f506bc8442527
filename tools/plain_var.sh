#!/bin/bash
# tools/plain_var.sh OUT variant... -- plain decode wall times with the product library and timing-only variants
out=gpurun_out/$1; mkdir -p $out; shift
for v in product "$@"; do
  for a in "2048 2048 2 0 NOISY3 8" "2048 2048 5 0 NOISY3 8" "2048 2048 7 2 LANDSAT16 4" "2048 2048 4 0 NOISY3 7" "2048 2048 8 2 LANDSAT16 4"; do
    if [ $v = product ]; then L=""; else L="qb3_amd/csrc/build/variants/libQB3_$v.so"; fi
    echo "== $v: $a" >> $out/pv.log
    QB3_LIB_PATH=$L PROBE_PLAIN=1 timeout -k 10 200 python tools/kernel_probe.py $a 1 2>&1 | grep -E "^decode_plain" | sed 's/.*wall ms/wall ms/' >> $out/pv.log
  done
done
cat $out/pv.log

#!/bin/bash
# tools/var_probe.sh OUT "W H B DT GEN MODE" [variant ...] -- kernel_probe with the product library and with timing-only variants (qb3_amd/csrc/build/variants)
out=gpurun_out/$1; mkdir -p $out; args=$2; shift 2
for v in product "$@"; do
  echo "== $v: $args" >> $out/var.log
  if [ $v = product ]; then timeout -k 10 200 python tools/kernel_probe.py $args 2>&1 | grep -E "^encode|^decode_index" >> $out/var.log
  else QB3_LIB_PATH=qb3_amd/csrc/build/variants/libQB3_$v.so timeout -k 10 200 python tools/kernel_probe.py $args 2>&1 | grep -E "^encode|^decode_index|Error|error" >> $out/var.log; fi
done

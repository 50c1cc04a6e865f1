#!/bin/bash
# C5's shape (1B clustered fp16 / 50M / k = 32) under several builds of the query kernels on ONE box, two rounds.  usage: tools/ab_c5.sh lib1 lib2 ... ("-" = shipped)
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset PT_HIP_LIB; else export PT_HIP_LIB=$L; fi
    echo "== lib $L"
    timeout -k 10 300 python tools/probe_r4.py c5 1e9 default 2>&1 | grep -v amdgpu.ids
  done
done

#!/bin/bash
# usage: variants.sh "NAME=FLAGS" ...   builds variant libs and prints kernel ms for the headline bench
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  make -s -C pgvector_amd/csrc OBJDIR=../../build/v_$name LIB=../../build/v_$name/libpgv_hip.so EXTRA="$flags" >/dev/null 2>&1 || { echo "$name build failed"; continue; }
  for i in 1 2; do
  PGV_HIP_LIB=$PWD/build/v_$name/libpgv_hip.so python bench.py --no-cpu-baseline --recall-queries 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$name', round(d['value']), round(d['roofline']['avg_launch_ms'],3), d['recall_at_10'])"
  done
done

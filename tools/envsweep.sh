#!/bin/bash
# usage: envsweep.sh <lib or ""> "VAR=val VAR2=val" ...   headline bench under each environment
lib=$1; shift
for spec in "$@"; do
  for i in 1 2; do
  env $spec PGV_HIP_LIB=$lib python bench.py --no-cpu-baseline --recall-queries 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$spec', round(d['value']), round(d['roofline']['avg_launch_ms'],3), d['recall_at_10'])"
  done
done

#!/bin/bash
# A/B of the batched step on ONE box: the working tree's library against build/ab/libpgv_hip_base.so
out=${1:-gpurun_out/ab_step}; mkdir -p $out; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round6.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" > $out/tests.log
PGV_HIP_LIB=$R/build/ab/libpgv_hip_base.so python tools/step_timeline.py > $out/timeline_base.txt 2>&1
python tools/step_timeline.py > $out/timeline_new.txt 2>&1
for rep in 1 2 3; do
  PGV_HIP_LIB=$R/build/ab/libpgv_hip_base.so python bench.py --child --steps 40 --warmup 5 2>/dev/null | tail -1 > $out/base_$rep.json
  python bench.py --child --steps 40 --warmup 5 2>/dev/null | tail -1 > $out/new_$rep.json
done

#!/bin/bash
# A/B of the single-query path on ONE box: the library of the working tree against build/ab/libpgv_hip_base.so
out=${1:-gpurun_out/ab_query}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd $R && timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" > $out/tests.log)
for rep in 1 2 3; do
  (cd $R && PGV_HIP_LIB=$R/build/ab/libpgv_hip_base.so python bench.py --section sweeps --section-out $out/base_$rep.json > $out/base_$rep.log 2>&1)
  (cd $R && python bench.py --section sweeps --section-out $out/new_$rep.json > $out/new_$rep.log 2>&1)
done
cd $R
PGV_HIP_LIB=$R/build/ab/libpgv_hip_base.so rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_base -o base -- python bench.py --section sweeps --soft-exit --section-out $out/pb.json > $out/prof_base.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_new -o new -- python bench.py --section sweeps --soft-exit --section-out $out/pn.json > $out/prof_new.log 2>&1
for d in base new; do mkdir -p $out/stats_$d; find $out/prof_$d -name "*stats*.csv" -exec cp {} $out/stats_$d/ \; ; rm -rf $out/prof_$d; done

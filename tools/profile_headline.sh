#!/bin/bash
# Profile the headline bench on the GPU box: a plain bench line, a rocprofv3
# kernel trace with stats, two separate PMC passes (FETCH_SIZE, WRITE_SIZE --
# never combined with each other or with sys/hip/hsa tracing), and a kernel trace
# of the round-2 micro-benchmarks (MFMA assignment, single-query path).
#
# usage (from the repo root, on the GPU box):  tools/profile_headline.sh <tag>
# writes gpurun_out/prof_<tag>/{bench.json,trace/,pmc_fetch/,pmc_write/,micro/,...}
# then:  python profiles/summarize_rocprof.py gpurun_out/prof_<tag> profiles/<tag>_summary.md
set -u
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
export PYTHONPATH=$root
python bench.py > "$out/bench.json" 2> "$out/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out/trace" -o trace --output-format csv -- \
    python "$root/bench.py" --no-cpu-baseline --no-sweeps --no-traffic > "$out/trace_bench.json" 2> "$out/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$out/pmc_fetch" -o fetch --output-format csv -- \
    python "$root/bench.py" --child --steps 5 --warmup 2 > "$out/fetch_bench.json" 2> "$out/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$out/pmc_write" -o write --output-format csv -- \
    python "$root/bench.py" --child --steps 5 --warmup 2 > "$out/write_bench.json" 2> "$out/write.err"
rocprofv3 --kernel-trace --stats -d "$out/micro" -o micro --output-format csv -- \
    python "$root/tools/bench_round2.py" > "$out/micro_bench.json" 2> "$out/micro.err"
cd "$root"
# the traces are large; keep the stats and the counter rows of our kernels only
find "$out" -name '*_agent_info.csv' -size +1M -delete
ls -la "$out" "$out"/*/ | head -60

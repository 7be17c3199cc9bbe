#!/bin/bash
# the 1 M x 1536 HNSW build of bench.py's hnsw section, REPS times on one box (phase seconds, recall)
out=gpurun_out/${1:-rep_hnsw}; mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
  timeout 300 python bench.py --sections hnsw --no-traffic > $out/r${rep}.json 2> $out/r${rep}.err
  python - <<PY
import json
d=json.loads(open("$out/r${rep}.json").read().strip().splitlines()[-1]); h=d["hnsw"]
print("rep ${rep}: build %.2f s" % h["build_secs"], {k: round(x,2) for k,x in h["build"]["phase_secs"].items()}, [v["recall_at_10"] for v in h["ef_search"].values()], h["parity"]["mismatches"])
PY
done

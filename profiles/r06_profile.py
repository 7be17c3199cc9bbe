#!/usr/bin/env python3
"""profiles/r06_profile.py -- tracked rocprofv3 evidence for every config's dominant kernel.

Runs ON THE GPU BOX (gpurun), one shape at a time, three rocprofv3 passes of the same short command each -- never a
--pmc pass together with a runtime / hip / memory trace (the pool refuses that), counters in passes of their own:
  trace   --kernel-trace --stats
  pmc_a   --pmc FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE
  pmc_b   --pmc WRITE_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
and condenses them into gpurun_out/r06prof/summary.json (+ one small CSV of per-kernel stats per shape); the raw
traces stay on the box.  profiles/r06_summarize.py turns summary.json into profiles/r06_rocprof_summary.md.

usage: python profiles/r06_profile.py [shape ...]     shapes: headline c2 hard c3full c5full c4 single dense
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "r06prof")
PY = sys.executable
BENCH = os.path.join(ROOT, "bench.py")

# shape -> (command, substring of the dominant kernel's name, what it is)
SHAPES = {
    "headline": ([PY, BENCH, "--child", "--steps", "10", "--warmup", "2", "--overlap-lanes", "0"], "mfma_scan_kernel<float, 0",
                 "headline 1 M x 1536 f32 L2, lists 1000, probes 10, 1024-query batches"),
    "c2": ([PY, BENCH, "--child", "--workload", "c2", "--steps", "10", "--warmup", "2", "--overlap-lanes", "0"], "mfma_scan_kernel<float, 0",
           "configs[1]: 1 M x 768 f32 L2, lists 1000, probes 10"),
    "hard": ([PY, BENCH, "--child", "--workload", "hard", "--steps", "10", "--warmup", "2", "--overlap-lanes", "0"], "mfma_scan_kernel<float, 0",
             "the headline's shape on the mid-difficulty data set (gen_hard)"),
    "c3full": ([PY, BENCH, "--section", "c3full", "--section-out", "/tmp/r06prof_c3.json", "--no-cpu-baseline", "--soft-exit"], "mfma_scan_kernel<float, 1",
               "configs[2] at full size: 10 M x 1536 f32 IP, lists 4096, probes 64, one GPU"),
    "c5full": ([PY, BENCH, "--section", "c5full", "--section-out", "/tmp/r06prof_c5.json", "--no-cpu-baseline", "--soft-exit"], "mfma_scan_kernel<__half, 0",
               "configs[4] at full size: 10 M x 3072 f16 L2, lists 4096, probes 64, one GPU"),
    "c4": ([PY, BENCH, "--section", "hnsw", "--section-out", "/tmp/r06prof_c4.json", "--no-cpu-baseline", "--soft-exit"], "hnsw_search_kernel",
           "configs[3]: HNSW 1 M x 1536 f32 cosine, m 16, GPU-built graph, ef_search 40 .. 1000, 20 000 queries in flight"),
    # (A/B: the 64-query form of the scan forced on for the headline: PGV_SCAN_WIDE=1, see run_pass)
    "headline_wide": ([PY, BENCH, "--child", "--steps", "10", "--warmup", "2", "--overlap-lanes", "0"], "mfma_scan_kernel<float, 0",
                      "headline with the 64-query form of the scan kernel forced on (PGV_SCAN_WIDE=1)"),
    "single": ([PY, os.path.join(ROOT, "tools", "exp_single_query.py"), "--rows", "1000000", "--lists", "1000", "--threads", "1", "--per", "3000"],
               "query_scan_kernel", "one backend, one query at a time (pgv_query_rank + pgv_query_scan): 1 M x 1536 f32, lists 1000, probes 10"),
    "dense": ([PY, os.path.join(ROOT, "tools", "exp_exact_topk.py")], "mfma_dense_kernel",
              "pgv_exact_topk: 1 M x 1536 f32 x 1024 queries, L2 and IP"),
}
PASSES = {
    "trace": ["--kernel-trace", "--stats"],
    "pmc_a": ["--kernel-trace", "--pmc", "FETCH_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_LDS_BANK_CONFLICT",
              "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    "pmc_b": ["--kernel-trace", "--pmc", "WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU"],
}


def short(name):
    name = name.replace("void ", "").replace("pgv::(anonymous namespace)::", "")
    return name.split("(")[0][:80]


def run_pass(shape, pname, cmd):
    d = os.path.join(OUT, shape, pname)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    env = dict(os.environ, TMPDIR="/tmp")
    if shape.endswith("_wide"):
        env["PGV_SCAN_WIDE"] = "1"
    t0 = time.time()
    with open(os.path.join(d, "stdout.txt"), "w") as so, open(os.path.join(d, "stderr.txt"), "w") as se:
        r = subprocess.run(["rocprofv3"] + PASSES[pname] + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd,
                           stdout=so, stderr=se, env=env, cwd="/tmp", timeout=900)
    return r.returncode, time.time() - t0, d


def dominant_launches(rows, want):
    """the launches of the dominant kernel that belong to the measured loop: those within 2 x of the longest-running
    group's median (build-time launches of the same kernel -- center ranking, k-means++ rounds -- are much shorter)"""
    mine = [r for r in rows if want in r["name"]]
    if not mine:
        return [], None
    # the instantiation with the most total time
    tot = defaultdict(float)
    for r in mine:
        tot[r["name"]] += r["dur"]
    name = max(tot, key=tot.get)
    mine = sorted([r for r in mine if r["name"] == name], key=lambda r: r["dur"])
    top = mine[-max(1, len(mine) // 4):]
    med = top[len(top) // 2]["dur"]
    return [r for r in mine if r["dur"] >= 0.5 * med], name


def main():
    shapes = sys.argv[1:] or list(SHAPES)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "summary.json")
    summary = json.load(open(path)) if os.path.exists(path) else {}
    for shape in shapes:
        cmd, want, what = SHAPES[shape]
        rec = {"what": what, "command": " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else c for c in cmd[1:]),
               "dominant": want, "passes": {}}
        for pname in PASSES:
            try:
                rc, secs, d = run_pass(shape, pname, cmd)
            except subprocess.TimeoutExpired:
                rec["passes"][pname] = {"error": "timeout"}
                continue
            rec["passes"][pname] = {"rc": rc, "secs": round(secs, 1)}
            if pname == "trace":
                files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
                if not files:
                    continue
                rows = [{"name": short(r["Kernel_Name"]), "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         "lds": int(r["LDS_Block_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": int(r.get("Accum_VGPR_Count", 0) or 0),
                         "sgpr": int(r["SGPR_Count"]), "grid": int(r["Grid_Size_X"]), "wg": int(r["Workgroup_Size_X"])}
                        for r in csv.DictReader(open(files[0]))]
                by = defaultdict(list)
                for r in rows:
                    by[r["name"]].append(r["dur"])
                total = sum(sum(v) for v in by.values())
                rec["kernels"] = [{"kernel": n, "calls": len(v), "total_ms": sum(v) / 1e3, "avg_us": sum(v) / len(v),
                                   "min_us": min(v), "max_us": max(v), "pct": 100.0 * sum(v) / total}
                                  for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]]
                dom, name = dominant_launches(rows, want)
                if dom:
                    rec["dominant_kernel"] = {"name": name, "launches": len(dom), "avg_us": sum(r["dur"] for r in dom) / len(dom),
                                              "min_us": min(r["dur"] for r in dom), "max_us": max(r["dur"] for r in dom),
                                              "lds_bytes": dom[-1]["lds"], "vgprs": dom[-1]["vgpr"], "agprs": dom[-1]["agpr"],
                                              "sgprs": dom[-1]["sgpr"], "grid_threads": dom[-1]["grid"], "workgroup": dom[-1]["wg"]}
                with open(os.path.join(OUT, "%s_kernel_stats.csv" % shape), "w") as f:
                    f.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
                    for krec in rec["kernels"]:
                        f.write("\"%s\",%d,%.3f,%.2f,%.2f,%.2f,%.2f\n" % (krec["kernel"], krec["calls"], krec["total_ms"], krec["avg_us"],
                                                                       krec["min_us"], krec["max_us"], krec["pct"]))
            else:
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if not files:
                    continue
                per = defaultdict(lambda: defaultdict(list))   # dispatch -> counter -> values
                meta = {}
                for r in csv.DictReader(open(files[0])):
                    did = r.get("Dispatch_Id") or r.get("Correlation_Id")
                    meta[did] = {"name": short(r["Kernel_Name"]), "dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3}
                    per[did][r["Counter_Name"]].append(float(r["Counter_Value"]))
                rows = [{"name": m["name"], "dur": m["dur"], "did": did} for did, m in meta.items()]
                dom, name = dominant_launches(rows, want)
                counters = defaultdict(list)
                for r in dom:
                    for c, v in per[r["did"]].items():
                        counters[c].append(sum(v))
                rec["passes"][pname]["launches"] = len(dom)
                rec["passes"][pname]["avg_us_profiled"] = sum(r["dur"] for r in dom) / len(dom) if dom else None
                rec["passes"][pname]["counters"] = {c: sum(v) / len(v) for c, v in counters.items()}
            # the bench's own line of the traced run, when the command printed one
            try:
                lines = [ln for ln in open(os.path.join(d, "stdout.txt")).read().splitlines() if ln.startswith("{")]
                if lines and pname == "trace":
                    rec["bench_line"] = json.loads(lines[-1])
            except Exception:  # noqa: BLE001
                pass
            shutil.rmtree(d, ignore_errors=True)     # raw traces stay out of the merge (64 MiB)
        for tmp, key in (("/tmp/r06prof_c3.json", "section"), ("/tmp/r06prof_c5.json", "section"), ("/tmp/r06prof_c4.json", "section")):
            if os.path.exists(tmp) and shape in ("c3full", "c5full", "c4") and tmp.endswith({"c3full": "c3.json", "c5full": "c5.json", "c4": "c4.json"}[shape]):
                try:
                    rec[key] = json.load(open(tmp))
                except Exception:  # noqa: BLE001
                    pass
        summary[shape] = rec
        json.dump(summary, open(path, "w"), indent=1)
        print(shape, json.dumps(rec.get("dominant_kernel")), {p: v.get("counters") for p, v in rec["passes"].items() if "counters" in v}, flush=True)


if __name__ == "__main__":
    main()

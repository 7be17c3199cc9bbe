#!/usr/bin/env python3
"""profiles/r06_summarize.py -- gpurun_out/r06prof/summary.json (profiles/r06_profile.py, run on the GPU box) ->
profiles/r06_rocprof_summary.md + profiles/r06_rocprof/ (summary.json and the per-shape kernel tables).

Per shape: the per-kernel table of the traced process, and for the config's dominant kernel its launches, avg / min / max
us, LDS, VGPRs, HBM bytes from the PMC passes (2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md; WRITE_SIZE)
beside the bytes the bench counted, the matrix pipes' busy share, LDS bank conflicts, occupancy, clock, and `frac`
recomputed from the TRACE's average duration (never from a profiled PMC pass: those run at lower clocks)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06prof")
DST = os.path.join(ROOT, "profiles", "r06_rocprof")
NCU, NSIMD = 256, 1024


def main():
    s = json.load(open(os.path.join(SRC, "summary.json")))
    os.makedirs(DST, exist_ok=True)
    for f in os.listdir(SRC):
        if f.endswith(".csv") or f == "summary.json":
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
    L = ["# rocprofv3 evidence per config, round 6 (`profiles/r06_profile.py` on one MI355X box)", "",
         "Three passes of the same short command per shape: `--kernel-trace --stats`; `--pmc FETCH_SIZE "
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE`; "
         "`--pmc WRITE_SIZE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU` (counters never together with a "
         "runtime trace).  Durations and `frac` come from the TRACE pass (the PMC passes run slower); counters are "
         "averages over the dominant kernel's measured launches.  HBM read bytes = 2 x FETCH_SIZE (the guide's gfx950 "
         "correction for wide coalesced reads); WRITE_SIZE as reported (uncalibrated).  Raw tables: `profiles/r06_rocprof/`.", ""]
    table = ["| shape | dominant kernel | launches | avg us (min - max) | LDS B | VGPRs | unique GB | 2 x FETCH GB | WRITE GB | "
             "frac of 8 TB/s (unique / fetched) | MFMA busy | LDS conflicts | waves / SIMD | GHz | wave cycles waiting |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    detail = []
    for shape, r in s.items():
        dk = r.get("dominant_kernel") or {}
        a = (r["passes"].get("pmc_a") or {}).get("counters") or {}
        b = (r["passes"].get("pmc_b") or {}).get("counters") or {}
        unique = None
        extra = ""
        line = r.get("bench_line") or {}
        sec = r.get("section") or {}
        if "roofline" in line:
            unique = line["roofline"].get("unique_bytes_per_launch")
            extra = "bench line of the traced run: %.0f QPS, recall %.4f, scan %.3f ms/launch (HIP events), passes %.3f" % (
                line["value"], line.get("recall_at_10", float("nan")), line["roofline"]["avg_launch_ms"], line["roofline"]["passes"])
        for key in ("full_c3", "full_c5"):
            if key in sec and "roofline" in sec[key]:
                unique = sec[key]["roofline"].get("unique_bytes_per_launch")
                extra = "section record of the traced run: %.0f QPS, recall %.4f, scan %.3f ms/launch (HIP events), passes %.3f" % (
                    sec[key]["qps"], sec[key]["recall_at_10"], sec[key]["roofline"]["avg_launch_ms"], sec[key]["roofline"]["passes"])
        if shape == "c4" and "hnsw" in sec:
            ef = sec["hnsw"]["ef_search"]
            extra = "section record of the traced run: " + ", ".join(
                "ef %s: %.0f QPS recall %.4f (%.0f GB/s of gathered rows)" % (e, v["qps"], v["recall_at_10"], v["scored_rows_GBps"])
                for e, v in ef.items())
        if shape == "single":
            extra = ("the four launches of one query: query_rank_kernel (also the k-means++ rounds of the build: its average is theirs), "
                     "query_lists_kernel, query_scan_kernel, query_head_kernel; ~11.8 k rows x 6 KB per scan")
        if shape == "dense":
            unique = 1_000_000 * 1536 * 4.0   # the rows once; the kernel is MFMA-bound: 2 n nq dim flops
            extra = "1 M x 1536 x 1024 queries = 3.15e12 flop per launch: %.1f TFLOP/s of 157.3 (fp32 MFMA)" % (
                2.0 * 1e6 * 1024 * 1536 / (dk["avg_us"] * 1e-6) / 1e12)
        fetch = 2.0 * a["FETCH_SIZE"] * 1024 if "FETCH_SIZE" in a else None
        write = b["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in b else None
        secs = dk.get("avg_us", 0) * 1e-6
        frac_u = unique / secs / 8e12 if unique and secs else None
        frac_f = fetch / secs / 8e12 if fetch and secs else None
        mfma = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * a["SQ_BUSY_CU_CYCLES"]) if a.get("SQ_BUSY_CU_CYCLES") else None
        conf = a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"] if a.get("SQ_LDS_IDX_ACTIVE") else None
        pa = r["passes"].get("pmc_a") or {}
        cycles = a["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in a else None
        ghz = cycles / (pa["avg_us_profiled"] * 1e3) if cycles and pa.get("avg_us_profiled") else None
        pb = r["passes"].get("pmc_b") or {}
        occ = None
        if "SQ_WAVE_CYCLES" in b and ghz and pb.get("avg_us_profiled"):
            occ = b["SQ_WAVE_CYCLES"] * 4.0 / (ghz * pb["avg_us_profiled"] * 1e3 * NSIMD)
        wait = b["SQ_WAIT_ANY"] / b["SQ_WAVE_CYCLES"] if b.get("SQ_WAVE_CYCLES") else None

        def f(x, fmt="%.3f"):
            return "-" if x is None else fmt % x
        table.append("| %s | `%s` | %s | %s (%s - %s) | %s | %s | %s | %s | %s | %s / %s | %s | %s | %s | %s | %s |" % (
            shape, dk.get("name", "?"), dk.get("launches", "-"), f(dk.get("avg_us"), "%.1f"), f(dk.get("min_us"), "%.1f"),
            f(dk.get("max_us"), "%.1f"), dk.get("lds_bytes", "-"), dk.get("vgprs", "-"), f(unique and unique / 1e9), f(fetch and fetch / 1e9),
            f(write and write / 1e9), f(frac_u), f(frac_f), f(mfma, "%.2f"), f(conf, "%.4f"), f(occ, "%.2f"), f(ghz, "%.2f"), f(wait, "%.2f")))
        detail += ["## %s -- %s" % (shape, r["what"]), "", "`%s`" % r["command"], ""]
        if extra:
            detail += [extra, ""]
        detail += ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
        for k in r.get("kernels", [])[:10]:
            detail.append("| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f |" % (k["kernel"][:70], k["calls"], k["total_ms"], k["avg_us"],
                                                                            k["min_us"], k["max_us"], k["pct"]))
        detail += ["", "counters over %s / %s launches of the dominant kernel (pass a / b, profiled avg %s / %s us): %s" % (
            pa.get("launches"), pb.get("launches"), f(pa.get("avg_us_profiled"), "%.1f"), f(pb.get("avg_us_profiled"), "%.1f"),
            ", ".join("%s %.4g" % kv for kv in sorted({**a, **b}.items()))), ""]
    L += table + [""] + [
        "Reading: `unique` = bytes of the rows some query of the batch probes, each once (what `roofline.frac` counts); "
        "`2 x FETCH` = what the L2 asked the fabric for (rows streamed once per <= 32-query group, query rows that missed, "
        "descriptors).  c4's kernel gathers whole rows at data-dependent addresses (no `unique`: every scored row is "
        "counted, bench `scored_rows_GBps`); its launch list mixes the build's searches with the timed ones, the section "
        "record carries the timed rates.  `dense` is MFMA-bound (flops in its section).", ""] + detail
    open(os.path.join(ROOT, "profiles", "r06_rocprof_summary.md"), "w").write("\n".join(L) + "\n")
    print("\n".join(table))


if __name__ == "__main__":
    main()

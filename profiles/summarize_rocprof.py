#!/usr/bin/env python3
"""Condense rocprofv3 CSVs (kernel trace + PMC passes) of one bench.py run into
the per-kernel summary that is committed under profiles/.

usage: summarize_rocprof.py <prof_dir> <out.md>
  <prof_dir>/trace/*_kernel_trace.csv           rocprofv3 --kernel-trace --stats
  <prof_dir>/pmc_fetch/*_counter_collection.csv rocprofv3 --pmc FETCH_SIZE --kernel-trace
  <prof_dir>/pmc_write/*_counter_collection.csv rocprofv3 --pmc WRITE_SIZE --kernel-trace
The timed list-scan launches are told apart from the center-ranking launches of
the same kernel by their dynamic LDS size and duration (the exact-scan used for
recall ground truth is listed separately).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("pgv::(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def load_trace(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((short(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                     int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])))
    return rows


def load_pmc(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.append((short(r["Kernel_Name"]), float(r["Counter_Value"]),
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["LDS_Block_Size"])))
    return out


def main():
    d, out = sys.argv[1], sys.argv[2]
    lines = ["# rocprofv3 summary: %s" % os.path.basename(os.path.normpath(d)), ""]
    bench = json.load(open(os.path.join(d, "trace_bench.json")))
    rf = bench["roofline"]
    lines += ["bench line of the traced run: value %.0f %s, scan kernel avg %.3f ms/launch (HIP events in bench.py), "
              "%.0f GB/s streamed from HBM (frac %.3f of 8 TB/s), %.0f GB/s of UNIQUE probed rows (roofline.achieved: every probed row "
              "counted once; SURVEY 8d's per-(query, row)-pair figure is algorithmic_bytes_per_launch / time and exceeds the "
              "physical rate), passes %s" % (
                  bench["value"], bench["unit"], rf["avg_launch_ms"], rf.get("achieved_streamed", rf.get("streamed_GBps", rf["achieved"])),
                  rf.get("frac_streamed", rf["frac"]), rf["achieved"],
                  "%.3f" % rf["passes"] if rf.get("passes") else "n/a"), ""]
    trace = load_trace(glob.glob(os.path.join(d, "trace", "*_kernel_trace.csv"))[0])
    by = defaultdict(list)
    for name, dur, lds, vgpr, grid, wg in trace:
        by[name].append(dur)
    total = sum(sum(v) for v in by.values())
    lines += ["## per-kernel totals (`rocprofv3 --kernel-trace --stats`, whole process incl. setup/build)", "",
              "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:18]:
        lines.append("| %s | %d | %.2f | %.1f | %.1f | %.1f | %.1f |" % (
            name, len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / total))
    # the timed list-scan launches: the `steps` launches of the widest scan kernel that are
    # neither the short center-ranking ones nor the single exact-scan
    steps = bench["steps"]
    # the list-scan kernel = the *scan_kernel instantiation with the most total time
    # (judged over the tail of the trace: the k-means++ rounds of the build use scan_kernel too)
    tail = [(name, dur) for name, dur, lds, vgpr, grid, wg in trace if "scan_kernel" in name][-2 * steps:]
    tail_time = defaultdict(int)
    for name, dur in tail:
        tail_time[name] += dur
    main = max(tail_time, key=lambda n: tail_time[n])
    scans = [(dur, lds, vgpr, grid) for name, dur, lds, vgpr, grid, wg in trace if name == main]
    timed = scans[-2 * steps:]  # alternating rank / scan launches of the timed loop
    list_scan = [s for s in timed if s[0] > 5 * min(t[0] for t in timed)]
    if not list_scan and timed and max(t[0] for t in timed) < 2 * min(t[0] for t in timed):
        # (round 4: the center ranking runs another instantiation -- every launch of this one is a list scan)
        list_scan = scans[-steps:]
    if list_scan:
        avg = sum(s[0] for s in list_scan) / len(list_scan)
        lines += ["", "## timed list-scan launches (%s)" % main, "",
                  "%d launches, avg %.3f ms (min %.3f, max %.3f), LDS %d B/workgroup, %d VGPRs, grid %d threads" % (
                      len(list_scan), avg / 1e6, min(s[0] for s in list_scan) / 1e6, max(s[0] for s in list_scan) / 1e6,
                      list_scan[0][1], list_scan[0][2], list_scan[0][3]),
                  "bench.py's live HIP-event average for the same kernel: %.3f ms" % bench["roofline"]["avg_launch_ms"]]
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        paths = glob.glob(os.path.join(d, tag, "*_counter_collection.csv"))
        if not paths:
            continue
        rows = [r for r in load_pmc(paths[0], counter) if r[0] == main]
        if not rows:
            continue
        b = json.load(open(os.path.join(d, tag.split("_")[1] + "_bench.json")))
        n = b["steps"]
        timed = rows[-2 * n:]
        big = [r for r in timed if r[2] > 5 * min(t[2] for t in timed)]
        if not big and timed and max(t[2] for t in timed) < 2 * min(t[2] for t in timed):
            big = rows[-n:]
        if not big:
            continue
        val = sum(r[1] for r in big) / len(big)
        rfb = b["roofline"]
        # (round 5: bench.py's stdout line is the compact one; older lines carried the byte counts themselves)
        stream = rfb.get("streamed_bytes_per_launch",
                         rfb.get("frac_streamed", rfb["frac"]) * rfb["peak"] * 1e9 * rfb["avg_launch_ms"] / 1e3)
        cfgb = b["config"]
        algo = rfb.get("algorithmic_bytes_per_launch",
                       # pairs scored per launch x row bytes: batch x (rows of `probes` average lists) x dim x 4
                       cfgb["batch_per_gpu"] * cfgb["probes"] * (cfgb["rows"] / cfgb["lists"]) * cfgb["dim"] * 4.0)
        lines += ["", "## %s over the timed list-scan launches" % counter, "",
                  "%d launches, avg %s = %.0f KB per launch (rocprofv3 reports KB)" % (len(big), counter, val)]
        if counter == "FETCH_SIZE":
            lines += ["gfx950 correction (MI355X_MICROARCH.md, HBM section): wide coalesced reads are tallied at half "
                      "their size -> HBM read traffic = 2 x FETCH_SIZE = %.2f GB per launch" % (2 * val * 1024 / 1e9),
                      "algorithmic bytes per launch %.2f GB, streamed rows per launch %.2f GB" % (algo / 1e9, stream / 1e9)]
        else:
            lines += ["= %.3f GB written per launch (uncalibrated on gfx950; output is 4 B per scored pair = %.3f GB)" % (
                val * 1024 / 1e9, algo / (4.0 * b["config"]["dim"]) * 4 / 1e9)]
    micro = glob.glob(os.path.join(d, "micro", "*_kernel_trace.csv"))
    if micro:
        mt = load_trace(micro[0])
        mby = defaultdict(list)
        for name, dur, lds, vgpr, grid, wg in mt:
            mby[name].append(dur)
        keep = [n for n in mby if any(t in n for t in ("mfma_argmin", "recheck", "chosen_distance", "argmin_kernel",
                                                        "center_norms", "redo_finish", "query_"))]
        lines += ["", "## round-2 micro-benchmarks (`tools/bench_round2.py` under `rocprofv3 --kernel-trace --stats`)", "",
                  "(`query_rank_kernel` also runs the 999 rounds of the micro-bench's k-means++ seeding -- one query against 50 000",
                  "samples, ~44 us each --, so its AVERAGE is not the single-query ranking's: the median and the share of launches",
                  "under 10 us tell them apart)", "",
                  "| kernel | calls | avg us | median us | min us | max us | launches < 10 us |", "|---|---|---|---|---|---|---|"]
        for name in sorted(keep, key=lambda n: -sum(mby[n])):
            v = sorted(mby[name])
            lines.append("| %s | %d | %.1f | %.1f | %.1f | %.1f | %d |" % (
                name, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3, sum(1 for x in v if x < 10000)))
        mj = os.path.join(d, "micro_bench.json")
        if os.path.exists(mj):
            try:
                mb = json.load(open(mj))
                lines += ["", "assignment cases (HIP-event time of the whole pgv_assign call, incl. recheck / redo / distances):", ""]
                for c in mb.get("assign", []):
                    lines.append("* %s: n %d, k %d, dim %d: %.2f ms = %.1f T MAC/s (%.0f TFLOP/s at 2 flop/MAC), rechecked %s, redone %s" % (
                        c["case"], c["n"], c["k"], c["dim"], c["ms"], c["tmac_per_s"], c["tflops_2flop"],
                        "%.3f" % c["recheck_fraction"] if c.get("recheck_fraction") is not None else "-",
                        "%.5f" % c["redo_fraction"] if c.get("redo_fraction") is not None else "-"))
                q = mb.get("query", {})
                if q:
                    lines += ["", "single-query path (1M x 1536, probes 10): " + json.dumps(q)]
            except Exception as e:
                lines.append("(micro_bench.json unreadable: %r)" % (e,))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

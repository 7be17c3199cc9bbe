"""The line bench.py leaves on stdout is what the driver parses (BENCH_rNN.json.parsed).  Round 4's line had grown to
~21 KB and came back `parsed: null`; this pins the contract: ONE line, strict JSON, under 4 096 bytes, nothing after
it on stdout, carrying roofline and cpu_baseline -- whatever the full record holds (no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = os.path.join(ROOT, "profiles", "r04", "bench_headline_final.json")   # a real, unabridged 21 KB record

CODE = r'''
import json, os, sys
sys.argv = ["bench.py"]
sys.path.insert(0, %r)
import bench
bench.write_detail = lambda full: ["(not written in the test)"]
full = json.load(open(%r))
%s
fd = os.dup(1)
os.dup2(2, 1)            # as bench.main does: stray prints go to stderr
print("a library greeting on stdout")
bench.emit_line(fd, full)
'''


def emit(mutation=""):
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, FULL, mutation)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


def check_contract(raw):
    assert raw.endswith(b"\n") and raw.count(b"\n") == 1, "exactly one line on stdout, nothing after it"
    assert len(raw) < 4096

    def strict(c):
        raise ValueError("non-JSON constant %s" % c)
    line = json.loads(raw.decode(), parse_constant=strict)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert "model" not in line["config"] and "workload" in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    return line


def test_the_line_is_one_compact_strict_json_object():
    line = check_contract(emit())
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    assert line["parity"] == {"mismatches": 0, "checked": 256, "page_built_index_mismatches": 0}
    assert set(line["other_configs"]) >= {"c2", "c3shard", "c5shard", "c4_hnsw", "c1_exact"}
    assert line["other_configs"]["c2"]["parity_mismatches"] == 0


def test_the_line_stays_under_the_cap_when_the_record_bloats():
    # NaN / inf, hundreds of failures, fat notes, extra configs: the line sheds optional parts, never the contract
    mutation = (
        "full['failures'] = ['section %d: ' % i + 'x' * 500 for i in range(300)]\n"
        "full['roofline']['traffic'] = float('nan')\n"
        "full['cpu_baseline']['sample'] = 'y' * 5000\n"
        "full['config']['workload'] = 'w' * 3000\n"
        "full['other_configs'].update({'extra%d' % i: dict(full['other_configs']['c2']) for i in range(40)})\n"
        "full['value'] = float('inf')\n")
    line = check_contract(emit(mutation))
    assert line["roofline"]["traffic"] is None and line["value"] is None
    assert line["failures"]


def test_compact_line_leaves_the_full_record_alone():
    import bench
    full = json.load(open(FULL))
    before = json.dumps(full, sort_keys=True)
    bench.compact_line(full)
    assert json.dumps(full, sort_keys=True) == before

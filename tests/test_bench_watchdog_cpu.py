"""bench.py must print its ONE JSON line whatever an optional section does (no GPU needed for this part):
  * optional sections are children with budgets; one that hangs is killed WITH ITS DESCENDANTS, what it had written
    is kept, and the line says where it was;
  * the watchdog (last resort for the parent itself) writes the line as far as it has got, names what did not return,
    and exits NON-ZERO like any failed run."""
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CODE = r'''
import sys, time, threading
sys.argv = ["bench.py"]
sys.path.insert(0, %r)
import bench
bench.WATCH.update({"line": %s, "fd": 1, "rank": %d, "section": "a section that hangs"})
threading.Thread(target=bench.watchdog, args=(1,), daemon=True).start()
time.sleep(60)
print("not reached")
'''


def run(line_literal, rank=0):
    return subprocess.run([sys.executable, "-c", CODE % (ROOT, line_literal, rank)], capture_output=True, text=True,
                          timeout=120)


def test_watchdog_prints_the_line_and_fails_the_run():
    r = run('{"metric": "m", "value": 2.5, "failures": ["earlier"]}')
    assert r.returncode == 2 and "not reached" not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 2.5 and line["failures"][0] == "earlier"
    assert "a section that hangs" in line["failures"][1] and line["bench_wall_secs"] == 1


def test_watchdog_without_a_line_fails_loudly():
    r = run("None")
    assert r.returncode == 3 and r.stdout.strip() == ""


def test_watchdog_is_quiet_on_other_ranks():
    r = run('{"metric": "m", "value": 1.0}', rank=1)
    assert r.returncode == 0 and r.stdout.strip() == ""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:   # a zombie waiting for init counts as gone
        return open("/proc/%d/stat" % pid).read().split(") ")[1][0] != "Z"
    except OSError:
        return False


def test_a_hung_section_is_killed_with_its_descendants_and_its_rows_are_kept(tmp_path, monkeypatch):
    import bench
    pidfile = tmp_path / "grandchild.pid"
    child = (
        "import json, os, subprocess, sys, time\n"
        "g = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"      # a backend process
        "open(%r, 'w').write(str(g.pid))\n"
        "json.dump({'failures': ['a row failed'], '_at': 'row 3 of 7', 'concurrent_backends': {'1': {'qps': 5.0}}},"
        " open(sys.argv[1], 'w'))\n"
        "time.sleep(600)\n" % str(pidfile))
    monkeypatch.setattr(bench, "section_cmd", lambda name, args, path: [sys.executable, "-c", child, path])
    line, failures = {"value": 1.0}, []
    t0 = time.time()
    bench.run_section("backends", types.SimpleNamespace(), 3, line, failures)
    assert time.time() - t0 < 15
    assert line["concurrent_backends"] == {"1": {"qps": 5.0}}            # what it had finished is in the line
    assert line["sections"]["backends"]["timed_out"] is True
    assert any("a row failed" in f for f in failures)
    assert any("not back after 3 s" in f and "row 3 of 7" in f and f.startswith("budget:") for f in failures)
    gpid = int(pidfile.read_text())
    for _ in range(50):
        if not _alive(gpid):
            break
        time.sleep(0.1)
    assert not _alive(gpid), "the section's own child processes must go with it"


def test_a_section_that_ends_is_merged_and_a_crashed_one_is_a_failure(tmp_path, monkeypatch):
    import bench
    ok = "import json, sys; json.dump({'failures': [], '_at': 'done', 'hnsw': {'build_secs': 12.0}}, open(sys.argv[1], 'w'))"
    monkeypatch.setattr(bench, "section_cmd", lambda name, args, path: [sys.executable, "-c", ok, path])
    line, failures = {}, []
    bench.run_section("hnsw", types.SimpleNamespace(), 30, line, failures)
    assert line["hnsw"] == {"build_secs": 12.0} and failures == [] and line["sections"]["hnsw"]["rc"] == 0
    crash = "import os; os._exit(7)"
    monkeypatch.setattr(bench, "section_cmd", lambda name, args, path: [sys.executable, "-c", crash, path])
    bench.run_section("configs", types.SimpleNamespace(), 30, line, failures)
    assert len(failures) == 1 and "exit code 7" in failures[0] and "nothing" in failures[0]


def test_the_mandatory_block_comes_before_every_child():
    """the order the verdict of round 3 asked for, checked on the source: cpu_baseline + parity in the parent before the
    first section starts, the backends last"""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index('line["cpu_baseline"] = base') < main.index("run_section(name, args")
    assert bench.SECTION_ORDER[-1] == "backends" and bench.SECTION_ORDER[0] == "configs"
    assert sum(bench.SECTION_BUDGET_S.values()) + 2 * 110 <= 1000

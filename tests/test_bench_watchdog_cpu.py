"""bench.py must print its ONE JSON line whatever an optional section does: the watchdog writes the line as far as it
has got, names the section that did not return, and ends the process (no GPU needed for this part)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


def test_watchdog_prints_the_line_and_exits():
    r = run('{"metric": "m", "value": 2.5, "failures": ["earlier"]}')
    assert r.returncode == 0 and "not reached" not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 2.5 and line["failures"][0] == "earlier"
    assert "a section that hangs" in line["failures"][1] and line["bench_wall_secs"] == 1


def test_watchdog_without_a_line_fails_loudly():
    r = run("None")
    assert r.returncode == 3 and r.stdout.strip() == ""


def test_watchdog_is_quiet_on_other_ranks():
    r = run('{"metric": "m", "value": 1.0}', rank=1)
    assert r.returncode == 0 and r.stdout.strip() == ""

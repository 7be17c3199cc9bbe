"""`python bench.py --gpus N` must start N ranks by itself (the reference's analogue: a parallel build
launches its own workers, src/ivfbuild.c:830-966) -- and must never print a 1-GPU number under "n_gpus": N.  No device
is needed for this part: --dry-launch --backend gloo forms the group and reports what it agreed on."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(argv, env=None, timeout=180):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    return r, (json.loads(lines[-1]) if lines else None), time.time() - t0


def test_gpus_2_without_world_size_starts_two_ranks_that_agree():
    r, line, _ = run(["--gpus", "2", "--backend", "gloo", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1            # ONE line on stdout
    assert line["n_gpus"] == 2 and line["world_agreed"] == 2 and line["ranks_seen"] == [0, 1]
    assert line["launcher"]["ranks"] == 2 and line["launcher"]["joined"] and line["launcher"]["exit_codes"] == [0, 0]
    assert "self-launch" in line["launcher"]["kind"]


def test_three_ranks():
    r, line, _ = run(["--gpus", "3", "--backend", "gloo", "--dry-launch"])
    assert r.returncode == 0 and line["world_agreed"] == 3 and line["ranks_seen"] == [0, 1, 2]


def test_a_rank_that_dies_fails_the_run_and_takes_the_others_down():
    r, line, secs = run(["--gpus", "3", "--backend", "gloo", "--dry-launch", "--startup-timeout", "60"],
                        env={"PGV_BENCH_TEST_FAULT": "die:1"})
    assert r.returncode == 2 and secs < 60
    assert line["value"] is None and line["n_gpus"] == 3
    assert any("rank 1 exited with code 7" in f for f in line["failures"])


def test_a_rank_that_never_joins_is_a_startup_timeout():
    r, line, secs = run(["--gpus", "2", "--backend", "gloo", "--dry-launch", "--startup-timeout", "8"],
                        env={"PGV_BENCH_TEST_FAULT": "hang:1"})
    assert r.returncode == 2 and secs < 60
    assert any("of 2 ranks joined the group within 8 s" in f for f in line["failures"])


def test_no_device_over_rccl_is_a_failure_not_a_one_gpu_run():
    """this container has no GPU: --gpus 2 over RCCL cannot run and must say so with exit code 2"""
    r, line, _ = run(["--gpus", "2"])
    assert r.returncode == 2
    assert line["value"] is None and line["n_gpus"] == 2 and line["failures"]
    assert "device" in line["failures"][0]


def test_world_size_that_disagrees_with_gpus_is_refused():
    r, line, _ = run(["--gpus", "4", "--backend", "gloo", "--dry-launch"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE is 2 but --gpus is 4" in line["failures"][0]


def test_under_torch_distributed_run_nothing_is_started_from_here():
    """the contract's own form: torch.distributed.run sets WORLD_SIZE, bench.py joins as a rank"""
    import socket
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--backend", "gloo", "--dry-launch"],
                       capture_output=True, text=True, timeout=180, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
    assert line["world_agreed"] == 2 and line["launcher"].startswith("external")

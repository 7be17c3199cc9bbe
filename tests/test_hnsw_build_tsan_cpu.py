"""The pipelined HNSW build's threads under ThreadSanitizer: pgv_host_hnsw_build's main thread, the helper that runs the
next batch's searches and the graph patches, and the one that scores the list records' pairs (pgvector_amd/host/
hnsw_build.c), against tests/c/mock_hip.c.  OpenMP is compiled out (tests/c/omp_stub/omp.h): libgomp's barriers are
invisible to the sanitizer, and the hand-offs under test are the pthread ones."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelined_build_has_no_data_race(tmp_path):
    exe = str(tmp_path / "hnsw_build_tsan")
    srcs = [os.path.join(ROOT, "tools", "hnsw_host_bench.c"), os.path.join(ROOT, "tests", "c", "mock_hip.c")]
    srcs += sorted(glob.glob(os.path.join(ROOT, "pgvector_amd", "host", "*.c")))
    cc = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=thread", "-rdynamic", "-Wno-unknown-pragmas",
                         "-I", os.path.join(ROOT, "tests", "c", "omp_stub"), "-I", os.path.join(ROOT, "include"),
                         "-I", os.path.join(ROOT, "pgvector_amd", "host")] + srcs +
                        ["-o", exe, "-lm", "-lpthread", "-lrt"], capture_output=True, text=True)
    if cc.returncode != 0 and "tsan" in cc.stderr.lower():
        pytest.skip("no ThreadSanitizer runtime in this toolchain")
    assert cc.returncode == 0, cc.stderr[-2000:]
    # 3000 elements, batches of up to 64: the helpers start at the first full batch (1024 linked elements)
    r = subprocess.run([exe, "3000", "64"], capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "rc 0 " in out, out[-2000:]
    assert "ThreadSanitizer" not in out, out[-4000:]

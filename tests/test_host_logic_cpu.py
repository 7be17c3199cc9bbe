"""The C host glue on the CPU: tests/c/host_logic_driver.c runs libpgv_host.so's HNSW build loop (against the
oracle's graph), page writers/stagers, the IVFFlat build / scan / iterative-scan drivers, vacuum and the
self-invalidating mirror against tests/c/mock_hip.c, a plain-C stand-in for the device entry points.  What is
tested is the host logic around the distance calls; the distances themselves are the GPU tests' business.
The stand-in is test infrastructure: it is compiled into the driver executable only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_logic_against_the_mock_device(tmp_path):
    libdir = os.path.join(ROOT, "pgvector_amd", "lib")
    oradir = os.path.join(ROOT, "oracle")
    exe = str(tmp_path / "host_logic_driver")
    # -rdynamic: the stand-in's pgv_* symbols live in the executable and must win over libpgv_hip.so's
    subprocess.run(["gcc", "-O1", "-Wall", "-rdynamic", "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "pgvector_amd", "host"), "-I", oradir,
                    os.path.join(ROOT, "tests", "c", "host_logic_driver.c"), os.path.join(ROOT, "tests", "c", "mock_hip.c"),
                    "-o", exe, "-L", libdir, "-lpgv_host", "-L", oradir, "-loracle", "-lm", "-lpthread",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath," + oradir], check=True)
    # the HNSW build's graph updates: on the "device" (the default: pgv_hnsw_link_*, whose replay -- csrc/hnsw_link_core.h --
    # the stand-in compiles from the source the GPU kernel is compiled from), the host-side replay on OpenMP threads
    # (PGV_HNSW_HOST_LINK=1), and that one with the new elements' SelectNeighbors on the host too (PGV_HNSW_HOST_SELECT=1):
    # every form must build the oracle's graph at batch 1 and pass the same scenarios
    for extra in ({}, {"PGV_HNSW_HOST_LINK": "1"}, {"PGV_HNSW_HOST_SELECT": "1"}):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, **extra))
        assert r.returncode == 0 and "HOST-LOGIC OK" in r.stdout, (extra, r.returncode, r.stdout[-500:], r.stderr[-2000:])

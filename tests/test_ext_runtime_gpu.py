"""The glue of ext/ executed on the real library: the same postmaster / backends / background worker of
tests/c/ext_driver.c (see tests/test_ext_runtime_cpu.py), linked against libpgv_hip.so -- k-means and assignment of the
build hooks on the device, mirrors staged by the worker process and imported by backend processes over hipIpc, the
pooler answering six backends from one pgv_search_batch, device state released when an ERROR longjmps out of a scan."""
import os
import subprocess

import pytest

from test_ext_runtime_cpu import build_driver

pytestmark = pytest.mark.gpu


def test_ext_glue_runs_on_the_gpu(tmp_path):
    exe = build_driver(str(tmp_path / "ext_driver_gpu"), [], extra_libs=["-lpgv_hip"])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    # the whole transcript where a gpurun call brings it back (the assertion below shows its tail only)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        with open(os.path.join(root, "gpurun_out", "ext_driver_gpu.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    assert r.returncode == 0 and "EXT-RUNTIME OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-4000:])
    assert "buffer refcount leak" not in r.stderr
    # one device here: the multi-device k-means' communicator path runs as a group of one through RCCL
    assert "k-means through a communicator of one (RCCL) = pgv_kmeans on the same stream" in r.stderr, r.stderr[-3000:]


REF_DRIVER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ext_driver_ref_gpu")


@pytest.mark.skipif(not os.path.exists(REF_DRIVER), reason="oracle/_ref/ext_driver_ref_gpu not built (the reference tree was absent)")
def test_the_references_own_gettuple_functions_drive_the_hooks_on_the_gpu():
    """oracle/_ref/ext_driver_ref_gpu (built by __graft_entry__.build() where the reference tree is mounted): the same
    program with the REFERENCE'S patched src/ivfscan.c, src/hnswscan.c and src/ivfkmeans.c, its src/ivfutils.c,
    src/hnswutils.c and src/vector.c linked in, against libpgv_hip.so.  Its phase "the
    reference's own ivfflatgettuple": the reference's scan code over the emulated pages = the oracle (vector.gpu off), and
    the hook lines inside ivfflatbeginscan / rescan / gettuple / endscan serving own-context, pooled and iterative scans
    from the real device (vector.gpu on).  Since round 5 the program also holds the reference's patched src/ivfinsert.c,
    src/ivfvacuum.c, src/hnswinsert.c, src/hnswvacuum.c, src/ivfbuild.c and src/hnswbuild.c: ivfflatbuild() and hnswbuild()
    themselves run CREATE INDEX over a stand-in heap with k-means, every argmin and the graph linking served by the device
    through the hook lines inside them (the CPU-side equalities with the oracle are tests/test_ext_runtime_cpu.py's)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([REF_DRIVER], capture_output=True, text=True, timeout=900, env=env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        with open(os.path.join(root, "gpurun_out", "ext_driver_ref_gpu.log"), "w") as f:
            f.write(r.stdout + "\n---- stderr ----\n" + r.stderr)
    assert r.returncode == 0 and "EXT-RUNTIME OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-4000:])
    assert any("the reference's own ivfflatgettuple" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert any("the reference's own hnswgettuple" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert any("the reference's own IvfflatKmeans" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert any("the reference's own ivfflatinsert" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert "removed by the reference's ivfflatbulkdelete" in r.stderr, r.stderr[-3000:]
    assert any("the reference's own hnswinsert" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert "removed by the reference's hnswbulkdelete" in r.stderr, r.stderr[-3000:]
    assert any("the reference's own ivfflatbuild" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert "argmins from the device, pages by the reference" in r.stderr, r.stderr[-3000:]
    assert any("the reference's own hnswbuild" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert "linked on the device in FlushPages, pages by the reference" in r.stderr, r.stderr[-3000:]
    # the device's batch-1 build handed to the reference's FlushPages = the reference's serial CPU build, byte for byte
    assert "vector_l2_ops: the hooks at vector.gpu_hnsw_build_batch = 1 hand FlushPages the reference's serial graph: 0 of" in r.stderr, r.stderr[-3000:]
    assert "the index the reference writes from it is the CPU build's, byte for byte" in r.stderr, r.stderr[-3000:]
    # vector.gpu_kmeans = off: Elkan's centers, the device's argmins -- no more than a row in a thousand may sit on a fence
    assert r.stderr.count("vector.gpu_kmeans = off -- Elkan's centers to the bit, the device's argmins:") == 4, r.stderr[-3000:]
    # halfvec_l2_ops on both access methods: the hooks take the real fp16 kernels
    assert any("the reference's own halfvec opclasses" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    assert "halfvec_l2_ops ivfflat with the hooks" in r.stderr and "halfvec_l2_ops hnsw with the hooks" in r.stderr
    # parallel CREATE INDEX: a leader and two workers, each with a device context of its own, flush their shares
    assert any("the reference's own parallel CREATE INDEX" in line and ": ok" in line for line in r.stderr.splitlines()), r.stderr[-3000:]
    lines = r.stderr.splitlines()
    par = lines[next(i for i, line in enumerate(lines) if "phase the reference's own halfvec opclasses" in line):]
    flushed = [int(line.split("path: ")[1].split()[0]) for line in par if "rows assigned on the device" in line]
    assert len(flushed) == 3 and sum(flushed) == 5539 and min(flushed) > 1800, flushed

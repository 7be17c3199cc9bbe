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

"""ext/ holds the glue of INTEGRATION.md as C a maintainer adds to the pgvector extension.  No PostgreSQL
headers exist in this environment: the files are type-checked against the stand-in declarations of ext/shim/
(every call into libpgv_hip must match include/pgv_hip.h exactly), their logic is exercised through the twins
over the emulated page image (pgvector_amd/host, tests/test_host_logic_cpu.py, the GPU tests)."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ext_glue_type_checks_against_the_abi():
    srcs = sorted(glob.glob(os.path.join(ROOT, "ext", "*.c")))
    assert len(srcs) >= 3
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-std=gnu11",
                        "-I" + os.path.join(ROOT, "ext", "shim"), "-I" + os.path.join(ROOT, "ext"),
                        "-I" + os.path.join(ROOT, "include")] + srcs, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_ext_glue_uses_only_the_public_abi():
    """nothing of this repository but include/pgv_hip.h is reachable from ext/ (no host-glue, no oracle)"""
    declared = set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "pgv_hip.h")).read()))
    for path in glob.glob(os.path.join(ROOT, "ext", "*.[ch]")):
        text = open(path).read()
        assert "pgv_host" not in text and "oracle" not in text, path
        for name in set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", text)):
            assert name in declared, (path, name)

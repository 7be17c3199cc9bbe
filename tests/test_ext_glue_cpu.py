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


# the one piece of host C the glue calls besides the ABI: the body of BuildGraph's insertion loop,
# which a maintainer adds to OBJS with ext/hnswbuild_gpu.c
HOST_BODY = {"pgv_host_hnsw_build", "pgv_host_hnsw_built_free", "pgv_host_last_error", "pgv_host_hnsw_set_cancel_check"}


def test_ext_glue_uses_only_the_public_abi():
    """nothing of this repository but include/pgv_hip.h is reachable from ext/ (no host-glue, no oracle) -- except
    hnswbuild_gpu.c's call of pgv_host_hnsw_build, declared in that file itself"""
    declared = set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "pgv_hip.h")).read()))
    host_h = open(os.path.join(ROOT, "pgvector_amd", "host", "pgv_host.h")).read()
    for path in glob.glob(os.path.join(ROOT, "ext", "*.[ch]")):
        text = open(path).read()
        allowed = HOST_BODY if path.endswith("hnswbuild_gpu.c") else set()
        assert "oracle" not in text, path
        assert '#include "pgv_host.h"' not in text, path
        for name in set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", text)):
            assert name in declared or name in allowed, (path, name)
        for name in allowed:
            # the declaration the glue carries must be the host header's own
            m = re.search(r"\b%s\s*\(([^;]*?)\);" % name, text, re.S)
            h = re.search(r"\b%s\s*\(([^;]*?)\);" % name, host_h, re.S)
            assert m and h and re.sub(r"\s+", "", m.group(1)) == re.sub(r"\s+", "", h.group(1)), name
        if allowed:
            # ... and so must the struct it fills
            def body(src):
                b = re.search(r"typedef struct pgv_hnsw_built\s*\{(.*?)\}\s*pgv_hnsw_built;", src, re.S).group(1)
                return re.sub(r"\s+", "", re.sub(r"/\*.*?\*/", "", b, flags=re.S))
            assert body(text) == body(host_h)

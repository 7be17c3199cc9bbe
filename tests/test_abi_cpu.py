"""CPU-side checks of the drop-in boundary: the library loads, exports every
symbol the header declares, and refuses to compute without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest

import pgvector_amd
from pgvector_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pgv_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libpgv_hip.so does not export %s" % n
    assert sorted(_lib.SYMBOLS) == names, "pgvector_amd/_lib.py SYMBOLS out of date with include/pgv_hip.h"


def test_abi_version():
    assert _lib.lib.pgv_abi_version() == 1


def test_product_does_not_reference_the_oracle():
    """the product must never route through oracle/ (it is test infrastructure)"""
    pkg = os.path.join(ROOT, "pgvector_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    assert not re.search(r"(import|from)\s+oracle|liboracle|pyoracle|#include\s*\"[^\"]*oracle", code), (f, line)


@pytest.mark.skipif(_lib.lib.pgv_device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(pgvector_amd.PgvError) as e:
        pgvector_amd.api.Context(0)
    assert e.value.code == _lib.PGV_ERR_DEVICE
    assert "no CPU path" in e.value.message

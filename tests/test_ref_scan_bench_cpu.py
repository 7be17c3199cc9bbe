"""oracle/_ref/ref_scan_bench_v*: the reference's OWN compiled scan (src/ivfscan.c + ivfutils.c + vector.c + halfvec.c +
halfutils.c, unpatched, built by oracle/Makefile `refbench`) as separate backend processes over an 8 KB page image --
the program bench.py's cpu_baseline leg times (kind "reference").  Here: the pages the PRODUCT'S page writer wrote,
read by the reference's scan code, give the oracle's answers (the third leg of the triangle product = oracle =
reference), for vector_l2_ops, vector_ip_ops and halfvec_l2_ops; and the timed phases count queries."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import CpuIvf, gen
from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_scan_bench_v3")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/ref_scan_bench_v3 not built (no reference tree)")


@pytest.mark.parametrize("ops,dtype,tname,oname", [(po.OPS_L2, po.ORA_F32, "f32", "l2"), (po.OPS_IP, po.ORA_F32, "f32", "ip"),
                                                   (po.OPS_L2, po.ORA_F16, "f16", "l2")])
def test_the_references_scan_over_the_products_pages_answers_like_the_oracle(tmp_path, ops, dtype, tname, oname):
    from pgvector_amd import _host
    ora = po.Oracle()
    n, dim, lists, probes, k, nq = 6000, 96, 24, 5, 10, 40
    data = gen(n, dim, seed=11, dist="clustered", dtype=dtype, clusters=24)
    ivf = CpuIvf(ora, ops, dtype, data, lists)
    rel = _host.Relation()
    rel.write_index(0 if dtype == po.ORA_F32 else 1, ivf.centers, ivf.list_offsets, ivf.vectors, ivf.tids)
    pages = np.ctypeslib.as_array((_host.C.c_uint8 * (rel.nblocks * 8192)).from_address(rel.rel.pages))
    pages.tofile(str(tmp_path / "pages.bin"))
    queries = gen(nq, dim, seed=12, dist="clustered", dtype=dtype, clusters=24)
    queries.tofile(str(tmp_path / "queries.bin"))
    r = subprocess.run([EXE, str(tmp_path / "pages.bin"), str(tmp_path / "queries.bin"), str(dim), str(nq), str(probes), str(k),
                        "3", "1.0", "0.5", str(tmp_path / "answers.bin"), tname, oname], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["procs"] == 3 and rec["queries"] > 0 and rec["single_queries"] > 0 and rec["qps"] > 0 and rec["single_qps"] > 0
    assert rec["blocks"] == rel.nblocks and rec["isa"] == "x86-64-v3"
    got = np.fromfile(str(tmp_path / "answers.bin"), dtype=np.uint64).reshape(nq, k)
    for i, q in enumerate(queries):
        wt, wd = ora.search(ivf.struct, q, probes, k)
        # the reference returns TIDs only: order must be the oracle's wherever its distances differ beyond float ties
        wd = np.asarray(wd, dtype=np.float64)
        j = 0
        while j < len(wt):
            e = j + 1
            while e < len(wt) and abs(wd[e] - wd[e - 1]) <= 4e-5 * max(abs(wd[e]), 1e-30):
                e += 1
            if e == len(wt):
                assert set(got[i][:j].tolist()) == set(wt[:j].tolist())
                break
            assert sorted(got[i][j:e].tolist()) == sorted(wt[j:e].tolist()), (i, j, e)
            j = e

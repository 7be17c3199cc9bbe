"""ext/pgvector-0.8.6-gpu.patch is the outer boundary as a maintainer would receive it: the hook lines of ext/pgv_gpu.h
as a patch against pgvector v0.8.6.  Where the reference tree is mounted (/root/reference: this container, not the GPU
box) the test
  * applies the committed patch to a fresh copy of the reference (`patch -p1`, no fuzz, no rejects),
  * checks that it is what ext/make_patch.py generates (the patch is not edited by hand),
  * compiles (-fsyntax-only -Wall -Werror) the PATCHED reference files -- ivfscan.c ivfbuild.c ivfkmeans.c ivfinsert.c
    ivfvacuum.c hnswscan.c hnswbuild.c hnswinsert.c hnswvacuum.c vector.c, i.e. the reference's own ivfflatgettuple,
    BuildCallback, InsertTuple ... with the hooks in them -- and the glue ext/*.c against the patched reference's OWN
    ivfflat.h / hnsw.h (not the trimmed copies under ext/shim/), with ext/shim/ standing in for the server headers
    (declarations only: pgshim.h + pgshim_ref.h).
Nothing of the reference is copied into the repository: the tree is read where it lies and patched in a temp dir."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "ext", "pgvector-0.8.6-gpu.patch")
PATCHED = ["ivfscan", "ivfbuild", "ivfkmeans", "ivfinsert", "ivfvacuum", "hnswscan", "hnswbuild", "hnswinsert", "hnswvacuum",
           "vector"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference tree is not mounted here")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    d = tmp_path_factory.mktemp("pgvector_patched")
    shutil.copytree(os.path.join(REF, "src"), d / "src")
    shutil.copy(os.path.join(REF, "Makefile"), d / "Makefile")
    r = subprocess.run(["patch", "-p1", "--fuzz=0", "--no-backup-if-mismatch", "-i", PATCH], cwd=d, capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert not glob.glob(str(d / "**" / "*.rej"), recursive=True)
    return d


def test_the_patch_applies_to_the_reference_and_touches_every_hook_site(patched):
    text = open(PATCH).read()
    for f in ("Makefile", "src/vector.c", "src/ivfflat.h", "src/ivfscan.c", "src/ivfbuild.c", "src/ivfkmeans.c",
              "src/ivfinsert.c", "src/ivfvacuum.c", "src/hnsw.h", "src/hnswscan.c", "src/hnswbuild.c", "src/hnswinsert.c",
              "src/hnswvacuum.c"):
        assert "+++ b/%s\n" % f in text, f
    gettuple = open(patched / "src" / "ivfscan.c").read()
    for call in ("PgvIvfflatBeginScan(index, so)", "PgvIvfflatRescan(so->gpu)", "PgvIvfflatGetTuple(scan)",
                 "PgvIvfflatAlreadyReturned(so->gpu, heaptid)", "PgvIvfflatEndScan(so->gpu)"):
        assert call in gettuple, call
    build = open(patched / "src" / "hnswbuild.c").read()
    assert "if (!PgvHnswBuildDefer(buildstate, element))\n\t\tInsertTupleInMemory(buildstate, element);" in build
    # every Pgv* function the patch calls is declared in ext/pgv_gpu.h and defined in ext/*.c
    import re
    called = set(re.findall(r"^\+.*?\b(Pgv[A-Za-z]+)\(", text, re.M))
    header = open(os.path.join(ROOT, "ext", "pgv_gpu.h")).read()
    glue = "".join(open(p).read() for p in glob.glob(os.path.join(ROOT, "ext", "*.c")))
    assert len(called) >= 15
    for name in called:
        assert re.search(r"\b%s\(" % name, header), name
        assert re.search(r"^%s\(" % name, glue, re.M), name


def test_the_committed_patch_is_what_make_patch_generates(tmp_path):
    out = tmp_path / "ext"
    out.mkdir()
    shutil.copy(os.path.join(ROOT, "ext", "make_patch.py"), out / "make_patch.py")
    r = subprocess.run([sys.executable, str(out / "make_patch.py"), REF], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(out / "pgvector-0.8.6-gpu.patch").read() == open(PATCH).read()


def _syntax(patched, srcs):
    return subprocess.run(["gcc", "-fsyntax-only", "-std=gnu11", "-Wall", "-Werror", "-I" + str(patched / "src"),
                           "-I" + os.path.join(ROOT, "ext"), "-I" + os.path.join(ROOT, "ext", "shim"),
                           "-I" + os.path.join(ROOT, "include")] + srcs, capture_output=True, text=True)


def test_the_patched_reference_files_compile_with_their_hooks(patched):
    r = _syntax(patched, [str(patched / "src" / (f + ".c")) for f in PATCHED])
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_glue_compiles_against_the_patched_references_own_headers(patched):
    srcs = sorted(glob.glob(os.path.join(ROOT, "ext", "*.c")))
    r = _syntax(patched, srcs)
    assert r.returncode == 0, r.stderr[-4000:]
    # ... and it was the reference's ivfflat.h / hnsw.h that were read, not the trimmed stand-ins
    deps = subprocess.run(["gcc", "-MM", "-std=gnu11", "-I" + str(patched / "src"), "-I" + os.path.join(ROOT, "ext"),
                           "-I" + os.path.join(ROOT, "ext", "shim"), "-I" + os.path.join(ROOT, "include")] + srcs,
                          capture_output=True, text=True).stdout
    assert str(patched / "src" / "ivfflat.h") in deps and str(patched / "src" / "hnsw.h") in deps
    assert os.path.join("ext", "shim", "ivfflat.h") not in deps and os.path.join("ext", "shim", "hnsw.h") not in deps

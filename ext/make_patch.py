#!/usr/bin/env python3
"""Generates ext/pgvector-0.8.6-gpu.patch: the hook lines of ext/pgv_gpu.h as a patch that APPLIES to pgvector v0.8.6.

    python ext/make_patch.py [/path/to/pgvector-0.8.6]        (default /root/reference; never modified, never vendored)

Each edit below is (file, text that must occur exactly once in the reference, replacement).  The reference tree is
copied to a temporary a/ and b/, b/ is edited, `diff -ruN a b` is the patch (-p1).  tests/test_ext_patch_cpu.py applies
the committed patch to a fresh copy of the reference and compiles the PATCHED reference files together with ext/*.c."""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
INC = '#include "pgv_gpu.h"\n'

EDITS = [
    # ------------------------------------------------------------------------------------------------ Makefile
    ("Makefile",
     "src/ivfutils.o src/ivfvacuum.o src/sparsevec.o src/vector.o\n",
     "src/ivfutils.o src/ivfvacuum.o src/sparsevec.o src/vector.o\n"
     "\n"
     "# MI355X distance path: the glue of ext/ (copied into src/) over libpgv_hip.so / libpgv_host.so\n"
     "PGV_AMD ?= /opt/pgvector_amd\n"
     "OBJS += src/pgv_context.o src/ivfscan_gpu.o src/ivfbuild_gpu.o src/hnswscan_gpu.o src/hnswbuild_gpu.o\n"
     "PG_CPPFLAGS += -I$(PGV_AMD)/include\n"
     "SHLIB_LINK += -L$(PGV_AMD)/lib -Wl,-rpath,$(PGV_AMD)/lib -lpgv_host -lpgv_hip\n"),
    # ------------------------------------------------------------------------------------------------ _PG_init
    ("src/vector.c", '#include "vector.h"\n', '#include "vector.h"\n' + INC),
    ("src/vector.c", "\tIvfflatInit();\n}\n", "\tIvfflatInit();\n\tPgvGpuInit();\n}\n"),
    # ------------------------------------------------------------------------------------------------ ivfflat.h
    ("src/ivfflat.h",
     "\tListInfo   *listInfo;\n",
     "\tListInfo   *listInfo;\n\tvoid\t   *gpu;\t\t\t/* PgvIvfBuild of ivfbuild_gpu.c, NULL: the CPU path */\n"),
    ("src/ivfflat.h",
     "\tIvfflatScanList *lists;\n}\t\t\tIvfflatScanOpaqueData;\n",
     "\tIvfflatScanList *lists;\n\n\tvoid\t   *gpu;\t\t\t/* PgvIvfScan of ivfscan_gpu.c, NULL: the CPU path */\n}\t\t\tIvfflatScanOpaqueData;\n"),
    # ------------------------------------------------------------------------------------------------ ivfscan.c
    ("src/ivfscan.c", '#include "ivfflat.h"\n', '#include "ivfflat.h"\n' + INC),
    ("src/ivfscan.c",
     "\tMemoryContextSwitchTo(oldCtx);\n\n\tscan->opaque = so;\n",
     "\tMemoryContextSwitchTo(oldCtx);\n\n\tso->gpu = PgvIvfflatBeginScan(index, so);\n\n\tscan->opaque = so;\n"),
    ("src/ivfscan.c",
     "\tso->first = true;\n\tpairingheap_reset(so->listQueue);\n\tso->listIndex = 0;\n",
     "\tso->first = true;\n\tpairingheap_reset(so->listQueue);\n\tso->listIndex = 0;\n\tPgvIvfflatRescan(so->gpu);\n"),
    ("src/ivfscan.c",
     "\tAssert(ScanDirectionIsForward(dir));\n\n\tif (so->first)\n\t{\n\t\tDatum\t\tvalue;\n\n\t\t/* Count index scan for stats */\n"
     "\t\tpgstat_count_index_scan(scan->indexRelation);\n",
     "\tAssert(ScanDirectionIsForward(dir));\n\n"
     "\t/* A scan the GPU path serves: 1 a tuple, 0 no more, -1 the rest of it is this function's */\n"
     "\tif (so->gpu && !so->first)\n\t{\n\t\tint\t\t\tr = PgvIvfflatGetTuple(scan);\n\n\t\tif (r >= 0)\n\t\t\treturn r != 0;\n\t}\n\n"
     "\tif (so->first)\n\t{\n\t\tDatum\t\tvalue;\n\n\t\t/* Count index scan for stats */\n"
     "\t\tpgstat_count_index_scan(scan->indexRelation);\n"),
    ("src/ivfscan.c",
     "\t\tvalue = GetScanValue(scan);\n\t\tIvfflatBench(\"GetScanLists\", GetScanLists(scan, value));\n",
     "\t\tvalue = GetScanValue(scan);\n\n"
     "\t\tif (so->gpu)\n\t\t{\n\t\t\tint\t\t\tr;\n\n"
     "\t\t\tso->value = value;\t/* the (normalised) query, where PgvIvfflatGetTuple reads it */\n"
     "\t\t\tr = PgvIvfflatGetTuple(scan);\n\t\t\tif (r >= 0)\n\t\t\t\treturn r != 0;\n\t\t}\n\n"
     "\t\tIvfflatBench(\"GetScanLists\", GetScanLists(scan, value));\n"),
    ("src/ivfscan.c",
     "\twhile (!tuplesort_gettupleslot(so->sortstate, true, false, so->mslot, NULL))\n\t{\n"
     "\t\tif (so->listIndex == so->maxProbes)\n\t\t\treturn false;\n\n"
     "\t\tIvfflatBench(\"GetScanItems\", GetScanItems(scan, so->value));\n\t}\n\n"
     "\theaptid = (ItemPointer) DatumGetPointer(slot_getattr(so->mslot, 2, &isnull));\n",
     "\tfor (;;)\n\t{\n"
     "\t\twhile (!tuplesort_gettupleslot(so->sortstate, true, false, so->mslot, NULL))\n\t\t{\n"
     "\t\t\tif (so->listIndex == so->maxProbes)\n\t\t\t\treturn false;\n\n"
     "\t\t\tIvfflatBench(\"GetScanItems\", GetScanItems(scan, so->value));\n\t\t}\n\n"
     "\t\theaptid = (ItemPointer) DatumGetPointer(slot_getattr(so->mslot, 2, &isnull));\n\n"
     "\t\t/* A pooled scan that was restaged under it restarted here: not twice what the GPU path gave out */\n"
     "\t\tif (so->gpu && PgvIvfflatAlreadyReturned(so->gpu, heaptid))\n\t\t\tcontinue;\n\t\tbreak;\n\t}\n"),
    ("src/ivfscan.c",
     "\tIvfflatScanOpaque so = (IvfflatScanOpaque) scan->opaque;\n\n\t/* Free any temporary files */\n",
     "\tIvfflatScanOpaque so = (IvfflatScanOpaque) scan->opaque;\n\n\tPgvIvfflatEndScan(so->gpu);\n\n\t/* Free any temporary files */\n"),
    # ------------------------------------------------------------------------------------------------ ivfkmeans.c
    ("src/ivfkmeans.c", '#include "ivfflat.h"\n', '#include "ivfflat.h"\n' + INC),
    ("src/ivfkmeans.c",
     "\telse\n\t\tElkanKmeans(index, samples, centers, typeInfo, memoryUsed);\n",
     "\telse if (!PgvIvfflatKmeans(index, samples, centers, typeInfo))\n\t\tElkanKmeans(index, samples, centers, typeInfo, memoryUsed);\n"),
    # ------------------------------------------------------------------------------------------------ ivfbuild.c
    ("src/ivfbuild.c", '#include "ivfflat.h"\n', '#include "ivfflat.h"\n' + INC),
    # AddTupleToSort: the argmin loop stays the CPU path; the tuplesort feed becomes IvfflatAddToSort (both use it)
    ("src/ivfbuild.c",
     "\t\tvalue = IvfflatNormValue(buildstate->typeInfo, buildstate->collation, value);\n\t}\n\n"
     "\t/* Find the list that minimizes the distance */\n",
     "\t\tvalue = IvfflatNormValue(buildstate->typeInfo, buildstate->collation, value);\n\t}\n\n"
     "\t/* The GPU path buffers the row; its argmin runs per batch (PgvIvfflatBuildFlush) */\n"
     "\tif (buildstate->gpu)\n\t{\n\t\tPgvIvfflatBuildAdd(buildstate, tid, value);\n\t\treturn;\n\t}\n\n"
     "\t/* Find the list that minimizes the distance */\n"),
    ("src/ivfbuild.c",
     "\tbuildstate->listCounts[closestCenter]++;\n#endif\n\n\t/* Create a virtual tuple */\n\tExecClearTuple(slot);\n"
     "\tslot->tts_values[0] = Int32GetDatum(closestCenter);\n",
     "\tbuildstate->listCounts[closestCenter]++;\n#endif\n\n\tIvfflatAddToSort(buildstate, closestCenter, tid, value);\n}\n\n"
     "/*\n * Add an assigned tuple to the sort\n */\nvoid\n"
     "IvfflatAddToSort(IvfflatBuildState * buildstate, int closestCenter, ItemPointer tid, Datum value)\n{\n"
     "\tTupleTableSlot *slot = buildstate->slot;\n\n\t/* Create a virtual tuple */\n\tExecClearTuple(slot);\n"
     "\tslot->tts_values[0] = Int32GetDatum(closestCenter);\n"),
    ("src/ivfbuild.c",
     "\tVectorArray centers = buildstate->centers;\n\tTupleTableSlot *slot = buildstate->slot;\n\n\t/* Detoast once for all calls */\n",
     "\tVectorArray centers = buildstate->centers;\n\n\t/* Detoast once for all calls */\n"),
    ("src/ivfbuild.c",
     "\tbuildstate->listInfo = palloc_array_checked(ListInfo, (Size) buildstate->lists);\n",
     "\tbuildstate->listInfo = palloc_array_checked(ListInfo, (Size) buildstate->lists);\n\tbuildstate->gpu = NULL;\n"),
    # a parallel participant (worker or leader-as-worker): its own batch buffer over the shared centers
    ("src/ivfbuild.c",
     "\tbuildstate.sortstate = ivfspool->sortstate;\n",
     "\tbuildstate.sortstate = ivfspool->sortstate;\n\tPgvIvfflatBuildBegin(&buildstate);\n"),
    ("src/ivfbuild.c",
     "\t\t\t\t\t\t\t\t\t   (void *) &buildstate, scan);\n\n\t/* Execute this worker's part of the sort */\n",
     "\t\t\t\t\t\t\t\t\t   (void *) &buildstate, scan);\n\tPgvIvfflatBuildFlush(&buildstate);\n\n"
     "\t/* Execute this worker's part of the sort */\n"),
    # the serial scan
    ("src/ivfbuild.c",
     "\t\telse\n\t\t\tbuildstate->reltuples = table_index_build_scan(buildstate->heap, buildstate->index, buildstate->indexInfo,\n"
     "\t\t\t\t\t\t\t\t\t\t\t\t\t\t   true, true, BuildCallback, (void *) buildstate, NULL);\n",
     "\t\telse\n\t\t{\n\t\t\tPgvIvfflatBuildBegin(buildstate);\n"
     "\t\t\tbuildstate->reltuples = table_index_build_scan(buildstate->heap, buildstate->index, buildstate->indexInfo,\n"
     "\t\t\t\t\t\t\t\t\t\t\t\t\t\t   true, true, BuildCallback, (void *) buildstate, NULL);\n"
     "\t\t\tPgvIvfflatBuildFlush(buildstate);\n\t\t}\n"),
    ("src/ivfbuild.c",
     "\tBuildIndex(heap, index, indexInfo, &buildstate, MAIN_FORKNUM);\n\n\tresult = palloc_object(IndexBuildResult);\n"
     "\tresult->heap_tuples = buildstate.reltuples;\n",
     "\tBuildIndex(heap, index, indexInfo, &buildstate, MAIN_FORKNUM);\n\tPgvNoteIndexChange(index);\t/* a rebuilt index is a new image */\n\n"
     "\tresult = palloc_object(IndexBuildResult);\n\tresult->heap_tuples = buildstate.reltuples;\n"),
    # ------------------------------------------------------------------------------------------------ insert / vacuum
    ("src/ivfinsert.c", '#include "ivfflat.h"\n', '#include "ivfflat.h"\n' + INC),
    ("src/ivfinsert.c",
     "\tInsertTuple(index, values, isnull, heap_tid);\n",
     "\tInsertTuple(index, values, isnull, heap_tid);\n\tPgvNoteIndexChange(index);\t/* device mirrors staged before now are stale */\n"),
    ("src/ivfvacuum.c", '#include "ivfflat.h"\n', '#include "ivfflat.h"\n' + INC),
    ("src/ivfvacuum.c",
     "\tFreeAccessStrategy(bas);\n\n\treturn stats;\n",
     "\tFreeAccessStrategy(bas);\n\n\tif (stats->tuples_removed > 0)\n\t\tPgvNoteIndexChange(index);\n\n\treturn stats;\n"),
    # ------------------------------------------------------------------------------------------------ hnsw.h
    ("src/hnsw.h",
     "\tHnswSupport support;\n}\t\t\tHnswScanOpaqueData;\n",
     "\tHnswSupport support;\n\n\tvoid\t   *gpu;\t\t\t/* PgvHnswScan of hnswscan_gpu.c, NULL: the CPU path */\n}\t\t\tHnswScanOpaqueData;\n"),
    ("src/hnsw.h",
     "\tchar\t   *hnswarea;\n}\t\t\tHnswBuildState;\n",
     "\tchar\t   *hnswarea;\n\n\tvoid\t   *gpu;\t\t\t/* PgvHnswBuild of hnswbuild_gpu.c, NULL: the CPU path */\n}\t\t\tHnswBuildState;\n"),
    # ------------------------------------------------------------------------------------------------ hnswscan.c
    ("src/hnswscan.c", '#include "hnsw.h"\n', '#include "hnsw.h"\n' + INC),
    ("src/hnswscan.c",
     "\tscan->opaque = so;\n\n\treturn scan;\n",
     "\tso->gpu = PgvHnswBeginScan(index);\n\n\tscan->opaque = so;\n\n\treturn scan;\n"),
    ("src/hnswscan.c",
     "\t\tso->w = GetScanItems(scan, value);\n\n\t\t/* Release shared lock */\n",
     "\t\t/* false: a NULL query, an iterative scan, no current mirror -- the reference's walk */\n"
     "\t\tif (!(so->gpu && PgvHnswGetScanItems(scan, value, &so->w)))\n\t\t\tso->w = GetScanItems(scan, value);\n\n"
     "\t\t/* Release shared lock */\n"),
    ("src/hnswscan.c",
     "\tHnswScanOpaque so = (HnswScanOpaque) scan->opaque;\n\n\tMemoryContextDelete(so->tmpCtx);\n",
     "\tHnswScanOpaque so = (HnswScanOpaque) scan->opaque;\n\n\tPgvHnswEndScan(so->gpu);\n\n\tMemoryContextDelete(so->tmpCtx);\n"),
    # ------------------------------------------------------------------------------------------------ hnswbuild.c
    ("src/hnswbuild.c", '#include "hnsw.h"\n', '#include "hnsw.h"\n' + INC),
    ("src/hnswbuild.c",
     "#ifdef HNSW_MEMORY\n\telog(INFO, \"memory: %zu MB\", buildstate->graph->memoryUsed / (1024 * 1024));\n#endif\n\n\tCreateMetaPage(buildstate);\n",
     "#ifdef HNSW_MEMORY\n\telog(INFO, \"memory: %zu MB\", buildstate->graph->memoryUsed / (1024 * 1024));\n#endif\n\n"
     "\t/* Link what the GPU path deferred (nothing in a parallel build or on the CPU path) */\n\tPgvHnswBuildLink(buildstate);\n\n"
     "\tCreateMetaPage(buildstate);\n"),
    ("src/hnswbuild.c",
     "\t/* Insert tuple */\n\tInsertTupleInMemory(buildstate, element);\n",
     "\t/* Insert tuple (false: not deferred -- the CPU path, or a participant of a parallel build) */\n"
     "\tif (!PgvHnswBuildDefer(buildstate, element))\n\t\tInsertTupleInMemory(buildstate, element);\n"),
    ("src/hnswbuild.c",
     "\tbuildstate->hnswleader = NULL;\n\tbuildstate->hnswshared = NULL;\n\tbuildstate->hnswarea = NULL;\n",
     "\tbuildstate->hnswleader = NULL;\n\tbuildstate->hnswshared = NULL;\n\tbuildstate->hnswarea = NULL;\n"
     "\tbuildstate->gpu = PgvHnswBuildBegin(buildstate);\n"),
    ("src/hnswbuild.c",
     "\tBuildIndex(heap, index, indexInfo, &buildstate, MAIN_FORKNUM);\n\n\tresult = palloc_object(IndexBuildResult);\n",
     "\tBuildIndex(heap, index, indexInfo, &buildstate, MAIN_FORKNUM);\n\tPgvNoteIndexChange(index);\n\n\tresult = palloc_object(IndexBuildResult);\n"),
    ("src/hnswinsert.c", '#include "hnsw.h"\n', '#include "hnsw.h"\n' + INC),
    ("src/hnswinsert.c",
     "\tHnswInsertTuple(index, values, isnull, heap_tid);\n\n\t/* Delete memory context */\n",
     "\tHnswInsertTuple(index, values, isnull, heap_tid);\n\tPgvNoteIndexChange(index);\n\n\t/* Delete memory context */\n"),
    ("src/hnswvacuum.c", '#include "hnsw.h"\n', '#include "hnsw.h"\n' + INC),
    ("src/hnswvacuum.c",
     "\tFreeVacuumState(&vacuumstate);\n\n\treturn vacuumstate.stats;\n",
     "\tFreeVacuumState(&vacuumstate);\n\tPgvNoteIndexChange(vacuumstate.index);\n\n\treturn vacuumstate.stats;\n"),
]


def apply_edits(root):
    for rel, old, new in EDITS:
        p = os.path.join(root, rel)
        s = open(p).read()
        if s.count(old) != 1:
            sys.exit("make_patch: %s: anchor occurs %d times (expected 1):\n%s" % (rel, s.count(old), old))
        open(p, "w").write(s.replace(old, new))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(HERE, "pgvector-0.8.6-gpu.patch")
    tmp = tempfile.mkdtemp(prefix="pgv_patch_")
    try:
        for side in ("a", "b"):
            os.makedirs(os.path.join(tmp, side))
            shutil.copytree(os.path.join(ref, "src"), os.path.join(tmp, side, "src"))
            shutil.copy(os.path.join(ref, "Makefile"), os.path.join(tmp, side, "Makefile"))
        apply_edits(os.path.join(tmp, "b"))
        r = subprocess.run(["diff", "-ruN", "a", "b"], cwd=tmp, capture_output=True, text=True)
        if r.returncode not in (0, 1):
            sys.exit(r.stderr)
        lines = [ln for ln in r.stdout.splitlines(True) if not ln.startswith("diff -ruN")]
        # no timestamps in the file headers: the patch is the same bytes whenever it is regenerated
        text = "".join(ln.split("\t")[0] + "\n" if ln.startswith(("--- a/", "+++ b/")) else ln for ln in lines)
        open(out, "w").write(text)
        print("wrote %s (%d bytes, %d files)" % (out, len(text), text.count("\n+++ b/") + text.startswith("+++ b/")))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

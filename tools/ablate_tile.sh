#!/bin/bash
# Where does the list-scan kernel's time go?  Builds two ablated copies of libpgv_hip.so next to
# the real one (never installed as the product library) and runs the headline bench on each:
#   ablate 1: rows are scored but only the first tile of a task is streamed  -> compute-only time
#   ablate 2: rows are streamed into LDS but never scored                    -> DMA-only time
#   ablate 3: as 1 and the distances are never stored; ablate 4: full kernel without the stores
# Results of ablated builds are wrong by construction; only roofline.avg_launch_ms is meaningful.
# usage (GPU box, repo root): tools/ablate_tile.sh [extra bench.py args]
set -u
root=$(pwd)
tag=$(echo "${PGV_EXTRA:-}" | tr -cd "A-Za-z0-9")
[ -n "${PGV_EXTRA:-}" ] && export PGV_HIP_LIB_FULL=$root/build/ablate0$tag/libpgv_hip.so && make -s -C pgvector_amd/csrc OBJDIR=../../build/ablate0$tag LIB=../../build/ablate0$tag/libpgv_hip.so EXTRA="$PGV_EXTRA" > /dev/null
for v in 1 2 3 4; do
    make -s -C pgvector_amd/csrc OBJDIR=../../build/ablate$v$tag LIB=../../build/ablate$v$tag/libpgv_hip.so \
        EXTRA="-DPGV_TILE_ABLATE=$v ${PGV_EXTRA:-}" > /dev/null || exit 1
done
run() {
    PGV_HIP_LIB=$1 python bench.py --no-cpu-baseline --recall-queries 8 "${@:2}" 2>/dev/null |
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('scan kernel avg %.3f ms/launch' % d['roofline']['avg_launch_ms'])"
}
echo -n "full:         "; run "${PGV_HIP_LIB_FULL:-}" "$@"
echo -n "compute-only: "; run $root/build/ablate1$tag/libpgv_hip.so "$@"
echo -n "dma-only:     "; run $root/build/ablate2$tag/libpgv_hip.so "$@"
echo -n "compute-only, no stores: "; run $root/build/ablate3$tag/libpgv_hip.so "$@"
echo -n "full, no stores:         "; run $root/build/ablate4$tag/libpgv_hip.so "$@"

#!/bin/bash
# builds tools/host_build_bench.c against the CPU stand-in of tests/c/mock_hip.c and runs it
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build
gcc -O2 -rdynamic -I$root/include -I$root/pgvector_amd/host $root/tools/host_build_bench.c $root/tests/c/mock_hip.c \
    -o $root/build/host_build_bench -L$root/pgvector_amd/lib -lpgv_host -lm -Wl,-rpath,$root/pgvector_amd/lib
exec $root/build/host_build_bench "$@"

#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (GPU sanitizers are not available on this pool): the host-only C++ API
# test (cpu backend, grids, concepts, TDV strategy protocol) and the plain-C oracle under its golden tests.
# The sanitized oracle replaces oracle/liboracle.so for the run and is rebuilt afterwards.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
g++ -std=c++20 -O1 -g -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -Wno-unknown-pragmas \
    -Iinclude -Iinclude/compat -Istencilstream_amd/csrc -Itests/cpp tests/cpp/host_api_test.cpp -o /tmp/host_api_test_san
ASAN_OPTIONS=detect_leaks=1 /tmp/host_api_test_san
gcc -O1 -g -fopenmp -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC \
    oracle/stencil_oracle.c -o oracle/liboracle.so -lm
trap 'rm -f oracle/liboracle.so; make -s -C oracle liboracle.so' EXIT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    python -m pytest tests/test_oracle_golden.py -x -q

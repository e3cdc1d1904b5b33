#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST library (parser, circuit, level packing, shard rule, preprocessor,
# client keys, radix planning) under the CPU test-suite.  GPU sanitizers are not available on the pool; the device code has
# its own debug build (libhelm_hip_check.so, tests/test_gpu_bounds_check.py).  Builds the sanitized library to a scratch file
# and puts the Makefile's back afterwards.   usage: bash tools/sanitize_host.sh      (from the repository root, ~2 min)
set -e
cd "$(dirname "$0")/../helm_amd/csrc"
cp libhelm_host.so /tmp/libhelm_host.so.orig
trap 'cp /tmp/libhelm_host.so.orig libhelm_host.so; touch libhelm_host.so' EXIT
g++ -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -march=x86-64-v3 -fopenmp -std=c++17 -fPIC -shared -o libhelm_host.so \
    helm_client.cpp helm_client64.cpp helm_client_wop.cpp host/*.cpp -L. -lhelm_hip -Wl,-rpath,'$ORIGIN'
cd ../..
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests -x -q -m "not gpu" --deselect tests/test_distributed_cpu.py --deselect tests/test_comm_handshake.py \
    --deselect tests/test_bench_launcher.py   # (multi-process tests start children that do not inherit the preload cleanly)

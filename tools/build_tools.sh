#!/bin/bash
# Builds the native measurement tools of tools/ into tools/bin/ (git-ignored; travels to the GPU box with the tree).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
g++ -std=c++17 -O2 -pthread -Iinclude tools/concurrent_bench.cpp -Lcircl_amd -lcirclhip \
    -Wl,-rpath,'$ORIGIN/../../circl_amd' -Wl,-rpath,/opt/rocm/lib -o tools/bin/concurrent_bench
echo built tools/bin/concurrent_bench

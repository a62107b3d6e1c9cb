#!/bin/bash
# A/B of the conv-GEMM variants on the bench's main shapes (one process per shape, all variants inside it)
set -e
for args in "--B 256 --T 64 --cin 1024 --cout 2048 --k 3" "--B 256 --T 128 --cin 1024 --cout 1024 --k 1" \
            "--B 256 --T 128 --cin 1024 --cout 1024 --k 3" "--B 256 --T 128 --cin 1024 --cout 3072 --k 1" \
            "--B 256 --T 128 --cin 512 --cout 512 --k 5" "--B 256 --T 32 --cin 1024 --cout 2048 --k 3" \
            "--B 256 --T 64 --cin 4096 --cout 4096 --k 1"; do
  timeout -k 10 120 python tools/gemm_bench.py $args "$@"
done

#!/bin/bash
# sample sclk / power with rocm-smi while one GEMM variant loops for a few seconds
v=${1:-dma}; shift
timeout -k 10 100 python tools/gemm_bench.py --variants $v --iters ${ITERS:-100000} "$@" &
pid=$!
sleep 12
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr '\n' ' '; echo
  sleep 1
done
wait $pid

#!/bin/bash
# Sweep the split-K plan knobs of the weight-gradient kernel and the number of side streams on the bench workload.
# usage (on the GPU box): bash tools/sweep_wgrad.sh "256 13 4" "64 13 4" ...   (ZS_WGRAD_WGS ZS_WGRAD_SLAB_COST ZS_SIDE_STREAMS)
for cfg in "$@"; do
  set -- $cfg
  echo "== ZS_WGRAD_WGS=$1 ZS_WGRAD_SLAB_COST=$2 ZS_SIDE_STREAMS=$3"
  ZS_WGRAD_WGS=$1 ZS_WGRAD_SLAB_COST=$2 ZS_SIDE_STREAMS=$3 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  frames/s %.0f' % (d['ms_per_step'], d['value']))"
done

#!/bin/bash
# Feasibility probe for micro-batch pipelining: do two independent half-batch training steps, issued from two processes,
# overlap on one MI355X (GEMM phases of one beside the GRU / elementwise phases of the other)?
# Compare 2 x (B=128) concurrently against 1 x (B=256).
S=${1:-300}
echo "== one process, B=256"
python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  frames/s %.0f' % (d['ms_per_step'], d['value']))"
echo "== one process, B=128"
python bench.py --batch 128 --steps $S --warmup 5 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  frames/s %.0f' % (d['ms_per_step'], d['value']))"
echo "== two processes, B=128 each, concurrently"
python bench.py --batch 128 --steps $((S*2)) --warmup 5 --no-cpu-baseline --no-kernel-events > /tmp/p1.json 2>/dev/null &
P1=$!
python bench.py --batch 128 --steps $((S*2)) --warmup 5 --no-cpu-baseline --no-kernel-events > /tmp/p2.json 2>/dev/null &
P2=$!
wait $P1 $P2
for f in /tmp/p1.json /tmp/p2.json; do python -c "import sys,json; d=json.loads(open('$f').read()); print('ms_per_step %.3f  frames/s %.0f' % (d['ms_per_step'], d['value']))"; done

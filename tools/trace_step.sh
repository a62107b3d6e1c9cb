set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace1; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 12 --warmup 3 > $O/train.json 2>/dev/null
cd $R
T=$(find $O/train -name "*kernel_trace.csv" | head -1)
python tools/step_chain.py $T > gpurun_out/chain1.txt
python tools/step_timeline.py $T > gpurun_out/timeline1.txt
rm -rf $O
cat gpurun_out/timeline1.txt

set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace2; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/g -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/e -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 10 --warmup 3 --no-graph > $O/e.json 2>/dev/null
cd $R
echo GRAPH; python tools/busy_time.py $(find $O/g -name "*kernel_trace.csv" | head -1)
echo EAGER; python tools/busy_time.py $(find $O/e -name "*kernel_trace.csv" | head -1); python -c "import json;print(json.load(open('$O/e.json'))['ms_per_step'])"
rm -rf $O

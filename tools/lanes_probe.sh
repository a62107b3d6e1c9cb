set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace3; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
ZS_LANES=2 rocprofv3 --kernel-trace --output-format csv -d $O/e -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 8 --warmup 3 --no-graph > $O/e.json 2>/dev/null
cd $R
echo "EAGER LANES=2"; python tools/busy_time.py $(find $O/e -name "*kernel_trace.csv" | head -1); python -c "import json;d=json.load(open('$O/e.json'));print(d['ms_per_step'], d['host_enqueue_ms_per_step'])"
rm -rf $O

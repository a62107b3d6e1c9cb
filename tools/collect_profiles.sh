#!/bin/bash
# Round-2 profile collection on the GPU box (writes under gpurun_out/prof_r02; copy the summaries into profiles/ afterwards).
# Counter passes are separate runs with --kernel-trace only, as the MI355X guide prescribes.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 20 --warmup 3 > $O/train.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/train_f -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/train_w -- python3 $R/bench.py --no-cpu-baseline --no-kernel-events --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/resynth -- python3 $R/bench.py --mode resynth --no-cpu-baseline --steps 2 --warmup 1 > $O/resynth.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/resynth_f -- python3 $R/bench.py --mode resynth --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/resynth_w -- python3 $R/bench.py --mode resynth --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stage2 -- python3 $R/tools/stage2_bench.py --iters 1 --tgat > $O/stage2.txt 2> /dev/null
cd $R
mkdir -p gpurun_out/r02_profiles
cp $(find $O/train -name "*kernel_stats.csv" | head -1) gpurun_out/r02_profiles/r02_bench_bf16_kernel_stats.csv
python tools/step_timeline.py $(find $O/train -name "*kernel_trace.csv" | head -1) > gpurun_out/r02_profiles/r02_step_timeline.txt
python tools/pmc_traffic.py $(find $O/train_f -name "*counter_collection.csv" | head -1) $(find $O/train_w -name "*counter_collection.csv" | head -1) gpurun_out/r02_profiles/r02_pmc_traffic.json > gpurun_out/r02_profiles/r02_pmc_traffic.txt
cp $(find $O/resynth -name "*kernel_stats.csv" | head -1) gpurun_out/r02_profiles/r02_resynth_kernel_stats.csv
python tools/pmc_traffic.py $(find $O/resynth_f -name "*counter_collection.csv" | head -1) $(find $O/resynth_w -name "*counter_collection.csv" | head -1) gpurun_out/r02_profiles/r02_resynth_pmc_traffic.json > gpurun_out/r02_profiles/r02_resynth_pmc_traffic.txt
cp $(find $O/stage2 -name "*kernel_stats.csv" | head -1) gpurun_out/r02_profiles/r02_stage2_kernel_stats.csv
cp $O/train.json gpurun_out/r02_profiles/r02_bench_train_under_rocprof.json
cp $O/resynth.json gpurun_out/r02_profiles/r02_bench_resynth_under_rocprof.json
cp $O/stage2.txt gpurun_out/r02_profiles/r02_stage2_bench.txt
rm -rf $O
ls -la gpurun_out/r02_profiles

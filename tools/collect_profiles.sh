#!/bin/bash
# Round-3 profile collection on the GPU box (writes under gpurun_out/prof_r03; copy the summaries into profiles/ afterwards).
# Counter passes are separate runs with --kernel-trace only, as the MI355X guide prescribes.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
P=$R/gpurun_out/r03_profiles
rm -rf $O $P; mkdir -p $O $P
cd /tmp; export TMPDIR=/tmp
B="--no-cpu-baseline --no-kernel-events --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py $B --steps 20 --warmup 3 > $O/train.json 2> /dev/null
echo "train trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/train_f -- python3 $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/train_w -- python3 $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2>&1
echo "train counters done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/resynth -- python3 $R/bench.py --mode resynth --no-cpu-baseline --steps 2 --warmup 1 > $O/resynth.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/resynth_f -- python3 $R/bench.py --mode resynth --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/resynth_w -- python3 $R/bench.py --mode resynth --no-cpu-baseline --steps 1 --warmup 1 > /dev/null 2>&1
echo "resynth done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dstep -- python3 $R/tools/dstep_bench.py 5 > $O/dstep.txt 2>&1
echo "dstep done"
cd $R
cp $(find $O/train -name "*kernel_stats.csv" | head -1) $P/r03_bench_bf16_kernel_stats.csv
python3 tools/step_timeline.py $(find $O/train -name "*kernel_trace.csv" | head -1) > $P/r03_step_timeline.txt
python3 tools/step_chain.py $(find $O/train -name "*kernel_trace.csv" | head -1) > $P/r03_step_chain.txt
python3 tools/pmc_traffic.py $(find $O/train_f -name "*counter_collection.csv" | head -1) $(find $O/train_w -name "*counter_collection.csv" | head -1) $P/r03_pmc_traffic.json > $P/r03_pmc_traffic.txt
cp $(find $O/resynth -name "*kernel_stats.csv" | head -1) $P/r03_resynth_kernel_stats.csv
python3 tools/resynth_trace.py $(find $O/resynth -name "*kernel_trace.csv" | head -1) 100 > $P/r03_resynth_batch.txt
python3 tools/pmc_traffic.py $(find $O/resynth_f -name "*counter_collection.csv" | head -1) $(find $O/resynth_w -name "*counter_collection.csv" | head -1) $P/r03_resynth_pmc_traffic.json > $P/r03_resynth_pmc_traffic.txt
cp $(find $O/dstep -name "*kernel_stats.csv" | head -1) $P/r03_stage2_dstep_kernel_stats.csv
python3 tools/kernel_stats_per.py $P/r03_stage2_dstep_kernel_stats.csv 7 30 > $P/r03_stage2_dstep_per_step.txt
grep "D step" $O/dstep.txt | tail -1 >> $P/r03_stage2_dstep_per_step.txt
cp $O/train.json $P/r03_bench_train_under_rocprof.json
cp $O/resynth.json $P/r03_bench_resynth_under_rocprof.json
rm -rf $O
ls -la $P

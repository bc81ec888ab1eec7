# Round 5 evidence in two gpurun calls (each under the 1200 s limit):
#   gpurun -- bash tools/r05_profile_all.sh bench   the driver's command un-profiled and under rocprofv3 --kernel-trace --stats
#   gpurun -- bash tools/r05_profile_all.sh sides   the PMC passes (tools/pmc_all.sh) and the side configurations
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
if [ "$1" = bench ]; then
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
echo "bench done: $(python3 -c "import json;d=json.load(open('$O/bench_driver_cmd.json'));print(d['value'], d['roofline']['frac'], d['roofline']['end_to_end_frac'])")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o ks -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd_under_rocprof.json 2> $O/rocprof.err
cd $R
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $F $O/bench_driver_cmd_kernel_stats.csv; rm -rf $O/kt
echo "rocprof stats done"; head -5 $O/bench_driver_cmd_kernel_stats.csv
else
bash tools/pmc_all.sh > $O/pmc_all.log 2>&1 || echo "pmc_all failed"
cp gpurun_out/pmc_final/traffic.json $O/traffic.json; cp gpurun_out/pmc_final/valu.csv $O/eval512_pmc_valu.csv; cp gpurun_out/pmc_final/mfma_busy_by_launch.csv $O/eval512_pmc_mfma_busy_by_launch.csv
echo "pmc done"
python3 bench.py --tasks 64 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_64fits.json 2>/dev/null
python3 bench.py --config c2 --steps 5 --warmup 1 > $O/bench_c2.json 2>/dev/null
python3 bench.py --config c5 --steps 3 --warmup 1 > $O/bench_c5.json 2>/dev/null
python3 tools/probe_c5_stages.py > $O/c5_stages.txt 2>&1
python3 tools/probe_latency.py > $O/probe_latency.txt 2>&1
python3 tools/run_c3.py 64 > $O/c3_full.txt 2>&1
python3 tools/run_c4.py 512 > $O/c4_full.txt 2>&1
python3 tools/probe_round_stages.py 2048 8 32 64 128 512 > $O/round_stages.txt 2>&1
echo "sides done"; cat $O/bench_64fits.json | head -c 300; echo; cat $O/c3_full.txt | tail -3; cat $O/c4_full.txt | head -2; cat $O/probe_latency.txt | tail -8
fi

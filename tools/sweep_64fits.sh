# 64 fits per GPU (the per-GPU share of the headline job on eight GPUs) against the tuning switches: fits/s and ms per step
set -e
for v in "X=1" "WV_GROUPS=2" "WV_BN_FILL=2048" "WV_BN_FILL=8192" "WV_SMALL_FILL=4096" "WV_SMALL_FILL=16384" "WV_SMALL_FILL=0" "WV_TRI_SPLIT_MIN=1" "X=2"; do
  echo -n "$v: " >> gpurun_out/r05_sweep_64fits.txt
  env $v python3 bench.py --tasks 64 --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],1))" >> gpurun_out/r05_sweep_64fits.txt
done
cat gpurun_out/r05_sweep_64fits.txt

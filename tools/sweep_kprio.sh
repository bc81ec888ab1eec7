# WV_KPRIO (long tiles of variable-K launches run their K loop at raised priority) against the default, alternating
O=gpurun_out/r05_kprio_ab.txt
rm -f $O
for v in "WV_KPRIO=0" "WV_KPRIO=1" "WV_KPRIO=0" "WV_KPRIO=1"; do
  echo "== $v" >> $O
  env $v python3 tools/probe_latency.py 2>&1 | grep -E "N= (2048|4096)" >> $O
  env $v python3 bench.py --tasks 64 --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('64 fits:', round(d['value'],2), 'fits/s', round(d['ms_per_step'],1), 'ms')" >> $O
done
cat $O

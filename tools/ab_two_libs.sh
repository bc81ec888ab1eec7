# A/B of two builds of the library inside one gpurun call (one box): gpurun_in/lib_base.so (copied before a change)
# against the in-tree build; alternating.  usage: bash tools/ab_two_libs.sh OUT.txt [full]
O=gpurun_out/${1:-ab_two_libs.txt}
rm -f $O
for rep in 1 2; do
for v in "WAVEOME_AMD_LIB=$PWD/gpurun_in/lib_base.so" "X=new"; do
  echo "== $v" >> $O
  env $v python3 tools/probe_latency.py 2>&1 | grep -E "N= (2048|4096)" >> $O
  env $v python3 bench.py --tasks 64 --batch 64 --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('64 fits:', round(d['value'],2), 'fits/s', round(d['ms_per_step'],1), 'ms')" >> $O
  if [ "$2" = full ]; then
  env $v python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-breakdown 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 fits:', round(d['value'],2), 'fits/s', round(d['ms_per_step'],1), 'ms')" >> $O
  fi
done
done
cat $O

O=gpurun_out/r05_solve_ab.txt; rm -f $O
for rep in 1 2; do
for v in "WAVEOME_AMD_LIB=$PWD/gpurun_in/lib_base.so" "X=new"; do
  echo "== $v" >> $O
  env $v python3 tools/probe_latency.py 2>&1 | grep -E "N= (4096)" >> $O
done; done
cat $O

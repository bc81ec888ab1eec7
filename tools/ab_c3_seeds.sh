# C3 (run-to-tolerance fits, N = 4096, 64 outputs) on three data seeds with two builds of the library on one box:
# gpurun_in/lib_base.so (the build before a change) against the in-tree build.  What it is for: a change of the last
# bits of any kernel re-rolls the paths of run-to-tolerance fits; this shows how much of a C3 wall-time difference
# between two builds is that, and how much is the kernels.   usage: bash tools/ab_c3_seeds.sh [OUT.txt]
O=gpurun_out/${1:-r05_c3_seeds_ab.txt}; rm -f $O
for seed in 0 1 2; do
for v in "WAVEOME_AMD_LIB=$PWD/gpurun_in/lib_base.so" "X=new"; do
  echo "== $v" >> $O
  env $v python3 tools/run_c3.py 64 $seed 2>&1 | grep "^C3" >> $O
done; done
cat $O

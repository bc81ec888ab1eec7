# kernel timelines of one evaluation with one fit in flight (N = 2048, 4096): gpurun -- bash tools/timeline_all.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/timeline; mkdir -p $O; cd $R
for N in 2048 4096; do
  rocprofv3 --kernel-trace --output-format csv -d $O/t$N -o tl -- python3 tools/timeline_one_fit.py run $N 1 > $O/run$N.log 2>&1
  F=$(find $O/t$N -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline_one_fit.py sum $F | head -40 > $O/timeline_one_fit_$N.txt
  rm -rf $O/t$N
done
head -22 $O/timeline_one_fit_2048.txt; head -22 $O/timeline_one_fit_4096.txt

# every sharded GPSearch path, one process vs 3 gloo ranks on the one GPU; the checksums of each pair must agree
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/rehearse_all.txt; : > $O
P=29700
for m in pen pen_iter pen_restart sweep search; do
  python tools/rehearse_ranks.py $m 2>/dev/null | grep "^$m" >> $O
  P=$((P+1))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port $P tools/rehearse_ranks.py $m 2>/dev/null | grep "rank 0/3" >> $O
done
cat $O

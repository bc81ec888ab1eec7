# one-fit / few-fit rounds against the tile-choice thresholds of pick_bn (wv_chol.hip); the probe prints WV_BN_FILL only,
# so every block is headed by its whole setting
for v in "X=1" "WV_SMALL_FILL=512" "WV_SMALL_FILL=1024" "WV_SMALL_FILL=2048" "WV_BN_FILL=256" "WV_BN_FILL=1024"; do
  echo "== $v" >> gpurun_out/r05_sweep_latency_tiles.txt
  env $v python3 tools/probe_latency.py 2>&1 | grep -E "N= (2048|4096)" >> gpurun_out/r05_sweep_latency_tiles.txt
done
cat gpurun_out/r05_sweep_latency_tiles.txt

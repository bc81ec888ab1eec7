import sys; sys.path.insert(0,'.')
from waveome_amd.engine import Context
c=Context(0)
for K in (16,32,64,128,256,512,1024):
    for beta_mode in (0,5):
        tf,ms=c.gemm_bench(1024,1024,K,True,True,128+1000*beta_mode,128,3)
        tiles=64*128
        per_tile_us = ms*1e3*512/tiles   # 512 concurrent WG slots
        print(f"K{K} mode{beta_mode}: {tf:6.1f} TF/s {ms*1e3:8.1f} us  -> {per_tile_us:6.2f} us per tile-slot, {per_tile_us/(K/16):5.2f} us/chunk", flush=True)

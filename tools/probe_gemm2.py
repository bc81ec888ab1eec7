import sys; sys.path.insert(0,'.')
from waveome_amd.engine import Context
c=Context(0)
for mode in (0,2,4,5,0,4):
    tf,ms=c.gemm_bench(1024,1024,1024,True,True,128+1000*mode,128,2)
    print(f"1024^3 x128 KK mode{mode} (0 full, 2 tri+beta, 4 tri only, 5 beta only): {tf:6.1f} TF/s {ms*1e3:9.1f} us", flush=True)

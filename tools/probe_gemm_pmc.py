import sys; sys.path.insert(0,'.')
from waveome_amd.engine import Context
c=Context(0)
for (M,N,K,nb,akc,bkc,mode) in [(1024,1024,1024,128,1,1,0),(1024,1024,1024,128,1,1,2),(2048,2048,2048,32,0,0,0),(2048,2048,2048,32,0,0,1)]:
    tf,ms=c.gemm_bench(M,N,K,bool(akc),bool(bkc),128+1000*mode,nb,1)
    print(f"M{M} K{K} nb{nb} akc{akc} mode{mode}: {tf:6.1f} TF/s {ms*1e3:9.1f} us", flush=True)

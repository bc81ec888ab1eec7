import sys; sys.path.insert(0,'.')
from waveome_amd.engine import Context
c=Context(0)
for (M,nb,akc,bkc,mode) in [(1024,256,1,0,6),(512,256,1,0,6),(1024,256,1,0,3)]:
    tf,ms=c.gemm_bench(M,M,M,bool(akc),bool(bkc),128+1000*mode,nb,3)
    print(f"M{M} nb{nb} akc{akc} bkc{bkc} mode{mode}: {tf:6.1f} TF/s {ms*1e3:8.1f} us", flush=True)

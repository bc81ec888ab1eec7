import sys; sys.path.insert(0,'.')
from waveome_amd.engine import Context
c=Context(0)
cfgs=[(128,128,2048,512),(128,128,8192,512),(2048,2048,2048,8),(2048,2048,2048,16),(1024,1024,1024,32),(512,512,512,64),(2048,2048,256,16),(4096,4096,4096,8)]
for (M,N,K,nb) in cfgs:
    for akc,bkc in ((True,True),(False,False)):
        tf,ms=c.gemm_bench(M,N,K,akc,bkc,128,nb,3)
        print(f"M{M} N{N} K{K} nb{nb} akc{int(akc)} bkc{int(bkc)}: {tf:6.1f} TF/s  {ms:8.3f} ms", flush=True)

import sys; sys.path.insert(0,'.')
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
c=Context(0)
print("mode 0 full, 1 lauum-like (lower tiles, contraction from the tile row), 2 syrk-like (lower tiles, beta = 1), 3 trimmed k < n0 + BN")
for (M,N,K,nb,akc,bkc,mode) in [(2048,2048,2048,32,0,0,0),(2048,2048,2048,32,0,0,1),(2048,2048,2048,128,0,0,1),
                               (1024,1024,1024,128,1,1,0),(1024,1024,1024,128,1,1,2),(1024,1024,1024,128,1,1,3),(1024,1024,1024,128,1,0,0),
                               (512,512,512,128,1,1,0),(512,512,512,128,1,1,2),(256,256,256,128,1,1,0)]:
    for bn in (128,):
        tf,ms=c.gemm_bench(M,N,K,bool(akc),bool(bkc),bn+1000*mode,nb,3)
        print(f"M{M} N{N} K{K} nb{nb} akc{akc} bkc{bkc} mode{mode} bn{bn}: {tf:6.1f} TF/s  {ms*1e3:9.1f} us", flush=True)

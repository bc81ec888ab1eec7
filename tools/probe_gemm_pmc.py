import sys; sys.path.insert(0,'.')
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
c=Context(0)
for (M,nb,akc,bkc,mode) in [(1024,256,0,0,0),(1024,256,0,0,3),(1024,256,0,0,2)]:
    tf,ms=c.gemm_bench(M,M,M,bool(akc),bool(bkc),128+1000*mode,nb,1)
    print(f"M{M} nb{nb} akc{akc} mode{mode}: {tf:6.1f} TF/s {ms*1e3:9.1f} us", flush=True)

import sys; sys.path.insert(0,'.')
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
c=Context(0)
for wps in (1,2,4):
    for mode in (3,4):
        print('wps',wps,'mode',mode,'(3=mfma 4x4x4 only, 4=8 mfma16+128 fma) valu/mfma TF', c.valu_f64_rate(20000,wps,mode), flush=True)

import sys; sys.path.insert(0,'.')
import numpy as np, ctypes as C
import _probes_lib  # noqa: F401  (diagnostic build: tools/libwv_probes.so)
from waveome_amd.engine import Context
from waveome_amd import _lib
c=Context(0)
t=np.zeros(4096,dtype=np.int32)
_lib.check(c.lib.wv_dbg_mfma4_probe(c.handle,_lib.as_ip(t)))
t=t.reshape(64,64)
np.save('gpurun_out/mfma4_table.npy',t)
print('multi-hit entries:', int((t>=64).sum()))
# for each A lane list the B lanes it pairs with and the D lane
for pa in range(0,64):
    pbs=np.nonzero(t[pa]>=0)[0]
    print(pa, [(int(pb), int(t[pa,pb])) for pb in pbs])

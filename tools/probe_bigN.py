import sys, time; sys.path.insert(0,'.')
import numpy as np
from oracle import gp_oracle as o
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem
N=int(sys.argv[1]); B=int(sys.argv[2])
rng=np.random.default_rng(5)
X=rng.uniform(-3,3,size=(N,4)); X=(X-X.mean(0))/X.std(0,ddof=1)
Y=np.stack([np.sin(2*X[:,0])+0.5*np.cos(X[:,1]*(1+p))+0.1*rng.normal(size=N) for p in range(2)],1)
kern=K.Sum([K.Matern32(active_dims=[d]) for d in range(4)])
ms=ModelSpec(kern,4,mean_function=K.ConstantMean())
ctx=Context(0); prob=Problem(ctx,X,Y,ms,max_batch=B)
print('workspace GB',prob.workspace_bytes/1e9,flush=True)
th=ms.theta(); th[ms.noise]=0.1
t0=time.perf_counter(); lml,st=prob.lml([0],th[None,:]); print('gpu lml',lml[0],st[0],time.perf_counter()-t0,flush=True)
t0=time.perf_counter(); ref=o.lml(ms.to_dict(),th,X,Y[:,0]); print('oracle lml',ref,time.perf_counter()-t0,'rel err',abs(lml[0]-ref)/abs(ref),flush=True)
cols=np.arange(B,dtype=np.int32)%2; U=np.tile(ms.unconstrain(),(B,1))
prob.value_grad(cols,U)
t0=time.perf_counter(); prob.value_grad(cols,U); dt=time.perf_counter()-t0
print(f'value+grad B={B}: {dt*1e3:.1f} ms -> {B*N**3/dt/1e12:.1f} TFLOP/s',flush=True)

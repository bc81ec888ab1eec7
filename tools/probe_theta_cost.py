import sys, time; sys.path.insert(0,'.')
import numpy as np, bench
from waveome_amd.engine import Context, Problem, lbfgs_opts
B=128
X,Y=bench.make_inputs(B); spec=bench.headline_spec(); ctx=Context(0)
prob=Problem(ctx,X,Y.T,spec,max_batch=B)
cols=np.arange(B,dtype=np.int32); U0=np.tile(spec.unconstrain(),(B,1))
def t(U,n=3):
    prob.value_grad(cols,U)
    t0=time.perf_counter()
    for _ in range(n): prob.value_grad(cols,U)
    return (time.perf_counter()-t0)/n*1e3
print('default theta ms', t(U0))
res=prob.fit(cols,U0,lbfgs_opts(maxiter=100,maxfun=100000,ftol=0,gtol=0))
print('fit evals', res['nfev'].sum(), 'nit', res['nit'].mean())
U1=res['u']
print('fitted theta ms', t(U1))
prob.timing(True); prob.timing_reset(); prob.value_grad(cols,U1); st=prob.stage_times(); prob.timing(False)
print({k:round(v['ms'],2) for k,v in st.items()})
prob.timing(True); prob.timing_reset(); prob.value_grad(cols,U0); st=prob.stage_times(); prob.timing(False)
print({k:round(v['ms'],2) for k,v in st.items()})
th=np.array([spec.constrain(u) for u in U1]); print('theta min/max per param', th.min(0), th.max(0))

import sys; sys.path.insert(0,'.')
import numpy as np
sys.path.insert(0,'tests')
from test_gpu_parity import _data, _with_priors
from oracle import gp_oracle as o
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, Problem, lbfgs_opts
ctx=Context(0)
N=200
X,Y=_data(11,N,P=7)
kern=K.SquaredExponential(active_dims=[0])+K.SquaredExponential(active_dims=[1])+K.Categorical(active_dims=[3])
ms=_with_priors(ModelSpec(kern,4,mean_function=K.ConstantMean()))
prob=Problem(ctx,X,Y,ms,max_batch=3)
cols=np.arange(7,dtype=np.int32); U0=np.tile(ms.unconstrain(),(7,1))
res=prob.fit(cols,U0,lbfgs_opts(maxiter=200))
d=ms.to_dict()
for t in range(7):
    ref=o.fit(d,X,Y[:,t],maxiter=200)
    print(t,'gpu loss %.10f nit %d nfev %d st %d | ref loss %.10f nit %d nfev %d %s'%(res['loss'][t],res['nit'][t],res['nfev'][t],res['status'][t],ref['loss'],ref['nit'],ref['nfev'],ref['message']))
    # evaluate GPU objective at ref optimum and oracle at gpu optimum
    l,g,s=prob.value_grad([t],ref['u'][None,:]); fo,go=o.objective(d,res['u'][t],X,Y[:,t])
    print('   gpu obj at ref opt %.10f (|g|max %.2e) ; oracle obj at gpu opt %.10f (|g|max %.2e)'%(l[0],np.abs(g).max(),fo,np.abs(go).max()))

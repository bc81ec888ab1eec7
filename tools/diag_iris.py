import sys; sys.path.insert(0,'.')
import numpy as np
from sklearn.datasets import load_iris
from oracle import gp_oracle as o
from waveome_amd import kernels as K
from waveome_amd.model_search import GPSearch
from waveome_amd.model_classes import PSVGP
from waveome_amd.regularization import full_kernel_build
from waveome_amd.engine import Context, Problem, lbfgs_opts
ctx=Context(0)
ir=load_iris(as_frame=True).frame
ir.columns=["sepal_length","sepal_width","petal_length","petal_width","species"]
ir["species"]=ir["species"].map({0:"setosa",1:"versicolor",2:"virginica"})
gps=GPSearch(X=ir[["sepal_length","sepal_width","species"]],Y=ir[["petal_length","petal_width"]],categorical_vars=["species"])
kern,_=full_kernel_build(cat_vars=gps.cat_idx,num_vars=gps.cont_idx,unit_idx=None,var_names=gps.feat_names,return_sum=True)
X=gps.X.to_numpy(); Y=gps.Y.to_numpy()
tmpl=PSVGP(X,Y[:,:1],mean_function=K.ConstantMean(),kernel=kern,context=ctx)
spec=tmpl._spec(); d=spec.to_dict()
prob=Problem(ctx,X,Y,spec,max_batch=2)
for p in range(2):
    evs=[]
    def fun(u):
        f,g=o.objective(d,u,X,Y[:,p]); evs.append(u.copy()); return f,g
    import scipy.optimize
    u0=spec.unconstrain()
    res=scipy.optimize.minimize(fun,u0,jac=True,method='L-BFGS-B',options={'maxiter':50000,'maxfun':50000})
    print('col',p,'oracle fit',res.fun,res.nit,res.nfev)
    U=np.array(evs)
    worst=0
    for i in range(0,len(U)):
        l,g,s=prob.value_grad([p],U[i][None,:]); fo,go=o.objective(d,U[i],X,Y[:,p])
        th=spec.constrain(U[i]); Ky=o.gram(d,th,X)+th[spec.noise]*np.eye(len(X))
        rel=abs(l[0]-fo)/max(1,abs(fo)); gerr=np.abs(g[0]-go).max()/max(1e-30,np.abs(go).max())
        if i%6==0 or rel>1e-9 or gerr>1e-6: print('  eval',i,'f',fo,'rel f err %.2e'%rel,'grad rel err %.2e'%gerr,'cond %.2e'%np.linalg.cond(Ky),'noise %.3e'%th[spec.noise], 'st',s[0])
print("---- detail at worst point")
p=1
evs=[]
def fun(u):
    f,g=o.objective(d,u,X,Y[:,p]); evs.append(u.copy()); return f,g
res=scipy.optimize.minimize(fun,spec.unconstrain(),jac=True,method='L-BFGS-B',options={'maxiter':50000})
u=evs[31]
l,g,s=prob.value_grad([p],u[None,:]); fo,go=o.objective(d,u,X,Y[:,p])
th=spec.constrain(u)
print('theta',th)
lp,lg=spec.log_prior(th,grad=True); print('prior val c++ %.12f py %.12f'%(lp,o.log_prior(d,th)))
print('prior grad c++',lg); print('prior grad py ',o.log_prior_grad(d,th))
for k,i in enumerate(spec.trainable_index):
    print(spec.param_names[i],'gpu %.10e oracle %.10e diff %.2e'%(g[0][k],go[k],g[0][k]-go[k]))
# repeat the GPU evaluation several times: deterministic?
for r in range(3):
    l2,g2,s2=prob.value_grad([p],u[None,:]); print('repeat maxdiff',np.abs(g2-g).max())

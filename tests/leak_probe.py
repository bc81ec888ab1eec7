"""By-hand check behind tests/test_gpu_parity.py::test_problem_lifecycles_return_their_device_memory: free device memory
(hipMemGetInfo) per create / use / close cycle of every allocating path, one path at a time, in pairs, and 200 cycles of
all of them together.  Run on a GPU box:  python tests/leak_probe.py"""
import sys, ctypes, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import test_gpu_parity as T
from waveome_amd import ModelSpec, kernels as K
from waveome_amd.engine import Context, FitQueue, Problem, lbfgs_opts
hip=ctypes.CDLL("libamdhip64.so")
def free():
    f,t=ctypes.c_size_t(),ctypes.c_size_t(); hip.hipMemGetInfo(ctypes.byref(f),ctypes.byref(t)); return f.value
ctx=Context(0); ctx32=Context(0,dtype="float32")
X,Y=T._data(77,300)
ms=ModelSpec(T._composite(),4,K.GaussianLikelihood(),K.ConstantMean())
ms2=ModelSpec(K.SquaredExponential(active_dims=[0])+K.Lin(active_dims=[1]),4,K.GaussianLikelihood(),K.ConstantMean())
def a():
    p=Problem(ctx,X,Y,ms,max_batch=3); p.close()
def b():
    p=Problem(ctx,X,Y,ms,max_batch=3); p.fit([0,1,2],np.tile(ms.unconstrain(),(3,1)),lbfgs_opts(maxiter=3)); p.close()
def c():
    p=Problem(ctx,X,Y,ms,max_batch=3); p.predict(0,ms.theta(),Xnew=X[:50]+0.1); p.close()
def c0():
    p=Problem(ctx,X,Y,ms,max_batch=3); p.predict(0,ms.theta()); p.close()
def d():
    p=Problem(ctx,X,Y,ms,max_batch=3); p.whitened_means(0,ms.theta(),[1,2,3]); p.close()
def e():
    m=Problem(ctx,X,Y,[ms,ms2],max_batch=4); m.close()
def f():
    m=Problem(ctx,X,Y,[ms,ms2],max_batch=4); q=FitQueue(m,lbfgs_opts(maxiter=50)); w=max(ms.n_trainable,ms2.n_trainable)
    U0=np.zeros((4,w)); U0[:2,:ms.n_trainable]=ms.unconstrain(); U0[2:,:ms2.n_trainable]=ms2.unconstrain()
    q.submit([0,1,0,2],[0,0,1,1],U0,tags=[10,11,12,13]); q.advance(); q.advance(); q.close(); m.close()
def g():
    p=Problem(ctx32,X,Y,ms2,max_batch=2); p.value_grad([0,1],np.tile(ms2.unconstrain(),(2,1))); p.close()
def h():
    c2=Context(0); c2.close()
def gr():
    p=Problem(ctx,X,Y,ms,max_batch=3); p.gram(ms.theta()); p.close()
for name,fn in [('create/close',a),('fit',b),('predict train',c0),('predict new',c),('whitened',d),('multi create',e),('queue',f),('f32 value_grad',g),('context',h),('gram',gr)]:
    fn(); fn(); f0=free()
    for i in range(20): fn()
    f1=free()
    print(f"{name:16s} {(f0-f1)/20/2**20:8.3f} MiB per cycle", flush=True)
def cyc():
    b(); c(); d(); f(); g()
cyc(); cyc(); f0=free()
for r in range(8):
    for i in range(25): cyc()
    import psutil
    print(f"combined: after {25*(r+1)} cycles free fell by {(f0-free())/2**20:.1f} MiB; host RSS {psutil.Process().memory_info().rss/2**20:.1f} MiB", flush=True)
import itertools
for name,fns in [('fit+predict new',(b,c)),('fit+whitened',(b,d)),('fit+queue',(b,f)),('fit+f32',(b,g)),('queue+f32',(f,g)),('predict+whitened',(c,d))]:
    for fn in fns: fn()
    f0=free()
    for i in range(20):
        for fn in fns: fn()
    print(f"{name:18s} {(f0-free())/20/2**20:8.3f} MiB per cycle", flush=True)

import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.optimize
import bench
from oracle import gp_oracle as o
from waveome_amd.engine import Context, Problem, lbfgs_opts
X, Y = bench.make_inputs(3); X, Y = X[:512], Y[:, :512]
ms = bench.headline_spec(); d = ms.to_dict(); u0 = ms.unconstrain()
prob = Problem(Context(0), X, Y.T, ms, max_batch=3)
cols = np.arange(3, dtype=np.int32)
res = prob.fit(cols, np.tile(u0, (3, 1)), lbfgs_opts(maxiter=50000, maxfun=50000))
refs = [scipy.optimize.minimize(lambda u, c=c: o.objective(d, u, X, Y[c]), u0, jac=True, method="L-BFGS-B", options=dict(maxiter=50000, maxfun=50000)) for c in range(3)]
tight = lbfgs_opts(maxiter=3000, maxfun=6000, ftol=0.0, gtol=1e-9)
pol_d = prob.fit(cols, res["u"], tight)
pol_s = prob.fit(cols, np.stack([r.x for r in refs]), tight)
for c in range(3):
    fo_d, go_d = o.objective(d, pol_d["u"][c], X, Y[c]); fo_s, go_s = o.objective(d, pol_s["u"][c], X, Y[c])
    print(c, "first:", res["loss"][c], res["nit"][c], res["status"][c], "scipy", refs[c].fun, refs[c].nit)
    print("   polish from device end:", pol_d["loss"][c], pol_d["nit"][c], pol_d["status"][c], "oracle f", fo_d, "|g|inf", np.max(np.abs(go_d)))
    print("   polish from scipy end: ", pol_s["loss"][c], pol_s["nit"][c], pol_s["status"][c], "oracle f", fo_s, "|g|inf", np.max(np.abs(go_s)))
    print("   theta d", np.round(ms.constrain(pol_d["u"][c]), 4)); print("   theta s", np.round(ms.constrain(pol_s["u"][c]), 4))

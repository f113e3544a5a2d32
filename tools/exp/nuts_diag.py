import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
import bench_inputs
from gpax_amd import _lib
from gpax_amd.infer.nuts import nuts_transition
from gpax_amd.models import ExactGP
X, y, _, _ = bench_inputs.synthetic_problem(25, 1, 4, seed=1)
m = ExactGP(1, "RBF"); m.X_train, m.y_train = m._set_data(X, y); m._data_version += 1
sites = m._sites(); dim = 3
def potential(u):
    v, g = m._log_joint(sites, u, 1e-6, jacobian=True)
    return (-v, -g) if np.isfinite(v) else (np.inf, np.zeros_like(u))
ra, rb = np.random.default_rng(5), np.random.default_rng(5)
native = m._native_transition(sites, 1e-6, rb)
u = np.zeros(3); U, g = potential(u)
ua, Ua, ga = u.copy(), U, g.copy(); ub, Ub, gb = u.copy(), U, g.copy()
im = np.ones(3)
for it in range(150):
    ua, Ua, ga, aa, na, da = nuts_transition(potential, ua, Ua, ga, 0.5, im, ra, 10)
    ub, Ub, gb, ab, nb, db = native(ub, Ub, gb, 0.5, im, rb, 10)
    if it % 10 == 0 or na != nb:
        print(it, na, nb, np.abs(ua-ub).max(), abs(Ua-Ub), abs(aa-ab), flush=True)
    if na != nb: break

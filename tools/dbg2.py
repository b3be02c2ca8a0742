import sys
sys.path.insert(0, '.')
import numpy as np
from oracle.ipm import _Cone
c = _Cone(2, 3)
rng = np.random.default_rng(0)
def rnd():
    u = rng.normal(size=c.m); u[:2] = abs(u[:2])+0.1
    q = u[2:].reshape(3,3); q[:,0] = np.linalg.norm(q[:,1:],axis=1) + abs(rng.normal(size=3)) + 0.1
    return u
s, z = rnd(), rnd()
W2inv, aW, aWi, lam = c.nt_scaling(s, z)
print("lam check", np.abs(aW(z) - aWi(s)).max())
print("W2inv check", np.abs(W2inv @ aW(aW(z)) - z).max())
u = rnd(); v = rng.normal(size=c.m)
print("div check", np.abs(c.prod(u, c.div(u, v)) - v).max())
d = rng.normal(size=c.m)
a = c.max_step(u, d)
print("step", a, c.min_eig(u + a*d), c.min_eig(u+0.99*a*d))
print(c.e())

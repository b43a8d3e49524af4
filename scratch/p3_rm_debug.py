"""where do the return-map variants differ? (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from _cases import synthetic_u
from test_domain_gpu import make_pair
P, D, G = make_pair((3, 2, 2), 3, None, "j2", "bspline")
G.dt_ = 0.5
u0 = synthetic_u(P, scale=0.012, seed=7)
u = synthetic_u(P, scale=0.02)
out = {}
for variant in ("lane", "2", "4"):
    os.environ["MIMI_HIP_P3_RETURN_MAP"] = variant
    G.ResetState()
    r0 = np.zeros(P.n_vdofs); G.AddDomainResidual(u, r0)          # virgin state
    G.DomainPostTimeAdvance(u0)
    eq = G.State("accumulated_plastic_strain").copy()
    r = np.zeros(P.n_vdofs); G.AddDomainResidual(u, r)
    out[variant] = (r0, eq, r)
for v in ("2", "4"):
    for name, a, b in zip(("r virgin", "eqps", "r committed"), out["lane"], out[v]):
        d = np.abs(a - b)
        print(v, name, "equal" if np.array_equal(a, b) else "DIFFER: %d of %d entries, max abs %.3e, max rel %.3e" % ((d > 0).sum(), d.size, d.max(), (d / np.abs(a).max()).max()))

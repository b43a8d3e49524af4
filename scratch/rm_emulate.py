"""Emulation of the J2 return-map solver (csrc/materials.hpp scalar_solve = solvers/newton.hpp:53-169) on the trial states of
the benchmark displacement, to compare ways of assigning points to lanes (round 5).  Counts residual
EVALUATIONS per point (the admissibility check at 0, the bracket end, the iterations); a wave costs the maximum over its
lanes.  No GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import iga

n_el, p = (6, 6, 6), 3
P = iga.Patch.block(n_el, p)
rng = np.random.default_rng(20241008)
u = 0.05 * rng.standard_normal(P.n_vdofs)
T = P.tables()
ue = u.reshape(-1, 3)[T["conn"]]                       # [e, a, i]
H = np.einsum("eai,eqaJ->eqiJ", ue, T["dN_dX"])
F = H + np.eye(3)
E_, nu = 2100.0, 0.3
G = E_ / (2 * (1 + nu)); K = E_ / (3 * (1 - 2 * nu))
A_, B_, n_ = 70.0, 140.0, 0.2835
eps = 0.5 * (F + np.swapaxes(F, -1, -2)) - np.eye(3)
tr = np.trace(eps, axis1=-2, axis2=-1)
s = 2 * G * (eps - tr[..., None, None] / 3 * np.eye(3))
q = np.sqrt(1.5) * np.sqrt((s * s).sum((-1, -2)))
q = q.reshape(-1)
tol = A_ * 1e-10

def hard(x):
    small = np.abs(x) < 1e-13
    xs = np.where(small, 1.0, x)
    pw = xs ** (n_ - 1.0)
    return np.where(small, A_, A_ + B_ * xs * pw), np.where(small, 0.0, B_ * n_ * pw)

def resid(x):
    h, hd = hard(x)
    return q_cur - 3 * G * x - h, -3 * G - hd

evals = np.zeros(q.size, int)
for idx in range(q.size):
    q_cur = q[idx]
    f0, _ = resid(0.0)
    n = 1
    if not f0 > tol:
        evals[idx] = n
        continue
    upper = (q_cur - A_) / (3 * G)
    lower = 0.0
    fl = f0
    fh, _ = resid(upper); n += 1
    if abs(fh) < 1e-10:
        evals[idx] = n; continue
    assert fl * fh <= 0
    xl, xh = (upper, lower) if fl > 0 else (lower, upper)
    x = 0.0
    dxo = abs(upper - lower); dx = dxo
    fv, df = resid(0.0)                                 # (x == lower: the evaluation at 0 is reused)
    it = 0
    while True:
        if (x - xh) * df - fv > 0 or (x - xl) * df - fv < 0 or abs(2 * fv) > abs(dxo * df):
            dxo = dx; dx = 0.5 * (xh - xl); x = xl + dx
        else:
            dxo = dx; dx = fv / df; x -= dx
        fv, df = resid(x); n += 1
        conv = abs(dx) < 1e-10 or abs(fv) < tol
        if fv < 0: xl = x
        else: xh = x
        it += 1
        if conv or it == 100: break
    evals[idx] = n
ne = len(T["conn"])
ev = evals.reshape(ne, 125)
print("points: mean evaluations %.2f, elastic %.1f %%, max %d" % (evals.mean(), 100 * (evals == 1).mean(), evals.max()))
def waves(a, w=64):
    a = np.concatenate([a, np.ones((-len(a)) % w, int)])
    return a.reshape(-1, w).max(1)
cur = sum(waves(ev[e]).sum() for e in range(ne)) / ne
print("as shipped (2 waves per element, lane = point): %.1f wave-evaluations per element" % cur)
for g in (1, 2, 4, 8):
    tot = 0
    for e0 in range(0, ne - ne % g, g):
        a = np.sort(ev[e0:e0 + g].ravel())[::-1]
        tot += waves(a).sum()
    print("sorted exactly over %d element(s): %.1f" % (g, tot / (ne - ne % g)))
# a key that is known before the solve: the overstress q - A (monotone in `upper`)
key = (q - A_).reshape(ne, 125)
for g in (1, 2, 4):
    tot = 0
    for e0 in range(0, ne - ne % g, g):
        k = key[e0:e0 + g].ravel(); a = ev[e0:e0 + g].ravel()
        order = np.argsort(k)          # ascending overstress: elastic first ... 
        tot += waves(a[order]).sum()
    print("sorted by overstress over %d element(s): %.1f" % (g, tot / (ne - ne % g)))
# buckets of the overstress
yk = key.ravel(); ye = evals
import collections
edges = [0, 1, 2, 5, 10, 20, 40, 80, 160, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (yk > lo) & (yk <= hi)
    if m.any():
        print("  overstress (%g, %g]: %5.1f %% of the points, evaluations mean %.1f min %d max %d" % (lo, hi, 100 * m.mean(), ye[m].mean(), ye[m].min(), ye[m].max()))
np.save("/tmp/rm_evals.npy", ev); np.save("/tmp/rm_key.npy", key)

# round 5, second idea: every lane iterates K times where it is; the points not converged by then ("stragglers") are collected
# over the workgroup (g elements) and finished packed into as few waves as they fill
its = np.maximum(ev - 2, 0)            # loop iterations (evaluations minus the two before the loop)
print("iterations per point: mean %.2f; histogram" % its.mean(), np.bincount(its.reshape(-1))[:40])
for g in (1, 2, 4):
    for K in (4, 5, 6, 7, 8):
        tot = 0; groups = 0
        for e0 in range(0, ne - ne % g, g):
            a = its[e0:e0 + g].reshape(-1)
            nw = 2 * g
            first = sum(min(K, waves(its[e0 + k]).max() if False else 0) for k in range(0))  # (unused)
            # phase 1: each wave runs min(K, its max) iterations
            p1 = sum(np.minimum(waves(its[e0 + k]), K).sum() for k in range(g))
            rest = np.sort(np.maximum(a - K, 0))[::-1]
            rest = rest[rest > 0]
            p2 = waves(rest).sum() if rest.size else 0      # packed, longest first
            tot += p1 + p2; groups += 1
        print("  %d element(s) per workgroup, K = %d: %.1f wave-iterations per element (shipped: %.1f)" % (g, K, tot / groups / g, sum(waves(its[e]).sum() for e in range(ne)) / ne))

"""What walking the LONG axis would be worth (round 5): the north-star patch (128 x 128 x 16 elements, the phase-1 kernels walk the
third axis: columns of 16) against the SAME problem handed over with its parametric axes rotated (y, z, x) -> columns of 128, the
caller's numbering and CSR kept through node_ids (the permuted-numbering route).  Same physics, same matrix; timings + a check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
name = sys.argv[1] if len(sys.argv) > 1 else "northstar"
n_el, p, material = bench.WORKLOADS[name]
P = mimi_amd.BSplinePatch.block(n_el, p)
pattern = CSRPattern.of_bspline_patch(P, on_device=True)
dev = torch.device("cuda", 0)
u = torch.from_numpy(bench.synthetic_u(P)).to(dev)

def timed(G, label):
    G.dt_ = 0.5
    r = torch.zeros(P.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    for _ in range(3):
        G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.SetPhaseTiming(True)
    acc = np.zeros(2)
    for _ in range(10):
        G.AddDomainResidualAndGrad(u, 1.0, r, A)
        acc += G.PhaseMs()
    acc /= 10
    G.SetPhaseTiming(False)
    G.Synchronize()
    t = time.perf_counter()
    for _ in range(20):
        G.AddDomainResidualAndGrad(u, 1.0, r, A)
    G.Synchronize()
    t_all = (time.perf_counter() - t) / 20 * 1e3
    r.zero_(); A.zero_(); G.AddDomainResidualAndGrad(u, 1.0, r, A); G.Synchronize()
    print("%-28s R+J %.3f ms (phase 1 %.3f + phase 2 %.3f)" % (label, t_all, acc[0], acc[1]), flush=True)
    return r.clone(), A.clone()

G0 = NonlinearSolid("d", bench.make_material(material), pattern, patch=P).Prepare()
r0, A0 = timed(G0, "as given (columns of %d)" % n_el[2])
del G0
torch.cuda.empty_cache()
# rotated axes: logical (0, 1, 2) = physical (1, 2, 0)
rot = (1, 2, 0)
nc = P.n_ctrl
ids = np.arange(P.n_nodes, dtype=np.int64).reshape(nc[2], nc[1], nc[0])           # [iz][iy][ix] -> caller's id
# P' lexicographic (i0' fastest) with i0' = iy, i1' = iz, i2' = ix: array [i2'][i1'][i0'] = [ix][iz][iy]
ids_rot = np.ascontiguousarray(ids.transpose(2, 0, 1)).reshape(-1)
ctrl = P.control_points[ids_rot]
Pr = mimi_amd.BSplinePatch([P.degrees[d] for d in rot], [P.knots[d] for d in rot], ctrl)
G1 = NonlinearSolid("d", bench.make_material(material), pattern, patch=Pr, node_ids=ids_rot).Prepare()
r1, A1 = timed(G1, "axes rotated (columns of %d)" % n_el[0])
print("residual rel. diff %.2e, matrix rel. diff %.2e" % (float((r1 - r0).abs().max() / r0.abs().max()), float((A1 - A0).abs().max() / A0.abs().max())))

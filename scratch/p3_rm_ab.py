"""same-box A/B of the degree-3 pre-pass's return mapping: MIMI_HIP_P3_RETURN_MAP = lane | 2 | 4 (elements per workgroup whose
return-map equations are solved together): residual-only assembly and DomainPostTimeAdvance of BASELINE configuration 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd.integrators import CSRPattern, NonlinearSolid
n_el, p, material = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
patch = mimi_amd.BSplinePatch.block(n_el, p)
pattern = CSRPattern.of_bspline_patch(patch, on_device=True)
G = NonlinearSolid("d", bench.make_material(material), pattern, patch=patch).Prepare()
G.dt_ = 0.5
dev = torch.device("cuda", 0)
u = torch.from_numpy(bench.synthetic_u(patch)).to(dev)
r = torch.zeros(patch.n_vdofs, dtype=torch.float64, device=dev)
def timed(fn, reps=5):
    fn(); G.Synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    G.Synchronize()
    return (time.perf_counter() - t) / reps * 1e3
sums = {}
for variant in ("lane", "2", "4", "lane", "2", "4"):
    os.environ["MIMI_HIP_P3_RETURN_MAP"] = variant
    G.ResetState()
    t_r = timed(lambda: G.AddDomainResidual(u, r))
    def commit():
        G.ResetState(); G.DomainPostTimeAdvance(u)
    def reset():
        G.ResetState()
    t_c = timed(commit) - timed(reset)
    G.ResetState(); r.zero_(); G.AddDomainResidual(u, r); G.Synchronize()
    sums[variant] = float(r.abs().sum())
    print("return map %-4s residual-only %.3f ms   commit %.3f ms   checksum %.17e" % (variant, t_r, t_c, sums[variant]), flush=True)
print("equal:", len(set(sums.values())) == 1)

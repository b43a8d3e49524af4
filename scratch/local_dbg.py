import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, mimi_amd
from mimi_amd import parallel
from mimi_amd.integrators import CSRPattern, NonlinearSolid
from mimi_amd.splines import PatchShape
n_el, p, world = (2, 3, 9), 3, 3
for rank in range(world):
    shape = PatchShape.block(n_el, p)
    sg = parallel.SlabShard(shape, None, rank, world)
    b, e = sg.element_box; ax = sg.axis
    below, above = sg.ghost_layers()
    lp = mimi_amd.BSplinePatch.block_slab(n_el, p, ax, b[ax] - below, e[ax] + above)
    print("rank", rank, "axis", ax, "box", b, e, "ghost", below, above, "local spans", lp.n_spans, flush=True)
    pattern = CSRPattern.of_bspline_patch(lp, on_device=True)
    shard = sg.localized(lp, pattern, ghost=(below, above))
    print("  local element box", shard.element_box, flush=True)
    g = NonlinearSolid("domain", bench.make_material("neohookean"), pattern, patch=lp, element_box=shard.element_box).Prepare()
    print("  path", g.path_, "elements", g.n_elements_, flush=True)
    dev = torch.device("cuda", 0)
    u = torch.zeros(lp.n_vdofs, dtype=torch.float64, device=dev)
    r = torch.zeros(lp.n_vdofs, dtype=torch.float64, device=dev)
    A = torch.zeros(pattern.nnz, dtype=torch.float64, device=dev)
    g.AddDomainResidualAndGrad(u, 1.0, r, A)
    g.Synchronize()
    print("  assembled", float(A.abs().sum()), flush=True)

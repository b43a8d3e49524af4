"""Sampled-row checks: complete CSR rows / residual entries of a few nodes out of the oracle's ELEMENT blocks
(oracle/ref_path.c element_residual_and_grad), for meshes the oracle cannot assemble as a whole."""
import numpy as np


def sample_nodes(n, n_random, seed):
    """multi-indices: the 8 corners, edge and face midpoints, the last nodes, random interior ones"""
    n = np.asarray(n)
    picks = set()
    for c in np.ndindex(2, 2, 2):
        picks.add(tuple(int(v) for v in np.array(c) * (n - 1)))
    mid = n // 2
    for d in range(3):
        for side in (0, 1):
            f = mid.copy()
            f[d] = side * (n[d] - 1)
            picks.add(tuple(int(v) for v in f))          # face centres
            g = f.copy()
            g[(d + 1) % 3] = 0
            picks.add(tuple(int(v) for v in g))          # edge midpoints
    for k in range(1, 5):                                # the last rows of the matrix
        picks.add((int(max(n[0] - 1 - k, 0)), int(n[1] - 1), int(n[2] - 1)))
        picks.add((int(n[0] - 1), int(max(n[1] - 1 - k, 0)), int(max(n[2] - 1 - k % 2, 0))))
    rng = np.random.default_rng(seed)
    target = len(picks) + n_random
    while len(picks) < target:
        picks.add(tuple(int(rng.integers(0, n[d])) for d in range(3)))
    return sorted(picks)


class SampledRows:
    """expected rows of the sampled nodes: `row(k, i, cols)` -> (values at the sorted columns `cols`, residual entry)"""

    def __init__(self, P, material, nodes, u, dt=0.5, u_commit=None):
        from oracle import ref_path as rp
        p = P.p[0]
        self.P, self.nodes = P, nodes
        n, m = np.array(P.n), np.array(P.m)
        self.node_ids = [int(mi[0] + n[0] * (mi[1] + n[1] * mi[2])) for mi in nodes]
        self.el_of_node, wanted = [], set()
        for mi in nodes:
            rng_d = [range(max(mi[d] - p, 0), min(mi[d], m[d] - 1) + 1) for d in range(3)]
            els = [(int(e0 + m[0] * (e1 + m[1] * e2)), (mi[0] - e0) + (p + 1) * ((mi[1] - e1) + (p + 1) * (mi[2] - e2)))
                   for e2 in rng_d[2] for e1 in rng_d[1] for e0 in rng_d[0]]
            self.el_of_node.append(els)
            wanted.update(e for e, _ in els)
        elements = np.array(sorted(wanted), dtype=np.int64)
        self.slot = {int(e): k for k, e in enumerate(elements)}
        self.D = rp.DomainOracle(P, material, elements=elements, with_a_ids=False, with_sparsity=False)
        self.D.set_dt(dt)
        self.elements = elements
        if u_commit is not None:
            # DomainPostTimeAdvance at u_commit on the sampled elements (nonlinear_solid.cpp:179-199): the blocks below
            # are then integrated from the committed state
            self.D.domain_post_time_advance(u_commit)
        self.blocks = {int(e): self.D.element_residual_and_grad(self.slot[int(e)], u, rp.TANGENT_EXACT) for e in elements}

    def row(self, k, i, cols):
        n_dof = self.P.n_dof
        cols = np.asarray(cols, dtype=np.int64)
        exp = np.zeros(len(cols))
        r_exp = 0.0
        for e, a in self.el_of_node[k]:
            R_e, K_e = self.blocks[e]          # rows / columns component-grouped: i n_dof + a (ElementData::v_dofs)
            conn = self.D.conn[self.slot[e]].astype(np.int64)
            r_exp += R_e[i * n_dof + a]
            for j in range(3):
                pos = np.searchsorted(cols, conn * 3 + j)
                assert np.array_equal(cols[pos], conn * 3 + j)
                np.add.at(exp, pos, K_e[i * n_dof + a, j * n_dof:(j + 1) * n_dof])
        return exp, r_exp

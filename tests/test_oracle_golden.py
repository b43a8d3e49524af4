"""Pins the oracle: the CPU restatement (oracle/) driven through the restated
gen-alpha / line-search-Newton callers must reproduce the reference's golden time
series (reference tests/test_nonlinear_solid.py:55-98, np.allclose defaults there;
we assert a much tighter 1e-9 absolute)."""
import os

import numpy as np
import pytest

from _cases import balken_oracle


@pytest.mark.parametrize("matname,refdir", [("neohook", "neohook_h1_p2"), ("j2", "j2_h1_p2"),
                                            ("j2simo", "j2_simo_h1_p2"), ("j2log", "j2_log_h1_p2")])
@pytest.mark.parametrize("tangent_mode", [0, 1], ids=["referenceFD", "exact"])
def test_golden_time_series(golden_dir, matname, refdir, tangent_mode):
    from oracle import harness as hz
    P, D, op, ode, dt = balken_oracle(matname, tangent_mode)
    x = np.zeros(P.n_vdofs)
    v = np.zeros_like(x)
    t = 0.0
    for i in range(10):
        t = ode.step(x, v, t, dt)
        ref = hz.golden_to_lexicographic(np.genfromtxt(os.path.join(golden_dir, "ref", refdir, f"x_{i}.txt")))
        assert np.allclose(x, ref)                       # the reference's own criterion
        assert np.abs(x - ref).max() < 1e-9, (i, np.abs(x - ref).max())
    if matname != "neohook":
        assert D.eqps.max() > 0.05                       # plasticity really active


@pytest.mark.parametrize("matname", ["neohook", "j2"])
def test_thread_count_invariance(matname):
    """reference tests/test_nthreads.py:78-122: same answers for nthreads in {1,2,3,4}."""
    xs = []
    for nt in (1, 2, 3, 4):
        P, D, op, ode, dt = balken_oracle(matname, 0, n_threads=nt)
        x = np.zeros(P.n_vdofs)
        v = np.zeros_like(x)
        t = 0.0
        for i in range(3):
            t = ode.step(x, v, t, dt)
        xs.append(x.copy())
    for x in xs[1:]:
        assert np.allclose(x, xs[0])

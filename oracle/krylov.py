"""TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

numpy restatement of the iterative linear solver the reference configures when "use_iterative_solver" is set
(src/mimi/py/py_nonlinear_solid.cpp:329-339): mfem::GMRESSolver (rel 1e-8, abs 1e-12, 300 iterations, m = 50) with an
mfem::DSmoother (Jacobi, one sweep from zero) preconditioner.  MFEM is an absent, un-pinned submodule of the
reference; this follows its published algorithm (mfem linalg/solvers.cpp, GMRESSolver::Mult; DSmoother::Mult) -- parity
unpinned by any reference fixture."""
import numpy as np


def gmres(A, b, rel_tol=1e-8, abs_tol=1e-12, max_iter=300, kdim=50, jacobi=True):
    """(x, iterations, final_norm, converged); A: scipy CSR; x starts at 0 (iterative_mode false)."""
    n = len(b)
    dinv = 1.0 / A.diagonal() if jacobi else np.ones(n)
    x = np.zeros(n)
    r = dinv * b
    beta = np.linalg.norm(r)
    goal = max(rel_tol * beta, abs_tol)
    if beta <= goal:
        return x, 0, beta, True
    m = kdim
    j = 1
    while j <= max_iter:
        V = np.zeros((m + 1, n))
        H = np.zeros((m + 1, m))
        cs, sn, s = np.zeros(m + 1), np.zeros(m + 1), np.zeros(m + 1)
        V[0] = r / beta
        s[0] = beta
        i = 0
        while i < m and j <= max_iter:
            w = dinv * (A @ V[i])
            for k in range(i + 1):                     # modified Gram-Schmidt
                H[k, i] = w @ V[k]
                w -= H[k, i] * V[k]
            H[i + 1, i] = np.linalg.norm(w)
            V[i + 1] = w / H[i + 1, i]
            for k in range(i):                         # ApplyPlaneRotation
                t = cs[k] * H[k, i] + sn[k] * H[k + 1, i]
                H[k + 1, i] = -sn[k] * H[k, i] + cs[k] * H[k + 1, i]
                H[k, i] = t
            dx, dy = H[i, i], H[i + 1, i]              # GeneratePlaneRotation
            if dy == 0.0:
                cs[i], sn[i] = 1.0, 0.0
            elif abs(dy) > abs(dx):
                t = dx / dy
                sn[i] = 1.0 / np.sqrt(1.0 + t * t)
                cs[i] = t * sn[i]
            else:
                t = dy / dx
                cs[i] = 1.0 / np.sqrt(1.0 + t * t)
                sn[i] = t * cs[i]
            H[i, i] = cs[i] * dx + sn[i] * dy
            H[i + 1, i] = 0.0
            s[i + 1] = -sn[i] * s[i]
            s[i] = cs[i] * s[i]
            resid = abs(s[i + 1])
            if resid <= goal:
                y = np.linalg.solve(np.triu(H[:i + 1, :i + 1]), s[:i + 1])
                return x + y @ V[:i + 1], j, resid, True
            i += 1
            j += 1
        y = np.linalg.solve(np.triu(H[:i, :i]), s[:i])
        x = x + y @ V[:i]
        r = dinv * (b - A @ x)
        beta = np.linalg.norm(r)
        if beta <= goal:
            return x, j - 1, beta, True
    return x, max_iter, beta, False


def cg(A, b, rel_tol=1e-8, abs_tol=1e-12, max_iter=1000, jacobi=True):
    """mfem::CGSolver::Mult with a DSmoother preconditioner, iterative_mode false (the mass solve of
    operators::NonlinearSolid, operators/nonlinear_solid.cpp:39-50,155).  (x, iterations, sqrt((r, M r)), converged)"""
    n = len(b)
    dinv = 1.0 / A.diagonal() if jacobi else np.ones(n)
    x = np.zeros(n)
    r = b.copy()
    z = dinv * r
    d = z.copy()
    nom = r @ z
    r0 = max(nom * rel_tol * rel_tol, abs_tol * abs_tol)
    if nom <= r0:
        return x, 0, np.sqrt(abs(nom)), True
    q = A @ d
    den = q @ d
    if not den > 0:
        return x, 0, np.sqrt(abs(nom)), False
    it = 1
    while True:
        alpha = nom / den
        x += alpha * d
        r -= alpha * q
        z = dinv * r
        betanom = r @ z
        if betanom <= r0:
            return x, it, np.sqrt(abs(betanom)), True
        if it >= max_iter:
            return x, it, np.sqrt(abs(betanom)), False
        beta = betanom / nom
        d = z + beta * d
        q = A @ d
        den = d @ q
        if not den > 0:
            return x, it, np.sqrt(abs(betanom)), False
        nom = betanom
        it += 1

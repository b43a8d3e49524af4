/* mimi_hip.h -- C ABI of libmimi_hip.so: MI355X (gfx950) element integration and
 * assembly for mimi's NURBS nonlinear solids.
 *
 * This is the drop-in boundary.  Each entry point replaces one virtual of the
 * reference's `mimi::integrators::NonlinearBase` (or the setup that feeds it); the
 * reference interface it stands in for is cited as path:line under
 * /root/reference/src/mimi/.  INTEGRATION.md shows the C++ subclass a mimi maintainer
 * would add to bind these.
 *
 * Conventions
 *  - plain C, no torch / HIP types in any signature.
 *  - every `const double*` / `double*` / index pointer argument may be a HOST or a
 *    DEVICE pointer; the library detects which (hipPointerGetAttributes).  Host
 *    buffers are staged through device scratch and the call is synchronous; device
 *    buffers are used in place and the call only enqueues work on the handle's
 *    stream (mimi_hip_*_set_stream / mimi_hip_*_synchronize).  ORDERING: the kernels read and write those buffers
 *    in stream order on the handle's stream only.  Work the caller has enqueued on them elsewhere (a zero fill of
 *    r / A, a copy into u) must be ordered before the call, and reads of the results behind it: give the handle the
 *    stream that work runs on (set_stream), or synchronise.  Several handles adding into one A need one stream.
 *  - u, r: fp64[n_vdofs], byVDIM ordering u[node*dim + c]   (py_nonlinear_solid.cpp:63,74)
 *  - A_values: fp64[nnz] of a CSR matrix with sorted columns whose structure
 *    (rowptr, col) was given at create time and never changes (precomputed.cpp:151-174,
 *    nonlinear_base.hpp:130).  Outputs are ACCUMULATED (+=), never overwritten.
 *  - return value: 0 on success, non-zero on error; mimi_hip_last_error() then holds
 *    the message the reference would have thrown as std::runtime_error
 *    (utils/print.hpp:47-56).  Not re-entrant per handle (like the reference:
 *    mutable work data, nonlinear_solid.hpp:42).
 */
#ifndef MIMI_HIP_H
#define MIMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIMI_HIP_ABI_VERSION 12

/* ---- errors ---------------------------------------------------------------- */
const char* mimi_hip_last_error(void);
int mimi_hip_abi_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never throws) */
int mimi_hip_device_count(void);

/* ---- materials (materials/materials.hpp, materials/material_hardening.hpp) -- */
enum mimi_hip_material_kind {
  MIMI_HIP_MAT_NEOHOOKEAN = 0, /* CompressibleOgdenNeoHookean  materials.hpp:118-140, materials.cpp:96-118 */
  MIMI_HIP_MAT_J2 = 1,         /* J2 (small strain, nonlinear isotropic hardening) materials.hpp:259-403 */
  MIMI_HIP_MAT_STVK = 2,       /* StVenantKirchhoff  materials.hpp:88-111, materials.cpp:72-94 */
  MIMI_HIP_MAT_J2LINEAR = 3,   /* J2Linear (linear isotropic + kinematic hardening)  materials.hpp:142-249 */
  MIMI_HIP_MAT_J2SIMO = 4,     /* J2Simo (finite strain, be / F_old state)  materials.hpp:406-557 */
  MIMI_HIP_MAT_J2LOG = 5       /* J2Log (logarithmic strain, Fp_inv state)  materials.hpp:559-753 */
};

enum mimi_hip_hardening_kind {
  MIMI_HIP_HARD_POWERLAW = 0,      /* material_hardening.hpp:79-98  */
  MIMI_HIP_HARD_VOCE = 1,          /* :100-121 */
  MIMI_HIP_HARD_JC = 2,            /* JohnsonCookHardening :123-143 */
  MIMI_HIP_HARD_JC_RATE = 3,       /* JohnsonCookRateDependentHardening :145-187 */
  MIMI_HIP_HARD_JC_TEMP_RATE = 4,  /* JohnsonCookTemperatureAndRateDependentHardening :189-280 */
  MIMI_HIP_HARD_JC_CONST_TEMP = 5  /* JohnsonCookConstantTemperatureHardening :282-346 */
};

typedef struct mimi_hip_material {
  int32_t kind;                 /* mimi_hip_material_kind */
  int32_t hardening;            /* mimi_hip_hardening_kind (J2 only) */
  /* MaterialBase (materials.hpp:31-38); set lambda/mu/K/G with mimi_hip_material_set_young_poisson */
  double density, lambda, mu, K, G;
  /* J2 thermo members (materials.hpp:268-273) */
  double heat_fraction, specific_heat, initial_temperature, melting_temperature;
  /* hardening parameters; unused ones are ignored */
  double sigma_y, n, eps0;                  /* PowerLaw (n shared with JohnsonCook) */
  double sigma_sat, strain_constant;        /* Voce */
  double A, B, C, eps0_dot;                 /* JohnsonCook: A + B eqps^n, rate term 1 + C ln(rate/eps0_dot) */
  double reference_temperature, m;          /* thermal softening exponent m */
  /* J2Linear (materials.hpp:149-151); its yield stress sigma_y_ is the field sigma_y above */
  double lin_isotropic_hardening, lin_kinematic_hardening;
} mimi_hip_material;

/* MaterialBase::SetYoungPoisson (materials.cpp:7-14) */
void mimi_hip_material_set_young_poisson(mimi_hip_material* mat, double young, double poisson);

/* ---- tangent mode ------------------------------------------------------------ */
enum mimi_hip_tangent_mode {
  MIMI_HIP_TANGENT_ANALYTIC = 0,     /* consistent tangent dP/dF, closed form (default) */
  MIMI_HIP_TANGENT_REFERENCE_FD = 1  /* the reference's element-level forward difference,
                                        step |u_i|*1e-8 or 1e-10 (nonlinear_solid.cpp:48-76) */
};

/* ---- domain integrator: integrators::NonlinearSolid ---------------------------- */
typedef struct mimi_hip_domain_s* mimi_hip_domain_t;

/* Flat form of what PrecomputedData hands the reference integrator
 * (utils/precomputed.hpp:58-130, utils/precomputed.cpp:39-330). */
typedef struct mimi_hip_domain_tables {
  int32_t dim;           /* 2 or 3 */
  int32_t n_elements;
  int32_t n_dof;         /* ElementData::n_dof, basis functions per element, <= 64 */
  int32_t n_quad;        /* ElementQuadData::n_quad, <= 125 */
  int64_t n_nodes;       /* global scalar dofs; n_vdofs = n_nodes*dim */
  const int32_t* dofs;   /* [n_elements][n_dof]   ElementData::dofs (precomputed.cpp:83) */
  const double* dN_dX;   /* [n_elements][n_quad][dim][n_dof]  QuadData::dN_dX, (n_dof x dim)
                            column-major per point (precomputed.cpp:316-321) */
  const double* weight_det; /* [n_elements][n_quad] integration_weight*det_dX_dxi (nonlinear_solid.hpp:79) */
  const int64_t* csr_rowptr; /* [n_vdofs+1] */
  const int32_t* csr_col;    /* [nnz], sorted within each row */
} mimi_hip_domain_tables;

/* NonlinearSolid ctor + Prepare() (nonlinear_solid.hpp:45-50, nonlinear_solid.cpp:31-46)
 * from flat per-point tables: works for any mesh / numbering / rational weights. */
int mimi_hip_domain_create(const mimi_hip_domain_tables* tables, const mimi_hip_material* material,
                           int device, mimi_hip_domain_t* out);

/* Tensor-product B-spline / NURBS patch description: the library builds the
 * 1-D basis tables, per-point inverse geometry Jacobians and weights itself
 * (replaces PrepareElementData + PrecomputeElementQuadData, precomputed.cpp:39-149,264-330)
 * and integrates by sum factorisation.  Lexicographic conventions: node
 * A = A0 + n0*(A1 + n1*A2), element e = e0 + m0*(e1 + m1*e2). */
typedef struct mimi_hip_bspline_patch {
  int32_t dim;
  int32_t degree[3];
  int32_t n_knots[3];
  const double* knots[3];        /* host pointers */
  const double* control_points;  /* host, [n_nodes][dim], lexicographic */
  const int64_t* node_ids;       /* host, [n_nodes] lexicographic -> global node id, or NULL (identity) */
  int32_t quadrature_order;      /* < 0: 2*p+3 (precomputed.cpp:284-286) */
  int32_t element_begin[3];      /* half-open element box [begin,end) integrated by THIS handle  */
  int32_t element_end[3];        /* (element sharding across GPUs); all zeros = whole patch       */
  const int64_t* csr_rowptr;     /* [n_vdofs+1] host or device */
  const int32_t* csr_col;        /* [nnz]       host or device */
  const double* weights;         /* host, [n_nodes] lexicographic NURBS weights, or NULL (all 1).  They must be a tensor
                                    product w[a0,a1,a2] = w0[a0] w1[a1] w2[a2] (arcs, cylinders, annuli, extrusions and
                                    revolutions are): the rational basis is then a tensor product too and every kernel
                                    family applies.  Other weights are refused here: hand such a patch to
                                    mimi_hip_domain_create as flat tables. */
} mimi_hip_bspline_patch;

int mimi_hip_domain_create_bspline(const mimi_hip_bspline_patch* patch, const mimi_hip_material* material,
                                   int device, mimi_hip_domain_t* out);

int mimi_hip_domain_destroy(mimi_hip_domain_t h);

/* public members dt_, first_effective_dt_, second_effective_dt_ that forms::Nonlinear
 * pushes before every call (nonlinear_base.hpp:23-25, forms/nonlinear.hpp:63-65) */
int mimi_hip_domain_set_dt(mimi_hip_domain_t h, double dt, double first_effective_dt,
                           double second_effective_dt);
int mimi_hip_domain_set_tangent_mode(mimi_hip_domain_t h, int mode);
/* stream = hipStream_t as void*; NULL = the handle's own (non-blocking) stream; MIMI_HIP_STREAM_NULL = the device's
 * null stream (hipStream_t 0: what e.g. torch's default stream is) */
#define MIMI_HIP_STREAM_NULL ((void*)(intptr_t)-1)
int mimi_hip_domain_set_stream(mimi_hip_domain_t h, void* stream);
int mimi_hip_domain_synchronize(mimi_hip_domain_t h);

/* AddDomainResidual(current_u, residual): r += R(u)   (nonlinear_solid.cpp:151-160) */
int mimi_hip_domain_add_residual(mimi_hip_domain_t h, const double* u, double* r);
/* AddDomainResidualAndGrad(current_u, grad_factor, residual, grad):
 * r += R(u); A_values += grad_factor * K(u)            (nonlinear_solid.cpp:162-177) */
int mimi_hip_domain_add_residual_and_grad(mimi_hip_domain_t h, const double* u, double grad_factor,
                                          double* r, double* A_values);
/* The same assembly with the old values of the matrix taken from a second array:
 *   r += R(u); A_out = A_base + grad_factor * K(u)
 * for WHOLE-PATCH handles only (refused on an element box / slab handle: "=" does not compose over boxes the way "+="
 * does -- a slab assembles with mimi_hip_domain_add_residual_and_grad into its copy of the base).  This is operators::NonlinearSolid::ResidualAndGrad's "jacobian_ values <- mass values, then
 * AddMultGrad" (operators/nonlinear_solid.cpp:257-258) without the copy pass: with both arrays on the device the row
 * gathers read A_base where "+=" would read A_out -- no extra traffic.  A_base == A_out is the plain "+=".  Routes
 * without a row gather (the atomics fallback of the general path) and host-resident arrays copy A_base into A_out first. */
int mimi_hip_domain_add_residual_and_grad_from(mimi_hip_domain_t h, const double* u, double grad_factor,
                                               double* r, const double* A_base, double* A_out);
/* DomainPostTimeAdvance(converged_u): commit material state (nonlinear_solid.cpp:179-199) */
int mimi_hip_domain_post_time_advance(mimi_hip_domain_t h, const double* u);

/* material state access (MaterialState of materials.hpp:278-286), for tests / output:
 * what = 0 accumulated plastic strain [n_el][n_q]; 1 temperature [n_el][n_q];
 *        2 the material's first state matrix [n_el][n_q][dim*dim] (column-major per point): plastic strain (J2,
 *          J2Linear), be_old (J2Simo), Fp_inv (J2Log);
 *        3 its second one: beta (J2Linear, materials.hpp:158), F_old (J2Simo, materials.hpp:434) */
int mimi_hip_domain_get_state(mimi_hip_domain_t h, int what, double* out, int64_t capacity);
int mimi_hip_domain_reset_state(mimi_hip_domain_t h);
/* The two-step form of mimi_hip_domain_add_residual_and_grad, for a caller that needs some rows before the others (the
 * rows a neighbour rank is waiting for: mimi_amd/parallel.py, bench.py): integrate() runs the integration kernels of the
 * whole handle -- element row pieces and element residual pieces into the handle's scratch, nothing into r / A_values --
 * and gather() adds the rows of the nodes in the box [node_begin, node_end) (global node indices per direction, inside
 * the nodes the handle's elements touch) into r / A_values.  Every node must be gathered exactly once per integrate()
 * for the sum of the calls to equal one add_residual_and_grad (bitwise: same kernels, same order of additions per row).
 * Two-phase tensor paths only (3-D, degree 2 or 3, structured CSR, analytic tangent), device-resident arguments only;
 * other handles report an error and the caller stays with the one-call form. */
int mimi_hip_domain_integrate(mimi_hip_domain_t h, const double* u);
int mimi_hip_domain_gather(mimi_hip_domain_t h, double grad_factor, double* r, double* A_values,
                           const int32_t node_begin[3], const int32_t node_end[3]);
/* measurement aid (bench.py): HIP events on the launch stream around the two phases of the last two-phase tangent
 * assembly -- phase 1 = the integration kernel(s), phase 2 = the row gather.  Off by default (three event records per
 * call when on). */
int mimi_hip_domain_set_phase_timing(mimi_hip_domain_t h, int on);
int mimi_hip_domain_phase_ms(mimi_hip_domain_t h, double* phase1_ms, double* phase2_ms);
/* the same with phase 1 split at the end of its material pre-pass (0 when the path has none): pre-pass, integration /
 * contraction kernel, row gather */
int mimi_hip_domain_phase_ms_detail(mimi_hip_domain_t h, double* prepass_ms, double* integration_ms, double* gather_ms);
/* sizes: what = 0 n_elements, 1 n_quad, 2 n_dof, 3 nnz, 4 n_vdofs, 5 path (0 general, 1 tensor), 6 CSR kind (0 any,
 * 1 structured lexicographic, 2 structured permuted), 7 kernel family of the last assembly on the handle (0 none yet,
 * 1 two-phase tensor degree 2, 2 two-phase tensor degree 3, 3 small-element tensor, 4 general) */
int64_t mimi_hip_domain_info(mimi_hip_domain_t h, int what);

/* ---- structured sparsity: PrecomputedData::PrepareSparsity (precomputed.cpp:151-174) ----
 * CSR pattern of a single B-spline patch with lexicographic node numbering, built on the
 * device.  Call once with col == NULL to fill rowptr ([n_vdofs+1], device or host) and get
 * nnz, then again with col ([nnz], device or host). */
int mimi_hip_bspline_sparsity(int32_t dim, const int32_t n_nodes_dir[3], const int32_t degree[3],
                              int device, int64_t* rowptr, int32_t* col, int64_t* nnz);
/* Row slice of the same pattern, for a rank that owns a box of nodes (one slab of the patch): rowptr keeps its full
 * length [n_vdofs+1], rows of nodes outside [node_begin, node_end) get zero length, and col / the value array hold
 * only the slice (nnz = the slice's entries).  A domain handle accepts such a pattern when the box covers every node
 * its elements touch: the integration kernels only ever write rows of those nodes.  Column indices stay global. */
int mimi_hip_bspline_sparsity_rows(int32_t dim, const int32_t n_nodes_dir[3], const int32_t degree[3],
                                   const int32_t node_begin[3], const int32_t node_end[3], int device,
                                   int64_t* rowptr, int32_t* col, int64_t* nnz);

/* ---- exchange step of a sharded assembly (no reference counterpart beyond the in-process reduction of
 * integrators/nonlinear_base.hpp:90-151) ----
 * The rows `rows[0..n_rows)` (distinct, device int64) of the residual `r` and of the CSR value array travel as one
 * message: [n_rows residual entries][the rows' values, row after row]; offsets[k] (device int64) = position of row k's
 * values in the message (n_rows + the lengths of the rows before it).  A_values == NULL: residual entries only.
 * All pointers are device pointers; the kernels are enqueued on `stream` (NULL or MIMI_HIP_STREAM_NULL: the null stream)
 * of the current device and return at once. */
int mimi_hip_rows_zero(void* stream, const int64_t* rowptr, const int64_t* rows, int64_t n_rows, double* r, double* A_values);
int mimi_hip_rows_pack(void* stream, const int64_t* rowptr, const int64_t* rows, const int64_t* offsets, int64_t n_rows,
                       const double* r, const double* A_values, double* message);
int mimi_hip_rows_unpack_add(void* stream, const int64_t* rowptr, const int64_t* rows, const int64_t* offsets, int64_t n_rows,
                             const double* message, double* r, double* A_values);
/* The same message with the rows trimmed to the entries the sender's elements can have written (ABI 12): of a shared row
 * only the columns inside the sender's own node planes are not zero by construction (3 of the 5 column planes of a degree-2
 * row), so only those travel: [n_rows residual entries][A_values[positions[0..n_positions)]].  positions (device int64,
 * distinct) are places in the value array; sender and receiver list the same (row, column) pairs in the same order, each in
 * its own array (mimi_amd/parallel.py InterfaceExchange builds both from the slab bounds).  A_values == NULL: residual
 * entries only. */
int mimi_hip_entries_pack(void* stream, const int64_t* rows, int64_t n_rows, const int64_t* positions, int64_t n_positions,
                          const double* r, const double* A_values, double* message);
int mimi_hip_entries_unpack_add(void* stream, const int64_t* rows, int64_t n_rows, const int64_t* positions, int64_t n_positions,
                                const double* message, double* r, double* A_values);

/* ---- contact integrator: integrators::MortarContact ------------------------------ */
typedef struct mimi_hip_contact_s* mimi_hip_contact_t;

enum mimi_hip_rigid_body_kind {
  MIMI_HIP_BODY_SPHERE = 0, /* params: centre[3], radius */
  MIMI_HIP_BODY_PLANE = 1,  /* params: point[3], unit normal[3] pointing out of the rigid half space */
  MIMI_HIP_BODY_SPLINE = 2  /* mimi_hip_contact_tables::spline: one B-spline / NURBS curve (2-D) or surface (3-D) */
};

/* NearestDistanceToSplines with its one boundary spline (coefficients/nearest_distance.hpp:215-288): para_dim = dim - 1.
 * Orientation as in the reference (Results::ComputeNormal, :139-184): the rigid normal is (t_y, -t_x) of the curve
 * tangent in 2-D, S_u x S_v in 3-D, and must point out of the rigid body. */
typedef struct mimi_hip_spline_body {
  int32_t para_dim;
  int32_t degree[2];             /* <= 5 */
  int32_t n_knots[2];
  const double* knots[2];        /* host */
  const double* control_points;  /* host, [n_ctrl][dim], first parametric direction fastest */
  const double* weights;         /* host, [n_ctrl], or NULL */
  int32_t kdtree_resolution;     /* PlantKdTree(resolution, nthreads) (:243-255): sample points per direction for the
                                    initial guess of the search */
  int32_t max_iterations;        /* Query::max_iterations (:29); <= 0: 50 */
} mimi_hip_spline_body;

/* Flat form of the boundary-element tables MortarContact::Prepare builds
 * (mortar_contact.cpp:19-133) plus the analytic rigid body standing in for
 * NearestDistanceBase (coefficients/nearest_distance.hpp:14-213). */
typedef struct mimi_hip_contact_tables {
  int32_t dim;
  int32_t n_faces;
  int32_t n_dof;        /* per face, <= 16 */
  int32_t n_quad;       /* <= 25 */
  int64_t n_nodes;
  const int32_t* dofs;  /* [n_faces][n_dof] global node ids of the marked boundary elements */
  const double* N;      /* [n_faces][n_quad][n_dof]          QuadData::N */
  const double* dN_dxi; /* [n_faces][n_quad][dim-1][n_dof]   QuadData::dN_dxi */
  const double* weight; /* [n_faces][n_quad]                 QuadData::integration_weight */
  const double* x_ref;  /* [n_nodes][dim] reference coordinates of the control points (byVDIM) */
  int32_t body_kind;
  double body[8];
  double penalty;       /* NearestDistanceBase::coefficient_ (nearest_distance.hpp:18) */
  const int64_t* csr_rowptr;
  const int32_t* csr_col;
  const mimi_hip_spline_body* spline;  /* body_kind == MIMI_HIP_BODY_SPLINE, else NULL */
} mimi_hip_contact_tables;

int mimi_hip_contact_create(const mimi_hip_contact_tables* tables, int device, mimi_hip_contact_t* out);
int mimi_hip_contact_destroy(mimi_hip_contact_t h);
int mimi_hip_contact_set_tangent_mode(mimi_hip_contact_t h, int mode);
int mimi_hip_contact_set_stream(mimi_hip_contact_t h, void* stream);
int mimi_hip_contact_synchronize(mimi_hip_contact_t h);
/* AddBoundaryResidual (mortar_contact.cpp:297-351) */
int mimi_hip_contact_add_residual(mimi_hip_contact_t h, const double* u, double* r);
/* AddBoundaryResidualAndGrad (mortar_contact.cpp:353-421) */
int mimi_hip_contact_add_residual_and_grad(mimi_hip_contact_t h, const double* u, double grad_factor,
                                           double* r, double* A_values);
/* GapNorm (mortar_contact.cpp:423-467) */
int mimi_hip_contact_gap_norm(mimi_hip_contact_t h, const double* u, double* out);
/* last_area_, last_pressure_, last_force_[dim] of the latest Add* call
 * (mortar_contact.hpp:33-35; BoundaryPostTimeAdvance mortar_contact.cpp:469-488):
 * out[0] = area, out[1] = pressure integral, out[2..2+dim) = force */
int mimi_hip_contact_last_history(mimi_hip_contact_t h, double* out5);
/* nodal average_pressure_ (mortar_contact.hpp:59), [n_marked]; returns n_marked via *n */
int mimi_hip_contact_get_pressure(mimi_hip_contact_t h, double* out, int64_t capacity, int64_t* n);

/* The rigid body moved (examples/nl_contact.py moves the curve's control points and calls plant_kd_tree again before
 * every step) and / or the penalty changed (scene.coefficient = ...): spline != NULL re-uploads the boundary spline and its
 * sampled initial guesses (same layout rules as at create time); penalty > 0 replaces coefficient_. */
int mimi_hip_contact_update_body(mimi_hip_contact_t h, const mimi_hip_spline_body* spline, double penalty);

/* Multi-GPU (one handle per rank over the faces of its element slab): the nodal area / gap of nodes shared between
 * slabs must be summed over the ranks before the pressure is formed (mortar_contact.cpp:195-261 runs over all marked
 * faces).  gap_area = pass 1 only; marked_nodes = the sorted global node ids behind the nodal arrays (out == NULL: count
 * only); nodal = read (set 0) or write (set 1) the nodal area / gap arrays [n_marked], host or device pointers;
 * add_residual_from_nodal = pressure + pass 2 (A_values == NULL: residual only). */
int mimi_hip_contact_gap_area(mimi_hip_contact_t h, const double* u);
int mimi_hip_contact_marked_nodes(mimi_hip_contact_t h, int32_t* out, int64_t capacity, int64_t* n);
int mimi_hip_contact_nodal(mimi_hip_contact_t h, int set, double* area, double* gap);
int mimi_hip_contact_add_residual_from_nodal(mimi_hip_contact_t h, const double* u, double grad_factor, double* r,
                                             double* A_values);

/* ---- the callers' steps around the assembly, device-resident ---------------------------
 * One handle per CSR pattern (rowptr / col host or device; device arrays are used in place and must outlive the
 * handle) and list of essential dofs (forms::Nonlinear's zero_dofs). */
typedef struct mimi_hip_linear_s* mimi_hip_linear_t;
int mimi_hip_linear_create(int64_t n, const int64_t* csr_rowptr, const int32_t* csr_col, const int64_t* ess_dofs,
                           int64_t n_ess, int device, mimi_hip_linear_t* out);
int mimi_hip_linear_destroy(mimi_hip_linear_t h);
/* NULL = the handle's own stream */
int mimi_hip_linear_set_stream(mimi_hip_linear_t h, void* hip_stream);
/* what: 0 rows, 1 stored entries, 2 consecutive rows that share one column list (3 or 2: the dofs of a node in the byVDIM
 * numbering of py_nonlinear_solid.cpp:63 -- the products read the list once per group; 1: any other pattern), 3 whether
 * that shared list is made of node triples 3 c, 3 c + 1, 3 c + 2 (1: the products read one index per node, a ninth of the
 * index bytes).  -1: bad argument. */
int64_t mimi_hip_linear_info(mimi_hip_linear_t h, int what);
/* forms::Nonlinear::AddMult / AddMultGrad tail (forms/nonlinear.hpp:76-80,112-115): r[ess] = 0 (r may be NULL);
 * A.EliminateRowCol(ess, DIAG_ONE) on the CSR values (A_values may be NULL).  Host or device pointers. */
int mimi_hip_linear_eliminate(mimi_hip_linear_t h, double* r, double* A_values);
/* y += alpha A x on the handle's pattern (mfem::SparseMatrix::AddMult: `mass_->Mult(a, y)`, `viscosity_->AddMult(v, y)` of
 * operators::NonlinearSolid::Mult / ResidualAndGrad, operators/nonlinear_solid.cpp:172-205,240-283).  Host or device. */
int mimi_hip_linear_add_mult(mimi_hip_linear_t h, const double* A_values, const double* x, double alpha, double* y);
/* The reference's iterative linear solver (py/py_nonlinear_solid.cpp:329-339: mfem::GMRESSolver + mfem::DSmoother,
 * rel 1e-8, abs 1e-12, 300 iterations; kdim <= 0 = mfem's default 50): x = 0 start, left-preconditioned restarted
 * GMRES, modified Gram-Schmidt, stops when the preconditioned residual estimate <= max(rel_tol * ||M b||, abs_tol).
 * use_jacobi = 0 runs it unpreconditioned.  A_values / b / x host or device. */
int mimi_hip_linear_gmres(mimi_hip_linear_t h, const double* A_values, const double* b, double* x, double rel_tol,
                          double abs_tol, int max_iter, int kdim, int use_jacobi, int32_t* iterations,
                          double* final_norm, int32_t* converged);
/* The mass solve of operators::NonlinearSolid (operators/nonlinear_solid.cpp:39-50,155; .hpp:38-42: mfem::CGSolver +
 * mfem::DSmoother, rel 1e-8, abs 1e-12, 1000 iterations): x = 0 start, preconditioned conjugate gradients, stops when
 * (r, M r) <= max(rel_tol^2 (r0, M r0), abs_tol^2); final_norm = sqrt of that product. */
int mimi_hip_linear_cg(mimi_hip_linear_t h, const double* A_values, const double* b, double* x, double rel_tol,
                       double abs_tol, int max_iter, int use_jacobi, int32_t* iterations, double* final_norm,
                       int32_t* converged);

#ifdef __cplusplus
}
#endif
#endif /* MIMI_HIP_H */

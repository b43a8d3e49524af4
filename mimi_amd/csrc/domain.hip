// C ABI of the domain integrator (include/mimi_hip.h): handle life cycle, table upload /
// generation, kernel dispatch.  Reference counterparts:
//   NonlinearSolid::Prepare                      integrators/nonlinear_solid.cpp:31-46
//   NonlinearSolid::AddDomainResidual            integrators/nonlinear_solid.cpp:151-160
//   NonlinearSolid::AddDomainResidualAndGrad     integrators/nonlinear_solid.cpp:162-177
//   NonlinearSolid::DomainPostTimeAdvance        integrators/nonlinear_solid.cpp:179-199
#include "domain.hpp"
#include "kernels_general.hpp"
#include "kernels_setup.hpp"
#include "kernels_tensor.hpp"
#include "kernels_tensor_2phase.hpp"
#include "kernels_tensor_wgs.hpp"
#include "kernels_tensor_wgsym.hpp"
#include "kernels_tensor_residual.hpp"
#include "kernels_tensor_small.hpp"
#include "tensor_dispatch.hpp"

#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>

namespace mimi_hip {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }

MaterialDev make_material_dev(const mimi_hip_material& m) {
  MaterialDev d{};
  d.m = m;
  d.const_temperature_contribution = 1.0;
  if (m.kind == MIMI_HIP_MAT_J2 || m.kind == MIMI_HIP_MAT_J2SIMO || m.kind == MIMI_HIP_MAT_J2LOG) {
    if (m.hardening < MIMI_HIP_HARD_POWERLAW || m.hardening > MIMI_HIP_HARD_JC_CONST_TEMP)
      fail("hardening missing for J2 / J2Simo / J2Log");  // materials.cpp:139-148,177-183,217-223
    d.sigma_y_ref = (m.hardening == MIMI_HIP_HARD_POWERLAW || m.hardening == MIMI_HIP_HARD_VOCE) ? m.sigma_y : m.A;
    if (m.hardening >= MIMI_HIP_HARD_JC_TEMP_RATE && m.reference_temperature > m.melting_temperature)
      fail("reference temperature, %g ,can't be bigger than melting temperature, %g .",
           m.reference_temperature, m.melting_temperature);  // material_hardening.hpp:228-238
    if (m.hardening == MIMI_HIP_HARD_JC_CONST_TEMP) {
      d.const_temperature_contribution =
          1.0 - std::pow((m.initial_temperature - m.reference_temperature)
                             / (m.melting_temperature - m.reference_temperature), m.m);
      if (d.const_temperature_contribution <= 0.0)
        fail("Invalid temperature contribution %g", d.const_temperature_contribution);
    }
  } else if (m.kind == MIMI_HIP_MAT_J2LINEAR) {
    d.sigma_y_ref = m.sigma_y;
  } else if (m.kind != MIMI_HIP_MAT_NEOHOOKEAN && m.kind != MIMI_HIP_MAT_STVK) {
    fail("unknown material kind %d", m.kind);
  }
  return d;
}

template<typename F>
static int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  } catch (...) {
    set_last_error("unknown error");
    return 1;
  }
}

static void check_status(mimi_hip_domain_s* h) {
  MH_HIP(hipMemcpyAsync(h->status_host, h->status_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  MH_HIP(hipStreamSynchronize(h->stream));
  const int s = *h->status_host;
  if (s) {
    MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
    if (s & 1) fail("ScalarSolve: root not bracketed by input bounds.");          // solvers/newton.hpp:81-93
    if (s & 2) fail("ScalarSolve: failed to converge in allotted iterations.");   // solvers/newton.hpp:120-132
    if (s & 4) fail("CSR pattern does not contain an element's dof block");
    if (s & 8) fail("geometry map has a non-positive Jacobian determinant");
    fail("device status %d", s);
  }
}

static void init_common(mimi_hip_domain_s* h, int device, const mimi_hip_material* material) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    fail("libmimi_hip: no HIP device visible -- this library has no CPU fallback");
  if (device < 0 || device >= count) fail("device %d out of range (%d visible)", device, count);
  h->device = device;
  MH_HIP(hipSetDevice(device));
  MH_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  h->mat = make_material_dev(*material);
  MH_HIP(hipMalloc(reinterpret_cast<void**>(&h->status_dev), sizeof(int)));
  MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
  MH_HIP(hipHostMalloc(reinterpret_cast<void**>(&h->status_host), sizeof(int), hipHostMallocDefault));
  *h->status_host = 0;
}

static bool material_has_state(int kind) { return kind != MIMI_HIP_MAT_NEOHOOKEAN && kind != MIMI_HIP_MAT_STVK; }
// closed-form tangents inside the kernels (every kernel family) vs the other materials (materials_other.hpp: general
// kernels, and the two-phase tensor kernels through the tangent record of the material pre-pass)
static bool material_closed_form(int kind) { return kind == MIMI_HIP_MAT_NEOHOOKEAN || kind == MIMI_HIP_MAT_J2; }

static void init_state(mimi_hip_domain_s* h) {
  h->n_pts = (int64_t)h->n_el * h->n_q;
  const int kind = h->mat.m.kind;
  if (!material_has_state(kind)) return;
  // CreateState (materials.cpp:120-133 J2Linear, 151-166 J2, 185-208 J2Simo, 225-252 J2Log): zero matrices / eqps,
  // T = initial; J2Simo: be_old = F_old = I; J2Log: Fp_inv = I
  const int dd = h->dim * h->dim;
  h->eqps.resize(h->n_pts);
  h->temperature.resize(h->n_pts);
  h->plastic_strain.resize(h->n_pts * dd);
  MH_HIP(hipMemsetAsync(h->eqps.ptr, 0, h->n_pts * sizeof(double), h->stream));
  MH_HIP(hipMemsetAsync(h->plastic_strain.ptr, 0, h->n_pts * dd * sizeof(double), h->stream));
  const bool two = kind == MIMI_HIP_MAT_J2LINEAR || kind == MIMI_HIP_MAT_J2SIMO;
  if (two) {
    h->state2.resize(h->n_pts * dd);
    MH_HIP(hipMemsetAsync(h->state2.ptr, 0, h->n_pts * dd * sizeof(double), h->stream));
  }
  std::vector<double> T(h->n_pts, kind == MIMI_HIP_MAT_J2LINEAR ? 0.0 : h->mat.m.initial_temperature);
  MH_HIP(hipMemcpyAsync(h->temperature.ptr, T.data(), h->n_pts * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (kind == MIMI_HIP_MAT_J2SIMO || kind == MIMI_HIP_MAT_J2LOG) {
    std::vector<double> ones(h->n_pts, 1.0);
    for (int i = 0; i < h->dim; ++i) {
      const size_t c = (size_t)i * (h->dim + 1);   // diagonal component of the SoA [component][point] layout
      MH_HIP(hipMemcpyAsync(h->plastic_strain.ptr + c * h->n_pts, ones.data(), h->n_pts * sizeof(double), hipMemcpyHostToDevice, h->stream));
      if (kind == MIMI_HIP_MAT_J2SIMO)
        MH_HIP(hipMemcpyAsync(h->state2.ptr + c * h->n_pts, ones.data(), h->n_pts * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    MH_HIP(hipStreamSynchronize(h->stream));
  }
  MH_HIP(hipStreamSynchronize(h->stream));
}

// pair_pos[e][a][b]: offset of column dofs[b] * dim inside row dofs[a] * dim (what the general and colour kernels scatter
// through); col = the CSR columns, host or device
static void build_pair_pos(mimi_hip_domain_s* h, const int32_t* col) {
  DeviceBuffer<int32_t> col_tmp;
  const int32_t* col_dev = col;
  if (!is_device_pointer(col)) {
    col_tmp.assign(col, h->nnz, h->stream);
    col_dev = col_tmp.ptr;
  }
  const int64_t total = (int64_t)h->n_el * h->n_dof * h->n_dof;
  h->pair_pos.resize(total);
  const int threads = 256;
  const int64_t blocks = (total + threads - 1) / threads;
  hipLaunchKernelGGL(pair_pos_kernel, dim3((unsigned)blocks), dim3(threads), 0, h->stream, h->n_el, h->n_dof,
                     h->dim, h->dofs.ptr, h->rowptr, col_dev, h->pair_pos.ptr, h->status_dev);
  MH_HIP(hipGetLastError());
  check_status(h);
}

// Handles whose CSR is the (possibly permuted) structured pattern do not build pair_pos at create time: the two-phase
// kernels never use it.  The fallback kernels get it here, from columns regenerated out of the pattern's closed form.
static void ensure_pair_pos(mimi_hip_domain_s* h) {
  if (h->pair_pos.ptr) return;
  if (!(h->structured_csr || h->structured_perm)) fail("pair positions were not built for this handle");
  SparsityDev S{};
  S.dim = h->dim;
  for (int d = 0; d < 3; ++d) {
    S.n[d] = d < h->dim ? h->n_ctrl[d] : 1;
    S.p[d] = d < h->dim ? h->degree[d] : 0;
    S.prefix[d] = nullptr;
  }
  DeviceBuffer<int32_t> col;
  col.resize((size_t)h->nnz);
  if (h->structured_csr) {
    const int64_t n_rows = h->n_vdofs;
    hipLaunchKernelGGL(structured_col_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, h->stream, S, n_rows, h->rowptr,
                       col.ptr, 0, h->status_dev);
  } else {
    if (h->degree[0] == 3)
      hipLaunchKernelGGL((permuted_col_kernel<uint16_t, 343>), dim3((unsigned)((h->n_nodes + 3) / 4)), dim3(256), 0, h->stream, S,
                         (int64_t)h->n_nodes, h->node_ids.ptr, h->rowptr, h->nbr_pos16.ptr, col.ptr);
    else
      hipLaunchKernelGGL((permuted_col_kernel<unsigned char, 125>), dim3((unsigned)((h->n_nodes + 3) / 4)), dim3(256), 0, h->stream, S,
                         (int64_t)h->n_nodes, h->node_ids.ptr, h->rowptr, h->nbr_pos.ptr, col.ptr);
  }
  MH_HIP(hipGetLastError());
  build_pair_pos(h, col.ptr);
  MH_HIP(hipStreamSynchronize(h->stream));
}

static void setup_csr(mimi_hip_domain_s* h, const int64_t* rowptr, const int32_t* col, bool need_pair_pos) {
  if (!rowptr || !col) fail("csr_rowptr / csr_col must be given");
  // rowptr: keep a device copy unless it already lives there
  if (is_device_pointer(rowptr)) {
    h->rowptr = rowptr;
  } else {
    h->rowptr_own.assign(rowptr, h->n_vdofs + 1, h->stream);
    h->rowptr = h->rowptr_own.ptr;
  }
  int64_t nnz = 0;
  MH_HIP(hipMemcpyAsync(&nnz, h->rowptr + h->n_vdofs, sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  MH_HIP(hipStreamSynchronize(h->stream));
  h->nnz = nnz;
  if (need_pair_pos) build_pair_pos(h, col);
}

static PatchDev patch_dev(mimi_hip_domain_s* h, const double* ctrl) {
  PatchDev P{};
  const int dim = h->dim;
  P.dim = dim;
  for (int d = 0; d < 3; ++d) {
    P.p[d] = d < dim ? h->degree[d] : 0;
    P.nq[d] = d < dim ? h->nq1[d] : 1;
    P.n_ctrl[d] = d < dim ? h->n_ctrl[d] : 1;
    P.box_begin[d] = d < dim ? h->el_begin[d] : 0;
    P.box_n[d] = d < dim ? h->el_end[d] - h->el_begin[d] : 1;
    P.B[d] = d < dim ? h->tab1d.ptr + h->tab_off_B[d] : nullptr;
    P.D[d] = d < dim ? h->tab1d.ptr + h->tab_off_D[d] : nullptr;
    P.W[d] = d < dim ? h->tab1d.ptr + h->tab_off_W[d] : nullptr;
    P.first[d] = d < dim ? h->first1d.ptr + h->first_off[d] : nullptr;
  }
  P.ctrl = ctrl;
  P.node_ids = h->node_ids.ptr;
  P.n_dof = h->n_dof;
  P.n_q = h->n_q;
  P.n_el = h->n_el;
  return P;
}

// reference-layout tables (utils/precomputed.cpp:316-321) from the compact geometry, on demand
static void ensure_general_tables(mimi_hip_domain_s* h) {
  ensure_pair_pos(h);
  if (h->dN_dX.ptr) return;
  if (!h->geo.ptr) fail("no tables to integrate with");
  const int64_t npts = (int64_t)h->n_el * h->n_q;
  h->dN_dX.resize((size_t)npts * h->n_dof * h->dim);
  h->wdet.resize((size_t)npts);
  PatchDev P = patch_dev(h, nullptr);
  const int threads = 256;
  const int64_t total = npts * h->n_dof;
  const int64_t blocks = (total + threads - 1) / threads;
  if (h->dim == 2)
    hipLaunchKernelGGL(expand_tables_kernel<2>, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->geo.ptr, h->dofs.ptr, h->dN_dX.ptr, h->wdet.ptr);
  else
    hipLaunchKernelGGL(expand_tables_kernel<3>, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->geo.ptr, h->dofs.ptr, h->dN_dX.ptr, h->wdet.ptr);
  MH_HIP(hipGetLastError());
}

static GeneralArgs general_args(mimi_hip_domain_s* h, const double* u, double* r, double* A, double gf) {
  GeneralArgs a{};
  a.n_el = h->n_el;
  a.n_dof = h->n_dof;
  a.n_q = h->n_q;
  a.dofs = h->dofs.ptr;
  a.dN_dX = h->dN_dX.ptr;
  a.wdet = h->wdet.ptr;
  a.rowptr = h->rowptr;
  a.pair_pos = h->pair_pos.ptr;
  a.u = u;
  a.r = r;
  a.A = A;
  a.grad_factor = gf;
  a.dt = h->dt;
  a.mat = h->mat;
  a.state = StateView{h->eqps.ptr, h->temperature.ptr, h->plastic_strain.ptr, h->n_pts, h->state2.ptr};
  a.status = h->status_dev;
  return a;
}

// two-phase general path: element blocks / residual vectors densely into scratch_k / scratch_r, then
// general_gather_kernel.  Needs n_el * n_tdof^2 doubles (77 GB at 128 x 128 x 16 p = 3) and the node -> element adjacency;
// falls back to the atomics when the scratch does not fit (MIMI_HIP_GENERAL_NO_TWO_PHASE=1 forces that)
static bool ensure_general_two_phase(mimi_hip_domain_s* h, bool with_k) {
  static const bool off = getenv("MIMI_HIP_GENERAL_NO_TWO_PHASE") && getenv("MIMI_HIP_GENERAL_NO_TWO_PHASE")[0] == '1';
  if (off || h->general_two_phase_failed) return false;
  const size_t n_tdof = (size_t)h->n_dof * h->dim;
  // what rules the path out is checked BEFORE anything is allocated (3-D degree >= 4 would be 1.1 MB per element)
  if (!h->adj_ptr.ptr) {
    // the gather kernel keeps one CSR row in LDS: rows longer than its image -> atomics
    std::vector<int64_t> rp((size_t)h->n_vdofs + 1);
    MH_HIP(hipMemcpy(rp.data(), h->rowptr, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    int64_t longest = 0;
    for (int64_t v = 0; v < h->n_vdofs; ++v) longest = std::max(longest, rp[v + 1] - rp[v]);
    if (longest > GG_MAX_ROW || h->n_dof > 64) {
      h->general_two_phase_failed = true;
      return false;
    }
  }
  const size_t need = with_k ? (size_t)h->n_el * n_tdof * n_tdof : 0;
  if (h->scratch_k.count < need) {
    size_t free_b = 0, total_b = 0;
    MH_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t have = h->scratch_k.count * sizeof(double);
    if (need * sizeof(double) > free_b + have || need * sizeof(double) > (total_b / 2)) {
      h->general_two_phase_failed = true;
      h->scratch_k.release();          // (whatever a residual-only call left: the atomics route needs none of it)
      h->scratch_r.release();
      return false;
    }
    h->scratch_k.resize(need);
  }
  h->scratch_r.resize((size_t)h->n_el * n_tdof);
  if (!h->adj_ptr.ptr) {
    const size_t n = (size_t)h->n_el * h->n_dof;
    std::vector<int32_t> dofs(n);
    MH_HIP(hipMemcpy(dofs.data(), h->dofs.ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int64_t n_nodes = h->n_vdofs / h->dim;
    std::vector<int64_t> ptr(n_nodes + 1, 0);
    for (size_t k = 0; k < n; ++k) ++ptr[dofs[k] + 1];
    for (int64_t v = 0; v < n_nodes; ++v) ptr[v + 1] += ptr[v];
    std::vector<int32_t> adj(n);
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (size_t k = 0; k < n; ++k)    // entry = (element << 6) | local node
      adj[fill[dofs[k]]++] = (int32_t)(((k / h->n_dof) << 6) | (k % h->n_dof));
    h->adj_ptr.assign(ptr.data(), ptr.size(), h->stream);
    h->adj.assign(adj.data(), adj.size(), h->stream);
  }
  return true;
}

// calls f(std::integral_constant<int, kind>) for the handle's material among the four that are not closed-form: the general
// kernels take the kind as a compile-time constant (one instantiation per material: no spilled registers)
template<class F>
static void by_other_kind(int kind, F f) {
  switch (kind) {
  case MIMI_HIP_MAT_STVK: f(std::integral_constant<int, MIMI_HIP_MAT_STVK>{}); break;
  case MIMI_HIP_MAT_J2LINEAR: f(std::integral_constant<int, MIMI_HIP_MAT_J2LINEAR>{}); break;
  case MIMI_HIP_MAT_J2SIMO: f(std::integral_constant<int, MIMI_HIP_MAT_J2SIMO>{}); break;
  default: f(std::integral_constant<int, MIMI_HIP_MAT_J2LOG>{}); break;
  }
}

#ifndef GEN_BIG_PP
#define GEN_BIG_PP 8
#define GEN_BIG_THREADS 512
#endif
template<int DIM>
static void launch_general(mimi_hip_domain_s* h, int grad, const GeneralArgs& a_in) {
  GeneralArgs a = a_in;
  static const bool no_mfma_env = getenv("MIMI_HIP_GENERAL_NO_MFMA") && getenv("MIMI_HIP_GENERAL_NO_MFMA")[0] == '1';
  (void)no_mfma_env;
  const bool two_phase = ensure_general_two_phase(h, grad != 0);
  if (!two_phase && grad) consume_base(h, a.A);      // (atomics into the values in place)
  const double* A_old = (h->A_base && a.A) ? h->A_base : a.A;
  a.scratch_k = (two_phase && grad) ? h->scratch_k.ptr : nullptr;
  a.scratch_r = two_phase ? h->scratch_r.ptr : nullptr;
  // (called after the element kernel of every route below)
  auto gather = [&]() {
    if (!two_phase) return;
    const int64_t n_rows = h->n_vdofs;
    auto kernel = grad ? general_gather_kernel<DIM, 1> : general_gather_kernel<DIM, 0>;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((n_rows + GG_WAVES - 1) / GG_WAVES)), dim3(64 * GG_WAVES), 0, h->stream, n_rows,
                       h->n_dof, h->rowptr, h->adj_ptr.ptr, h->adj.ptr, h->pair_pos.ptr, h->scratch_k.ptr, h->scratch_r.ptr,
                       a.grad_factor, A_old, a.A, a.r);
    MH_HIP(hipGetLastError());
  };
  size_t lds = general_lds_bytes(DIM, h->n_dof, h->n_q, grad);
  if (lds > 160 * 1024) fail("element too large for LDS (%zu bytes)", lds);
  // tangent assemblies of the materials without a closed-form tangent: w det P, w det dP/dF per point from a kernel of their own
  auto material_prepass = [&]() {
    constexpr int DD = DIM * DIM;
    h->mat_rec.resize((size_t)h->n_el * h->n_q * (DD + DD * DD));
    a.mat_rec = h->mat_rec.ptr;
    by_other_kind(h->mat.m.kind, [&](auto K) {
      constexpr int FK = decltype(K)::value;
      hipLaunchKernelGGL((general_material_kernel<DIM, FK>), dim3(h->n_el), dim3(64), (size_t)h->n_dof * DIM * sizeof(double), h->stream, a);
      MH_HIP(hipGetLastError());
    });
  };
  // small elements (one pass of the node-pair phase fits one wave: 2-D p <= 3, 3-D p = 1): one wave per element, four
  // elements per workgroup (MIMI_HIP_GENERAL_NO_WPE=1: one workgroup per element as for the large ones)
  static const bool no_wpe = getenv("MIMI_HIP_GENERAL_NO_WPE") && getenv("MIMI_HIP_GENERAL_NO_WPE")[0] == '1';
  const bool wpe = !no_wpe && grad != 2 && h->n_dof * ((h->n_dof + 2) / 3) <= 128 && h->n_q <= 64;
  if (wpe) {
    a.lds_per_element = (int)((lds + 15) / 16 * 16);
    lds = (size_t)a.lds_per_element * 4;
    auto gow = [&](auto kernel) {
      if (lds > 64 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
      hipLaunchKernelGGL(kernel, dim3((unsigned)((h->n_el + 3) / 4)), dim3(256), lds, h->stream, a);
      MH_HIP(hipGetLastError());
    };
    const bool other = !material_closed_form(h->mat.m.kind);
    if (!other) {
      if (grad == 0) gow(domain_general_kernel<DIM, 0, 3, 256, 0, 0, 1>); else gow(domain_general_kernel<DIM, 1, 3, 256, 0, 0, 1>);
    } else if (grad == 1) {
      material_prepass();
      gow(domain_general_kernel<DIM, 1, 3, 256, 1, 0, 1>);
    } else {
      by_other_kind(h->mat.m.kind, [&](auto K) {
        constexpr int FK = decltype(K)::value;
        gow(domain_general_kernel<DIM, 0, 3, 256, FK, 0, 1>);
      });
    }
    gather();
    return;
  }
  auto go = [&](auto kernel, int threads = 256) {
    if (lds > 64 * 1024)
      ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
    hipLaunchKernelGGL(kernel, dim3(h->n_el), dim3(threads), lds, h->stream, a);
    MH_HIP(hipGetLastError());
  };
  // MIMI_HIP_GENERAL_NO_MFMA=1: the vector-pipe node-pair phase also for 64-node elements (A/B comparisons)
  static const bool no_mfma = getenv("MIMI_HIP_GENERAL_NO_MFMA") && getenv("MIMI_HIP_GENERAL_NO_MFMA")[0] == '1';
  if (!material_closed_form(h->mat.m.kind)) {
    // the other materials (materials_other.hpp): the tangent assembly takes P and dP/dF from the material pre-pass (one lean
    // kernel per material), the residual-only and reference-FD assemblies evaluate the stress in the element kernel
    if (grad == 1) {
      material_prepass();
      if (DIM == 3 && h->n_dof == 64 && !no_mfma) go(domain_general_kernel<3, 1, GEN_BIG_PP, GEN_BIG_THREADS, 1, 1>, GEN_BIG_THREADS);
      else if (h->n_dof * h->n_dof > 3 * 256) go(domain_general_kernel<DIM, 1, GEN_BIG_PP, GEN_BIG_THREADS, 1>, GEN_BIG_THREADS);
      else go(domain_general_kernel<DIM, 1, 3, 256, 1>);
    } else {
      by_other_kind(h->mat.m.kind, [&](auto K) {
        constexpr int FK = decltype(K)::value;
        if (grad == 0) go(domain_general_kernel<DIM, 0, 3, 256, FK>);
        else go(domain_general_kernel<DIM, 2, 3, 256, FK>);
      });
    }
    gather();
    return;
  }
  if (grad == 0) go(domain_general_kernel<DIM, 0>);
  else if (grad == 1 && DIM == 3 && h->n_dof == 64 && !no_mfma) go(domain_general_kernel<3, 1, GEN_BIG_PP, GEN_BIG_THREADS, 0, 1>, GEN_BIG_THREADS);
  else if (grad == 1 && h->n_dof * h->n_dof > 3 * 256) go(domain_general_kernel<DIM, 1, GEN_BIG_PP, GEN_BIG_THREADS>, GEN_BIG_THREADS);
  else if (grad == 1) go(domain_general_kernel<DIM, 1, 3>);
  else go(domain_general_kernel<DIM, 2>);
  gather();
}

// small elements on the tensor path (kernels_tensor_small.hpp): element kernel from the 1-D tables, then the general
// path's gather (adjacency, pair positions).  mode 0 residual, 1 residual + tangent, 2 post time advance
template<int DIM, int P>
static void launch_tensor_small_dp(mimi_hip_domain_s* h, int mode, TensorArgs a) {
  using S = SmallShape<DIM, P>;
  const bool other = !material_closed_form(h->mat.m.kind);
  const size_t lds = (size_t)4 * (mode == 1 ? S::total1 : S::total0) * sizeof(double);
  const unsigned blocks = (unsigned)((h->n_el + 3) / 4);
  auto go = [&](auto kernel) {
    if (lds > 64 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, h->stream, a, (int)h->n_el);
    MH_HIP(hipGetLastError());
  };
  if (!other) {
    if (mode == 0) go(tensor_small_kernel<DIM, P, 0, 0>);
    else if (mode == 1) go(tensor_small_kernel<DIM, P, 0, 1>);
    else go(tensor_small_kernel<DIM, P, 0, 2>);
  } else {
    by_other_kind(h->mat.m.kind, [&](auto K) {
      constexpr int FK = decltype(K)::value;
      if (mode == 0) go(tensor_small_kernel<DIM, P, FK, 0>);
      else if (mode == 1) go(tensor_small_kernel<DIM, P, FK, 1>);
      else go(tensor_small_kernel<DIM, P, FK, 2>);
    });
  }
}

static bool launch_tensor_small(mimi_hip_domain_s* h, int mode, const double* u, double* r, double* A, double gf) {
  TensorArgs a = tensor_args(h, u, r, A, gf);
  if (mode != 2) {
    ensure_pair_pos(h);
    if (!ensure_general_two_phase(h, mode == 1)) return false;     // (blocks do not fit: the caller takes the general kernels)
    a.scratch_k = h->scratch_k.ptr;
    a.scratch_r = h->scratch_r.ptr;
  }
  const int p = h->degree[0];
  if (h->dim == 2) {
    if (p == 1) launch_tensor_small_dp<2, 1>(h, mode, a);
    else if (p == 2) launch_tensor_small_dp<2, 2>(h, mode, a);
    else launch_tensor_small_dp<2, 3>(h, mode, a);
  } else {
    launch_tensor_small_dp<3, 1>(h, mode, a);
  }
  if (mode == 2) return true;
  const int64_t n_rows = h->n_vdofs;
  const double* A_old = (h->A_base && A) ? h->A_base : A;
  const unsigned gblocks = (unsigned)((n_rows + GG_WAVES - 1) / GG_WAVES);
  if (h->dim == 2) {
    auto kernel = mode == 1 ? general_gather_kernel<2, 1> : general_gather_kernel<2, 0>;
    hipLaunchKernelGGL(kernel, dim3(gblocks), dim3(64 * GG_WAVES), 0, h->stream, n_rows, h->n_dof, h->rowptr, h->adj_ptr.ptr, h->adj.ptr,
                       h->pair_pos.ptr, h->scratch_k.ptr, h->scratch_r.ptr, gf, A_old, A, r);
  } else {
    auto kernel = mode == 1 ? general_gather_kernel<3, 1> : general_gather_kernel<3, 0>;
    hipLaunchKernelGGL(kernel, dim3(gblocks), dim3(64 * GG_WAVES), 0, h->stream, n_rows, h->n_dof, h->rowptr, h->adj_ptr.ptr, h->adj.ptr,
                       h->pair_pos.ptr, h->scratch_k.ptr, h->scratch_r.ptr, gf, A_old, A, r);
  }
  MH_HIP(hipGetLastError());
  return true;
}

// A_base (tangent assemblies only): nullptr = the plain "A += gf K"; otherwise A = A_base + gf K on the rows of the handle's
// nodes.  Both arrays on the device: the row gathers read A_base where they would read A (no extra pass); any other
// residence: A_base is copied into (the staging copy of) A first.
static void run_domain(mimi_hip_domain_s* h, const double* u, double* r, double* A, double gf, bool with_grad,
                       const double* A_base = nullptr) {
  MH_HIP(hipSetDevice(h->device));
  if (!u || !r || (with_grad && !A)) fail("null vector argument");
  h->integrated = false;   // the element pieces of an earlier mimi_hip_domain_integrate are overwritten by this call
  Mirror<double> mu = Mirror<double>::in(u, h->n_vdofs, h->stage_u, h->stream);
  Mirror<double> mr = Mirror<double>::inout(r, h->n_vdofs, h->stage_r, h->stream);
  Mirror<double> mA;
  h->A_base = nullptr;
  if (with_grad && A_base && A_base != A) {
    if (is_device_pointer(A) && is_device_pointer(A_base)) {
      mA = Mirror<double>::inout(A, h->nnz, h->stage_A, h->stream);
      h->A_base = A_base;
    } else if (is_device_pointer(A)) {
      mA = Mirror<double>::inout(A, h->nnz, h->stage_A, h->stream);
      MH_HIP(hipMemcpyAsync(A, A_base, (size_t)h->nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
      // host output: its staging copy starts from the base instead of from A's own contents
      h->stage_A.resize(h->nnz);
      MH_HIP(hipMemcpyAsync(h->stage_A.ptr, A_base, (size_t)h->nnz * sizeof(double), hipMemcpyDefault, h->stream));
      mA.dev = h->stage_A.ptr;
      mA.host = A;
      mA.count = h->nnz;
      mA.stage = &h->stage_A;
    }
  } else if (with_grad) {
    mA = Mirror<double>::inout(A, h->nnz, h->stage_A, h->stream);
  }
  const int grad = !with_grad ? 0 : (h->tangent_mode == MIMI_HIP_TANGENT_REFERENCE_FD ? 2 : 1);
  // (a 3-D degree-2 / 3 patch whose CSR is not the structured pattern is not tensor_usable: the general kernels take it)
  if (tensor_small(h) && grad != 2 && launch_tensor_small(h, grad, mu.dev, mr.dev, mA.dev, gf)) {
    // (2-D, degree 1: element kernel from the 1-D tables + the general gather)
    h->last_family = 3;
  } else if (tensor_usable(h) && !tensor_small(h) && grad != 2) {
    h->last_family = launch_tensor(h, grad, mu.dev, mr.dev, mA.dev, gf);
  } else {
    h->last_family = 4;
    ensure_general_tables(h);
    GeneralArgs a = general_args(h, mu.dev, mr.dev, mA.dev, gf);
    if (h->dim == 2) launch_general<2>(h, grad, a); else launch_general<3>(h, grad, a);
  }
  h->A_base = nullptr;
  mr.finish(h->stream);
  if (with_grad) mA.finish(h->stream);
  const bool any_host = mu.host || mr.host || (with_grad && mA.host);
  if (any_host) check_status(h);  // synchronous for host-resident arguments
}

}  // namespace mimi_hip

using namespace mimi_hip;

mimi_hip_domain_s::~mimi_hip_domain_s() {
  for (auto& ev : phase_ev)
    if (ev) (void)hipEventDestroy(ev);
  if (status_dev) (void)hipFree(status_dev);
  if (status_host) (void)hipHostFree(status_host);
  if (own_stream) (void)hipStreamDestroy(own_stream);
}

extern "C" {

const char* mimi_hip_last_error(void) { return g_last_error.c_str(); }
int mimi_hip_abi_version(void) { return MIMI_HIP_ABI_VERSION; }

int mimi_hip_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return count;
}

void mimi_hip_material_set_young_poisson(mimi_hip_material* mat, double young, double poisson) {
  // MaterialBase::SetYoungPoisson (materials.cpp:7-14)
  mat->lambda = young * poisson / ((1 + poisson) * (1 - 2 * poisson));
  mat->mu = young / (2.0 * (1.0 + poisson));
  mat->G = mat->mu;
  mat->K = young / (3.0 * (1.0 - (2.0 * poisson)));
}

int mimi_hip_domain_create(const mimi_hip_domain_tables* t, const mimi_hip_material* material, int device,
                           mimi_hip_domain_t* out) {
  return guarded([&] {
    if (!t || !material || !out) fail("null argument");
    if (t->dim != 2 && t->dim != 3) fail("Unsupported Dim: %d", t->dim);
    if (t->n_dof < 1 || t->n_dof > 64) fail("n_dof %d out of range [1,64]", t->n_dof);
    if (t->n_quad < 1 || t->n_quad > 125) fail("n_quad %d out of range [1,125]", t->n_quad);
    if (t->n_elements < 1) fail("no elements");
    auto h = std::make_unique<mimi_hip_domain_s>();
    init_common(h.get(), device, material);
    h->dim = t->dim;
    h->n_el = t->n_elements;
    h->n_dof = t->n_dof;
    h->n_q = t->n_quad;
    h->n_nodes = t->n_nodes;
    h->n_vdofs = t->n_nodes * t->dim;
    h->path = 0;
    const size_t n_tdof = (size_t)t->n_dof * t->dim;
    h->dofs.assign(t->dofs, (size_t)t->n_elements * t->n_dof, h->stream);
    h->dN_dX.assign(t->dN_dX, (size_t)t->n_elements * t->n_quad * n_tdof, h->stream);
    h->wdet.assign(t->weight_det, (size_t)t->n_elements * t->n_quad, h->stream);
    setup_csr(h.get(), t->csr_rowptr, t->csr_col, true);
    init_state(h.get());
    *out = h.release();
  });
}

int mimi_hip_domain_create_bspline(const mimi_hip_bspline_patch* p, const mimi_hip_material* material, int device,
                                   mimi_hip_domain_t* out) {
  return guarded([&] {
    if (!p || !material || !out) fail("null argument");
    if (p->dim != 2 && p->dim != 3) fail("Unsupported Dim: %d", p->dim);
    auto h = std::make_unique<mimi_hip_domain_s>();
    init_common(h.get(), device, material);
    const int dim = p->dim;
    h->dim = dim;
    Tables1D t1[3];
    int pmax = 0;
    for (int d = 0; d < dim; ++d) pmax = std::max(pmax, p->degree[d]);
    if (pmax < 1 || pmax > 3) fail("degree %d unsupported (1..3)", pmax);
    // precomputed.cpp:284-290: order = 2*GetOrder()+3 when negative; order/2+1 points / direction
    const int order = p->quadrature_order < 0 ? 2 * pmax + 3 : p->quadrature_order;
    const int nq = order / 2 + 1;
    int64_t n_nodes = 1;
    h->n_dof = 1;
    h->n_q = 1;
    // NURBS weights: accepted when they are a tensor product of 1-D weights (then the rational basis factorises)
    std::vector<double> w1d[3];
    if (p->weights) {
      int64_t nc[3] = {1, 1, 1}, stride[3] = {1, 1, 1};
      for (int d = 0; d < dim; ++d) nc[d] = p->n_knots[d] - p->degree[d] - 1;
      stride[1] = nc[0];
      stride[2] = nc[0] * nc[1];
      const double c = p->weights[0];
      if (!(c > 0.0)) fail("NURBS weights must be positive");
      for (int d = 0; d < dim; ++d) {
        w1d[d].resize(nc[d]);
        for (int64_t a = 0; a < nc[d]; ++a) w1d[d][a] = p->weights[a * stride[d]];   // the line through node 0
      }
      const double cpow = dim == 3 ? c * c : c;
      for (int64_t a2 = 0; a2 < nc[2]; ++a2)
        for (int64_t a1 = 0; a1 < nc[1]; ++a1)
          for (int64_t a0 = 0; a0 < nc[0]; ++a0) {
            const double w = p->weights[a0 + a1 * stride[1] + a2 * stride[2]];
            const double prod = w1d[0][a0] * w1d[1][a1] * (dim == 3 ? w1d[2][a2] : 1.0);
            if (!(w > 0.0)) fail("NURBS weights must be positive");
            if (std::fabs(w * cpow - prod) > 1e-12 * std::fabs(prod))
              fail("NURBS weights are not a tensor product of 1-D weights (node %lld): pass this patch as flat tables "
                   "(mimi_hip_domain_create)", (long long)(a0 + a1 * stride[1] + a2 * stride[2]));
          }
    }
    for (int d = 0; d < dim; ++d) {
      if (p->degree[d] < 1 || p->degree[d] > 3) fail("degree %d unsupported (1..3)", p->degree[d]);
      t1[d] = make_tables_1d(p->knots[d], p->n_knots[d], p->degree[d], nq, p->weights ? w1d[d].data() : nullptr);
      h->degree[d] = p->degree[d];
      h->nq1[d] = nq;
      h->n_ctrl[d] = t1[d].n_ctrl;
      h->el_total[d] = t1[d].n_spans;
      n_nodes *= t1[d].n_ctrl;
      h->n_dof *= p->degree[d] + 1;
      h->n_q *= nq;
    }
    if (h->n_q > 125) fail("n_quad %d out of range", h->n_q);
    bool whole = true;
    for (int d = 0; d < 3; ++d) whole = whole && p->element_begin[d] == 0 && p->element_end[d] == 0;
    h->n_el = 1;
    for (int d = 0; d < dim; ++d) {
      h->el_begin[d] = whole ? 0 : p->element_begin[d];
      h->el_end[d] = whole ? h->el_total[d] : p->element_end[d];
      if (h->el_begin[d] < 0 || h->el_end[d] > h->el_total[d] || h->el_begin[d] >= h->el_end[d])
        fail("element box [%d,%d) invalid in direction %d (%d spans)", h->el_begin[d], h->el_end[d], d, h->el_total[d]);
      h->n_el *= h->el_end[d] - h->el_begin[d];
    }
    h->n_nodes = n_nodes;
    h->n_vdofs = n_nodes * dim;

    // 1-D tables -> one device buffer:  per direction B, D, W ; first[] in a second buffer
    std::vector<double> tab;
    std::vector<int32_t> first;
    size_t offW[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) {
      h->tab_off_B[d] = tab.size();
      tab.insert(tab.end(), t1[d].B.begin(), t1[d].B.end());
      h->tab_off_D[d] = tab.size();
      tab.insert(tab.end(), t1[d].D.begin(), t1[d].D.end());
      offW[d] = tab.size();
      tab.insert(tab.end(), t1[d].w.begin(), t1[d].w.end());
      h->first_off[d] = first.size();
      first.insert(first.end(), t1[d].first.begin(), t1[d].first.end());
    }
    h->first_is_identity = true;
    for (int d = 0; d < dim; ++d)
      for (int k = 0; k < (int)t1[d].first.size(); ++k) h->first_is_identity = h->first_is_identity && (t1[d].first[k] == k);
    h->tab1d.assign(tab.data(), tab.size(), h->stream);
    h->first1d.assign(first.data(), first.size(), h->stream);
    DeviceBuffer<double> ctrl;
    ctrl.assign(p->control_points, (size_t)n_nodes * dim, h->stream);
    if (p->node_ids) h->node_ids.assign(p->node_ids, (size_t)n_nodes, h->stream);

    for (int d = 0; d < dim; ++d) h->tab_off_W[d] = offW[d];
    PatchDev P = patch_dev(h.get(), ctrl.ptr);

    const int64_t npts = (int64_t)h->n_el * h->n_q;
    h->geo.resize((size_t)npts * (dim * dim + 1));
    {
      const int threads = 256;
      const int64_t blocks = (npts + threads - 1) / threads;
      if (dim == 2)
        hipLaunchKernelGGL(geometry_kernel<2>, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->geo.ptr, h->status_dev);
      else
        hipLaunchKernelGGL(geometry_kernel<3>, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->geo.ptr, h->status_dev);
      MH_HIP(hipGetLastError());
      check_status(h.get());
    }
    // element connectivity (always) + reference-layout tables (general path / FD mode only)
    const char* keep_env = getenv("MIMI_HIP_KEEP_GENERAL");
    const char* path_env = getenv("MIMI_HIP_FORCE_GENERAL");
    const bool force_general = path_env && path_env[0] == '1';
    const bool tensor_ok = tensor_supported(dim, h->degree, nq);
    const bool keep_general = force_general || !tensor_ok || (keep_env && keep_env[0] == '1');
    h->path = (tensor_ok && !force_general) ? 1 : 0;
    h->dofs.resize((size_t)h->n_el * h->n_dof);
    {
      const size_t n_tdof = (size_t)h->n_dof * dim;
      DeviceBuffer<double> scratch_g, scratch_w;
      double* gptr = nullptr;
      double* wptr = nullptr;
      if (keep_general) {
        h->dN_dX.resize((size_t)npts * n_tdof);
        h->wdet.resize((size_t)npts);
        gptr = h->dN_dX.ptr;
        wptr = h->wdet.ptr;
      }
      // connectivity-only launch when the tables are not kept: reuse the kernel per element chunk
      const int threads = 256;
      if (keep_general) {
        const int64_t total = npts * h->n_dof;
        const int64_t blocks = (total + threads - 1) / threads;
        if (dim == 2)
          hipLaunchKernelGGL(expand_tables_kernel<2>, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->geo.ptr, h->dofs.ptr, gptr, wptr);
        else
          hipLaunchKernelGGL(expand_tables_kernel<3>, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->geo.ptr, h->dofs.ptr, gptr, wptr);
      } else {
        const int64_t total = (int64_t)h->n_el * h->n_dof;
        const int64_t blocks = (total + threads - 1) / threads;
        hipLaunchKernelGGL(connectivity_kernel, dim3((unsigned)blocks), dim3(threads), 0, h->stream, P, h->dofs.ptr);
      }
      MH_HIP(hipGetLastError());
      MH_HIP(hipStreamSynchronize(h->stream));
    }
    setup_csr(h.get(), p->csr_rowptr, p->csr_col, false);
    // lexicographic numbering + the closed-form pattern => CSR positions are arithmetic
    h->structured_csr = false;
    if (!p->node_ids && dim == 3 && !(getenv("MIMI_HIP_NO_STRUCTURED") && getenv("MIMI_HIP_NO_STRUCTURED")[0] == '1')) {
      SparsityDev S{};
      S.dim = dim;
      DeviceBuffer<int64_t> prefix[3];
      for (int d = 0; d < 3; ++d) {
        S.n[d] = h->n_ctrl[d];
        S.p[d] = h->degree[d];
        std::vector<int64_t> pre(S.n[d] + 1, 0);
        prefix[d].assign(pre.data(), pre.size(), h->stream);  // unused by the check kernel
        S.prefix[d] = prefix[d].ptr;
        // a row-sliced pattern must hold the rows of every node this handle's elements touch
        S.req_lo[d] = t1[d].first[h->el_begin[d]];
        S.req_hi[d] = t1[d].first[h->el_end[d] - 1] + h->degree[d] + 1;
      }
      S.partial = 1;
      DeviceBuffer<int32_t> col_tmp;
      const int32_t* col_dev = p->csr_col;
      if (!is_device_pointer(p->csr_col)) {
        col_tmp.assign(p->csr_col, h->nnz, h->stream);
        col_dev = col_tmp.ptr;
      }
      MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
      const int64_t n_rows = h->n_vdofs;
      hipLaunchKernelGGL(structured_col_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, h->stream, S, n_rows,
                         h->rowptr, const_cast<int32_t*>(col_dev), 1, h->status_dev);
      MH_HIP(hipGetLastError());
      MH_HIP(hipMemcpyAsync(h->status_host, h->status_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      MH_HIP(hipStreamSynchronize(h->stream));
      h->structured_csr = (*h->status_host == 0);
      MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
    }
    // permuted numbering (node_ids given): is the caller's CSR the permuted structured pattern?
    h->structured_perm = false;
    const bool degree3 = dim == 3 && h->degree[0] == 3 && h->degree[1] == 3 && h->degree[2] == 3;
    if (p->node_ids && dim == 3 && ((h->degree[0] <= 2 && h->degree[1] <= 2 && h->degree[2] <= 2) || degree3) &&
        !(getenv("MIMI_HIP_NO_STRUCTURED") && getenv("MIMI_HIP_NO_STRUCTURED")[0] == '1')) {
      SparsityDev S{};
      S.dim = dim;
      for (int d = 0; d < 3; ++d) {
        S.n[d] = h->n_ctrl[d];
        S.p[d] = h->degree[d];
        S.prefix[d] = nullptr;
      }
      DeviceBuffer<int32_t> col_tmp;
      const int32_t* col_dev = p->csr_col;
      if (!is_device_pointer(p->csr_col)) {
        col_tmp.assign(p->csr_col, h->nnz, h->stream);
        col_dev = col_tmp.ptr;
      }
      MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
      if (degree3) {
        h->nbr_pos16.resize((size_t)n_nodes * 343);
        hipLaunchKernelGGL((permuted_window_kernel<uint16_t, 343>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, h->stream, S,
                           (int64_t)n_nodes, h->node_ids.ptr, h->rowptr, col_dev, h->nbr_pos16.ptr, h->status_dev);
      } else {
        h->nbr_pos.resize((size_t)n_nodes * 125);
        hipLaunchKernelGGL((permuted_window_kernel<unsigned char, 125>), dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, h->stream, S,
                           (int64_t)n_nodes, h->node_ids.ptr, h->rowptr, col_dev, h->nbr_pos.ptr, h->status_dev);
      }
      MH_HIP(hipGetLastError());
      MH_HIP(hipMemcpyAsync(h->status_host, h->status_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      MH_HIP(hipStreamSynchronize(h->stream));
      h->structured_perm = (*h->status_host == 0);
      MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
    }
    // degree 3 has the two-phase tensor kernels only: anything else about the handle (numbering, pattern) -> general path
    if (h->path == 1 && !tensor_usable(h.get())) h->path = 0;
    // pair positions now unless this handle will run the two-phase kernels (then on demand, ensure_pair_pos)
    if (!tensor_usable(h.get()) || tensor_small(h.get())) build_pair_pos(h.get(), p->csr_col);
    init_state(h.get());
    MH_HIP(hipStreamSynchronize(h->stream));
    *out = h.release();
  });
}

int mimi_hip_domain_destroy(mimi_hip_domain_t h) {
  return guarded([&] {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
  });
}

int mimi_hip_domain_set_dt(mimi_hip_domain_t h, double dt, double first_effective_dt, double second_effective_dt) {
  return guarded([&] {
    if (!h) fail("null handle");
    h->dt = dt;
    h->first_effective_dt = first_effective_dt;
    h->second_effective_dt = second_effective_dt;
  });
}

int mimi_hip_domain_set_tangent_mode(mimi_hip_domain_t h, int mode) {
  return guarded([&] {
    if (!h) fail("null handle");
    if (mode != MIMI_HIP_TANGENT_ANALYTIC && mode != MIMI_HIP_TANGENT_REFERENCE_FD) fail("bad tangent mode %d", mode);
    h->tangent_mode = mode;
  });
}

int mimi_hip_domain_set_stream(mimi_hip_domain_t h, void* stream) {
  return guarded([&] {
    if (!h) fail("null handle");
    h->stream = stream == MIMI_HIP_STREAM_NULL ? nullptr : (stream ? reinterpret_cast<hipStream_t>(stream) : h->own_stream);
  });
}

// the two-step form of a tangent assembly: phase 1 of the whole handle, then phase 2 over parts of its nodes
static void require_two_phase(mimi_hip_domain_s* h) {
  const bool ok = h->dim == 3 && tensor_usable(h) && !tensor_small(h) && h->tangent_mode == MIMI_HIP_TANGENT_ANALYTIC;
  if (!ok) fail("integrate / gather: only on the two-phase tensor paths (3-D, degree 2 or 3, structured CSR, analytic tangent)");
}

int mimi_hip_domain_integrate(mimi_hip_domain_t h, const double* u) {
  return guarded([&] {
    if (!h || !u) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    require_two_phase(h);
    if (!is_device_pointer(u)) fail("integrate / gather: device-resident arguments only");
    h->phase_select = 1;
    try {
      launch_tensor(h, 1, u, nullptr, nullptr, 0.0);
    } catch (...) {
      h->phase_select = 0;
      throw;
    }
    h->phase_select = 0;
    h->integrated = true;
  });
}

int mimi_hip_domain_gather(mimi_hip_domain_t h, double grad_factor, double* r, double* A_values, const int32_t node_begin[3],
                           const int32_t node_end[3]) {
  return guarded([&] {
    if (!h || !r || !A_values || !node_begin || !node_end) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    require_two_phase(h);
    if (!h->integrated) fail("gather: mimi_hip_domain_integrate has not run on this handle");
    if (!is_device_pointer(r) || !is_device_pointer(A_values)) fail("integrate / gather: device-resident arguments only");
    for (int d = 0; d < 3; ++d) {
      const int lo = h->el_begin[d], hi = h->el_end[d] + h->degree[d];     // nodes the handle's elements touch
      if (node_begin[d] < lo || node_end[d] > hi || node_begin[d] >= node_end[d])
        fail("gather: node range [%d,%d) in direction %d is not inside the handle's nodes [%d,%d)", node_begin[d], node_end[d], d, lo, hi);
      h->gather_begin[d] = node_begin[d];
      h->gather_end[d] = node_end[d];
    }
    h->phase_select = 2;
    try {
      launch_tensor(h, 1, nullptr, r, A_values, grad_factor);
    } catch (...) {
      h->phase_select = 0;
      throw;
    }
    h->phase_select = 0;
  });
}

int mimi_hip_domain_synchronize(mimi_hip_domain_t h) {
  return guarded([&] {
    if (!h) fail("null handle");
    MH_HIP(hipSetDevice(h->device));
    check_status(h);
  });
}

int mimi_hip_domain_add_residual(mimi_hip_domain_t h, const double* u, double* r) {
  return guarded([&] {
    if (!h) fail("null handle");
    run_domain(h, u, r, nullptr, 0.0, false);
  });
}

int mimi_hip_domain_add_residual_and_grad(mimi_hip_domain_t h, const double* u, double grad_factor, double* r,
                                          double* A_values) {
  return guarded([&] {
    if (!h) fail("null handle");
    run_domain(h, u, r, A_values, grad_factor, true);
  });
}

int mimi_hip_domain_add_residual_and_grad_from(mimi_hip_domain_t h, const double* u, double grad_factor, double* r,
                                               const double* A_base, double* A_out) {
  return guarded([&] {
    if (!h) fail("null handle");
    if (!A_base) fail("null vector argument");
    // "A_out = A_base + gf K" does not compose over element boxes the way "+=" does (a second box would overwrite the rows it
    // shares with the first, or -- on the routes that copy the base first -- the whole array): whole-patch handles only
    // (tables-created handles, el_total 1, are whole by construction)
    if (A_base != A_out)
      for (int d = 0; d < h->dim; ++d)
        if (h->el_begin[d] != 0 || h->el_end[d] != h->el_total[d])
          fail("mimi_hip_domain_add_residual_and_grad_from needs a whole-patch handle (this one holds elements [%d,%d) of %d in "
               "direction %d): assemble element boxes with mimi_hip_domain_add_residual_and_grad into a copy of the base",
               h->el_begin[d], h->el_end[d], h->el_total[d], d);
    try {
      run_domain(h, u, r, A_out, grad_factor, true, A_base);
    } catch (...) {
      h->A_base = nullptr;
      throw;
    }
  });
}

int mimi_hip_domain_post_time_advance(mimi_hip_domain_t h, const double* u) {
  return guarded([&] {
    if (!h) fail("null handle");
    if (!material_has_state(h->mat.m.kind)) return;  // has_states_ == false (nonlinear_solid.cpp:182-183)
    MH_HIP(hipSetDevice(h->device));
    h->integrated = false;   // (the state the stored pieces were integrated with is about to change)
    Mirror<double> mu = Mirror<double>::in(u, h->n_vdofs, h->stage_u, h->stream);
    if (tensor_small(h)) {
      launch_tensor_small(h, 2, mu.dev, nullptr, nullptr, 0.0);
    } else if (h->path == 1) {
      launch_tensor_post(h, mu.dev);
    } else {
      ensure_general_tables(h);
      GeneralArgs a = general_args(h, mu.dev, nullptr, nullptr, 0.0);
      const size_t lds = (size_t)h->n_dof * h->dim * sizeof(double);
      const bool other = !material_closed_form(h->mat.m.kind);
      void (*kernel)(GeneralArgs) =
          h->dim == 2 ? (other ? post_time_advance_general_kernel<2, 1> : post_time_advance_general_kernel<2, 0>)
                      : (other ? post_time_advance_general_kernel<3, 1> : post_time_advance_general_kernel<3, 0>);
      hipLaunchKernelGGL(kernel, dim3(h->n_el), dim3(256), lds, h->stream, a);
      MH_HIP(hipGetLastError());
    }
    if (mu.host) check_status(h);
  });
}

int mimi_hip_domain_get_state(mimi_hip_domain_t h, int what, double* out, int64_t capacity) {
  return guarded([&] {
    if (!h || !out) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    if (!material_has_state(h->mat.m.kind)) fail("material has no state");
    const int dd = h->dim * h->dim;
    const int64_t need = (what == 2 || what == 3) ? h->n_pts * dd : h->n_pts;
    if (capacity < need) fail("state buffer too small (%lld < %lld)", (long long)capacity, (long long)need);
    MH_HIP(hipStreamSynchronize(h->stream));
    if (what == 0) {
      MH_HIP(hipMemcpy(out, h->eqps.ptr, need * sizeof(double), hipMemcpyDeviceToHost));
    } else if (what == 1) {
      MH_HIP(hipMemcpy(out, h->temperature.ptr, need * sizeof(double), hipMemcpyDeviceToHost));
    } else if (what == 2 || what == 3) {
      if (what == 3 && !h->state2.ptr) fail("material has no second state matrix");
      std::vector<double> soa(need);
      MH_HIP(hipMemcpy(soa.data(), what == 2 ? h->plastic_strain.ptr : h->state2.ptr, need * sizeof(double), hipMemcpyDeviceToHost));
      for (int64_t pt = 0; pt < h->n_pts; ++pt)
        for (int c = 0; c < dd; ++c) out[pt * dd + c] = soa[(int64_t)c * h->n_pts + pt];
    } else {
      fail("unknown state id %d", what);
    }
  });
}

int mimi_hip_domain_set_phase_timing(mimi_hip_domain_t h, int on) {
  return guarded([&] {
    if (!h) fail("null handle");
    MH_HIP(hipSetDevice(h->device));
    if (on)
      for (auto& ev : h->phase_ev)
        if (!ev) MH_HIP(hipEventCreate(&ev));
    h->phase_timing = on != 0;
  });
}

int mimi_hip_domain_phase_ms(mimi_hip_domain_t h, double* phase1_ms, double* phase2_ms) {
  return guarded([&] {
    if (!h || !phase1_ms || !phase2_ms) fail("null argument");
    if (!h->phase_timing) fail("phase timing is off (mimi_hip_domain_set_phase_timing)");
    MH_HIP(hipSetDevice(h->device));
    MH_HIP(hipEventSynchronize(h->phase_ev[2]));
    float a = 0.f, b = 0.f;
    MH_HIP(hipEventElapsedTime(&a, h->phase_ev[0], h->phase_ev[1]));
    MH_HIP(hipEventElapsedTime(&b, h->phase_ev[1], h->phase_ev[2]));
    *phase1_ms = a;
    *phase2_ms = b;
  });
}

int mimi_hip_domain_phase_ms_detail(mimi_hip_domain_t h, double* prepass_ms, double* integration_ms, double* gather_ms) {
  return guarded([&] {
    if (!h || !prepass_ms || !integration_ms || !gather_ms) fail("null argument");
    if (!h->phase_timing) fail("phase timing is off (mimi_hip_domain_set_phase_timing)");
    MH_HIP(hipSetDevice(h->device));
    MH_HIP(hipEventSynchronize(h->phase_ev[2]));
    float a = 0.f, b = 0.f, c = 0.f;
    if (h->phase_has_prepass) {
      MH_HIP(hipEventElapsedTime(&a, h->phase_ev[0], h->phase_ev[3]));
      MH_HIP(hipEventElapsedTime(&b, h->phase_ev[3], h->phase_ev[1]));
    } else {
      MH_HIP(hipEventElapsedTime(&b, h->phase_ev[0], h->phase_ev[1]));
    }
    MH_HIP(hipEventElapsedTime(&c, h->phase_ev[1], h->phase_ev[2]));
    *prepass_ms = a;
    *integration_ms = b;
    *gather_ms = c;
  });
}

int mimi_hip_domain_reset_state(mimi_hip_domain_t h) {
  return guarded([&] {
    if (!h) fail("null handle");
    MH_HIP(hipSetDevice(h->device));
    init_state(h);
  });
}

int64_t mimi_hip_domain_info(mimi_hip_domain_t h, int what) {
  if (!h) return -1;
  switch (what) {
  case 0: return h->n_el;
  case 1: return h->n_q;
  case 2: return h->n_dof;
  case 3: return h->nnz;
  case 4: return h->n_vdofs;
  case 5: return h->path;
  case 6: return h->structured_csr ? 1 : (h->structured_perm ? 2 : 0);
  case 7: return h->last_family;
  default: return -1;
  }
}

}  // extern "C"

// node_begin / node_end == nullptr: every row; else only the rows of the nodes in that box (others get zero length)
static int bspline_sparsity(int32_t dim, const int32_t n_nodes_dir[3], const int32_t degree[3], const int32_t* node_begin,
                            const int32_t* node_end, int device, int64_t* rowptr, int32_t* col, int64_t* nnz_out) {
  return guarded([&] {
    if (dim != 2 && dim != 3) fail("Unsupported Dim: %d", dim);
    if ((node_begin == nullptr) != (node_end == nullptr)) fail("node_begin and node_end go together");
    if (!rowptr || !nnz_out) fail("null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) fail("libmimi_hip: no HIP device visible");
    MH_HIP(hipSetDevice(device));
    SparsityDev S{};
    S.dim = dim;
    int64_t n_nodes = 1;
    DeviceBuffer<int64_t> prefix[3];
    for (int d = 0; d < 3; ++d) {
      S.n[d] = d < dim ? n_nodes_dir[d] : 1;
      S.p[d] = d < dim ? degree[d] : 0;
      n_nodes *= S.n[d];
      std::vector<int64_t> pre(S.n[d] + 1, 0);
      for (int A = 0; A < S.n[d]; ++A) {
        const int lo = std::max(A - S.p[d], 0), hi = std::min(A + S.p[d], S.n[d] - 1);
        pre[A + 1] = pre[A] + (hi - lo + 1);
      }
      prefix[d].assign(pre.data(), pre.size(), nullptr);
      S.prefix[d] = prefix[d].ptr;
    }
    const int64_t n_rows = n_nodes * dim;
    DeviceBuffer<int64_t> rp_tmp;
    int64_t* rp_dev = rowptr;
    const bool rp_host = !is_device_pointer(rowptr);
    if (rp_host) {
      rp_tmp.resize(n_rows + 1);
      rp_dev = rp_tmp.ptr;
    }
    {
      const int threads = 256;
      const int64_t blocks = (n_nodes + 1 + threads - 1) / threads;
      hipLaunchKernelGGL(structured_rowptr_kernel, dim3((unsigned)blocks), dim3(threads), 0, nullptr, S, n_nodes, rp_dev);
      MH_HIP(hipGetLastError());
    }
    int64_t nnz = 0;
    if (node_begin) {
      // keep the lengths of the box's rows, drop the others; the scan runs on the host (n_rows + 1 integers, set-up only)
      for (int d = 0; d < dim; ++d)
        if (node_begin[d] < 0 || node_end[d] > S.n[d] || node_begin[d] >= node_end[d])
          fail("node box [%d,%d) invalid in direction %d (%d nodes)", node_begin[d], node_end[d], d, S.n[d]);
      std::vector<int64_t> full(n_rows + 1), local(n_rows + 1);
      MH_HIP(hipMemcpy(full.data(), rp_dev, (n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
      local[0] = 0;
      for (int64_t A = 0; A < n_nodes; ++A) {
        const int Am[3] = {(int)(A % S.n[0]), (int)((A / S.n[0]) % S.n[1]), (int)(A / ((int64_t)S.n[0] * S.n[1]))};
        bool in = true;
        for (int d = 0; d < dim; ++d) in = in && Am[d] >= node_begin[d] && Am[d] < node_end[d];
        for (int i = 0; i < dim; ++i) {
          const int64_t r = A * dim + i;
          local[r + 1] = local[r] + (in ? full[r + 1] - full[r] : 0);
        }
      }
      MH_HIP(hipMemcpy(rp_dev, local.data(), (n_rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    MH_HIP(hipMemcpy(&nnz, rp_dev + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost));
    *nnz_out = nnz;
    if (rp_host) MH_HIP(hipMemcpy(rowptr, rp_dev, (n_rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (col) {
      DeviceBuffer<int32_t> col_tmp;
      int32_t* col_dev = col;
      const bool col_host = !is_device_pointer(col);
      if (col_host) {
        col_tmp.resize(nnz);
        col_dev = col_tmp.ptr;
      }
      const int threads = 256;
      const int64_t blocks = (n_rows + 3) / 4;
      hipLaunchKernelGGL(structured_col_kernel, dim3((unsigned)blocks), dim3(threads), 0, nullptr, S, n_rows, rp_dev,
                         col_dev, 0, (int*)nullptr);
      MH_HIP(hipGetLastError());
      MH_HIP(hipDeviceSynchronize());
      if (col_host) MH_HIP(hipMemcpy(col, col_dev, nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    MH_HIP(hipDeviceSynchronize());
  });
}

extern "C" {

int mimi_hip_bspline_sparsity(int32_t dim, const int32_t n_nodes_dir[3], const int32_t degree[3], int device,
                              int64_t* rowptr, int32_t* col, int64_t* nnz_out) {
  return bspline_sparsity(dim, n_nodes_dir, degree, nullptr, nullptr, device, rowptr, col, nnz_out);
}

int mimi_hip_bspline_sparsity_rows(int32_t dim, const int32_t n_nodes_dir[3], const int32_t degree[3],
                                   const int32_t node_begin[3], const int32_t node_end[3], int device, int64_t* rowptr,
                                   int32_t* col, int64_t* nnz_out) {
  if (!node_begin || !node_end) return guarded([&] { fail("node_begin / node_end must be given"); });
  return bspline_sparsity(dim, n_nodes_dir, degree, node_begin, node_end, device, rowptr, col, nnz_out);
}

}  // extern "C"

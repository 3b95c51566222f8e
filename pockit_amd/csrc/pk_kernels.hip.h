// pk_kernels.hip.h -- hand-written CDNA4 (gfx950) kernels of the NLP-callback evaluator.
//
// The generated model code (pockit_amd/codegen.py) supplies, per phase, a struct P with
// straight-line fp64 functions (mid_g, mid_int, mid_jac, front_jac, ...) and a struct Gen that
// dispatches on the phase id.  Everything about *how* the work is mapped to the GPU lives here.
//
// Work decomposition: a *tile* is a run of consecutive mesh intervals of
// one pattern with at most 64 collocation nodes; one 64-lane wavefront owns one tile:
//   phase A  lane = node: coalesced 8-byte loads of the trajectory vector (states/controls are
//            stored node-contiguous per variable), model evaluation in registers, the per-node
//            derivative values that are needed K times are staged in LDS ([segment][lane]);
//   phase B  lane = output position: every I-expanded segment of the tile is a contiguous run of
//            nj * K^2 doubles in the output array; the wave streams them out in 512-byte coalesced
//            stores, reading the staged values from LDS (broadcast within an interval).
// Four independent waves share a 256-thread workgroup (one __syncthreads between the phases).
// One extra workgroup per launch handles the boundary nodes and the system-level scalars.
//
// Reference semantics restated by each kernel are cited at the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "pk_abi.h"

struct PkSys {
  const double* s;     // static parameters
  const double* I;     // integrals (valid only when the pre-pass ran)
  double sigma;
  const double* lams;  // multipliers of the system constraints = lambda[0 .. n_sys)
};

namespace pk {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- per-phase scalars: static parameters, t0/tf with boundary substitution ------------------
// (reference: phasebase.py:839-851  _value_basic)
template <class P>
__device__ __forceinline__ void phase_scalars(const PkArgs& A, const PkPhase& ph, double* s, double& dt,
                                              double& mt) {
  const double* __restrict__ sx = A.x + A.l_s;
#pragma unroll
  for (int i = 0; i < P::NS; ++i) s[i] = sx[i];
  const double* __restrict__ xp = A.x + ph.x_off;
  const double t0 = P::t0(xp, ph.L, s), tf = P::tf(xp, ph.L, s);
  dt = tf - t0;
  mt = (tf + t0) / 2;
}

template <class P>
__device__ __forceinline__ double phase_dt(const PkArgs& A, int phase) {
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, A.phase[phase], s, dt, mt);
  return dt;
}

// ---- middle-stage arguments of node q: [x_i(q) | u_j(q) | t(q) | s]  with FIXED/FUNC boundary
// values substituted (the reference overwrites x in place; we never write to x) ----------------
template <class P>
__device__ __forceinline__ void load_node(const PkArgs& A, const PkPhase& ph, const double* s, double dt,
                                          double mt, int q, double* a, double& tau, double& w) {
  const double* __restrict__ xp = A.x + ph.x_off;
#pragma unroll
  for (int i = 0; i < P::NX; ++i) a[i] = xp[i * ph.state_len + q];
  const double* __restrict__ up = xp + P::NX * ph.state_len;
#pragma unroll
  for (int i = 0; i < P::NU; ++i) a[P::NX + i] = up[i * ph.L_m + q];
  if (q == 0) P::fix_front(a, s);
  if (P::SCHEME == 1 && q == ph.L_m - 1) P::fix_back(a, s);
  tau = A.db[ph.tau_off + q];
  w = A.db[ph.w_off + q];
  a[P::NX + P::NU] = (tau - 0.5) * dt + mt;
#pragma unroll
  for (int i = 0; i < P::NS; ++i) a[P::NX + P::NU + 1 + i] = s[i];
}

struct TileGeom {
  int K, stride, R, nq, nown;
};

template <class P>
__device__ __forceinline__ TileGeom tile_geom(const PkTile& tl) {
  TileGeom g;
  g.K = tl.K;
  g.stride = tl.K - P::SCHEME;           // nodes an interval adds (LGL shares its end node)
  g.R = g.stride;                        // defect rows per interval
  g.nq = tl.nj == 0 ? 0 : tl.nj * g.stride + P::SCHEME;
  g.nown = (P::SCHEME && !tl.last) ? g.nq - 1 : g.nq;   // LGL: the shared end node belongs to the next tile
  return g;
}

// ============================================================================================
// pre-pass: integrand values -> per-tile partial sums of w * phi      (phasebase.py:997-1006)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_int(const PkArgs& A, const PkTile& tl, double* __restrict__, int lane) {
  if (P::INT_N == 0) return;
  const PkPhase& ph = A.phase[tl.phase];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  double o[P::INT_N > 0 ? P::INT_N : 1];
#pragma unroll
  for (int r = 0; r < P::INT_N; ++r) o[r] = 0.0;
  if (lane < g.nown) {
    double a[P::NARG], tau, w;
    load_node<P>(A, ph, s, dt, mt, tl.q0 + lane, a, tau, w);
    P::mid_int(a, o);
#pragma unroll
    for (int r = 0; r < P::INT_N; ++r) o[r] *= w;
  }
  const int t = blockIdx.x * PK_WAVES_PER_BLOCK + (threadIdx.x >> 6);
#pragma unroll
  for (int r = 0; r < P::INT_N; ++r) {
    const double v = wave_sum(o[r]);
    if (lane == 0 && tl.nj > 0) A.partial[(size_t)t * PK_NRED + r] = v;
  }
}

// ============================================================================================
// constraints: collocation defects  x_q - x_end - dt * (d/2) * I_hat f   and path-constraint values
// (phasebase.py:1008-1021; the K x K block product is the batched small GEMV of the path)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_g(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.phase[tl.phase];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  double xr[P::NX];
  if (lane < g.nq) {
    const int q = tl.q0 + lane;
    double a[P::NARG], tau, w, o[P::G_NOUT];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    P::mid_g(a, o);
#pragma unroll
    for (int i = 0; i < P::NX; ++i) {
      sv[i * PK_WAVE + lane] = o[i];
      xr[i] = a[i];
    }
    if (lane < g.nown) {
#pragma unroll
      for (int j = 0; j < P::NC; ++j) A.out[ph.path_off + j * ph.L_m + q] = o[P::NX + j];
    }
  }
  __syncthreads();
  const int nrows = tl.nj * g.R;
  if (lane < nrows) {
    const int jj = lane / g.R, r = lane - jj * g.R;
    const PkKind& kd = A.kind[tl.kidf];
    const double* __restrict__ full = A.db + kd.full_off + r * g.K;
    const double width = A.db[ph.width_off + tl.j0 + jj];
    const int endslot = tl.q0 + (jj + 1) * g.stride;
    const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
    const double* __restrict__ xp = A.x + ph.x_off;
    const double* __restrict__ f = sv + jj * g.stride;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) {
      double acc = 0.0;
      for (int c = 0; c < g.K; ++c) acc += (full[c] * width * 0.5) * f[i * PK_WAVE + c];
      double xe = xp[i * ph.state_len + endslot];
      if (endslot == back_slot) xe = P::back_value(i, xe, s);
      A.out[ph.g_off + i * ph.L_d + tl.r0 + lane] = (xr[i] - xe) - acc * dt;
    }
  }
}

// ============================================================================================
// dense objective gradient: per-node variable slots + per-tile partial sums for the slots shared
// by all nodes (t0, tf, static parameters)              (phasebase.py:1036-1068, systembase.py:625-657)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_grad(const PkArgs& A, const PkTile& tl, double* __restrict__, int lane) {
  const PkPhase& ph = A.phase[tl.phase];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  double orr[P::GR_NR > 0 ? P::GR_NR : 1];
#pragma unroll
  for (int r = 0; r < P::GR_NR; ++r) orr[r] = 0.0;
  if (lane < g.nown) {
    const int q = tl.q0 + lane;
    double a[P::NARG], tau, w, ov[P::NX + P::NU];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    if (q == 0)
      P::front_grad(a, tau, dt, w, sy, nullptr, ov, orr);
    else if (P::SCHEME == 1 && q == ph.L_m - 1)
      P::back_grad(a, tau, dt, w, sy, nullptr, ov, orr);
    else
      P::mid_grad(a, tau, dt, w, sy, nullptr, ov, orr);
    double* __restrict__ gp = A.out + ph.x_off;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) gp[i * ph.state_len + q] = ov[i];
#pragma unroll
    for (int i = 0; i < P::NU; ++i) gp[P::NX * ph.state_len + i * ph.L_m + q] = ov[P::NX + i];
  }
  const int t = blockIdx.x * PK_WAVES_PER_BLOCK + (threadIdx.x >> 6);
#pragma unroll
  for (int r = 0; r < P::GR_NR; ++r) {
    const double v = wave_sum(orr[r]);
    if (lane == 0 && tl.nj > 0) A.partial[(size_t)t * PK_NRED + r] = v;
  }
}

// ============================================================================================
// Jacobian / Hessian streaming phase: out[base_e + offI + p] = -(I_hat[t] d/2) * sv_e[col(p)] (* lambda[row(p)])
// (phasebase.py:1120-1124 and 1280-1285 -- the gather-multiply-concatenate that dominates the reference)
// ============================================================================================
template <class P, int NI, bool WITH_LAMBDA>  // WITH_LAMBDA: Hessian (uses P::H_state)
__device__ __forceinline__ void stream_expanded(const PkArgs& A, const PkPhase& ph, const PkTile& tl,
                                                const TileGeom& g, const double* __restrict__ sv,
                                                const int64_t* __restrict__ segb, int lane) {
  if (NI == 0) return;
  const PkKind& kd = A.kind[tl.kid];
  const int nnz = kd.nnzI;
  const int tot = tl.nj * nnz;
  if (tot == 0) return;
  const int32_t* __restrict__ rc = A.ib + kd.irc_off;
  const double* __restrict__ iv = A.db + kd.iv_off;
  const double* __restrict__ wd = A.db + ph.width_off + tl.j0;
  const uint32_t magic = 0xFFFFFFFFu / (uint32_t)nnz + 1u;   // p / nnz for p < 2^16
  double* __restrict__ out = A.out;
  for (int p = lane; p < tot; p += PK_WAVE) {
    const int jj = (int)__umulhi((uint32_t)p, magic);
    const int t = p - jj * nnz;
    const int r = rc[2 * t], c = rc[2 * t + 1];
    const double val = -(iv[t] * wd[jj] * 0.5);
    const double* __restrict__ col = sv + jj * g.stride + c;
    const size_t at = (size_t)tl.offI + p;
    if (WITH_LAMBDA) {
      const double* __restrict__ lam = A.lam + ph.g_off + tl.r0 + jj * g.R + r;
#pragma unroll
      for (int e = 0; e < NI; ++e)
        out[segb[e] + at] = val * lam[P::H_state(e) * ph.L_d] * col[e * PK_WAVE];
    } else {
#pragma unroll
      for (int e = 0; e < NI; ++e) out[segb[e] + at] = val * col[e * PK_WAVE];
    }
  }
}

template <class P>
__device__ __forceinline__ void tile_jac(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.phase[tl.phase];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  const int64_t* __restrict__ segb = A.lb + ph.jseg_off;
  if (lane < g.nq) {
    const int q = tl.q0 + lane;
    double a[P::NARG], tau, w, o[P::J_NI + P::J_NN + 1];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    P::mid_jac(a, tau, dt, w, sy, nullptr, o);
#pragma unroll
    for (int e = 0; e < P::J_NI; ++e) sv[e * PK_WAVE + lane] = o[e];
    if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::J_NN; ++e) A.out[segb[P::J_NI + e] + (q - ph.mid_lo)] = o[P::J_NI + e];
    }
  }
  __syncthreads();
  if (tl.nj == 0) return;
  {  // constant translation entries of every state (phasebase.py:1077)
    const PkKind& kd = A.kind[tl.kid];
    const int tot = tl.nj * kd.nnzT;
    const double* __restrict__ tv = A.db + kd.tv_off;
    const int64_t* __restrict__ tb = A.lb + ph.jt_off;
    for (int p = lane; p < tot; p += PK_WAVE) {
      const double v = tv[p % kd.nnzT];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) A.out[tb[i] + tl.offT + p] = v;
    }
  }
  stream_expanded<P, P::J_NI, false>(A, ph, tl, g, sv, segb, lane);
}

template <class P>
__device__ __forceinline__ void tile_hess(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.phase[tl.phase];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  const int64_t* __restrict__ segb = A.lb + ph.hseg_off;
  if (lane < g.nq) {
    const int q = tl.q0 + lane;
    double a[P::NARG], tau, w, o[P::H_NI + P::H_NN + 1], lp[P::NC > 0 ? P::NC : 1];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
    for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + q];
    P::mid_hess(a, tau, dt, w, sy, lp, o);
#pragma unroll
    for (int e = 0; e < P::H_NI; ++e) sv[e * PK_WAVE + lane] = o[e];
    if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::H_NN; ++e) A.out[segb[P::H_NI + e] + (q - ph.mid_lo)] = o[P::H_NI + e];
    }
  }
  __syncthreads();
  if (tl.nj == 0) return;
  stream_expanded<P, P::H_NI, true>(A, ph, tl, g, sv, segb, lane);
}

// ---- boundary-node evaluation for the edge workgroup -----------------------------------------
template <class P>
__device__ __forceinline__ void load_edge(const PkArgs& A, int phase, int back, double* s, double* a, double& tau,
                                          double& dt, double& w, double* lp) {
  const PkPhase& ph = A.phase[phase];
  double mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const int q = back ? ph.L_m - 1 : 0;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  if (A.lam != nullptr) {
#pragma unroll
    for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + q];
  }
}

__device__ __forceinline__ void scatter_items(const PkArgs& A, const double* __restrict__ E) {
  for (int it = threadIdx.x; it < A.n_items; it += PK_BLOCK) {
    const PkItem m = A.items[it];
    double v = m.coef * E[m.eid];
    if (m.lam >= 0) v *= A.lam[m.lam];
    A.out[m.pos] = v;
  }
}

// deterministic block sum of partial[t * PK_NRED + r], t in [lo, hi)
__device__ __forceinline__ double block_sum_partials(const PkArgs& A, int lo, int hi, int r, double* red) {
  double v = 0.0;
  for (int t = lo + (int)threadIdx.x; t < hi; t += PK_BLOCK) v += A.partial[(size_t)t * PK_NRED + r];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int k = 0; k < PK_WAVES_PER_BLOCK; ++k) tot += red[k];
  return tot;
}

// ============================================================================================
// kernels
// ============================================================================================
#define PK_TILE_PROLOGUE()                                                            \
  extern __shared__ double pk_lds[];                                                  \
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                         \
  const int ti = blockIdx.x * PK_WAVES_PER_BLOCK + wave;                              \
  PkTile tl;                                                                          \
  if (ti < A.n_tiles) {                                                               \
    tl = A.tile[ti];                                                                  \
  } else {                                                                            \
    tl = A.tile[A.n_tiles > 0 ? A.n_tiles - 1 : 0];                                   \
    tl.nj = 0;                                                                        \
  }

template <class Gen>
__device__ __forceinline__ void kernel_int(const PkArgs& A) {
  PK_TILE_PROLOGUE();
  Gen::tile_int(tl.phase, A, tl, pk_lds, lane);
}

// single workgroup: I_k = dt * sum of partials; optionally f = F_o(I, s)   (systembase.py:592-605)
template <class Gen>
__device__ __forceinline__ void kernel_intfin(const PkArgs& A) {
  __shared__ double red[PK_WAVES_PER_BLOCK];
  for (int n = 0; n < Gen::N_INT && !(A.flags & 4); ++n) {
    const int k = Gen::int_phase(n);
    const double sum = block_sum_partials(A, A.phase[k].tile_lo, A.phase[k].tile_hi, Gen::int_slot(n), red);
    if (threadIdx.x == 0) A.Ibuf[Gen::int_global(n)] = sum * Gen::phase_dt(k, A);
  }
  __syncthreads();
  if (threadIdx.x == 0 && (A.flags & 1)) {
    __threadfence_block();
    const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
    A.out[0] = Gen::sys_objective(sy);
  }
}

template <class Gen>
__device__ __forceinline__ void kernel_g(const PkArgs& A) {
  if ((int)blockIdx.x == (A.n_tiles + PK_WAVES_PER_BLOCK - 1) / PK_WAVES_PER_BLOCK) {
    if (threadIdx.x == 0 && A.n_sys > 0 && !(A.flags & 2)) {   // system constraints C(I, s)   (systembase.py:607-611)
      const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
      Gen::sys_constraints(sy, A.out);
    }
    return;
  }
  PK_TILE_PROLOGUE();
  Gen::tile_g(tl.phase, A, tl, pk_lds + wave * Gen::LDS_G, lane);
}

template <class Gen>
__device__ __forceinline__ void kernel_grad(const PkArgs& A) {
  PK_TILE_PROLOGUE();
  Gen::tile_grad(tl.phase, A, tl, pk_lds, lane);
}

// single workgroup: reduce the per-tile partials into the shared slots  (systembase.py:654-657)
template <class Gen>
__device__ __forceinline__ void kernel_gradfin(const PkArgs& A) {
  __shared__ double red[PK_WAVES_PER_BLOCK];
  __shared__ double tot[PK_NPHASE * PK_NRED];
  for (int k = 0; k < PK_NPHASE; ++k)
    for (int r = 0; r < Gen::gr_nr(k); ++r) {
      const double v = block_sum_partials(A, A.phase[k].tile_lo, A.phase[k].tile_hi, r, red);
      if (threadIdx.x == 0) tot[k * PK_NRED + r] = v;
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int z = 0; z < A.n_gz; ++z) A.out[A.ib[A.gz_off + z]] = 0.0;
    for (int k = 0; k < PK_NPHASE; ++k)
      for (int r = 0; r < Gen::gr_nr(k); ++r) A.out[A.ib[A.phase[k].red_off + r]] += tot[k * PK_NRED + r];
    if (!(A.flags & 2)) {
      const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
      double gs[PK_NS];
      Gen::sys_grad_static(sy, gs);
      for (int i = 0; i < A.n_s; ++i) A.out[A.l_s + i] += gs[i];
    }
  }
}

template <class Gen>
__device__ __forceinline__ void kernel_jac(const PkArgs& A) {
  if ((int)blockIdx.x == (A.n_tiles + PK_WAVES_PER_BLOCK - 1) / PK_WAVES_PER_BLOCK) {
    extern __shared__ double pk_lds[];
    const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (A.flags & 2) return;   // secondary shard: boundary nodes / system level belong to the primary
    for (int li = wave; li < Gen::NLISTS; li += PK_WAVES_PER_BLOCK)
      if (lane == 0) Gen::edge_jac(li, A, sy, pk_lds);
    __syncthreads();
    scatter_items(A, pk_lds);
    return;
  }
  PK_TILE_PROLOGUE();
  Gen::tile_jac(tl.phase, A, tl, pk_lds + wave * Gen::LDS_J, lane);
}

template <class Gen>
__device__ __forceinline__ void kernel_hess(const PkArgs& A) {
  if ((int)blockIdx.x == (A.n_tiles + PK_WAVES_PER_BLOCK - 1) / PK_WAVES_PER_BLOCK) {
    extern __shared__ double pk_lds[];
    const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (A.flags & 2) return;
    for (int li = wave; li < Gen::NLISTS; li += PK_WAVES_PER_BLOCK)
      if (lane == 0) Gen::edge_hess(li, A, sy, pk_lds);
    __syncthreads();
    scatter_items(A, pk_lds);
    return;
  }
  PK_TILE_PROLOGUE();
  Gen::tile_hess(tl.phase, A, tl, pk_lds + wave * Gen::LDS_H, lane);
}

}  // namespace pk

#define PK_DEFINE_KERNELS(GEN)                                                                       \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_int(PkArgs A) { pk::kernel_int<GEN>(A); }         \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_intfin(PkArgs A) { pk::kernel_intfin<GEN>(A); }   \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_g(PkArgs A) { pk::kernel_g<GEN>(A); }             \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_grad(PkArgs A) { pk::kernel_grad<GEN>(A); }       \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_gradfin(PkArgs A) { pk::kernel_gradfin<GEN>(A); } \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_jac(PkArgs A) { pk::kernel_jac<GEN>(A); }         \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_hess(PkArgs A) { pk::kernel_hess<GEN>(A); }

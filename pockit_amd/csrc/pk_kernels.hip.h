// pk_kernels.hip.h -- hand-written CDNA4 (gfx950) kernels of the NLP-callback evaluator.
//
// The generated model code (pockit_amd/codegen.py) supplies, per phase, a struct P with
// straight-line fp64 functions (mid_g, mid_int, mid_jac, mid_xall, front_jac, ...) and a struct Gen
// that dispatches on the phase id.  Everything about *how* the work is mapped to the GPU lives here.
//
// Work decomposition: a *tile* is a run of consecutive mesh intervals of
// one pattern with at most 64 collocation nodes; one 64-lane wavefront owns one tile (and one role, below):
//   phase A  lane = node: coalesced 8-byte loads of the trajectory vector (states/controls are
//            stored node-contiguous per variable), model evaluation in registers, the per-node
//            values that are needed by all K rows of the interval are staged in LDS ([segment][lane]);
//            in the same memory round trip the wave copies its tile's small pattern tables (K <= 8: at most
//            64 entries each), the base offsets of its output segments (kept in a VGPR pair, lane e = segment e)
//            and, for the Hessian, its rows of the multipliers;
//   phase B  lane = output position: every I-expanded segment of the tile is a contiguous run of
//            nj * K^2 doubles in the output array; the wave streams them out in coalesced stores of 16 bytes
//            per lane (two consecutive positions), reading staged values and tables from LDS only.  Phase B contains NO vector load (loads and
//            stores share vmcnt and return out of order: one load would make every wait a wait for all stores)
//            and needs no workgroup barrier (a wave stages for itself only).
// The tile record is wave-uniform: it is read through the scalar cache into SGPRs (readfirstlane +
// constant address space).
// Four waves of ONE role on four consecutive tiles share a 256-thread workgroup.  Roles: the whole tile (pk_g,
// pk_jac, pk_hess, unsplit pk_xall), or -- split x-part -- values (g, grad f, integrand sums, constant entries of J)
// and Jacobian (evaluated entries of J), next to the Hessian role in the single-launch cycle pk_cycle.
// Extra workgroups handle the boundary nodes / system-level scalars and the sums over all nodes.
// Sums over all nodes (integrals, gradient entries of t0/tf/static parameters) are reduced
// wave -> workgroup in the tile kernels and workgroup -> total by ONE workgroup in a fixed order
// (bit-reproducible): pk_fin in a launch of its own, or -- pk_cycle -- a workgroup of the same launch that receives
// the partial sums through self-flagging hand-off slots (handoff_put / fin_handoff).
// Output stores are agent-scope (written through L2), see put().
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

// Group tag of the generated per-group model functions (mid_jac_g(Grp<2>, ...)): a role whose derivative set is large is
// evaluated, staged and streamed in groups of segments, one after the other inside the wave (codegen.split_groups).
template <int G>
struct Grp {};

// Tables that no kernel ever writes (segment bases, tile records) are read through the constant address
// space: uniform addresses then become scalar loads (s_load, SGPR results) and -- being invariant -- the
// compiler may hoist them above stores and barriers instead of paying a vector-memory round trip right before
// the streaming stores that need them.
#define PK_CONST_AS __attribute__((address_space(4)))
typedef const int64_t PK_CONST_AS* pk_cbase_t;
__device__ __forceinline__ pk_cbase_t const_bases(const int64_t* p) { return (pk_cbase_t)(uintptr_t)p; }
__device__ __forceinline__ PkTile load_tile(const PkTile* p) {
  const int32_t PK_CONST_AS* src = (const int32_t PK_CONST_AS*)(uintptr_t)p;
  PkTile t;
  int32_t* dst = reinterpret_cast<int32_t*>(&t);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(PkTile) / sizeof(int32_t)); ++k) dst[k] = src[k];
  return t;
}

// Base offsets of the N output segments of a phase, held by the wave itself: lane e keeps base e in a VGPR pair
// (ONE coalesced vector load, issued together with the node loads of phase A) and v_readlane hands base e to the
// scalar unit where a store needs it.  Read through the constant address space instead, the scalar loads are
// re-materialised by the register allocator right at their use AFTER the evaluation -- a scalar-cache miss of
// 0.3-0.5 us in front of the translation, the streaming and the Hessian stores (ISA + wave timeline).
template <int N>
struct SegBases {
  int lo, hi;
  pk_cbase_t mem;
  __device__ __forceinline__ void load(const int64_t* __restrict__ lb, int off, int lane) {
    mem = const_bases(lb + off);
    if (N <= PK_WAVE) {
      const long long v = lane < N ? (long long)lb[off + lane] : 0ll;
      lo = (int)v;
      hi = (int)(v >> 32);
    }
  }
  // The load has to be WAITED FOR in phase A, where only loads are in flight: vmcnt counts loads and stores
  // together and they return out of order with respect to each other, so a first use in phase B would make the
  // compiler wait for every store issued before it (s_waitcnt vmcnt(0), a write round trip).
  __device__ __forceinline__ void settle() {
    if (N <= PK_WAVE) asm volatile("" : "+v"(lo), "+v"(hi));
  }
  __device__ __forceinline__ int64_t operator[](int e) const {      // e: wave-uniform (a constant after unrolling)
    if (N > PK_WAVE) return mem[e];
    const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, e), h = (unsigned)__builtin_amdgcn_readlane(hi, e);
    return (int64_t)(((unsigned long long)h << 32) | l);
  }
};

// Phase A -> phase B of a tile wave.  A wave stages values for ITSELF only (its own LDS rows, its own copy of the
// kind tables), LDS executes a wave's instructions in order and a wave runs in lockstep, so no workgroup barrier
// is needed between the phases: s_barrier made the Jacobian waves wait 0.8 us for the slowest wave of the
// workgroup (wave timeline).  Only the compiler must keep the order.
// values loaded in phase A for use in phase B are pinned (waited for) before the first store goes out, see SegBases
template <int N>
__device__ __forceinline__ void settle(double (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// Between two group passes of a role the node arguments are made opaque to the compiler (no instruction is emitted): every
// pass then evaluates its own straight-line function from registers it must assume new.  Without this the optimizer merges
// the common subexpressions of ALL passes into one block and keeps them live across the passes -- the register pressure the
// groups exist to bound (orbit_transfer's compact Hessian: 256 VGPRs + 92 AGPRs + scratch whatever the group size).
template <int N>
__device__ __forceinline__ void fresh_args(double (&a)[N], double& tau, double& dt, double& w) {
  settle(a);
  asm volatile("" : "+v"(tau), "+v"(w));
  (void)dt;      // (wave-uniform, in SGPRs: the few terms of dt alone may be shared between the passes)
}

// End of the load part of phase A: every vector load of the wave has returned (s_waitcnt vmcnt(0), explicit so that
// it holds on EVERY control-flow path -- the compiler's own waits sit inside the `lane < nodes` branch, and the
// pending-load state that leaks around that branch turns into a vmcnt(0) behind the first stores of phase B).
__device__ __forceinline__ void loads_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCD-aware workgroup -> tile-block mapping.  The hardware deals consecutive workgroup ids round-robin to the 8
// XCDs, each with its own L2; neighbouring tiles write neighbouring pieces of the same output arrays (and of the
// same 128-byte lines where a run does not end on a line boundary).  Giving every XCD one contiguous range of
// tile blocks lets its L2 merge those pieces into full lines before they are written back.  `first` workgroups
// (boundary / finalize workgroups) keep their ids; returns the tile-block index of workgroup `wg`.
// array extent for "one entry per phase": a system without phases (static parameters only,
// tests/test_base/test_system_base.py:10-20 of the reference) still needs a non-empty array type
#define PK_NPHASE_DIM (PK_NPHASE > 0 ? PK_NPHASE : 1)
#define PK_XCDS 8
__device__ __forceinline__ int xcd_ids_below(int n, int y) { return (n + PK_XCDS - 1 - y) / PK_XCDS; }   // ids < n on XCD y
__device__ __forceinline__ int xcd_tile_block(int wg, int first, int total) {
  const int x = wg % PK_XCDS;
  int start = 0;
#pragma unroll
  for (int y = 0; y < PK_XCDS; ++y)
    if (y < x) start += xcd_ids_below(total, y) - xcd_ids_below(first, y);     // tile blocks on XCD y
  return start + xcd_ids_below(wg, x) - xcd_ids_below(first, x);
}

// Developer tracing: PK_MARK(k) stores the constant-rate device clock (s_memrealtime: 100 MHz, the same on every XCD)
// of lane 0 at checkpoint k of the wave's trace record.  Records: [tile][role 0 values / whole tile, 1 Jacobian,
// 2 Hessian], then three for pk_cycle's boundary-J, boundary-H and finalize workgroups.
// Diagnostic launch switches (POCKIT_AMD_DEBUG_FLAGS, tools/cycle_flags.sh: parts of a launch switched off to time the
// rest) exist in developer builds only (models generated with POCKIT_AMD_TRACE=1): PK_DIAG is a compile-time false in
// production code objects, so no kernel carries the tests.
#ifdef PK_TRACE
#define PK_DIAG(bits) ((A.flags & (bits)) != 0)
#else
#define PK_DIAG(bits) false
#endif
#ifdef PK_TRACE
#define PK_TRACE_REC(role) const int pk_trec = tl.pad >= 0 ? tl.pad * 3 + (role) : -1
#define PK_MARK_AT(rec, k)                                                                                   \
  do {                                                                                                      \
    if (A.trace != nullptr && (threadIdx.x & 63) == 0 && (rec) >= 0)                                        \
      A.trace[(size_t)(rec) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();                                 \
  } while (0)
#define PK_MARK(k) PK_MARK_AT(pk_trec, k)
#define PK_TRACE_PARAM , int pk_trec
#define PK_TRACE_ARG , pk_trec
#else
#define PK_TRACE_PARAM
#define PK_TRACE_ARG
#define PK_TRACE_REC(role) do { } while (0)
#define PK_MARK_AT(rec, k) do { } while (0)
#define PK_MARK(k) do { } while (0)
#endif

// All output stores go through put(): agent-scope stores (`sc1`, written through the XCD's L2).  With plain stores the
// 10-140 MB a launch writes stay dirty in the L2s until the end-of-kernel release writes them back -- after the last wave,
// with nothing to overlap: write-through spreads that over the kernel's life (MI355X, pk_cycle: quadrotor 2000x6 134k -> 161k
// cycles/s, brachistochrone 1250x8 135k -> 154k, humanoid 5000x8 38.5k -> 39.6k; system scope `sc0 sc1` measures the same;
// nontemporal stores are SLOWER than plain ones inside the Infinity Cache -- they lose the L2's merging of the partial lines
// neighbouring tiles share; profiles/DESIGN_history_r01_r03.md section 5).
// cache-policy bits of the 16-byte streaming stores: "sc1", or -- code objects whose launch writes more than the Infinity
// Cache holds -- "nt" (codegen.py chooses by the size of the mesh)
#ifndef PK_STREAM_FLAGS
#define PK_STREAM_FLAGS "sc1"
#endif

__device__ __forceinline__ void put(double* __restrict__ p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum over the 64 lanes of a wave, returned in every lane.  Data-parallel-primitive (DPP) moves keep the
// six steps in the vector ALU (2 v_mov_dpp + 1 v_add_f64 each); the xor-butterfly on __shfl_xor goes through
// ds_bpermute, whose LDS-path latency made every reduction a ~700-cycle serial chain (wave timeline,
// tools/wave_trace.py).  Fixed association: pairs, quads, rows of 16, rows 0+1 / 2+3, halves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_take<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_take<0x114, 0xf>(v);   // row_shr:4   (lanes without a source add 0)
  v += dpp_take<0x118, 0xf>(v);   // row_shr:8   -> lanes 12..15 of every row hold the row sum
  v += dpp_take<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_take<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// The sums over the wave of N per-lane values -> dst[0 .. N) (written by lane 0): ALL trees first -- N independent chains of DPP
// moves and adds the scheduler interleaves -- then one block of stores.  Tree by tree with the conditional store in between,
// every tree was a serial chain of its own: a model with 12 static parameters ran 13 of them one after the other in front
// of the hand-off of its partial sums (drone_stabilization).  Same operations per sum: bit-identical.
template <int N>
__device__ __forceinline__ void wave_sums_to(double* v, double* __restrict__ dst, int lane) {
  if constexpr (N > 0) {
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = wave_sum(v[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < N; ++r) dst[r] = v[r];
    }
  }
}

// ---- fp64 matrix cores for the contractions that ARE matrix products: an interval with 64 < K <= 256 points --------
// Where an interval takes a whole workgroup, its K x K integration / interpolation blocks times the [K x n] per-node values
// are small GEMMs (phasebase.py:1008-1012 I_m.dot(F), :1339-1372 the augmented products).  v_mfma_f64_16x16x4_f64 computes
// a 16 x 16 tile of C += A(16 x 4) B(4 x 16) per instruction; on gfx950 its FLOP rate equals the fp64 VALU's, what it saves
// is operand traffic -- one A and one B register per lane feed 1024 multiply-adds, where the scalar loop issues a global
// load and n LDS reads per row and column.  Lane maps (cdna_hip_programming.md): A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], D[row = (lane >> 4) + 4 reg][col = lane & 15].
// One wave computes rows [16 rb, 16 rb + 16) of  C[r][j] = sum_k (Amat[r * lda + k] * scale) * Bcol[j * ldb + k]  for the
// columns [16 cb, 16 cb + 16); store(r, j, value) is called for every element of C the lane holds (r < nrows, j < ncols).
// MEASURED (tools/big_k_sweep.py, profiles/r03_bigk_mfma_vs_valu.txt): the matrix-core form LOSES to the VALU loops at every
// order up to 256 -- mesh error estimation 46 vs 26 us at K = 128, 204 vs 97 us at K = 256; the cycle's defect product does
// not register at all (the launch is bound by streaming K^2 entries per segment).  Three reasons: on gfx950 the f64 MFMA's
// FLOP rate equals the fp64 VALU's (it only saves operand traffic); the products are skinny, n = n_x <= 10 pads to the
// tile's 16 columns; and a workgroup-wide interval is latency-bound (one wave per SIMD), not FLOP-bound.  So the VALU form
// is the default and POCKIT_AMD_BIG_MFMA=1 (at code generation) compiles this one, kept under test.
#ifndef PK_BIG_MFMA
#define PK_BIG_MFMA 0
#endif
typedef double pk_d4 __attribute__((ext_vector_type(4)));
// The sum over k is taken in an order that suits the loads: in the step s of a group of 16 k's, the lanes of quarter q
// (= lane >> 4) supply k = k0 + 4 q + s -- element s of the FOUR CONSECUTIVE values the lane fetched in one go (a 32-byte
// piece of its row of A from global memory: the four quarters of a row read one whole 128-byte line; the same four k's of
// its column of B from LDS).  A and B use the same order, so the product is the same sum; element by element, with the
// hardware's own k = lane >> 4 order, every step waited for a strided 8-byte global load (3 x slower than the VALU loop).
template <class Store>
__device__ __forceinline__ void mfma_rows(const double* __restrict__ Amat, int lda, int nrows, int kdim, double scale,
                                          const double* __restrict__ Bcol, int ldb, int ncols, int rb, int cb, int lane,
                                          Store&& store) {
  pk_d4 acc = {0.0, 0.0, 0.0, 0.0};
  const int arow = rb * 16 + (lane & 15), bcol = cb * 16 + (lane & 15), kq = lane >> 4;
  const bool a_ok = arow < nrows, b_ok = bcol < ncols;
  const double* __restrict__ ap = Amat + (size_t)(a_ok ? arow : 0) * lda;
  const double* __restrict__ bp = Bcol + (size_t)(b_ok ? bcol : 0) * ldb;
  // 64 k's per round: the 16 loads of the lane's row of A are issued together (a wave of a workgroup-wide interval has the
  // SIMD to itself: nothing else hides the latency of a load), then the 16 products with B read from LDS as they go
  for (int k0 = 0; k0 < kdim; k0 += 64) {
    double a4[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 16 * g + 4 * kq + u;
        a4[g][u] = (a_ok && k < kdim) ? ap[k] : 0.0;
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (k0 + 16 * g >= kdim) break;                       // (wave-uniform)
      double b4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 16 * g + 4 * kq + u;
        b4[u] = (b_ok && k < kdim) ? bp[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[g][u] * scale, b4[u], acc, 0, 0, 0);
    }
  }
  if (b_ok) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = rb * 16 + kq + 4 * reg;
      if (r < nrows) store(r, bcol, acc[reg]);
    }
  }
}

// ---- in-launch hand-off of the per-workgroup partial sums (pk_cycle) --------------------------------------
// The tile workgroups of a pk_cycle launch publish their partial sums while the SAME launch's finalize workgroup
// waits for them, so the sums over all nodes cost no second launch.  Protocol: every slot of cpart / cpart2 holds
// PK_EMPTY between launches; a tile workgroup overwrites its slots with ONE agent-scope 64-bit store each (written
// through the XCD's L2, fire and forget: no fence, no counter, no wait on the publishing side); the finalize
// workgroup polls its slots with agent-scope loads until none is PK_EMPTY, takes the values and puts PK_EMPTY back
// (ordered before the next launch by the end of this one).  The data word is its own flag, so there is nothing to
// order against it -- an arrival counter would need a release that writes back the whole dirty L2 (the variant
// DESIGN.md section 5 measured at +6 us).  Forward progress: the publishers never wait for anything, the poller
// occupies one workgroup slot; the poll is bounded (PK_POLL_LIMIT), after which the slot reads as NaN, the launch ends with
// NaN in f / the gradient slots instead of hanging AND counts the event in PkArgs.status: the library turns it into an error.
#define PK_POLL_SLEEP 4       // s_sleep between two poll rounds (x 64 cycles)
#define PK_POLL_LIMIT (1 << 24)     // poll rounds of >= 0.5 us each: several seconds, far beyond any launch's duration
__device__ __forceinline__ void handoff_put(unsigned long long* slot, double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  if (b == PK_EMPTY) b = 0x7FF8000000000000ull;     // (a NaN either way)
  __hip_atomic_store(slot, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// f may land in pinned host memory that the host polls (pk_runtime.cpp, wait_result): one system-scope store
__device__ __forceinline__ void put_f(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long handoff_peek(unsigned long long* slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void handoff_clear(unsigned long long* slot) {
  __hip_atomic_store(slot, (unsigned long long)PK_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
__device__ __forceinline__ double phase_dt(const PkArgs& A) {
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, A.ph[P::INDEX], s, dt, mt);
  return dt;
}

// ---- middle-stage arguments of node q: [x_i(q) | u_j(q) | t(q) | s]  with FIXED/FUNC boundary
// values substituted (the reference overwrites x in place; we never write to x).  Every lane of a wave
// calls it (indices are clamped), so that the loads of all lanes -- nodes, the LGR end slot one past the
// tile's last node, idle lanes -- are issued together with the table loads of the tile ------------
template <class P>
__device__ __forceinline__ void load_node(const PkArgs& A, const PkPhase& ph, const double* s, double dt,
                                          double mt, int q, double* a, double& tau, double& w) {
  const double* __restrict__ xp = A.x + ph.x_off;
  const int qs = min(q, ph.state_len - 1), qm = min(q, ph.L_m - 1);
#pragma unroll
  for (int i = 0; i < P::NX; ++i) a[i] = xp[i * ph.state_len + qs];
  const double* __restrict__ up = xp + P::NX * ph.state_len;
#pragma unroll
  for (int i = 0; i < P::NU; ++i) a[P::NX + i] = up[i * ph.L_m + qm];
  if (q == 0) P::fix_front(a, s);
  if (P::SCHEME == 1 && q == ph.L_m - 1) P::fix_back(a, s);
  tau = A.db[ph.tau_off + qm];
  w = A.db[ph.w_off + qm];
  a[P::NX + P::NU] = (tau - 0.5) * dt + mt;
#pragma unroll
  for (int i = 0; i < P::NS; ++i) a[P::NX + P::NU + 1 + i] = s[i];
}

// floor(p / d) for p < 2^16 from the host-computed magic number of d (PkTile.magic*): umulhi(p, ceil(2^32 / d)).
// d == 1 has no 32-bit magic (2^32): the host stores 0 for it and the quotient is p itself -- one wave-uniform select
// (scalar) and one v_and_or per division.  (Without it the tables of LGR K = 1 / LGL K = 2 tiles holding more than
// one interval were indexed with jj = 0 for every position.)
__device__ __forceinline__ int magic_div(uint32_t p, uint32_t magic) {
  const uint32_t all = magic == 0u ? 0xFFFFFFFFu : 0u;
  return (int)(__umulhi(p, magic) | (p & all));
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

// ---- shared pieces of the tile kernels ---------------------------------------------------------

// The tile's kind tables, copied to LDS by the wave itself while its node loads are in flight: every
// table of a pattern with K <= 8 has at most 64 entries (one per lane).  After the barrier the defect,
// translation and streaming phases then read LDS only -- no dependent global round trips between the
// evaluation and the stores (measured on MI355X: each such round trip costs the latency-bound kernels
// 0.6-0.8 us).  Larger K keeps the tables in global memory.
// PK_TAB_CAP: entries of the largest staged pattern table, a COMPILE-TIME constant of the code object -- 64 (every
// interval has K <= 8: one entry per lane, the common case) or 256 (K <= 16: R K <= 256 entries, four per lane), chosen by
// the code generator from the mesh (codegen.py).  As a run-time value it cost the 12k-node cycle 13 % (every LDS address
// of the staging area then needs address arithmetic instead of an immediate offset: 211k -> 184k cycles/s).
#ifndef PK_TAB_CAP
#define PK_TAB_CAP 64
#endif
#define PK_TAB_ROUNDS (PK_TAB_CAP / PK_WAVE)
struct TabRegs {
  double iv[PK_TAB_ROUNDS], full[PK_TAB_ROUNDS], tv, wd;
  int rc[PK_TAB_ROUNDS];
};

struct TileTabs {
  bool staged;
  const double* __restrict__ iv;    // LDS copies (valid when staged): [cap], [cap], [64], [64]
  const double* __restrict__ full;
  const double* __restrict__ tv;
  const double* __restrict__ wd;
  const int* __restrict__ rc;       // r | c << 16, [cap]
};

// doubles of one wave's table block in dynamic LDS
#define PK_TAB_WIDTH (2 * PK_TAB_CAP + 2 * PK_WAVE + PK_TAB_CAP / 2)
// start of the per-wave model staging area: behind the table blocks of the workgroup's waves
#define PK_STAGE(A) (pk_lds + PK_WAVES_PER_BLOCK * PK_TAB_WIDTH)

__device__ __forceinline__ bool tabs_fit(const PkArgs& A, const PkTile& tl, const TileGeom& g) {
  return tl.nnzI <= PK_TAB_CAP && g.R * g.K <= PK_TAB_CAP && tl.nnzT <= PK_WAVE && !PK_DIAG(4096);
}

// Entry e = lane + 64 u of a table is loaded by lane `lane` (u = 0 .. cap / 64 - 1): one round for K <= 8, up to four
// for K <= 16; the guards on u are wave-uniform (scalar branches), so low-order tiles issue nothing extra.
__device__ __forceinline__ TabRegs tabs_issue(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                              bool fit, int lane) {
  TabRegs t;
#pragma unroll
  for (int u = 0; u < PK_TAB_ROUNDS; ++u) { t.iv[u] = 0.0; t.full[u] = 0.0; t.rc[u] = 0; }
  t.tv = 0.0; t.wd = 0.0;
  if (!fit) return t;
  const int nfull = g.R * g.K;
#pragma unroll
  for (int u = 0; u < PK_TAB_ROUNDS; ++u) {
    const int e = lane + PK_WAVE * u;
    if (u == 0 || tl.nnzI > PK_WAVE * u) {
      if (e < tl.nnzI) {
        t.iv[u] = A.db[tl.iv_off + e];
        t.rc[u] = A.ib[tl.irc_off + 2 * e] | (A.ib[tl.irc_off + 2 * e + 1] << 16);
      }
    }
    if (u == 0 || nfull > PK_WAVE * u) {
      if (e < nfull) t.full[u] = A.db[tl.full_off + e];
    }
  }
  if (lane < tl.nnzT) t.tv = A.db[tl.tv_off + lane];
  if (lane < tl.nj) t.wd = A.db[ph.width_off + tl.j0 + lane];
  return t;
}

__device__ __forceinline__ TileTabs tabs_commit(const PkArgs& A, const PkTile& tl, const TileGeom& g, const TabRegs& t,
                                                bool fit, int lane) {
  extern __shared__ double pk_lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int cap = PK_TAB_CAP;
  double* __restrict__ iv = pk_lds + wave * PK_TAB_WIDTH;
  double* __restrict__ full = iv + cap;
  double* __restrict__ tv = full + cap;
  double* __restrict__ wd = tv + PK_WAVE;
  int* __restrict__ rc = reinterpret_cast<int*>(wd + PK_WAVE);
  if (fit) {
    const int nfull = g.R * g.K;
#pragma unroll
    for (int u = 0; u < PK_TAB_ROUNDS; ++u) {
      const int e = lane + PK_WAVE * u;
      if (u == 0 || tl.nnzI > PK_WAVE * u) { iv[e] = t.iv[u]; rc[e] = t.rc[u]; }
      if (u == 0 || nfull > PK_WAVE * u) full[e] = t.full[u];
    }
    tv[lane] = t.tv;
    wd[lane] = t.wd;
  }
  return TileTabs{fit, iv, full, tv, wd, rc};
}

// x at the end slot of the interval whose defect row this lane writes after the barrier: lane (jj+1)*stride of
// the wave holds it (every lane loaded its clamped slot); only a full LGR tile (64 nodes) reaches past the wave
template <class P>
__device__ __forceinline__ void defect_ends(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                            const double* a, double* xe, int lane) {
  const int jj = min(magic_div((uint32_t)lane, tl.magicR), max(tl.nj - 1, 0));
  const int src = (jj + 1) * g.stride;
#pragma unroll
  for (int i = 0; i < P::NX; ++i) xe[i] = __shfl(a[i], src & (PK_WAVE - 1), PK_WAVE);
  if (src >= PK_WAVE && lane < tl.nj * g.R) {
    const double* __restrict__ xp = A.x + ph.x_off;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) xe[i] = xp[i * ph.state_len + tl.q0 + src];
  }
}

#define PK_DOT_UNROLL_MAX 96
// one defect row: acc_i += sum_c (I_hat[r, c] * d / 2) * f_i(c), K known at compile time
template <class P, int K>
__device__ __forceinline__ void defect_dot(const double* __restrict__ full, const double* __restrict__ f, double width,
                                           double* acc) {
  double a[K];
#pragma unroll
  for (int c = 0; c < K; ++c) a[c] = full[c] * width * 0.5;      // (I_hat * d) / 2 as the reference scales it
#pragma unroll
  for (int c = 0; c < K; ++c) {
#pragma unroll
    for (int i = 0; i < P::NX; ++i) acc[i] += a[c] * f[i * PK_WAVE + c];
  }
}

template <class P>
__device__ __forceinline__ void defect_loop(const double* __restrict__ full, const double* __restrict__ f, double width,
                                            double* acc, int K) {
#pragma unroll 4
  for (int c = 0; c < K; ++c) {
    const double a = full[c] * width * 0.5;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) acc[i] += a * f[i * PK_WAVE + c];
  }
}
template <class P, int K>
__device__ __forceinline__ void defect_dot_k(const double* __restrict__ full, const double* __restrict__ f, double width,
                                             double* acc) {
  if constexpr (K * P::NX <= PK_DOT_UNROLL_MAX) defect_dot<P, K>(full, f, width, acc);
  else defect_loop<P>(full, f, width, acc, K);
}

// collocation defects of the tile's rows:  (x_q - x_end) - dt * sum_c (I_hat[r,c] d/2) f_i(c)
// f staged in LDS as fsv[i * 64 + lane]                  (phasebase.py:1008-1012; batched small GEMV)
template <class P, bool STAGED>
__device__ __forceinline__ void write_defects(const PkArgs& A, const PkPhase& ph, const PkTile& tl,
                                              const TileGeom& g, const TileTabs& T, const double* s, double dt,
                                              const double* __restrict__ fsv, const double* xr, double* xe,
                                              int lane) {
  const int nrows = tl.nj * g.R;
  if (lane >= nrows) return;
  const int jj = magic_div((uint32_t)lane, tl.magicR), r = lane - jj * g.R;
  const int endslot = tl.q0 + (jj + 1) * g.stride;
  const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
  const double* __restrict__ f = fsv + jj * g.stride;
  double acc[P::NX];
#pragma unroll
  for (int i = 0; i < P::NX; ++i) acc[i] = 0.0;
  if (STAGED) {
    const double width = T.wd[jj];
    const double* __restrict__ full = T.full + r * g.K;
    // K <= 8: fully unrolled so that all LDS reads of a row are in flight together -- while a row's NX * K values fit the
    // register file comfortably (PK_DOT_UNROLL_MAX doubles; 16 states x 8 points would hold 256 VGPRs for the reads alone)
    switch (g.K) {
      case 1: defect_dot_k<P, 1>(full, f, width, acc); break;
      case 2: defect_dot_k<P, 2>(full, f, width, acc); break;
      case 3: defect_dot_k<P, 3>(full, f, width, acc); break;
      case 4: defect_dot_k<P, 4>(full, f, width, acc); break;
      case 5: defect_dot_k<P, 5>(full, f, width, acc); break;
      case 6: defect_dot_k<P, 6>(full, f, width, acc); break;
      case 7: defect_dot_k<P, 7>(full, f, width, acc); break;
      case 8: defect_dot_k<P, 8>(full, f, width, acc); break;
      default:         // 9 <= K <= 16, tables staged in LDS as well
        defect_loop<P>(full, f, width, acc, g.K);
        break;
    }
  } else {
    const double* __restrict__ full = A.db + tl.full_off + r * g.K;
    const double width = A.db[ph.width_off + tl.j0 + jj];
#pragma unroll 4
    for (int c = 0; c < g.K; ++c) {
      const double a = full[c] * width * 0.5;      // (I_hat * d) / 2 as the reference scales it
#pragma unroll
      for (int i = 0; i < P::NX; ++i) acc[i] += a * f[i * PK_WAVE + c];
    }
  }
#pragma unroll
  for (int i = 0; i < P::NX; ++i) {
    if (endslot == back_slot) xe[i] = P::back_value(i, xe[i], s);
    put(&A.o_g[ph.g_off + i * ph.L_d + tl.r0 + lane], (xr[i] - xe[i]) - acc[i] * dt);
  }
}

// ---- the same for one CHUNK of the states of a WIDE model (more than PK_WIDE_NX states; see dyn_pass): the chunk fetches its
// end-slot values (wave shuffles of the node values `a`, the tile's last LGR interval from memory) and keeps its row sums
// in CN register pairs.  Same products, same order per row.
#define PK_WIDE_NX 16
// ROW0: row of state I0 in the staged rows `fsv` (I0 when all states are staged; 0 when the rows hold this chunk only)
template <class P, bool STAGED, int I0, int CN, int ROW0 = I0>
__device__ __forceinline__ void defect_chunk(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                             const TileTabs& T, const double* s, double dt, const double* __restrict__ fsv,
                                             const double* a, int lane) {
  const int nrows = tl.nj * g.R;
  const int jj = min(magic_div((uint32_t)lane, tl.magicR), max(tl.nj - 1, 0)), r = lane - jj * g.R;
  const int src = (jj + 1) * g.stride;
  double xe[CN];
#pragma unroll
  for (int e = 0; e < CN; ++e) xe[e] = __shfl(a[I0 + e], src & (PK_WAVE - 1), PK_WAVE);      // (every lane takes part)
  if (lane >= nrows) return;
  if (src >= PK_WAVE) {
    const double* __restrict__ xp = A.x + ph.x_off;
#pragma unroll
    for (int e = 0; e < CN; ++e) xe[e] = xp[(I0 + e) * ph.state_len + tl.q0 + src];
  }
  const int endslot = tl.q0 + src;
  const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
  const double* __restrict__ f = fsv + jj * g.stride;
  const double* __restrict__ full = STAGED ? T.full + r * g.K : A.db + tl.full_off + r * g.K;
  const double width = STAGED ? T.wd[jj] : A.db[ph.width_off + tl.j0 + jj];
  double acc[CN];
#pragma unroll
  for (int e = 0; e < CN; ++e) acc[e] = 0.0;
#pragma unroll 4
  for (int c = 0; c < g.K; ++c) {
    const double wc = full[c] * width * 0.5;      // (I_hat * d) / 2 as the reference scales it
#pragma unroll
    for (int e = 0; e < CN; ++e) acc[e] += wc * f[(ROW0 + e) * PK_WAVE + c];
  }
#pragma unroll
  for (int e = 0; e < CN; ++e) {
    if (endslot == back_slot) xe[e] = P::back_value(I0 + e, xe[e], s);
    put(&A.o_g[ph.g_off + (I0 + e) * ph.L_d + tl.r0 + lane], (a[I0 + e] - xe[e]) - acc[e] * dt);
  }
}
// constant translation entries of every state (phasebase.py:1077)
// a WIDE model walks its states in chunks of PK_TRANS_CHUNK: one run pointer (an SGPR pair) per state of the chunk instead of
// one per state of the model (128 states asked for 256 SGPRs)
#define PK_TRANS_CHUNK 16
template <class P, bool STAGED, int I0>
__device__ __forceinline__ void write_translation_wide(const PkArgs& A, const PkTile& tl, const TileTabs& T,
                                                       const SegBases<P::NX>& tbase, int lane) {
  if constexpr (I0 < P::NX) {
    constexpr int CN = P::NX - I0 < PK_TRANS_CHUNK ? P::NX - I0 : PK_TRANS_CHUNK;
    const int tot = tl.nj * tl.nnzT;
    const double* __restrict__ tvg = A.db + tl.tv_off;
    double* __restrict__ run[CN];
    if constexpr (P::NX > PK_WAVE) {
      // (more states than a wave has lanes: the bases come through the scalar cache; the chunk's pointer is made opaque so
      //  that the scalar loads of ALL chunks are not hoisted to the top of the wave -- 128 states: 256 SGPRs, spilled)
      pk_cbase_t tb = tbase.mem + I0;
      asm volatile("" : "+s"(tb));
#pragma unroll
      for (int i = 0; i < CN; ++i) run[i] = A.o_jac + (tb[i] + tl.offT);
    } else {
#pragma unroll
      for (int i = 0; i < CN; ++i) run[i] = A.o_jac + (tbase[I0 + i] + tl.offT);
    }
    for (uint32_t p = lane; p < (uint32_t)tot; p += PK_WAVE) {
      const int t = (int)p - magic_div(p, tl.magicT) * tl.nnzT;
      const double v = STAGED ? T.tv[t] : tvg[t];
#pragma unroll
      for (int i = 0; i < CN; ++i) put(&run[i][p], v);
    }
    write_translation_wide<P, STAGED, I0 + PK_TRANS_CHUNK>(A, tl, T, tbase, lane);
  }
}
template <class P, bool STAGED>
__device__ __forceinline__ void write_translation(const PkArgs& A, const PkPhase& ph, const PkTile& tl,
                                                  const TileTabs& T, const SegBases<P::NX>& tbase, int lane) {
  if constexpr (P::NX > PK_WIDE_NX) return write_translation_wide<P, STAGED, 0>(A, tl, T, tbase, lane);
  const int tot = tl.nj * tl.nnzT;
  const double* __restrict__ tvg = A.db + tl.tv_off;
  // wave-uniform run starts (SGPR pairs) + ONE 32-bit lane offset shared by all states: the stores take the
  // `saddr + voffset` form and no 64-bit address is computed per store
  double* __restrict__ run[P::NX];
#pragma unroll
  for (int i = 0; i < P::NX; ++i) run[i] = A.o_jac + (tbase[i] + tl.offT);
  for (uint32_t p = lane; p < (uint32_t)tot; p += PK_WAVE) {
    const int t = (int)p - magic_div(p, tl.magicT) * tl.nnzT;
    const double v = STAGED ? T.tv[t] : tvg[t];
#pragma unroll
    for (int i = 0; i < P::NX; ++i) put(&run[i][p], v);
  }
}

// streaming phase: out[base_e + offI + p] = -(I_hat[t] d/2) * sv_e[col(p)] (* lambda[row(p)])
// (phasebase.py:1120-1124 and 1280-1285 -- the gather-multiply-concatenate that dominates the reference)
// lam_s: the tile's multiplier rows staged in LDS as lam_s[state * 64 + row]   (Hessian only)
// E0: index of the group's first segment among the role's I-expanded segments (0: the role is one group)
// S0: the state whose multiplier row is row 0 of lam_s (a wide model stages the rows of the group's states only)
template <class P, int NI, bool HESS, bool STAGED, int E0, int S0, class Bases>
__device__ __forceinline__ void stream_loop(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                            const TileTabs& T, const double* __restrict__ sv,
                                            const double* __restrict__ lam_s, const Bases& bases,
                                            double* __restrict__ out, int lane) {
  // the tile's run in every segment: a wave-uniform pointer (SGPR pair); with the 32-bit position p as the only
  // per-lane part the stores take the `saddr + voffset` form (no 64-bit address arithmetic per store)
  double* __restrict__ run[NI > 0 ? NI : 1];
#pragma unroll
  for (int e = 0; e < NI; ++e) run[e] = out + (bases[E0 + e] + tl.offI);
  const int nnz = tl.nnzI;
  const int tot = tl.nj * nnz;
  if (STAGED) {
    // A lane takes two CONSECUTIVE positions and writes them with one 16-byte store per segment (the same sc1
    // flavour as put()): half the store instructions of the 8-byte variant below -- humanoid 5000x8 +2.3 %,
    // brachistochrone 1250x8 +3 %, quadrotor 2000x6 +0..8 % against 8-byte stores.
    // All LDS reads of the pair come before its first store (stores are ordered against every other memory operation).
    typedef double pk_d2 __attribute__((ext_vector_type(2)));
    for (uint32_t p0 = 2 * lane; p0 < (uint32_t)tot; p0 += 2 * PK_WAVE) {
      double v[2][NI > 0 ? NI : 1];
      const bool pair = p0 + 1 < (uint32_t)tot;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t pc = (u == 0 || pair) ? p0 + u : p0;
        const int jj = magic_div(pc, tl.magicI);
        const int t = (int)pc - jj * nnz;
        const int rc = T.rc[t];
        const double val = -(T.iv[t] * T.wd[jj] * 0.5);
        const double* __restrict__ col = sv + jj * g.stride + (rc >> 16);
        const double* __restrict__ lam = lam_s + jj * g.R + (rc & 0xFFFF);
#pragma unroll
        for (int e = 0; e < NI; ++e)
          v[u][e] = HESS ? val * lam[(P::H_state(E0 + e) - S0) * PK_WAVE] * col[e * PK_WAVE] : val * col[e * PK_WAVE];
      }
#pragma unroll
      for (int e = 0; e < NI; ++e) {
        if (pair) {
          const pk_d2 w = {v[0][e], v[1][e]};
          // The trailing s_nop is REQUIRED: gfx9-class hardware reads the data registers of a store wider than 64 bits
          // over more than one cycle, and a VALU write of those registers needs a wait state behind the store.  The
          // compiler inserts it for stores it knows; it cannot see into an asm statement, and the next iteration's
          // first v_mul would overwrite `w` in the window.  Found at 40k nodes (humanoid, several waves per SIMD): 7 %
          // of the pk_cycle launches had the first double of four neighbouring 16-byte stores of one segment replaced
          // by the following product (tests/test_gpu_parity.py::test_cycle_is_bit_stable_at_forty_thousand_nodes).
          asm volatile("global_store_dwordx4 %0, %1, off " PK_STREAM_FLAGS "\n\ts_nop 1" : : "v"(&run[e][p0]), "v"(w));
        } else {
          put(&run[e][p0], v[0][e]);
        }
      }
    }
    return;
  }
  const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
  const double* __restrict__ ivg = A.db + tl.iv_off;
  const double* __restrict__ wdg = A.db + ph.width_off + tl.j0;
  for (uint32_t p = lane; p < (uint32_t)tot; p += PK_WAVE) {
    const int jj = magic_div(p, tl.magicI);   // p / nnz (p < 2^16)
    const int t = (int)p - jj * nnz;
    const int r = rcg[2 * t], c = rcg[2 * t + 1];
    const double val = -(ivg[t] * wdg[jj] * 0.5);
    const double* __restrict__ col = sv + jj * g.stride + c;
    if (HESS) {
      const double* __restrict__ lam = lam_s + jj * g.R + r;
#pragma unroll
      for (int e = 0; e < NI; ++e) put(&run[e][p], val * lam[(P::H_state(E0 + e) - S0) * PK_WAVE] * col[e * PK_WAVE]);
    } else {
#pragma unroll
      for (int e = 0; e < NI; ++e) put(&run[e][p], val * col[e * PK_WAVE]);
    }
  }
}

template <class P, int NI, bool HESS, bool STAGED, int E0 = 0, int S0 = 0, class Bases>
__device__ __forceinline__ void stream_expanded(const PkArgs& A, const PkPhase& ph, const PkTile& tl,
                                                const TileGeom& g, const TileTabs& T, const double* __restrict__ sv,
                                                const double* __restrict__ lam_s, const Bases& segb,
                                                double* __restrict__ out, int lane) {
  if (NI == 0 || tl.nj * tl.nnzI == 0) return;
  stream_loop<P, NI, HESS, STAGED, E0, S0>(A, ph, tl, g, T, sv, lam_s, segb, out, lane);
}

// Phase B is compiled twice -- tables staged in LDS (K <= 8) or read from global memory -- and the wave branches
// ONCE: on gfx9-class hardware loads and stores share one counter (vmcnt) and return out of order with respect to
// each other, so a single global load in a loop of phase B makes the compiler wait for ALL outstanding stores
// (s_waitcnt vmcnt(0): a full write round trip of 0.4-0.8 us per loop iteration, found in the ISA of the
// translation loop and in front of the streaming loop).  The staged variant contains no global load at all.
#define PK_PHASE_B(T, CALL)       \
  do {                            \
    if ((T).staged) {             \
      constexpr bool STAGED = true;  \
      CALL;                       \
    } else {                      \
      constexpr bool STAGED = false; \
      CALL;                       \
    }                             \
  } while (0)

// per-node gradient entries: own variable slots directly, shared slots into orr   (systembase.py:646-657)
template <class P>
__device__ __forceinline__ void node_gradient_eval(const PkPhase& ph, int q, const double* a, double tau, double dt,
                                                   double w, const PkSys& sy, double* ov, double* orr, bool have_mid) {
  if (q == 0)
    P::front_grad(a, tau, dt, w, sy, nullptr, ov, orr);
  else if (P::SCHEME == 1 && q == ph.L_m - 1)
    P::back_grad(a, tau, dt, w, sy, nullptr, ov, orr);
  else if (!have_mid)
    P::mid_grad(a, tau, dt, w, sy, nullptr, ov, orr);
}

template <class P>
__device__ __forceinline__ void node_gradient_store(const PkArgs& A, const PkPhase& ph, int q, const double* ov) {
  double* __restrict__ gp = A.o_grad + ph.x_off;
#pragma unroll
  for (int i = 0; i < P::NX; ++i) put(&gp[i * ph.state_len + q], ov[i]);
#pragma unroll
  for (int i = 0; i < P::NU; ++i) put(&gp[P::NX * ph.state_len + i * ph.L_m + q], ov[P::NX + i]);
}

template <class P>
__device__ __forceinline__ void node_gradient(const PkArgs& A, const PkPhase& ph, int q, const double* a,
                                              double tau, double dt, double w, const PkSys& sy, double* ov,
                                              double* orr, bool have_mid) {
  node_gradient_eval<P>(ph, q, a, tau, dt, w, sy, ov, orr, have_mid);
  node_gradient_store<P>(A, ph, q, ov);
}

// ============================================================================================
// integrand values -> per-wave sums of w * phi      (phasebase.py:997-1006)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_int(const PkArgs& A, const PkTile& tl, double* __restrict__,
                                         double* __restrict__ wint, double* __restrict__, int lane) {
  if (P::INT_N == 0) return;
  const PkPhase& ph = A.ph[P::INDEX];
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
  wave_sums_to<P::INT_N>(o, wint, lane);
}

// ============================================================================================
// WIDE models (P::WIDE: more than PK_WIDE_NX states): the dynamics values and the defect rows in PASSES over chunks of
// states (codegen.py D_c0 / D_cn, at most codegen.WIDE_CHUNK states each).  The reference loops over the states with no limit
// (phasebase.py:1008-1012); a wave that staged the values of ALL states needed 64 n_x doubles of LDS (80 states: the
// whole 160 KiB of a workgroup) and kept every node argument in registers.  A pass loads the node arguments ITS functions
// and rows read (load_node is called afresh: what the pass does not use is dead code), evaluates the chunk's dynamics
// (P::mid_dyn_g, its own joint CSE), stages D_cn rows, and writes the chunk's defect rows -- same products, same order per
// row as write_defects.  The passes run one after the other in a wave (pk_g, pk_xall, a cycle whose roles are not
// pass-parallel) or each as a wave of its own (pk_cycle of a model with Gen::GROUPED: tile_dyn_pick).
// ============================================================================================
template <class P, int C>
__device__ __forceinline__ void dyn_pass(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                         const TileTabs& T, const double* s, double dt, double mt,
                                         double* __restrict__ sv, int lane) {
  constexpr int I0 = P::D_c0(C), CN = P::D_cn(C);
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, tl.q0 + lane, a, tau, w);
  if (lane < g.nq) {
    double o[CN];
    P::mid_dyn_g(Grp<C>{}, a, o);
#pragma unroll
    for (int e = 0; e < CN; ++e) sv[e * PK_WAVE + lane] = o[e];
  }
  wave_lds_sync();
  PK_PHASE_B(T, (defect_chunk<P, STAGED, I0, CN, 0>(A, ph, tl, g, T, s, dt, sv, a, lane)));
}
template <class P, int C = 0>
__device__ __forceinline__ void dyn_passes(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                           const TileTabs& T, const double* s, double dt, double mt,
                                           double* __restrict__ sv, int lane) {
  if constexpr (C < P::D_NG) {
    dyn_pass<P, C>(A, ph, tl, g, T, s, dt, mt, sv, lane);
    wave_lds_sync();                        // (the next pass overwrites the rows this one read)
    dyn_passes<P, C + 1>(A, ph, tl, g, T, s, dt, mt, sv, lane);
  }
}
// one pass as a wave of its own (pk_cycle: the workgroups of a tile block are [Jacobian passes | values | dynamics passes |
// Hessian passes]); `chunk` is wave-uniform
template <class P, int C>
__device__ __forceinline__ void tile_dyn_one(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  dyn_pass<P, C>(A, ph, tl, g, T, s, dt, mt, sv, lane);
}
template <class P, int C = 0>
__device__ __forceinline__ void tile_dyn_pick(int chunk, const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  if constexpr (C < P::D_NG) {
    if (chunk == C) tile_dyn_one<P, C>(A, tl, sv, lane);
    else tile_dyn_pick<P, C + 1>(chunk, A, tl, sv, lane);
  }
}

// ============================================================================================
// constraints: collocation defects and path-constraint values      (phasebase.py:1008-1021)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_g(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                       double* __restrict__, double* __restrict__, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  if constexpr (P::WIDE) {      // the path-constraint values, then the defects in passes over chunks of states
    const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
    if constexpr (P::NC > 0) {
      double a[P::NARG], tau, w, o[P::NC];
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
      if (lane < g.nown) {
        P::mid_path(a, o);
#pragma unroll
        for (int j = 0; j < P::NC; ++j) put(&A.o_g[ph.path_off + j * ph.L_m + q], o[j]);
      }
    }
    wave_lds_sync();
    dyn_passes<P>(A, ph, tl, g, T, s, dt, mt, sv, lane);
  } else {
    double a[P::NARG], tau, w, xr[P::NX], xe[P::NX];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
    defect_ends<P>(A, ph, tl, g, a, xe, lane);
    settle(xe);
    loads_done();
#pragma unroll
    for (int i = 0; i < P::NX; ++i) xr[i] = a[i];
    if (lane < g.nq) {
      double o[P::G_NOUT];
      P::mid_g(a, o);
#pragma unroll
      for (int i = 0; i < P::NX; ++i) sv[i * PK_WAVE + lane] = o[i];
      if (lane < g.nown) {
#pragma unroll
        for (int j = 0; j < P::NC; ++j) put(&A.o_g[ph.path_off + j * ph.L_m + q], o[P::NX + j]);
      }
    }
    wave_lds_sync();
    PK_PHASE_B(T, (write_defects<P, STAGED>(A, ph, tl, g, T, s, dt, sv, xr, xe, lane)));
  }
}

// ============================================================================================
// dense objective gradient      (phasebase.py:1036-1068, systembase.py:625-657)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_grad(const PkArgs& A, const PkTile& tl, double* __restrict__,
                                          double* __restrict__, double* __restrict__ wgrad, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
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
    node_gradient<P>(A, ph, q, a, tau, dt, w, sy, ov, orr, false);
  }
  wave_sums_to<P::GR_NR>(orr, wgrad, lane);
}

// ============================================================================================
// Jacobian      (phasebase.py:1070-1152)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_jac_single(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::J_NI + P::J_NN> segb;
  SegBases<P::NX> tbase;
  segb.load(A.lb, ph.jseg_off, lane);
  tbase.load(A.lb, ph.jt_off, lane);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  segb.settle();
  tbase.settle();
  loads_done();
  if (lane < g.nq) {
    double o[P::J_NI + P::J_NN + 1];
    P::mid_jac(a, tau, dt, w, sy, nullptr, o);
#pragma unroll
    for (int e = 0; e < P::J_NI; ++e) sv[e * PK_WAVE + lane] = o[e];
    if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::J_NN; ++e) put(&A.o_jac[segb[P::J_NI + e] + (q - ph.mid_lo)], o[P::J_NI + e]);
    }
  }
  wave_lds_sync();
  if (tl.nj == 0) return;
  PK_PHASE_B(T, (write_translation<P, STAGED>(A, ph, tl, T, tbase, lane),
                 stream_expanded<P, P::J_NI, false, STAGED>(A, ph, tl, g, T, sv, nullptr, segb, A.o_jac, lane)));
}


// ---- a role evaluated in GROUPS of segments (codegen.split_groups; P::J_NG / P::H_NG > 1) ------------------------------
// One pass per group inside the wave: evaluate the group's entries at the node (its own straight-line function with its own
// joint CSE), stage its I-expanded segment values in the wave's LDS rows, stream their K^2-fold runs, next group.  Output
// positions, tile ownership and operation order per entry are those of the single-pass code: a segment is one contiguous run
// of the output array whatever pass writes it.  What a pass keeps live is bounded by the group (LDS rows, the streaming
// loop's value registers and run pointers), so the number of derivative entries of a model is unbounded, as in the reference
// (phasebase.py:1083-1124, 1234-1285 loop over any number of entries).  Between two passes the wave orders its own LDS
// accesses (wave_lds_sync); the node arguments stay in registers, so a later pass issues no vector load.
template <class P, int G, bool STAGED>
__device__ __forceinline__ void jac_group_b(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                            const TileTabs& T, const double* __restrict__ jsv,
                                            const SegBases<P::J_NI + P::J_NN>& segb, int lane) {
  stream_expanded<P, P::J_gni(G), false, STAGED, P::J_gi0(G)>(A, ph, tl, g, T, jsv, nullptr, segb, A.o_jac, lane);
}
template <class P, int G, bool ONE = false>
__device__ __forceinline__ void jac_groups(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                           const TileTabs& T, double (&a)[P::NARG], double& tau, double& dt, double& w,
                                           const PkSys& sy, double* __restrict__ jsv,
                                           const SegBases<P::J_NI + P::J_NN>& segb, int lane, int q, bool live,
                                           const double* s = nullptr, double mt = 0.0) {
  if constexpr (G < P::J_NG) {
    constexpr int NI = P::J_gni(G), N0 = P::J_gn0(G), NN = P::J_gnn(G);
    // A WIDE model (P::WIDE) fetches its node arguments anew in every pass: only what THIS pass's function reads is loaded
    // (the other loads are dead code), so the registers a pass holds do not grow with the number of states -- 128 states kept
    // alive across the passes were 260 VGPRs before any arithmetic.  (The loads queue behind the previous pass's stores.)
    if constexpr (P::WIDE && !ONE) load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    else if constexpr (G > 0 && !ONE) fresh_args(a, tau, dt, w);
    if (live) {
      double o[NI + NN + 1];
      P::mid_jac_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
#pragma unroll
      for (int e = 0; e < NI; ++e) jsv[e * PK_WAVE + lane] = o[e];
      if (NN > 0 && lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < NN; ++e) put(&A.o_jac[segb[P::J_NI + N0 + e] + (q - ph.mid_lo)], o[NI + e]);
      }
    }
    if constexpr (NI > 0) {
      wave_lds_sync();
      if (tl.nj != 0) PK_PHASE_B(T, (jac_group_b<P, G, STAGED>(A, ph, tl, g, T, jsv, segb, lane)));
      if constexpr (!ONE) wave_lds_sync();      // (the next pass overwrites the rows this one streamed from)
    }
    if constexpr (!ONE) jac_groups<P, G + 1>(A, ph, tl, g, T, a, tau, dt, w, sy, jsv, segb, lane, q, live, s, mt);
  }
}

template <class P>
__device__ __forceinline__ void tile_jac_grouped(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::J_NI + P::J_NN> segb;
  SegBases<P::NX> tbase;
  segb.load(A.lb, ph.jseg_off, lane);
  tbase.load(A.lb, ph.jt_off, lane);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  segb.settle();
  tbase.settle();
  loads_done();
  wave_lds_sync();
  if (tl.nj != 0) PK_PHASE_B(T, (write_translation<P, STAGED>(A, ph, tl, T, tbase, lane)));
  jac_groups<P, 0>(A, ph, tl, g, T, a, tau, dt, w, sy, sv, segb, lane, q, lane < g.nq, s, mt);
}

template <class P>
__device__ __forceinline__ void tile_jac(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                         double* __restrict__, double* __restrict__, int lane) {
  if constexpr (P::J_NG == 1 && !P::WIDE) tile_jac_single<P>(A, tl, sv, lane);
  else tile_jac_grouped<P>(A, tl, sv, lane);
}

// pk_cycle of a model evaluated in groups: every PASS of the Jacobian / Hessian role is a wave of its own (the workgroups of
// a tile block are [Jacobian pass 0 .. | values | Hessian pass 0 ..], kernel_cycle) -- the passes of a role are independent
// of each other (each evaluates its own group from the node), so running them side by side instead of one after the other
// multiplies the waves in flight by the number of groups; the stand-alone kernels (pk_jac, pk_hess, pk_xall) keep the loop.
// `grp` is wave-uniform (it comes from the workgroup index): the pick is a chain of scalar branches.
template <class P, int G>
__device__ __forceinline__ void tile_jac_one(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::J_NI + P::J_NN> segb;
  segb.load(A.lb, ph.jseg_off, lane);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  segb.settle();
  loads_done();
  jac_groups<P, G, true>(A, ph, tl, g, T, a, tau, dt, w, sy, sv, segb, lane, q, lane < g.nq);
}
template <class P, int G = 0>
__device__ __forceinline__ void tile_jac_pick(int grp, const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  if constexpr (G < P::J_NG) {
    if (grp == G) tile_jac_one<P, G>(A, tl, sv, lane);
    else tile_jac_pick<P, G + 1>(grp, A, tl, sv, lane);
  }
}

// ============================================================================================
// compact Jacobian (transcription.SystemPlan.jacc): the reference emits one triplet per nonzero of
// the integration matrix for every derivative entry of a dynamics function (phasebase.py:885-887,1120-1124); an entry
// whose column is the same on every node (t_0, t_f, a static parameter) thereby lands K times on one (row, column).
// Such "dense-column" entries are contracted with the integration block here, like the defects of pk_g:
//   out[row r] = -sum_c (I_hat[r, c] d/2) e(c)  (+ what the translation block adds for FUNC boundary values)
// with e(c) taken from the front / back node's own expression on those nodes; entries with a per-node column keep the
// reference's expanded form (they are unique).  LDS: [JC_NI expanded | JC_ND dense-column] x 64 lanes.
// ============================================================================================
template <class P, bool STAGED>
__device__ __forceinline__ void write_dense_rows(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                                 const TileTabs& T, const double* __restrict__ a,
                                                 const double* __restrict__ dsv, const SegBases<P::JC_NI + P::JC_ND + P::JC_NN>& segb,
                                                 int lane) {
  constexpr int ND = P::JC_ND > 0 ? P::JC_ND : 1;
  if (P::JC_ND == 0) return;
  const int nrows = tl.nj * g.R;
  if (lane >= nrows) return;
  const int jj = magic_div((uint32_t)lane, tl.magicR), r = lane - jj * g.R;
  const double* __restrict__ f = dsv + jj * g.stride;
  const double* __restrict__ full = STAGED ? T.full + r * g.K : A.db + tl.full_off + r * g.K;
  const double width = STAGED ? T.wd[jj] : A.db[ph.width_off + tl.j0 + jj];
  double acc[ND], tf[ND], tb[ND];
#pragma unroll
  for (int e = 0; e < P::JC_ND; ++e) acc[e] = 0.0;
#pragma unroll 4
  for (int c = 0; c < g.K; ++c) {
    const double w = full[c] * width * 0.5;      // (I_hat * d) / 2 as the reference scales it
#pragma unroll
    for (int e = 0; e < P::JC_ND; ++e) acc[e] += w * f[e * PK_WAVE + c];
  }
  P::jacc_tdense(a, tf, tb);                     // (functions of the static parameters: the same on every lane)
  const bool first = tl.r0 + lane == 0, last_iv = tl.j0 + jj == ph.n_int - 1;
#pragma unroll
  for (int e = 0; e < P::JC_ND; ++e)
    put(&A.o_jac[segb[P::JC_NI + e] + tl.r0 + lane], (first ? tf[e] : 0.0) + (last_iv ? tb[e] : 0.0) - acc[e]);
}

template <class P>
__device__ __forceinline__ void tile_jacc_single(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::JC_NI + P::JC_ND + P::JC_NN> segb;
  SegBases<P::NX> tbase;
  segb.load(A.lb, ph.jcseg_off, lane);
  tbase.load(A.lb, ph.jct_off, lane);
  double* __restrict__ dsv = sv + P::JC_NI * PK_WAVE;
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  segb.settle();
  tbase.settle();
  loads_done();
  if (lane < g.nq) {
    double o[P::JC_NI + P::JC_ND + P::JC_NN + 1];
    P::mid_jacc(a, tau, dt, w, sy, nullptr, o);
    if (P::JC_ND > 0) {              // the boundary nodes carry their own expressions of the dense-column entries
      if (q == 0) P::front_jacc_dense(a, tau, dt, w, sy, nullptr, o + P::JC_NI);
      else if (P::SCHEME == 1 && q == ph.L_m - 1) P::back_jacc_dense(a, tau, dt, w, sy, nullptr, o + P::JC_NI);
    }
#pragma unroll
    for (int e = 0; e < P::JC_NI + P::JC_ND; ++e) sv[e * PK_WAVE + lane] = o[e];
    if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::JC_NN; ++e)
        put(&A.o_jac[segb[P::JC_NI + P::JC_ND + e] + (q - ph.mid_lo)], o[P::JC_NI + P::JC_ND + e]);
    }
  }
  wave_lds_sync();
  if (tl.nj == 0) return;
  PK_PHASE_B(T, (write_translation<P, STAGED>(A, ph, tl, T, tbase, lane),
                 write_dense_rows<P, STAGED>(A, ph, tl, g, T, a, dsv, segb, lane),
                 stream_expanded<P, P::JC_NI, false, STAGED>(A, ph, tl, g, T, sv, nullptr, segb, A.o_jac, lane)));
}

// the compact Jacobian in groups (see jac_groups): runs of expanded segments (kind 0: staged + streamed), of dense-column
// segments (kind 1: staged, contracted with the integration block per defect row) and of per-node segments (kind 2)
template <class P, int G, bool STAGED>
__device__ __forceinline__ void jacc_group_b(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                             const TileTabs& T, const double* __restrict__ a, const double* __restrict__ sv,
                                             const SegBases<P::JC_NI + P::JC_ND + P::JC_NN>& segb, int lane) {
  constexpr int KIND = P::JC_gk(G), LO = P::JC_g0(G), CN = P::JC_gn(G);
  if constexpr (KIND == 0) {
    stream_expanded<P, CN, false, STAGED, LO>(A, ph, tl, g, T, sv, nullptr, segb, A.o_jac, lane);
  } else {
    const int nrows = tl.nj * g.R;
    if (lane >= nrows) return;
    const int jj = magic_div((uint32_t)lane, tl.magicR), r = lane - jj * g.R;
    const double* __restrict__ f = sv + jj * g.stride;
    const double* __restrict__ full = STAGED ? T.full + r * g.K : A.db + tl.full_off + r * g.K;
    const double width = STAGED ? T.wd[jj] : A.db[ph.width_off + tl.j0 + jj];
    double acc[CN], tf[CN], tb[CN];
#pragma unroll
    for (int e = 0; e < CN; ++e) acc[e] = 0.0;
#pragma unroll 4
    for (int c = 0; c < g.K; ++c) {
      const double wt = full[c] * width * 0.5;      // (I_hat * d) / 2 as the reference scales it
#pragma unroll
      for (int e = 0; e < CN; ++e) acc[e] += wt * f[e * PK_WAVE + c];
    }
    P::jacc_tdense_g(Grp<G>{}, a, tf, tb);
    const bool first = tl.r0 + lane == 0, last_iv = tl.j0 + jj == ph.n_int - 1;
#pragma unroll
    for (int e = 0; e < CN; ++e)
      put(&A.o_jac[segb[P::JC_NI + LO + e] + tl.r0 + lane], (first ? tf[e] : 0.0) + (last_iv ? tb[e] : 0.0) - acc[e]);
  }
}
// ONE: this pass only, on node arguments the caller loaded (a pass-parallel launch: tile_jacc_part)
template <class P, int G, bool ONE = false>
__device__ __forceinline__ void jacc_groups(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                            const TileTabs& T, double (&a)[P::NARG], double& tau, double& dt, double& w,
                                            const PkSys& sy, double* __restrict__ sv,
                                            const SegBases<P::JC_NI + P::JC_ND + P::JC_NN>& segb, int lane, int q, bool live,
                                            const double* s, double mt) {
  if constexpr (G < P::JC_NG) {
    constexpr int KIND = P::JC_gk(G), LO = P::JC_g0(G), CN = P::JC_gn(G);
    if constexpr (P::WIDE && !ONE) load_node<P>(A, ph, s, dt, mt, q, a, tau, w);      // (see jac_groups)
    else if constexpr (G > 0 && !ONE) fresh_args(a, tau, dt, w);
    if (live) {
      double o[CN + 1];
      P::mid_jacc_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
      if constexpr (KIND == 1) {      // the boundary nodes carry their own expressions of the dense-column entries
        if (q == 0) P::front_jacc_dense_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
        else if (P::SCHEME == 1 && q == ph.L_m - 1) P::back_jacc_dense_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
      }
      if constexpr (KIND < 2) {
#pragma unroll
        for (int e = 0; e < CN; ++e) sv[e * PK_WAVE + lane] = o[e];
      } else if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < CN; ++e) put(&A.o_jac[segb[P::JC_NI + P::JC_ND + LO + e] + (q - ph.mid_lo)], o[e]);
      }
    }
    if constexpr (KIND < 2) {
      wave_lds_sync();
      if (tl.nj != 0) PK_PHASE_B(T, (jacc_group_b<P, G, STAGED>(A, ph, tl, g, T, a, sv, segb, lane)));
      wave_lds_sync();
    }
    if constexpr (!ONE) jacc_groups<P, G + 1>(A, ph, tl, g, T, a, tau, dt, w, sy, sv, segb, lane, q, live, s, mt);
  }
}

template <class P>
__device__ __forceinline__ void tile_jacc_grouped(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::JC_NI + P::JC_ND + P::JC_NN> segb;
  SegBases<P::NX> tbase;
  segb.load(A.lb, ph.jcseg_off, lane);
  tbase.load(A.lb, ph.jct_off, lane);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  segb.settle();
  tbase.settle();
  loads_done();
  wave_lds_sync();
  if (tl.nj != 0) PK_PHASE_B(T, (write_translation<P, STAGED>(A, ph, tl, T, tbase, lane)));
  jacc_groups<P, 0>(A, ph, tl, g, T, a, tau, dt, w, sy, sv, segb, lane, q, lane < g.nq, s, mt);
}

template <class P>
__device__ __forceinline__ void tile_jacc(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                          double* __restrict__, double* __restrict__, int lane) {
  if constexpr (P::JC_gk(0) < 0) tile_jacc_single<P>(A, tl, sv, lane);      // (one pass over all kinds)
  else tile_jacc_grouped<P>(A, tl, sv, lane);
}

// The compact Jacobian of a model evaluated in groups in a PASS-PARALLEL launch (pk_cyclec, pk_jacc of a code object with
// Gen::GROUPED): workgroup `sub` of the `stride` workgroups a tile block has for this role runs the passes G with
// G % stride == sub (the compact layout may have more passes than the reference layout's Jacobian role has workgroups);
// sub 0 also writes the translation entries.  The passes are independent of each other, as in tile_jac_pick.
template <class P, int G>
__device__ __forceinline__ void tile_jacc_one(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::JC_NI + P::JC_ND + P::JC_NN> segb;
  SegBases<P::NX> tbase;
  segb.load(A.lb, ph.jcseg_off, lane);
  if constexpr (G == 0) tbase.load(A.lb, ph.jct_off, lane);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);      // (what THIS pass reads: the rest is dead code)
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  segb.settle();
  if constexpr (G == 0) tbase.settle();
  loads_done();
  wave_lds_sync();
  if constexpr (G == 0) {
    if (tl.nj != 0) PK_PHASE_B(T, (write_translation<P, STAGED>(A, ph, tl, T, tbase, lane)));
  }
  jacc_groups<P, G, true>(A, ph, tl, g, T, a, tau, dt, w, sy, sv, segb, lane, q, lane < g.nq, nullptr, 0.0);
}
template <class P, int G>
__device__ __forceinline__ void jacc_round_robin(int sub, int stride, const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                                 int lane) {
  if constexpr (G < P::JC_NG) {
    if (G % stride == sub) tile_jacc_one<P, G>(A, tl, sv, lane);
    jacc_round_robin<P, G + 1>(sub, stride, A, tl, sv, lane);
  }
}
template <class P>
__device__ __forceinline__ void tile_jacc_part(int sub, int stride, const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                               int lane) {
  if constexpr (P::JC_gk(0) < 0) {      // (one pass over all kinds: the role's first workgroup)
    if (sub == 0) tile_jacc_single<P>(A, tl, sv, lane);
  } else {
    jacc_round_robin<P, 0>(sub, stride, A, tl, sv, lane);
  }
}

// ============================================================================================
// Hessian of the Lagrangian      (phasebase.py:1211-1337, systembase.py:735-835)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_hess_single(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::H_NI + P::H_NN> segb;
  segb.load(A.lb, ph.hseg_off, lane);
  double* __restrict__ lam_s = sv + P::H_NI * PK_WAVE;       // the tile's defect multipliers, [state][row]
  PK_TRACE_REC(2);
  PK_MARK(0);
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  // (S0, SN: the states whose defect multipliers the pass reads -- all of them, or for a WIDE model the range its segments
  //  belong to: codegen.py H_gs0 / H_gsn)
  constexpr int S0 = P::H_gs0(0), SN = P::H_gsn(0);
  double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], lrow[SN > 0 ? SN : 1];
  const int row = min(tl.r0 + lane, ph.L_d - 1);
#pragma unroll
  for (int i = 0; i < SN; ++i) lrow[i] = A.lam[ph.g_off + (S0 + i) * ph.L_d + row];
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
  for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
#pragma unroll
  for (int i = 0; i < SN; ++i) lam_s[i * PK_WAVE + lane] = lrow[i];
  segb.settle();
  loads_done();
  PK_MARK(1);
  if (lane < g.nq) {
    double o[P::H_NI + P::H_NN + 1];
    PK_MARK(2);
    P::mid_hess(a, tau, dt, w, sy, lp, o);
    PK_MARK(3);
#pragma unroll
    for (int e = 0; e < P::H_NI; ++e) sv[e * PK_WAVE + lane] = o[e];
    if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::H_NN; ++e) put(&A.o_hess[segb[P::H_NI + e] + (q - ph.mid_lo)], o[P::H_NI + e]);
    }
  }
  PK_MARK(4);
  wave_lds_sync();
  PK_MARK(5);
  if (tl.nj == 0) return;
  PK_PHASE_B(T, (stream_expanded<P, P::H_NI, true, STAGED, 0, S0>(A, ph, tl, g, T, sv, lam_s, segb, A.o_hess, lane)));
  PK_MARK(8);
#ifdef PK_TRACE
  __builtin_amdgcn_s_waitcnt(0);      // all stores acknowledged
  PK_MARK(9);
#endif
}

// the Hessian role in groups (see jac_groups): lam_s = the tile's defect multipliers, staged once behind the group rows
// (a WIDE model: the rows of the states pass G's segments belong to, staged by the pass itself -- hess_rows)
template <class P, int G>
__device__ __forceinline__ void hess_rows(const PkArgs& A, const PkPhase& ph, const PkTile& tl, double* __restrict__ lam_s,
                                          int lane) {
  constexpr int S0 = P::H_gs0(G), SN = P::H_gsn(G);
  if constexpr (P::H_gni(G) > 0 && SN > 0) {
    const int row = min(tl.r0 + lane, ph.L_d - 1);
    double lrow[SN];
#pragma unroll
    for (int i = 0; i < SN; ++i) lrow[i] = A.lam[ph.g_off + (S0 + i) * ph.L_d + row];
#pragma unroll
    for (int i = 0; i < SN; ++i) lam_s[i * PK_WAVE + lane] = lrow[i];
  }
}
template <class P, int G, bool STAGED>
__device__ __forceinline__ void hess_group_b(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                             const TileTabs& T, const double* __restrict__ sv,
                                             const double* __restrict__ lam_s, const SegBases<P::H_NI + P::H_NN>& segb,
                                             int lane) {
  stream_expanded<P, P::H_gni(G), true, STAGED, P::H_gi0(G), P::H_gs0(G)>(A, ph, tl, g, T, sv, lam_s, segb, A.o_hess, lane);
}
template <class P, int G, bool ONE = false>
__device__ __forceinline__ void hess_groups(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                            const TileTabs& T, double (&a)[P::NARG], double& tau, double& dt, double& w,
                                            const PkSys& sy, double (&lp)[P::NC > 0 ? P::NC : 1], double* __restrict__ sv,
                                            double* __restrict__ lam_s, const SegBases<P::H_NI + P::H_NN>& segb,
                                            int lane, int q, bool live, const double* s = nullptr, double mt = 0.0) {
  if constexpr (G < P::H_NG) {
    constexpr int NI = P::H_gni(G), N0 = P::H_gn0(G), NN = P::H_gnn(G);
    if constexpr (P::WIDE && !ONE) {
      // a WIDE model: every pass fetches its own node arguments, path multipliers and the defect-multiplier rows of the
      // states ITS segments belong to (see jac_groups; the rows of 128 states were 64 KB of LDS per wave)
      hess_rows<P, G>(A, ph, tl, lam_s, lane);
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
      for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
    } else if constexpr (G > 0 && !ONE) {
      fresh_args(a, tau, dt, w);
      settle(lp);
    }
    if (live) {
      double o[NI + NN + 1];
      P::mid_hess_g(Grp<G>{}, a, tau, dt, w, sy, lp, o);
#pragma unroll
      for (int e = 0; e < NI; ++e) sv[e * PK_WAVE + lane] = o[e];
      if (NN > 0 && lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < NN; ++e) put(&A.o_hess[segb[P::H_NI + N0 + e] + (q - ph.mid_lo)], o[NI + e]);
      }
    }
    if constexpr (NI > 0) {
      wave_lds_sync();
      if (tl.nj != 0) PK_PHASE_B(T, (hess_group_b<P, G, STAGED>(A, ph, tl, g, T, sv, lam_s, segb, lane)));
      if constexpr (!ONE) wave_lds_sync();      // (the next pass overwrites the rows this one streamed from)
    }
    if constexpr (!ONE) hess_groups<P, G + 1>(A, ph, tl, g, T, a, tau, dt, w, sy, lp, sv, lam_s, segb, lane, q, live, s, mt);
  }
}

template <class P>
__device__ __forceinline__ void tile_hess_grouped(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::H_NI + P::H_NN> segb;
  segb.load(A.lb, ph.hseg_off, lane);
  double* __restrict__ lam_s = sv + P::H_GMAX * PK_WAVE;       // the tile's defect multipliers, [state][row]
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1];
  if constexpr (P::WIDE) {      // (every pass loads what it reads: hess_groups)
    const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
    segb.settle();
    loads_done();
    wave_lds_sync();
    hess_groups<P, 0>(A, ph, tl, g, T, a, tau, dt, w, sy, lp, sv, lam_s, segb, lane, q, lane < g.nq, s, mt);
  } else {
    double lrow[P::NX];
    const int row = min(tl.r0 + lane, ph.L_d - 1);
#pragma unroll
    for (int i = 0; i < P::NX; ++i) lrow[i] = A.lam[ph.g_off + i * ph.L_d + row];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
    for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
    const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
#pragma unroll
    for (int i = 0; i < P::NX; ++i) lam_s[i * PK_WAVE + lane] = lrow[i];
    segb.settle();
    loads_done();
    hess_groups<P, 0>(A, ph, tl, g, T, a, tau, dt, w, sy, lp, sv, lam_s, segb, lane, q, lane < g.nq);
  }
}

template <class P>
__device__ __forceinline__ void tile_hess(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                          double* __restrict__, double* __restrict__, int lane) {
  if constexpr (P::H_NG == 1) tile_hess_single<P>(A, tl, sv, lane);
  else tile_hess_grouped<P>(A, tl, sv, lane);
}

template <class P, int G>
__device__ __forceinline__ void tile_hess_one(const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::H_NI + P::H_NN> segb;
  segb.load(A.lb, ph.hseg_off, lane);
  double* __restrict__ lam_s = sv + P::H_GMAX * PK_WAVE;       // the tile's defect multipliers, [state][row]
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  constexpr int S0 = P::H_gs0(G), SN = P::H_gsn(G);           // (the states whose rows the pass reads: all, or -- WIDE -- its own)
  double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], lrow[SN > 0 ? SN : 1];
  const int row = min(tl.r0 + lane, ph.L_d - 1);
  constexpr bool needs_rows = P::H_gni(G) > 0;                 // (a pass of per-node segments only reads no defect multiplier)
  if (needs_rows) {
#pragma unroll
    for (int i = 0; i < SN; ++i) lrow[i] = A.lam[ph.g_off + (S0 + i) * ph.L_d + row];
  }
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
  for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  if (needs_rows) {
#pragma unroll
    for (int i = 0; i < SN; ++i) lam_s[i * PK_WAVE + lane] = lrow[i];
  }
  segb.settle();
  loads_done();
  hess_groups<P, G, true>(A, ph, tl, g, T, a, tau, dt, w, sy, lp, sv, lam_s, segb, lane, q, lane < g.nq);
}
template <class P, int G = 0>
__device__ __forceinline__ void tile_hess_pick(int grp, const PkArgs& A, const PkTile& tl, double* __restrict__ sv, int lane) {
  if constexpr (G < P::H_NG) {
    if (grp == G) tile_hess_one<P, G>(A, tl, sv, lane);
    else tile_hess_pick<P, G + 1>(grp, A, tl, sv, lane);
  }
}

// ============================================================================================
// fused x-callbacks: f (integrand sums), grad f, g and J of one tile from ONE evaluation of the node
// (one joint CSE over all model functions; x read once).  Used by pk_eval_cycle_dev.
// LDS: [NX dynamics values | J_NI Jacobian segments] x 64 lanes.
// ============================================================================================
// CJ: the launch serves the COMPACT Jacobian layout (pk_cycle, flags bit 9): the values wave leaves the translation entries
// to the wave that runs tile_jacc (the compact layout has runs of its own for them)
template <class P, int ROLE, bool STAGED, bool CJ>
__device__ __forceinline__ void xall_phase_b(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                             const TileTabs& T, const double* s, double dt,
                                             const double* __restrict__ sv, const double* __restrict__ jsv,
                                             const double* xr, double* xe,
                                             const SegBases<P::J_NI + P::J_NN>& segb, const SegBases<P::NX>& tbase,
                                             int lane PK_TRACE_PARAM) {
  if (ROLE != 2 && !PK_DIAG(8192)) write_defects<P, STAGED>(A, ph, tl, g, T, s, dt, sv, xr, xe, lane);
  PK_MARK(6);
  if (tl.nj == 0) return;
  // the constant translation entries of J go out with the VALUES wave: the Jacobian wave's streaming is the longest
  // chain of the launch (wave timeline), the values wave has ~1 us of slack after its defect rows
  if (ROLE != 2 && !CJ && !PK_DIAG(16384)) write_translation<P, STAGED>(A, ph, tl, T, tbase, lane);
  PK_MARK(7);
  if (ROLE == 1) return;
  if (!PK_DIAG(32768))
    stream_expanded<P, P::J_NI, false, STAGED>(A, ph, tl, g, T, jsv, nullptr, segb, A.o_jac, lane);
}

// ROLE 0: the wave produces everything of its tile.  ROLE 1 / 2 (split launch): two waves of two different
// workgroups share a tile -- 1 writes the values (integrand sums, gradient, path constraints, defects) and the
// constant translation entries of J, 2 the evaluated part of the Jacobian (N segments, I-expanded segments); the role is a compile-time constant, so each
// wave's copy of the inlined model evaluation keeps only what its outputs need and its serial chain is roughly
// halved.  A workgroup holds four waves of ONE role (four consecutive tiles).
// pub_blk >= 0 (pk_cycle, roles 0 / 1): the workgroup hands its partial sums to the launch's finalize workgroup as
// soon as they exist (handoff_put) -- before its own staging, defect and streaming work.
template <class P, int ROLE, bool CJ>
__device__ __forceinline__ void tile_xall_single(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                                 double* __restrict__ wint, double* __restrict__ wgrad, int lane,
                                                 int pub_blk) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  PK_TRACE_REC(ROLE == 2 ? 1 : 0);
  PK_MARK(0);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::J_NI + P::J_NN> segb;
  SegBases<P::NX> tbase;
  if (ROLE != 1) segb.load(A.lb, ph.jseg_off, lane);
  if (ROLE != 2) tbase.load(A.lb, ph.jt_off, lane);
  double* __restrict__ jsv = sv + P::NX * PK_WAVE;
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w, xr[P::NX], xe[P::NX];
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  if (ROLE != 1) segb.settle();
  if (ROLE != 2) tbase.settle();
  PK_MARK(1);
  if (ROLE != 2) {
    defect_ends<P>(A, ph, tl, g, a, xe, lane);
    settle(xe);
  }
  loads_done();
#pragma unroll
  for (int i = 0; i < P::NX; ++i) xr[i] = a[i];
  double oi[P::INT_N > 0 ? P::INT_N : 1], orr[P::GR_NR > 0 ? P::GR_NR : 1];
#pragma unroll
  for (int r = 0; r < P::INT_N; ++r) oi[r] = 0.0;
#pragma unroll
  for (int r = 0; r < P::GR_NR; ++r) orr[r] = 0.0;
  double og[P::G_NOUT], oj[P::J_NI + P::J_NN + 1], ov[P::NX + P::NU];
  const bool live = lane < g.nq && !PK_DIAG(512);   // (bit 9: diagnostic switch, skip the evaluation phase)
  if (live) {
    double ot[P::GR_NR > 0 ? P::GR_NR : 1], op[P::INT_N > 0 ? P::INT_N : 1];
    PK_MARK(2);
    P::mid_xall(a, tau, dt, w, sy, og, oj, ov, ot, op);
    PK_MARK(3);
    if (ROLE != 2 && lane < g.nown) {
#pragma unroll
      for (int r = 0; r < P::INT_N; ++r) oi[r] = op[r] * w;
#pragma unroll
      for (int r = 0; r < P::GR_NR; ++r) orr[r] = ot[r];
      node_gradient_eval<P>(ph, q, a, tau, dt, w, sy, ov, orr, true);   // boundary nodes re-evaluate their own entries
    }
  }
  if (ROLE != 2) {
    wave_sums_to<P::INT_N>(oi, wint, lane);
    wave_sums_to<P::GR_NR>(orr, wgrad, lane);
    PK_MARK(10);
    if (pub_blk >= 0) {
      __syncthreads();                                      // (all four waves of the workgroup have this role)
      if ((int)threadIdx.x < PK_NRED) {                     // lanes of wave 0, whose wint / wgrad rows start the arrays
        double vi = 0.0, vg = 0.0;
#pragma unroll
        for (int wv = 0; wv < PK_WAVES_PER_BLOCK; ++wv) {   // same order as publish_block_partials
          vi += wint[wv * PK_NRED + threadIdx.x];
          vg += wgrad[wv * PK_NRED + threadIdx.x];
        }
        handoff_put(A.cpart + (size_t)pub_blk * PK_NRED + threadIdx.x, vi);
        handoff_put(A.cpart2 + (size_t)pub_blk * PK_NRED + threadIdx.x, vg);
      }
      PK_MARK(11);
    }
  }
  if (live) {
    if (ROLE != 2) {
      if (lane < g.nown) node_gradient_store<P>(A, ph, q, ov);      // (after the sums went out: they feed the finalize chain)
      PK_MARK(12);
#pragma unroll
      for (int i = 0; i < P::NX; ++i) sv[i * PK_WAVE + lane] = og[i];
    }
    if (ROLE != 1) {
#pragma unroll
      for (int e = 0; e < P::J_NI; ++e) jsv[e * PK_WAVE + lane] = oj[e];
    }
    PK_MARK(13);
    if (lane < g.nown) {
      if (ROLE != 2) {
#pragma unroll
        for (int j = 0; j < P::NC; ++j) put(&A.o_g[ph.path_off + j * ph.L_m + q], og[P::NX + j]);
      }
      if (ROLE != 1 && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < P::J_NN; ++e) put(&A.o_jac[segb[P::J_NI + e] + (q - ph.mid_lo)], oj[P::J_NI + e]);
      }
    }
  }
  PK_MARK(4);
  wave_lds_sync();
  PK_MARK(5);
  if (PK_DIAG(256)) return;   // diagnostic build switches: skip the phases after the staging (all / one by one)
  PK_PHASE_B(T, (xall_phase_b<P, ROLE, STAGED, CJ>(A, ph, tl, g, T, s, dt, sv, jsv, xr, xe, segb, tbase, lane PK_TRACE_ARG)));
  PK_MARK(8);
#ifdef PK_TRACE
  __builtin_amdgcn_s_waitcnt(0);      // all stores acknowledged
  PK_MARK(9);
#endif
}

// The fused x-part of a model whose Jacobian is evaluated in groups: the VALUES part (g, grad f, integrand sums; P::mid_xval)
// exactly as in the single-pass code, then the Jacobian segments group by group (jac_groups, the passes of pk_jac).
// A WIDE model (P::WIDE): mid_xval leaves the dynamics out; they and the defect rows follow in passes over chunks of states
// (dyn_passes) -- unless PP says that those passes are waves of their own in this launch (pk_cycle, tile_dyn_pick).
template <class P, int ROLE, bool STAGED, bool CJ>
__device__ __forceinline__ void xval_phase_b(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                             const TileTabs& T, const double* s, double dt, const double* __restrict__ sv,
                                             const double* xr, double* xe, const SegBases<P::NX>& tbase, int lane) {
  if constexpr (!P::WIDE) write_defects<P, STAGED>(A, ph, tl, g, T, s, dt, sv, xr, xe, lane);
  if (!CJ && tl.nj != 0) write_translation<P, STAGED>(A, ph, tl, T, tbase, lane);
}
template <class P, int ROLE, bool CJ, bool PP = false>
__device__ __forceinline__ void tile_xall_grouped(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                                  double* __restrict__ wint, double* __restrict__ wgrad, int lane,
                                                  int pub_blk) {
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  PK_TRACE_REC(ROLE == 2 ? 1 : 0);
  PK_MARK(0);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  SegBases<P::J_NI + P::J_NN> segb;
  SegBases<P::NX> tbase;
  if (ROLE != 1) segb.load(A.lb, ph.jseg_off, lane);
  if (ROLE != 2) tbase.load(A.lb, ph.jt_off, lane);
  double* __restrict__ jsv = sv + P::XROWS * PK_WAVE;      // (XROWS: rows of the dynamics values -- NX, or a WIDE model's chunk)
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
  if (ROLE != 1) segb.settle();
  if (ROLE != 2) tbase.settle();
  loads_done();
  PK_MARK(1);
  const bool live = lane < g.nq;
  if (ROLE != 2) {
    double oi[P::INT_N > 0 ? P::INT_N : 1], orr[P::GR_NR > 0 ? P::GR_NR : 1];
#pragma unroll
    for (int r = 0; r < P::INT_N; ++r) oi[r] = 0.0;
#pragma unroll
    for (int r = 0; r < P::GR_NR; ++r) orr[r] = 0.0;
    double og[P::G_NOUT], ov[P::NX + P::NU];
    constexpr bool SUMS_FIRST = P::NX + P::NU + P::NC <= 48;      // (values the stores keep in registers across the sums)
    if (live) {
      double ot[P::GR_NR > 0 ? P::GR_NR : 1], op[P::INT_N > 0 ? P::INT_N : 1];
      PK_MARK(2);
      P::mid_xval(a, tau, dt, w, sy, sv + lane, PK_WAVE, og, ov, ot, op);      // (dynamics values: straight into the LDS rows)
      PK_MARK(3);
      if (lane < g.nown) {
#pragma unroll
        for (int r = 0; r < P::INT_N; ++r) oi[r] = op[r] * w;
#pragma unroll
        for (int r = 0; r < P::GR_NR; ++r) orr[r] = ot[r];
        node_gradient_eval<P>(ph, q, a, tau, dt, w, sy, ov, orr, true);   // boundary nodes re-evaluate their own entries
        // The per-node stores (gradient entries, path values) go out at once where the model is so wide that they would crowd
        // the registers across the sums (SUMS_FIRST false); otherwise BEHIND the hand-off of the partial sums: the finalize
        // workgroup's chain -- poll across the XCDs while the launch is draining (2.4 us from the last publish to "seen"),
        // then its own sums -- ends the launch of a model with many sums (drone_stabilization: values waves handed off at
        // 4.3-5.3 us, the finalize workgroup ended at 10.2 us, everything else at 7.5; profiles/r05_n_drone_wave_timeline_wide.txt)
        if constexpr (!SUMS_FIRST) {
          node_gradient_store<P>(A, ph, q, ov);
#pragma unroll
          for (int j = 0; j < P::NC; ++j) put(&A.o_g[ph.path_off + j * ph.L_m + q], og[P::NX + j]);
        }
      }
    }
    PK_MARK(12);
    wave_sums_to<P::INT_N>(oi, wint, lane);
    wave_sums_to<P::GR_NR>(orr, wgrad, lane);
    PK_MARK(10);
    if (pub_blk >= 0) {
      __syncthreads();                                      // (all four waves of the workgroup have this role)
      if ((int)threadIdx.x < PK_NRED) {                     // same order as publish_block_partials
        double vi = 0.0, vg = 0.0;
#pragma unroll
        for (int wv = 0; wv < PK_WAVES_PER_BLOCK; ++wv) {
          vi += wint[wv * PK_NRED + threadIdx.x];
          vg += wgrad[wv * PK_NRED + threadIdx.x];
        }
        handoff_put(A.cpart + (size_t)pub_blk * PK_NRED + threadIdx.x, vi);
        handoff_put(A.cpart2 + (size_t)pub_blk * PK_NRED + threadIdx.x, vg);
      }
    }
    if constexpr (SUMS_FIRST) {
      if (live && lane < g.nown) {
        node_gradient_store<P>(A, ph, q, ov);
#pragma unroll
        for (int j = 0; j < P::NC; ++j) put(&A.o_g[ph.path_off + j * ph.L_m + q], og[P::NX + j]);
      }
    }
    // the end-slot state values of the defect rows only now (wave shuffles of the node values; a full LGR tile reaches one
    // slot past the wave): the node's own values ARE its row's x_q (a[0 .. NX) until the Jacobian passes make them opaque).
    PK_MARK(11);
    double xe[P::WIDE ? 1 : P::NX];
    if constexpr (!P::WIDE) defect_ends<P>(A, ph, tl, g, a, xe, lane);
    wave_lds_sync();
    PK_MARK(5);
    PK_PHASE_B(T, (xval_phase_b<P, ROLE, STAGED, CJ>(A, ph, tl, g, T, s, dt, sv, a, xe, tbase, lane)));
    PK_MARK(7);
    if constexpr (P::WIDE && !PP) dyn_passes<P>(A, ph, tl, g, T, s, dt, mt, sv, lane);
#ifdef PK_TRACE
    __builtin_amdgcn_s_waitcnt(0);      // all stores acknowledged
    PK_MARK(9);
#endif
  } else {
    wave_lds_sync();                                        // (the table blocks the wave staged for itself)
  }
  if (ROLE != 1) jac_groups<P, 0>(A, ph, tl, g, T, a, tau, dt, w, sy, jsv, segb, lane, q, live, s, mt);
}

template <class P, int ROLE, bool CJ = false, bool PP = false>
__device__ __forceinline__ void tile_xall(const PkArgs& A, const PkTile& tl, double* __restrict__ sv,
                                          double* __restrict__ wint, double* __restrict__ wgrad, int lane,
                                          int pub_blk) {
  if constexpr (P::J_NG == 1 && !P::WIDE) tile_xall_single<P, ROLE, CJ>(A, tl, sv, wint, wgrad, lane, pub_blk);
  else tile_xall_grouped<P, ROLE, CJ, PP>(A, tl, sv, wint, wgrad, lane, pub_blk);
}

#ifdef PK_BIG
// ============================================================================================
// Intervals with more points than a wavefront has lanes (K > 64): ONE WORKGROUP per interval.
// The reference has no limit on num_point (radau/discretization.py:488-521); the wave tiles above do (lane = node).
// Such an interval takes the first slot of a tile block (the host leaves the other three empty); its workgroup -- one
// per role, as for ordinary tiles -- walks the nodes with all 256 threads (phase A: evaluation, per-node outputs,
// staging of the per-node values in LDS rows of PK_BIG_MAX doubles), synchronizes, and walks the defect rows, the
// translation entries and the K^2 entries of every I-expanded segment (phase B; tables from global memory, 8-byte
// stores).  Same arithmetic, same operation order as the tile path.  Compiled only into code objects whose mesh has such
// an interval (codegen.py), so ordinary meshes carry none of it.
// An interval with more than PK_BIG_MAX points does not fit the workgroup's LDS rows: its staged rows live in a slot of a
// global staging buffer instead (PkArgs.big_stage; written and read by the waves of ONE workgroup, on one CU, either side
// of a workgroup barrier), rows of PkArgs.big_row doubles.  Everything else is the same code.
// ============================================================================================
#define PK_BIG_MAX 256
// PK_BIG_GLOBAL (code generator): the rows of EVERY workgroup-wide interval live in the staging buffer -- a model whose rows of
// PK_BIG_MAX doubles do not fit a workgroup's LDS (a workgroup-wide interval keeps rows of every state: 2 n_x + group rows)
#ifndef PK_BIG_GLOBAL
#define PK_BIG_GLOBAL 0
#endif
// where an interval of K points stages its rows: (row length, base) -- sub-slot u as in pk_abi.h
struct BigStage { int KS; double* base; };
__device__ __forceinline__ BigStage big_stage(const PkArgs& A, int K, int slot, int u, double* __restrict__ lds) {
  if (K <= PK_BIG_MAX && !PK_BIG_GLOBAL) return BigStage{PK_BIG_MAX, lds};
  return BigStage{A.big_row, A.big_stage + ((size_t)slot * 4 + (size_t)u) * (size_t)A.big_slot};
}
// ---- workgroup-wide intervals of a model whose derivative set is evaluated in groups (P::J_NG / P::H_NG > 1; see
// jac_groups): every pass walks the interval's nodes again (a thread owns several nodes, so the node arguments are loaded per
// pass), stages its group's rows, synchronizes the workgroup, walks the K^2 entries of its segments and synchronizes again.
template <class P, int G>
__device__ __forceinline__ void big_jac_groups(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const double* s, double dt,
                                               double mt, const PkSys& sy, double* __restrict__ js, int KS, pk_cbase_t segb,
                                               double width, int nq, int nown) {
  if constexpr (G < P::J_NG) {
    constexpr int I0 = P::J_gi0(G), NI = P::J_gni(G), N0 = P::J_gn0(G), NN = P::J_gnn(G);
    const int t = threadIdx.x;
    for (int c = t; c < nq; c += PK_BLOCK) {
      const int q = tl.q0 + c;
      double a[P::NARG], tau, w, o[NI + NN + 1];
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
      P::mid_jac_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
#pragma unroll
      for (int e = 0; e < NI; ++e) js[e * KS + c] = o[e];
      if (NN > 0 && c < nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < NN; ++e) put(&A.o_jac[segb[P::J_NI + N0 + e] + (q - ph.mid_lo)], o[NI + e]);
      }
    }
    if constexpr (NI > 0) {
      __syncthreads();
      const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
      const double* __restrict__ ivg = A.db + tl.iv_off;
      for (int p = t; p < tl.nnzI; p += PK_BLOCK) {
        const int c = rcg[2 * p + 1];
        const double val = -(ivg[p] * width * 0.5);
#pragma unroll
        for (int e = 0; e < NI; ++e) put(&A.o_jac[segb[I0 + e] + tl.offI + p], val * js[e * KS + c]);
      }
      __syncthreads();
    }
    big_jac_groups<P, G + 1>(A, ph, tl, s, dt, mt, sy, js, KS, segb, width, nq, nown);
  }
}
template <class P, int G>
__device__ __forceinline__ void big_hess_groups(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const double* s, double dt,
                                                double mt, const PkSys& sy, double* __restrict__ lds, int KS, pk_cbase_t segb,
                                                double width, int nq, int nown) {
  if constexpr (G < P::H_NG) {
    constexpr int I0 = P::H_gi0(G), NI = P::H_gni(G), N0 = P::H_gn0(G), NN = P::H_gnn(G);
    const int t = threadIdx.x;
    for (int c = t; c < nq; c += PK_BLOCK) {
      const int q = tl.q0 + c;
      double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], o[NI + NN + 1];
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
      for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
      P::mid_hess_g(Grp<G>{}, a, tau, dt, w, sy, lp, o);
#pragma unroll
      for (int e = 0; e < NI; ++e) lds[e * KS + c] = o[e];
      if (NN > 0 && c < nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < NN; ++e) put(&A.o_hess[segb[P::H_NI + N0 + e] + (q - ph.mid_lo)], o[NI + e]);
      }
    }
    if constexpr (NI > 0) {
      __syncthreads();
      const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
      const double* __restrict__ ivg = A.db + tl.iv_off;
      for (int p = t; p < tl.nnzI; p += PK_BLOCK) {
        const int r = rcg[2 * p], c = rcg[2 * p + 1];
        const double val = -(ivg[p] * width * 0.5);
#pragma unroll
        for (int e = 0; e < NI; ++e) {
          const double lam = A.lam[ph.g_off + P::H_state(I0 + e) * ph.L_d + tl.r0 + r];
          put(&A.o_hess[segb[I0 + e] + tl.offI + p], val * lam * lds[e * KS + c]);
        }
      }
      __syncthreads();
    }
    big_hess_groups<P, G + 1>(A, ph, tl, s, dt, mt, sy, lds, KS, segb, width, nq, nown);
  }
}

// the dynamics values of ALL states of a wide model, chunk function by chunk function (workgroup-wide intervals keep rows of
// every state)
template <class P, int C = 0>
__device__ __forceinline__ void dyn_all(const double* a, double* __restrict__ rows, int stride) {
  if constexpr (C < P::D_NG) {
    constexpr int I0 = P::D_c0(C), CN = P::D_cn(C);
    double o[CN];
    P::mid_dyn_g(Grp<C>{}, a, o);
#pragma unroll
    for (int e = 0; e < CN; ++e) rows[(I0 + e) * stride] = o[e];
    dyn_all<P, C + 1>(a, rows, stride);
  }
}
template <class P, int ROLE>
__device__ __forceinline__ void big_xall(const PkArgs& A, const PkTile& tl, double* __restrict__ lds,
                                         double* __restrict__ wint, double* __restrict__ wgrad, int pub_blk) {
  const PkPhase& ph = A.ph[P::INDEX];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int K = tl.K, stride = K - P::SCHEME, R = stride;
  const BigStage bs = big_stage(A, K, tl.stage, ROLE == 2 ? 1 : 0, lds);
  const int KS = bs.KS;
  const int nq = stride + P::SCHEME;
  const int nown = (P::SCHEME && !tl.last) ? nq - 1 : nq;
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  double* __restrict__ xs = bs.base;                      // [NX][KS] node values (boundary values substituted)
  double* __restrict__ fs = bs.base + P::NX * KS;         // [NX][KS] dynamics values
  double* __restrict__ js = bs.base + 2 * P::NX * KS;     // [J_NI][KS] Jacobian segment values
  pk_cbase_t segb = const_bases(A.lb + ph.jseg_off);
  pk_cbase_t tb = const_bases(A.lb + ph.jt_off);
  double oi[P::INT_N > 0 ? P::INT_N : 1], orr[P::GR_NR > 0 ? P::GR_NR : 1];
#pragma unroll
  for (int r = 0; r < P::INT_N; ++r) oi[r] = 0.0;
#pragma unroll
  for (int r = 0; r < P::GR_NR; ++r) orr[r] = 0.0;
  for (int c = t; c < nq; c += PK_BLOCK) {
    const int q = tl.q0 + c;
    double a[P::NARG], tau, w;
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    double og[P::G_NOUT], oj[P::J_NI + P::J_NN + 1], ov[P::NX + P::NU];
    double ot[P::GR_NR > 0 ? P::GR_NR : 1], op[P::INT_N > 0 ? P::INT_N : 1];
    if (ROLE != 2) {
#pragma unroll
      for (int i = 0; i < P::NX; ++i) xs[i * KS + c] = a[i];
    }
    if constexpr (P::J_NG == 1 && !P::WIDE) P::mid_xall(a, tau, dt, w, sy, og, oj, ov, ot, op);
    else if (ROLE != 2) {
      P::mid_xval(a, tau, dt, w, sy, fs + c, KS, og, ov, ot, op);   // (grouped: dynamics values straight into the rows,
      if constexpr (P::WIDE) dyn_all<P>(a, fs + c, KS);             //  the Jacobian pass by pass below)
    }
    if (ROLE != 2) {
      if (c < nown) {
#pragma unroll
        for (int r = 0; r < P::INT_N; ++r) oi[r] += op[r] * w;
        node_gradient_eval<P>(ph, q, a, tau, dt, w, sy, ov, ot, true);    // boundary nodes re-evaluate their own entries
#pragma unroll
        for (int r = 0; r < P::GR_NR; ++r) orr[r] += ot[r];
        node_gradient_store<P>(A, ph, q, ov);
#pragma unroll
        for (int j = 0; j < P::NC; ++j) put(&A.o_g[ph.path_off + j * ph.L_m + q], og[P::NX + j]);
      }
      if constexpr (P::J_NG == 1 && !P::WIDE) {
#pragma unroll
        for (int i = 0; i < P::NX; ++i) fs[i * KS + c] = og[i];
      }
    }
    if constexpr (P::J_NG == 1 && !P::WIDE) {
      if (ROLE != 1) {
#pragma unroll
        for (int e = 0; e < P::J_NI; ++e) js[e * KS + c] = oj[e];
        if (c < nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
          for (int e = 0; e < P::J_NN; ++e) put(&A.o_jac[segb[P::J_NI + e] + (q - ph.mid_lo)], oj[P::J_NI + e]);
        }
      }
    }
  }
  if (ROLE != 2) {
#pragma unroll
    for (int r = 0; r < P::INT_N; ++r) {
      const double v = wave_sum(oi[r]);
      if (lane == 0) wint[wave * PK_NRED + r] = v;
    }
#pragma unroll
    for (int r = 0; r < P::GR_NR; ++r) {
      const double v = wave_sum(orr[r]);
      if (lane == 0) wgrad[wave * PK_NRED + r] = v;
    }
  }
  __syncthreads();                                         // the staged rows are complete, the wave sums are in place
  if (ROLE != 2 && pub_blk >= 0 && t < PK_NRED) {          // pk_cycle: hand the block's sums to the finalize workgroup
    double vi = 0.0, vg = 0.0;
#pragma unroll
    for (int wv = 0; wv < PK_WAVES_PER_BLOCK; ++wv) {
      vi += wint[wv * PK_NRED + t];
      vg += wgrad[wv * PK_NRED + t];
    }
    handoff_put(A.cpart + (size_t)pub_blk * PK_NRED + t, vi);
    handoff_put(A.cpart2 + (size_t)pub_blk * PK_NRED + t, vg);
  }
  const double width = A.db[ph.width_off + tl.j0];
  if (ROLE != 2) {
    const double* __restrict__ full = A.db + tl.full_off;
    const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
    const int endslot = tl.q0 + stride;
#if PK_BIG_MFMA
    // defects: (x_q - x_end) - dt * sum_c (I_hat[r,c] d/2) f_i(c) -- the R x K block times the K x NX dynamics values on the
    // fp64 matrix cores, 16 rows per wave and step
    for (int rb = wave; rb < (R + 15) / 16; rb += PK_WAVES_PER_BLOCK)
      for (int cb = 0; cb < (P::NX + 15) / 16; ++cb)
        mfma_rows(full, K, R, K, width * 0.5, fs, KS, P::NX, rb, cb, lane, [&](int r, int i, double acc) {
          double xe = P::SCHEME ? xs[i * KS + (K - 1)] : A.x[ph.x_off + i * ph.state_len + endslot];
          if (endslot == back_slot) xe = P::back_value(i, xe, s);
          put(&A.o_g[ph.g_off + i * ph.L_d + tl.r0 + r], (xs[i * KS + r] - xe) - acc * dt);
        });
#else
    for (int r = t; r < R; r += PK_BLOCK) {                // defects: (x_q - x_end) - dt * sum_c (I_hat[r,c] d/2) f_i(c)
      double acc[P::NX];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) acc[i] = 0.0;
      for (int c = 0; c < K; ++c) {
        const double ac = full[r * K + c] * width * 0.5;
#pragma unroll
        for (int i = 0; i < P::NX; ++i) acc[i] += ac * fs[i * KS + c];
      }
#pragma unroll
      for (int i = 0; i < P::NX; ++i) {
        double xe = P::SCHEME ? xs[i * KS + (K - 1)] : A.x[ph.x_off + i * ph.state_len + endslot];
        if (endslot == back_slot) xe = P::back_value(i, xe, s);
        put(&A.o_g[ph.g_off + i * ph.L_d + tl.r0 + r], (xs[i * KS + r] - xe) - acc[i] * dt);
      }
    }
#endif
    const double* __restrict__ tvg = A.db + tl.tv_off;     // constant translation entries
    for (int p = t; p < tl.nnzT; p += PK_BLOCK) {
      const double v = tvg[p];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) put(&A.o_jac[tb[i] + tl.offT + p], v);
    }
  }
  if constexpr (P::J_NG == 1 && !P::WIDE) {
    if (ROLE != 1) {
      const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
      const double* __restrict__ ivg = A.db + tl.iv_off;
      for (int p = t; p < tl.nnzI; p += PK_BLOCK) {
        const int c = rcg[2 * p + 1];
        const double val = -(ivg[p] * width * 0.5);
#pragma unroll
        for (int e = 0; e < P::J_NI; ++e) put(&A.o_jac[segb[e] + tl.offI + p], val * js[e * KS + c]);
      }
    }
  } else if (ROLE != 1) {
    big_jac_groups<P, 0>(A, ph, tl, s, dt, mt, sy, js, KS, segb, width, nq, nown);
  }
}

template <class P>
__device__ __forceinline__ void big_hess(const PkArgs& A, const PkTile& tl, double* __restrict__ lds0) {
  const PkPhase& ph = A.ph[P::INDEX];
  const int t = threadIdx.x;
  const int K = tl.K, stride = K - P::SCHEME;
  const BigStage bs = big_stage(A, K, tl.stage, 2, lds0);
  const int KS = bs.KS;
  double* __restrict__ lds = bs.base;
  const int nq = stride + P::SCHEME;
  const int nown = (P::SCHEME && !tl.last) ? nq - 1 : nq;
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  pk_cbase_t segb = const_bases(A.lb + ph.hseg_off);
  if constexpr (P::H_NG > 1) {
    big_hess_groups<P, 0>(A, ph, tl, s, dt, mt, sy, lds, KS, segb, A.db[ph.width_off + tl.j0], nq, nown);
  } else {
  for (int c = t; c < nq; c += PK_BLOCK) {
    const int q = tl.q0 + c;
    double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], o[P::H_NI + P::H_NN + 1];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
    for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
    P::mid_hess(a, tau, dt, w, sy, lp, o);
#pragma unroll
    for (int e = 0; e < P::H_NI; ++e) lds[e * KS + c] = o[e];
    if (c < nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::H_NN; ++e) put(&A.o_hess[segb[P::H_NI + e] + (q - ph.mid_lo)], o[P::H_NI + e]);
    }
  }
  __syncthreads();
  const double width = A.db[ph.width_off + tl.j0];
  const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
  const double* __restrict__ ivg = A.db + tl.iv_off;
  for (int p = t; p < tl.nnzI; p += PK_BLOCK) {
    const int r = rcg[2 * p], c = rcg[2 * p + 1];
    const double val = -(ivg[p] * width * 0.5);
#pragma unroll
    for (int e = 0; e < P::H_NI; ++e) {
      const double lam = A.lam[ph.g_off + P::H_state(e) * ph.L_d + tl.r0 + r];
      put(&A.o_hess[segb[e] + tl.offI + p], val * lam * lds[e * KS + c]);
    }
  }
  }
}

// integrals only (pk_int, the prepass of models whose system functions are nonlinear in the integrals): the interval's
// nodes in the thread order of big_xall, per-wave sums into wint
template <class P>
__device__ __forceinline__ void big_int(const PkArgs& A, const PkTile& tl, double* __restrict__ wint) {
  if (P::INT_N == 0) return;
  const PkPhase& ph = A.ph[P::INDEX];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int nq = tl.K, nown = (P::SCHEME && !tl.last) ? nq - 1 : nq;
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  double oi[P::INT_N > 0 ? P::INT_N : 1];
#pragma unroll
  for (int r = 0; r < P::INT_N; ++r) oi[r] = 0.0;
  for (int c = t; c < nown; c += PK_BLOCK) {
    double a[P::NARG], tau, w, o[P::INT_N > 0 ? P::INT_N : 1];
    load_node<P>(A, ph, s, dt, mt, tl.q0 + c, a, tau, w);
    P::mid_int(a, o);
#pragma unroll
    for (int r = 0; r < P::INT_N; ++r) oi[r] += o[r] * w;
  }
#pragma unroll
  for (int r = 0; r < P::INT_N; ++r) {
    const double v = wave_sum(oi[r]);
    if (lane == 0) wint[wave * PK_NRED + r] = v;
  }
}

// auxiliary pass of the outer-product path (tile_aux below) for such an interval
template <class P>
__device__ __forceinline__ void big_aux(const PkArgs& A, const PkTile& tl) {
  if (P::A_NN == 0) return;
  const PkPhase& ph = A.ph[P::INDEX];
  const int nq = tl.K, nown = (P::SCHEME && !tl.last) ? nq - 1 : nq;
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  pk_cbase_t segb = const_bases(A.lb + ph.aseg_off);
  for (int c = threadIdx.x; c < nown; c += PK_BLOCK) {
    const int q = tl.q0 + c;
    if (q >= ph.mid_lo && q < ph.mid_hi) {
      double a[P::NARG], tau, w, o[P::A_NN + 1];
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
      P::mid_aux(a, tau, dt, w, sy, nullptr, o);
#pragma unroll
      for (int e = 0; e < P::A_NN; ++e) A.o_aux[segb[e] + (q - ph.mid_lo)] = o[e];
    }
  }
}

template <class P, int G>
__device__ __forceinline__ void big_jacc_groups(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const double* s, double dt,
                                                double mt, const PkSys& sy, double* __restrict__ sv, int KS, pk_cbase_t segb,
                                                double width, int nq, int nown) {
  if constexpr (G < P::JC_NG) {
    constexpr int KIND = P::JC_gk(G), LO = P::JC_g0(G), CN = P::JC_gn(G);
    const int t = threadIdx.x, K = tl.K, R = K - P::SCHEME;
    for (int c = t; c < nq; c += PK_BLOCK) {
      const int q = tl.q0 + c;
      double a[P::NARG], tau, w, o[CN + 1];
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
      P::mid_jacc_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
      if constexpr (KIND == 1) {      // the boundary nodes carry their own expressions of the dense-column entries
        if (q == 0) P::front_jacc_dense_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
        else if (P::SCHEME == 1 && q == ph.L_m - 1) P::back_jacc_dense_g(Grp<G>{}, a, tau, dt, w, sy, nullptr, o);
      }
      if constexpr (KIND < 2) {
#pragma unroll
        for (int e = 0; e < CN; ++e) sv[e * KS + c] = o[e];
      } else if (c < nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
        for (int e = 0; e < CN; ++e) put(&A.o_jac[segb[P::JC_NI + P::JC_ND + LO + e] + (q - ph.mid_lo)], o[e]);
      }
    }
    if constexpr (KIND < 2) {
      __syncthreads();
      if constexpr (KIND == 0) {
        const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
        const double* __restrict__ ivg = A.db + tl.iv_off;
        for (int p = t; p < tl.nnzI; p += PK_BLOCK) {
          const int c = rcg[2 * p + 1];
          const double val = -(ivg[p] * width * 0.5);
#pragma unroll
          for (int e = 0; e < CN; ++e) put(&A.o_jac[segb[LO + e] + tl.offI + p], val * sv[e * KS + c]);
        }
      } else {
        const double* __restrict__ full = A.db + tl.full_off;
        double a[P::NARG], tau, w, tf[CN], tbk[CN];
        load_node<P>(A, ph, s, dt, mt, tl.q0, a, tau, w);        // (the static parameters, for the boundary shares)
        P::jacc_tdense_g(Grp<G>{}, a, tf, tbk);
        const bool last_iv = tl.j0 == ph.n_int - 1;
        for (int r = t; r < R; r += PK_BLOCK) {
          double acc[CN];
#pragma unroll
          for (int e = 0; e < CN; ++e) acc[e] = 0.0;
          for (int c = 0; c < K; ++c) {
            const double wgt = full[r * K + c] * width * 0.5;
#pragma unroll
            for (int e = 0; e < CN; ++e) acc[e] += wgt * sv[e * KS + c];
          }
          const bool first = tl.r0 + r == 0;
#pragma unroll
          for (int e = 0; e < CN; ++e)
            put(&A.o_jac[segb[P::JC_NI + LO + e] + tl.r0 + r], (first ? tf[e] : 0.0) + (last_iv ? tbk[e] : 0.0) - acc[e]);
        }
      }
      __syncthreads();
    }
    big_jacc_groups<P, G + 1>(A, ph, tl, s, dt, mt, sy, sv, KS, segb, width, nq, nown);
  }
}

// compact Jacobian (tile_jacc above) of such an interval: the expanded and the dense-column segment values of all nodes are
// staged ([JC_NI + JC_ND][KS], sub-slot 3 beyond 256 points), then the translation entries, the dense rows (one K-term product
// per defect row and dense-column segment) and the K^2 entries of every expanded segment are walked by all 256 threads
template <class P>
__device__ __forceinline__ void big_jacc(const PkArgs& A, const PkTile& tl, double* __restrict__ lds0) {
  const PkPhase& ph = A.ph[P::INDEX];
  const int t = threadIdx.x;
  const int K = tl.K, stride = K - P::SCHEME, R = stride;
  const int nq = stride + P::SCHEME;
  const int nown = (P::SCHEME && !tl.last) ? nq - 1 : nq;
  const BigStage bs = big_stage(A, K, tl.stage, 3, lds0);
  const int KS = bs.KS;
  double* __restrict__ sv = bs.base;
  double* __restrict__ dsv = bs.base + P::JC_NI * KS;
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  pk_cbase_t segb = const_bases(A.lb + ph.jcseg_off);
  pk_cbase_t tb = const_bases(A.lb + ph.jct_off);
  if constexpr (P::JC_gk(0) >= 0) {      // evaluated in groups: the translation entries, then pass by pass
    const double* __restrict__ tvg = A.db + tl.tv_off;
    for (int p = t; p < tl.nnzT; p += PK_BLOCK) {
      const double v = tvg[p];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) put(&A.o_jac[tb[i] + tl.offT + p], v);
    }
    big_jacc_groups<P, 0>(A, ph, tl, s, dt, mt, sy, sv, KS, segb, A.db[ph.width_off + tl.j0], nq, nown);
  } else {
  for (int c = t; c < nq; c += PK_BLOCK) {
    const int q = tl.q0 + c;
    double a[P::NARG], tau, w, o[P::JC_NI + P::JC_ND + P::JC_NN + 1];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
    P::mid_jacc(a, tau, dt, w, sy, nullptr, o);
    if (P::JC_ND > 0) {              // the boundary nodes carry their own expressions of the dense-column entries
      if (q == 0) P::front_jacc_dense(a, tau, dt, w, sy, nullptr, o + P::JC_NI);
      else if (P::SCHEME == 1 && q == ph.L_m - 1) P::back_jacc_dense(a, tau, dt, w, sy, nullptr, o + P::JC_NI);
    }
#pragma unroll
    for (int e = 0; e < P::JC_NI + P::JC_ND; ++e) sv[e * KS + c] = o[e];
    if (c < nown && q >= ph.mid_lo && q < ph.mid_hi) {
#pragma unroll
      for (int e = 0; e < P::JC_NN; ++e)
        put(&A.o_jac[segb[P::JC_NI + P::JC_ND + e] + (q - ph.mid_lo)], o[P::JC_NI + P::JC_ND + e]);
    }
  }
  __syncthreads();
  const double width = A.db[ph.width_off + tl.j0];
  const double* __restrict__ tvg = A.db + tl.tv_off;       // constant translation entries
  for (int p = t; p < tl.nnzT; p += PK_BLOCK) {
    const double v = tvg[p];
#pragma unroll
    for (int i = 0; i < P::NX; ++i) put(&A.o_jac[tb[i] + tl.offT + p], v);
  }
  if (P::JC_ND > 0) {
    constexpr int ND = P::JC_ND > 0 ? P::JC_ND : 1;
    const double* __restrict__ full = A.db + tl.full_off;
    double a[P::NARG], tau, w, tf[ND], tbk[ND];
    load_node<P>(A, ph, s, dt, mt, tl.q0, a, tau, w);        // (the static parameters, for the boundary shares)
    P::jacc_tdense(a, tf, tbk);
    const bool last_iv = tl.j0 == ph.n_int - 1;
    for (int r = t; r < R; r += PK_BLOCK) {
      double acc[ND];
#pragma unroll
      for (int e = 0; e < P::JC_ND; ++e) acc[e] = 0.0;
      for (int c = 0; c < K; ++c) {
        const double wgt = full[r * K + c] * width * 0.5;
#pragma unroll
        for (int e = 0; e < P::JC_ND; ++e) acc[e] += wgt * dsv[e * KS + c];
      }
      const bool first = tl.r0 + r == 0;
#pragma unroll
      for (int e = 0; e < P::JC_ND; ++e)
        put(&A.o_jac[segb[P::JC_NI + e] + tl.r0 + r], (first ? tf[e] : 0.0) + (last_iv ? tbk[e] : 0.0) - acc[e]);
    }
  }
  const int32_t* __restrict__ rcg = A.ib + tl.irc_off;
  const double* __restrict__ ivg = A.db + tl.iv_off;
  for (int p = t; p < tl.nnzI; p += PK_BLOCK) {
    const int c = rcg[2 * p + 1];
    const double val = -(ivg[p] * width * 0.5);
#pragma unroll
    for (int e = 0; e < P::JC_NI; ++e) put(&A.o_jac[segb[e] + tl.offI + p], val * sv[e * KS + c]);
  }
  }
}

// does tile block `blk` hold an interval with more points than a wave has lanes?  (wave-uniform; tl0 receives its record)
__device__ __forceinline__ bool big_block(const PkTile* tiles, int n_tiles, int blk, PkTile& tl0) {
  const int t0 = blk * PK_WAVES_PER_BLOCK;
  if (t0 >= n_tiles) return false;
  tl0 = load_tile(tiles + t0);
  return tl0.nj > 0 && tl0.K > PK_WAVE;
}
#endif

// ============================================================================================
// auxiliary pass of the outer-product path: quadrature-weighted gradient entries of the integrals
// (one value per middle node), consumed by pk_outer       (systembase.py:625-644; easyderiv.py:393-430)
// ============================================================================================
template <class P>
__device__ __forceinline__ void tile_aux(const PkArgs& A, const PkTile& tl, double* __restrict__,
                                         double* __restrict__, double* __restrict__, int lane) {
  if (P::A_NN == 0) return;
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  pk_cbase_t segb = const_bases(A.lb + ph.aseg_off);
  if (lane < g.nown) {
    const int q = tl.q0 + lane;
    if (q >= ph.mid_lo && q < ph.mid_hi) {
      double a[P::NARG], tau, w, o[P::A_NN + 1];
      load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
      P::mid_aux(a, tau, dt, w, sy, nullptr, o);
#pragma unroll
      for (int e = 0; e < P::A_NN; ++e) A.o_aux[segb[e] + (q - ph.mid_lo)] = o[e];
    }
  }
}

// ============================================================================================
// compact Hessian: the multipliers are contracted with the integration block first,
//   mu_i(node) = sum_r (I_hat[r, c] d/2) * lambda[row r of state i]      (a K-term product per node and state)
// and every (row, col) position of a node gets ONE value (entries summed symbolically), so the output is
// one run of L_m - 1 doubles per distinct position class instead of N*K^2 triplets per derivative entry.
// ============================================================================================
template <class P>
__device__ __forceinline__ void interval_mu(const PkArgs& A, const PkPhase& ph, int K, const double* __restrict__ full,
                                            double width, int ld, int c, double* mu) {
  const int R = K - P::SCHEME;
  const double* __restrict__ lam = A.lam + ph.g_off + ld;
  for (int r = 0; r < R; ++r) {
    const double a = full[r * K + c] * width * 0.5;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) mu[i] += a * lam[i * ph.L_d + r];
  }
}

// mu of node q (phase-local), summed over the one (LGR) or two (LGL interior mesh point) intervals holding it
template <class P>
__device__ __forceinline__ void node_mu(const PkArgs& A, const PkPhase& ph, int j, int c, double* mu) {
#pragma unroll
  for (int i = 0; i < P::NX; ++i) mu[i] = 0.0;
  const int32_t* __restrict__ ivK = A.ib + ph.ivK_off;
  const int32_t* __restrict__ ivF = A.ib + ph.ivfull_off;
  const int32_t* __restrict__ ivL = A.ib + ph.ivld_off;
  interval_mu<P>(A, ph, ivK[j], A.db + ivF[j], A.db[ph.width_off + j], ivL[j], c, mu);
  if (P::SCHEME == 1 && c == 0 && j > 0)
    interval_mu<P>(A, ph, ivK[j - 1], A.db + ivF[j - 1], A.db[ph.width_off + j - 1], ivL[j - 1], ivK[j - 1] - 1, mu);
}

// mu of the tile's node `lane` from LDS: the tile's multiplier rows lam_s[state][row] and (STAGED) its integration block
template <class P, bool STAGED>
__device__ __forceinline__ void tile_mu(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                        const TileTabs& T, const double* __restrict__ lam_s, int jj, int c, double* mu) {
  const double* __restrict__ full = STAGED ? T.full : A.db + tl.full_off;
  const double width = STAGED ? T.wd[jj] : A.db[ph.width_off + tl.j0 + jj];
  const double* __restrict__ lam = lam_s + jj * g.R;
#pragma unroll 4
  for (int r = 0; r < g.R; ++r) {
    const double a = full[r * g.K + c] * width * 0.5;       // (I_hat * d) / 2, the reference's scaling
#pragma unroll
    for (int i = 0; i < P::NX; ++i) mu[i] += a * lam[i * PK_WAVE + r];
  }
}

// the per-node values of the compact Hessian, chunk by chunk (one chunk unless the model is large: codegen.split_chunks)
// (sub, stride: a pass-parallel launch gives a tile block `stride` workgroups for this role; workgroup `sub` takes the chunks
//  G with G % stride == sub.  stride 0: all chunks in this wave)
template <class P, int G>
__device__ __forceinline__ void hessc_chunks(const PkArgs& A, const PkPhase& ph, const long long* __restrict__ segb, double (&a)[P::NARG],
                                             double& tau, double& dt, double& w, const PkSys& sy,
                                             double (&lp)[P::NC > 0 ? P::NC : 1], double (&mu)[P::NX], int q, int sub = 0,
                                             int stride = 0) {
  if constexpr (G < P::HC_NG) {
    constexpr int C0 = P::HC_c0(G), CN = P::HC_cn(G);
    if constexpr (G > 0) {
      fresh_args(a, tau, dt, w);
      settle(lp);
      settle(mu);
    }
    if (stride == 0 || G % stride == sub) {
      double o[CN + 1];
      P::mid_hessc_g(Grp<G>{}, a, tau, dt, w, sy, lp, mu, nullptr, nullptr, o);
#pragma unroll
      for (int e = 0; e < CN; ++e) put(&A.o_hess[segb[C0 + e] + (q - ph.mid_lo)], o[e]);
    }
    hessc_chunks<P, G + 1>(A, ph, segb, a, tau, dt, w, sy, lp, mu, q, sub, stride);
  }
}

// ---- the compact Hessian of a WIDE model (P::WIDE): every chunk of outputs is a pass of its own that fetches the node
// arguments its expressions read and contracts the multipliers of the states THEY refer to -- the list P::HC_mus(G, k),
// k < P::HC_nmu(G), from the code generator; row k of the staged multiplier rows and mu[k] belong to state HC_mus(G, k).
// (All states at once were n_x rows of LDS and n_x register pairs of mu.)
template <class P, int G>
__device__ __forceinline__ void interval_mu_l(const PkArgs& A, const PkPhase& ph, int K, const double* __restrict__ full,
                                              double width, int ld, int c, double* mu) {
  constexpr int NM = P::HC_nmu(G);
  const int R = K - P::SCHEME;
  const double* __restrict__ lam = A.lam + ph.g_off + ld;
  for (int r = 0; r < R; ++r) {
    const double a = full[r * K + c] * width * 0.5;
#pragma unroll
    for (int k = 0; k < NM; ++k) mu[k] += a * lam[P::HC_mus(G, k) * ph.L_d + r];
  }
}
template <class P, int G>
__device__ __forceinline__ void node_mu_l(const PkArgs& A, const PkPhase& ph, int j, int c, double* mu) {
  constexpr int NM = P::HC_nmu(G);
#pragma unroll
  for (int k = 0; k < NM; ++k) mu[k] = 0.0;
  const int32_t* __restrict__ ivK = A.ib + ph.ivK_off;
  const int32_t* __restrict__ ivF = A.ib + ph.ivfull_off;
  const int32_t* __restrict__ ivL = A.ib + ph.ivld_off;
  interval_mu_l<P, G>(A, ph, ivK[j], A.db + ivF[j], A.db[ph.width_off + j], ivL[j], c, mu);
  if (P::SCHEME == 1 && c == 0 && j > 0)
    interval_mu_l<P, G>(A, ph, ivK[j - 1], A.db + ivF[j - 1], A.db[ph.width_off + j - 1], ivL[j - 1], ivK[j - 1] - 1, mu);
}
template <class P, bool STAGED, int NM>
__device__ __forceinline__ void tile_mu_n(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                          const TileTabs& T, const double* __restrict__ lam_s, int jj, int c, double* mu) {
  const double* __restrict__ full = STAGED ? T.full : A.db + tl.full_off;
  const double width = STAGED ? T.wd[jj] : A.db[ph.width_off + tl.j0 + jj];
  const double* __restrict__ lam = lam_s + jj * g.R;
#pragma unroll 4
  for (int r = 0; r < g.R; ++r) {
    const double a = full[r * g.K + c] * width * 0.5;       // (I_hat * d) / 2, the reference's scaling
#pragma unroll
    for (int k = 0; k < NM; ++k) mu[k] += a * lam[k * PK_WAVE + r];
  }
}
template <class P, int G>
__device__ __forceinline__ void hessc_passes_wide(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                                  const TileTabs& T, const double* s, double dt, double mt, const PkSys& sy,
                                                  double* __restrict__ lam_s, const long long* __restrict__ segb, int lane,
                                                  int sub = 0, int stride = 0) {
  if constexpr (G < P::HC_NG) {
    if (stride == 0 || G % stride == sub) {
    constexpr int C0 = P::HC_c0(G), CN = P::HC_cn(G), NM = P::HC_nmu(G), NMD = NM > 0 ? NM : 1;
    const int q = tl.q0 + lane;
    const int row = min(tl.r0 + lane, ph.L_d - 1);
    double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], lrow[NMD], mu[NMD];
#pragma unroll
    for (int k = 0; k < NM; ++k) lrow[k] = A.lam[ph.g_off + P::HC_mus(G, k) * ph.L_d + row];
    load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
    for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
#pragma unroll
    for (int k = 0; k < NM; ++k) lam_s[k * PK_WAVE + lane] = lrow[k];
    wave_lds_sync();
    if (lane < g.nown && q >= ph.mid_lo && q < ph.mid_hi) {
      const int jj = min(magic_div((uint32_t)lane, tl.magicR), max(tl.nj - 1, 0)), c = lane - jj * g.stride;
#pragma unroll
      for (int k = 0; k < NM; ++k) mu[k] = 0.0;
      if (T.staged) tile_mu_n<P, true, NM>(A, ph, tl, g, T, lam_s, jj, c, mu);
      else tile_mu_n<P, false, NM>(A, ph, tl, g, T, lam_s, jj, c, mu);
      if (P::SCHEME == 1 && c == 0 && tl.j0 + jj > 0) {       // LGL: a mesh point also closes the interval before it
        if (jj > 0) {
          if (T.staged) tile_mu_n<P, true, NM>(A, ph, tl, g, T, lam_s, jj - 1, g.K - 1, mu);
          else tile_mu_n<P, false, NM>(A, ph, tl, g, T, lam_s, jj - 1, g.K - 1, mu);
        } else {                                              // (that interval belongs to the tile before this one)
          const int32_t* __restrict__ ivK = A.ib + ph.ivK_off;
          const int32_t* __restrict__ ivF = A.ib + ph.ivfull_off;
          const int32_t* __restrict__ ivL = A.ib + ph.ivld_off;
          const int jp = tl.j0 - 1;
          interval_mu_l<P, G>(A, ph, ivK[jp], A.db + ivF[jp], A.db[ph.width_off + jp], ivL[jp], ivK[jp] - 1, mu);
        }
      }
      double o[CN + 1];
      P::mid_hessc_g(Grp<G>{}, a, tau, dt, w, sy, lp, mu, nullptr, nullptr, o);
#pragma unroll
      for (int e = 0; e < CN; ++e) put(&A.o_hess[segb[C0 + e] + (q - ph.mid_lo)], o[e]);
    }
    wave_lds_sync();                        // (the next pass overwrites the rows this one read)
    }
    hessc_passes_wide<P, G + 1>(A, ph, tl, g, T, s, dt, mt, sy, lam_s, segb, lane, sub, stride);
  }
}
#ifdef PK_BIG
// ... and of an interval with more points than a wave has lanes (alone in its tile): 64 nodes at a time, global-memory form
template <class P, int G>
__device__ __forceinline__ void hessc_big_wide(const PkArgs& A, const PkPhase& ph, const PkTile& tl, const TileGeom& g,
                                               const double* s, double dt, double mt, const PkSys& sy,
                                               const long long* __restrict__ segb, int lane) {
  if constexpr (G < P::HC_NG) {
    constexpr int C0 = P::HC_c0(G), CN = P::HC_cn(G), NM = P::HC_nmu(G);
    for (int c = lane; c < g.nown; c += PK_WAVE) {
      const int q = tl.q0 + c;
      if (q >= ph.mid_lo && q < ph.mid_hi) {
        double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], mu[NM > 0 ? NM : 1], o[CN + 1];
        load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
        for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + q];
        const int jj = c / g.stride;
        node_mu_l<P, G>(A, ph, tl.j0 + jj, c - jj * g.stride, mu);
        P::mid_hessc_g(Grp<G>{}, a, tau, dt, w, sy, lp, mu, nullptr, nullptr, o);
#pragma unroll
        for (int e = 0; e < CN; ++e) put(&A.o_hess[segb[C0 + e] + (q - ph.mid_lo)], o[e]);
      }
    }
    hessc_big_wide<P, G + 1>(A, ph, tl, g, s, dt, mt, sy, segb, lane);
  }
}
#endif

// One wave per tile: lane = node.  The tile's rows of the defect multipliers (lane = row, coalesced) and its integration
// block are staged in LDS while the node loads are in flight, so the K-term contraction per node and state reads LDS only
// (the first version read lambda and the block from global memory inside the loop: K (1 + NX) dependent-latency loads per
// lane, 7.7 us at 40k nodes for 8.6 MB of output).  An interval with more than 64 points -- alone in its tile -- is walked
// 64 nodes at a time with the global-memory form.
template <class P>
__device__ __forceinline__ void tile_hessc(const PkArgs& A, const PkTile& tl, double* __restrict__ lam_s,
                                           double* __restrict__, double* __restrict__, int lane, int sub = 0, int stride = 0) {
  if (P::HC_NN == 0) return;
  if (stride > 0 && (sub >= P::HC_NG || (tl.K > PK_WAVE && sub > 0))) return;      // (no chunk left for this workgroup; an interval
  if (tl.K > PK_WAVE) stride = 0;                                                  //  with more than 64 points: all chunks in one)
  const PkPhase& ph = A.ph[P::INDEX];
  const TileGeom g = tile_geom<P>(tl);
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkSys sy{s, A.Ibuf, A.sigma, A.lam};
  // The base offsets of the HC_NN output runs are staged in the wave's LDS behind its multiplier rows (coalesced load, one
  // broadcast ds_read at every store).  As scalar loads -- an SGPR pair per output, all hoisted to the top by the scheduler --
  // they spilled 40 SGPRs and cost the kernel a private segment; held in a VGPR pair and handed over by v_readlane they
  // were hoisted just the same (123 SGPR spills).
  long long* __restrict__ segb = reinterpret_cast<long long*>(lam_s + P::HC_LROWS * PK_WAVE);      // (HC_LROWS: NX, or -- WIDE -- the
  for (int e = lane; e < P::HC_NN; e += PK_WAVE) segb[e] = (long long)A.lb[ph.hcseg_off + e];      //  most states a pass refers to)
  if constexpr (P::WIDE) {
#ifdef PK_BIG
    if (tl.K > PK_WAVE) {
      wave_lds_sync();
      return hessc_big_wide<P, 0>(A, ph, tl, g, s, dt, mt, sy, segb, lane);
    }
#endif
    const bool fitw = tabs_fit(A, tl, g);
    const TabRegs trw = tabs_issue(A, ph, tl, g, fitw, lane);
    const TileTabs Tw = tabs_commit(A, tl, g, trw, fitw, lane);
    return hessc_passes_wide<P, 0>(A, ph, tl, g, Tw, s, dt, mt, sy, lam_s, segb, lane, sub, stride);
  }
#ifdef PK_BIG      // (such intervals exist only on meshes whose code object is generated with PK_BIG)
  if (tl.K > PK_WAVE) {
    wave_lds_sync();
    for (int c = lane; c < g.nown; c += PK_WAVE) {
      const int q = tl.q0 + c;
      if (q >= ph.mid_lo && q < ph.mid_hi) {
        double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], mu[P::NX];
        load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
        for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + q];
        const int jj = c / g.stride;
        node_mu<P>(A, ph, tl.j0 + jj, c - jj * g.stride, mu);
        hessc_chunks<P, 0>(A, ph, segb, a, tau, dt, w, sy, lp, mu, q);
      }
    }
    return;
  }
#endif
  const bool fit = tabs_fit(A, tl, g);
  const TabRegs tr = tabs_issue(A, ph, tl, g, fit, lane);
  const int q = tl.q0 + lane;
  double a[P::NARG], tau, w, lp[P::NC > 0 ? P::NC : 1], lrow[P::NX], mu[P::NX];
  const int row = min(tl.r0 + lane, ph.L_d - 1);
#pragma unroll
  for (int i = 0; i < P::NX; ++i) lrow[i] = A.lam[ph.g_off + i * ph.L_d + row];
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
  for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + min(q, ph.L_m - 1)];
  const TileTabs T = tabs_commit(A, tl, g, tr, fit, lane);
#pragma unroll
  for (int i = 0; i < P::NX; ++i) lam_s[i * PK_WAVE + lane] = lrow[i];
  loads_done();
  wave_lds_sync();
  if (lane >= g.nown || q < ph.mid_lo || q >= ph.mid_hi) return;
  const int jj = min(magic_div((uint32_t)lane, tl.magicR), max(tl.nj - 1, 0)), c = lane - jj * g.stride;
#pragma unroll
  for (int i = 0; i < P::NX; ++i) mu[i] = 0.0;
  if (T.staged) tile_mu<P, true>(A, ph, tl, g, T, lam_s, jj, c, mu);
  else tile_mu<P, false>(A, ph, tl, g, T, lam_s, jj, c, mu);
  if (P::SCHEME == 1 && c == 0 && tl.j0 + jj > 0) {       // LGL: a mesh point also closes the interval before it
    if (jj > 0) {
      if (T.staged) tile_mu<P, true>(A, ph, tl, g, T, lam_s, jj - 1, g.K - 1, mu);
      else tile_mu<P, false>(A, ph, tl, g, T, lam_s, jj - 1, g.K - 1, mu);
    } else {                                              // (that interval belongs to the tile before this one)
      const int32_t* __restrict__ ivK = A.ib + ph.ivK_off;
      const int32_t* __restrict__ ivF = A.ib + ph.ivfull_off;
      const int32_t* __restrict__ ivL = A.ib + ph.ivld_off;
      const int jp = tl.j0 - 1;
      interval_mu<P>(A, ph, ivK[jp], A.db + ivF[jp], A.db[ph.width_off + jp], ivL[jp], ivK[jp] - 1, mu);
    }
  }
  hessc_chunks<P, 0>(A, ph, segb, a, tau, dt, w, sy, lp, mu, q, sub, stride);
}
template <class P>
__device__ __forceinline__ void tile_hessc_part(int sub, int stride, const PkArgs& A, const PkTile& tl, double* __restrict__ lam_s,
                                                int lane) {
  tile_hessc<P>(A, tl, lam_s, nullptr, nullptr, lane, sub, stride);
}

// boundary node of the compact Hessian: also the contracted multipliers of the boundary columns
template <class P>
__device__ __forceinline__ void load_edge_c(const PkArgs& A, int back, double* s, double* a, double& tau, double& dt,
                                            double& w, double* lp, double* mu, double* ltf, double* ltb) {
  const PkPhase& ph = A.ph[P::INDEX];
  double mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const int q = back ? ph.L_m - 1 : 0;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
#pragma unroll
  for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + q];
  const int N = ph.n_int;
  const int32_t* __restrict__ ivK = A.ib + ph.ivK_off;
  const int32_t* __restrict__ ivL = A.ib + ph.ivld_off;
  if (back && P::SCHEME == 1) node_mu<P>(A, ph, N - 1, ivK[N - 1] - 1, mu);
  else node_mu<P>(A, ph, 0, 0, mu);
  const int Rl = ivK[N - 1] - P::SCHEME;
#pragma unroll
  for (int i = 0; i < P::NX; ++i) {
    const double* __restrict__ lam = A.lam + ph.g_off + i * ph.L_d;
    ltf[i] = lam[0];                                 // T_f = (row 0, +1)
    double sum = 0.0;
    for (int r = 0; r < Rl; ++r) sum += lam[ivL[N - 1] + r];
    ltb[i] = -sum;                                   // T_b = (rows of the last interval, -1)
  }
}

// ---- boundary-node evaluation for the edge workgroup -----------------------------------------
template <class P>
__device__ __forceinline__ void load_edge(const PkArgs& A, int back, double* s, double* a, double& tau,
                                          double& dt, double& w, double* lp) {
  const PkPhase& ph = A.ph[P::INDEX];
  double mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const int q = back ? ph.L_m - 1 : 0;
  load_node<P>(A, ph, s, dt, mt, q, a, tau, w);
  if (A.lam != nullptr) {
#pragma unroll
    for (int j = 0; j < P::NC; ++j) lp[j] = A.lam[ph.path_off + j * ph.L_m + q];
  }
}

// item `threadIdx.x` and its multiplier are fetched before the boundary expressions are evaluated
struct ItemPref {
  PkItem m;
  double lam;
  bool ok;
};
__device__ __forceinline__ ItemPref prefetch_item(const PkArgs& A, const PkItem* __restrict__ items, int n_items) {
  ItemPref ip;
  ip.ok = (int)threadIdx.x < n_items;
  ip.lam = 1.0;
  if (ip.ok) {
    ip.m = items[threadIdx.x];
    if (ip.m.lam >= 0) ip.lam = A.lam[ip.m.lam];
  }
  return ip;
}
__device__ __forceinline__ void scatter_items(const PkArgs& A, const PkItem* __restrict__ items, int n_items,
                                              const ItemPref& ip, const double* __restrict__ E,
                                              double* __restrict__ out) {
  if (ip.ok) out[ip.m.pos] = ip.m.coef * E[ip.m.eid] * ip.lam;
  for (int it = threadIdx.x + PK_BLOCK; it < n_items; it += PK_BLOCK) {
    const PkItem m = items[it];
    double v = m.coef * E[m.eid];
    if (m.lam >= 0) v *= A.lam[m.lam];
    out[m.pos] = v;
  }
}

// The four wave sums of a workgroup -> one partial per workgroup.  The host pads every phase's tile
// list to a multiple of PK_WAVES_PER_BLOCK (empty tiles), so a workgroup never mixes phases.
__device__ __forceinline__ void publish_block_partials(double* __restrict__ partial, const double* __restrict__ wred,
                                                       int blk) {
  __syncthreads();
  if ((int)threadIdx.x < PK_NRED) {
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < PK_WAVES_PER_BLOCK; ++w) v += wred[w * PK_NRED + threadIdx.x];
    partial[(size_t)blk * PK_NRED + threadIdx.x] = v;
  }
}

// deterministic sum over the workgroups of phase k (fixed shape: strided thread sums, wave shuffle
// tree, 4-way LDS combine)
__device__ __forceinline__ double block_sum_partials(const PkArgs& A, const double* __restrict__ partial, int k,
                                                     int r, double* red) {
  const int blo = A.ph[k].tile_lo / PK_WAVES_PER_BLOCK, bhi = A.ph[k].tile_hi / PK_WAVES_PER_BLOCK;
  double v = 0.0;
  for (int b = blo + (int)threadIdx.x; b < bhi; b += PK_BLOCK) v += partial[(size_t)b * PK_NRED + r];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < PK_WAVES_PER_BLOCK; ++w) tot += red[w];
  return tot;
}

// ============================================================================================
// kernels
// ============================================================================================
// EDGE = 1: workgroup 0 is the boundary/system workgroup (dispatched first: its serial chain is the
// longest of the launch), tile workgroups follow.
#define PK_TILE_PROLOGUE(EDGE)                                                        \
  const int blk = pk::xcd_tile_block((int)blockIdx.x, (EDGE), (int)gridDim.x);        \
  PK_TILE_PROLOGUE_AT()
// PK_TILE_PROLOGUE_AT: the caller has defined `blk`, the workgroup's tile block (four consecutive tiles)
#define PK_TILE_PROLOGUE_AT() PK_TILE_PROLOGUE_FROM(A.tile, A.n_tiles)
// (TILES, NTILES: where the tile list comes from -- the PkArgs in the kernarg segment, or pk_cycle's preloaded copies)
#define PK_TILE_PROLOGUE_FROM(TILES, NTILES)                                          \
  extern __shared__ double pk_lds[];                                                  \
  __shared__ double wint[PK_WAVES_PER_BLOCK * PK_NRED];                               \
  __shared__ double wgrad[PK_WAVES_PER_BLOCK * PK_NRED];                              \
  /* wave-uniform on purpose (readfirstlane): the tile record then comes through the scalar cache into */ \
  /* SGPRs and everything derived from it is scalar arithmetic */                      \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; \
  const int ti = blk * PK_WAVES_PER_BLOCK + wave;                                     \
  PkTile tl;                                                                          \
  if (ti < (NTILES)) {                                                                \
    tl = load_tile((TILES) + ti);                                                     \
    tl.pad = ti;                                                                      \
  } else {                                                                            \
    tl = load_tile((TILES) + ((NTILES) > 0 ? (NTILES)-1 : 0));                        \
    tl.nj = 0;                                                                        \
    tl.pad = -1;                                                                      \
  }                                                                                   \
  if (lane < PK_NRED) {                                                               \
    wint[wave * PK_NRED + lane] = 0.0;                                                \
    wgrad[wave * PK_NRED + lane] = 0.0;                                               \
  }

#define PK_IS_EDGE_BLOCK() (blockIdx.x == 0)

// mode 0: Jacobian, 1: Hessian, 2: auxiliary buffer, 3: compact Hessian, 4: compact Jacobian
template <class Gen>
__device__ __forceinline__ void edge_block(const PkArgs& A, int mode, bool with_g, const PkItem* __restrict__ items,
                                           int n_items) {
  extern __shared__ double pk_lds[];
  const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
  if (A.flags & 2) return;   // secondary shard: boundary nodes / system level belong to the primary
  if (with_g && threadIdx.x == PK_BLOCK - 1 && A.n_sys > 0) Gen::sys_constraints(sy, A.o_g);   // systembase.py:607-611
  const ItemPref ip = prefetch_item(A, items, n_items);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int li = wave; li < Gen::NLISTS; li += PK_WAVES_PER_BLOCK)
    if (lane == 0) {
      if (mode == 1) Gen::edge_hess(li, A, sy, pk_lds);
      else if (mode == 2) Gen::edge_aux(li, A, sy, pk_lds);
      else if (mode == 3) Gen::edge_hessc(li, A, sy, pk_lds);
      else if (mode == 4) Gen::edge_jacc(li, A, sy, pk_lds);
      else Gen::edge_jac(li, A, sy, pk_lds);
    }
  __syncthreads();
  scatter_items(A, items, n_items, ip, pk_lds, (mode == 1 || mode == 3) ? A.o_hess : (mode == 2 ? A.o_aux : A.o_jac));
}


template <class Gen>
__device__ __forceinline__ void kernel_int(const PkArgs& A) {
  PK_TILE_PROLOGUE(0);
#ifdef PK_BIG
  PkTile tl0;
  if (big_block(A.tile, A.n_tiles, blk, tl0)) Gen::bigi(tl0.phase, A, tl0, wint);
  else
#endif
  Gen::tile_int(tl.phase, A, tl, pk_lds, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane);
  publish_block_partials(A.partial, wint, blk);
}

template <class Gen>
__device__ __forceinline__ void kernel_g(const PkArgs& A) {
  if (PK_IS_EDGE_BLOCK()) {
    if (threadIdx.x == 0 && A.n_sys > 0 && !(A.flags & 2)) {
      const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
      Gen::sys_constraints(sy, A.o_g);
    }
    return;
  }
  PK_TILE_PROLOGUE(1);
  Gen::tile_g(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_G, wint, wgrad, lane);
}

template <class Gen>
__device__ __forceinline__ void kernel_grad(const PkArgs& A) {
  PK_TILE_PROLOGUE(0);
  Gen::tile_grad(tl.phase, A, tl, pk_lds, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane);
  publish_block_partials(A.partial2, wgrad, blk);
}

template <class Gen>
__device__ __forceinline__ void kernel_jac(const PkArgs& A) {
  if (PK_IS_EDGE_BLOCK()) return edge_block<Gen>(A, 0, false, A.items, A.n_items);
  PK_TILE_PROLOGUE(1);
  Gen::tile_jac(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_J, wint, wgrad, lane);
}

template <class Gen>
__device__ __forceinline__ void fin_body(const PkArgs& A);
template <class Gen>
__device__ __forceinline__ void fin_handoff(const PkArgs& A);

template <class Gen>
__device__ __forceinline__ void kernel_hess(const PkArgs& A) {
  if (PK_IS_EDGE_BLOCK()) return edge_block<Gen>(A, 1, false, A.items, A.n_items);
  if (blockIdx.x == 1) {
    // cycle mode: the reductions of the preceding pk_xall launch ride along in a workgroup of their own
    if (A.flags & (8 | 16)) fin_body<Gen>(A);
    return;
  }
  if constexpr (Gen::GROUPED) {
    // a model evaluated in groups: every pass of a tile block is a workgroup of its own, as in pk_cycle -- the grid has
    // H_NGMAX workgroups per block (md.hess_subs); the Hessian callback of such a model is one pass long, not all of them
    constexpr int NH = Gen::H_NGMAX;
    const int slot = pk::xcd_tile_block((int)blockIdx.x, 2, (int)gridDim.x);
    const int blk = slot / NH, pass = slot - blk * NH;
    PK_TILE_PROLOGUE_AT();
#ifdef PK_BIG
    PkTile tl0;
    if (big_block(A.tile, A.n_tiles, blk, tl0)) {      // (a workgroup-wide interval walks its passes itself)
      if (pass == 0) Gen::bigh(tl0.phase, A, tl0, PK_STAGE(A));
      return;
    }
#endif
    Gen::tile_hessg(tl.phase, pass, A, tl, PK_STAGE(A) + wave * Gen::LDS_H, lane);
  } else {
    PK_TILE_PROLOGUE(2);
#ifdef PK_BIG
    PkTile tl0;
    if (big_block(A.tile, A.n_tiles, blk, tl0)) return Gen::bigh(tl0.phase, A, tl0, PK_STAGE(A));
#endif
    Gen::tile_hess(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_H, wint, wgrad, lane);
  }
}

// (a model evaluated in groups, Gen::GROUPED: every pass of a tile block is a workgroup of its own, as in pk_hess -- the grid
//  has HC_NGMAX / JC_NGMAX workgroups per block: md.hessc_subs / md.jacc_subs)
template <class Gen>
__device__ __forceinline__ void kernel_hessc(const PkArgs& A) {
  if (PK_IS_EDGE_BLOCK()) return edge_block<Gen>(A, 3, false, A.items, A.n_items);
  if constexpr (Gen::GROUPED && Gen::HC_NGMAX > 1) {
    constexpr int NP = Gen::HC_NGMAX;
    const int slot = pk::xcd_tile_block((int)blockIdx.x, 1, (int)gridDim.x);
    const int blk = slot / NP, pass = slot - blk * NP;
    PK_TILE_PROLOGUE_AT();
    Gen::tile_hesscp(tl.phase, pass, NP, A, tl, PK_STAGE(A) + wave * Gen::LDS_G, lane);
  } else {
    PK_TILE_PROLOGUE(1);
    Gen::tile_hessc(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_G, wint, wgrad, lane);
  }
}

template <class Gen>
__device__ __forceinline__ void kernel_jacc(const PkArgs& A) {
  if (PK_IS_EDGE_BLOCK()) return edge_block<Gen>(A, 4, false, A.items, A.n_items);
  if constexpr (Gen::GROUPED && Gen::JC_NGMAX > 1) {
    constexpr int NP = Gen::JC_NGMAX;
    const int slot = pk::xcd_tile_block((int)blockIdx.x, 1, (int)gridDim.x);
    const int blk = slot / NP, pass = slot - blk * NP;
    PK_TILE_PROLOGUE_AT();
#ifdef PK_BIG
    PkTile tl0;
    if (big_block(A.tile, A.n_tiles, blk, tl0)) {      // (a workgroup-wide interval walks its passes itself)
      if (pass == 0) Gen::bigjc(tl0.phase, A, tl0, PK_STAGE(A));
      return;
    }
#endif
    Gen::tile_jaccp(tl.phase, pass, NP, A, tl, PK_STAGE(A) + wave * Gen::LDS_JC, lane);
  } else {
    PK_TILE_PROLOGUE(1);
#ifdef PK_BIG
    PkTile tl0;
    if (big_block(A.tile, A.n_tiles, blk, tl0)) return Gen::bigjc(tl0.phase, A, tl0, PK_STAGE(A));
#endif
    Gen::tile_jacc(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_JC, wint, wgrad, lane);
  }
}

template <class Gen>
__device__ __forceinline__ void kernel_aux(const PkArgs& A) {
  if (PK_IS_EDGE_BLOCK()) return edge_block<Gen>(A, 2, false, A.items, A.n_items);
  PK_TILE_PROLOGUE(1);
#ifdef PK_BIG
  PkTile tl0;
  if (big_block(A.tile, A.n_tiles, blk, tl0)) return Gen::biga(tl0.phase, A, tl0);
#endif
  Gen::tile_aux(tl.phase, A, tl, pk_lds, wint, wgrad, lane);
}


// ---- mesh error estimation (reference: phasebase.py:1339-1372  _error_estimation_data_continuous) -----
// One wavefront per GROUP of consecutive mesh intervals of one phase and one K: W = K + 1 lanes per interval (the
// interval's augmented nodes), floor(64 / W) intervals per wave -- at K = 8 seven intervals share a wave instead of one
// wave of 9 live lanes each.  Per interval: its K (+1) state values and K control values are staged in LDS, interpolated
// to the K + 1 nodes of the augmented rule (lane = augmented node), the dynamics are evaluated there, and the two sides of
// the integral-form collocation equation on the augmented rule are written out:  T_aug x  and  dt (I_aug d/2) f.  The
// host compares them per interval (np.allclose semantics) and runs the hp-refinement logic (pockit_amd/refine.py).
// LDS per wave: (2 NX + NU) x 64 doubles, private to the wave (no workgroup barrier between the three steps).
// ---- the same for a WIDE model (P::WIDE), in passes over chunks of states (see dyn_pass): a pass interpolates the node
// arguments ITS dynamics functions read straight from x (the K + 1 values of an interval are shared by its lanes: cache hits;
// what the pass does not read is dead code), stages the chunk's dynamics values -- D_cn rows instead of 2 n_x + n_u -- and writes
// the chunk's rows of both sides.  Same sums in the same order as interval_err.
template <class P>
__device__ __forceinline__ double err_x(const PkPhase& ph, const double* __restrict__ xp, const double* s, int i, int slot,
                                        int back_slot) {
  double v = xp[i * ph.state_len + slot];
  if (slot == 0) v = P::front_value(i, v, s);
  if (slot == back_slot) v = P::back_value(i, v, s);
  return v;
}
// the node arguments chunk C's dynamics functions read (codegen.py: P::D_na(C) of them, argument P::D_ar(C, k) -- states and
// controls only; time and static parameters are always set), interpolated to augmented node a.  Short constant-index loops:
// a loop over ALL states around a model-sized switch was left rolled by the compiler, and the argument array went to scratch.
template <class P, int C>
__device__ __forceinline__ void err_args(const PkArgs& A, const PkPhase& ph, const PkErrIv& iv, const double* s, double dt,
                                         double mt, int a, double* arg) {
  constexpr int NA = P::D_na(C);
  const int K = iv.K, ncx = K + 1 - P::SCHEME, na = K + 1;
  const double* __restrict__ Vx = A.errdb + iv.tab_off;
  const double* __restrict__ Vu = Vx + na * ncx;
  const double* __restrict__ xp = A.x + ph.x_off;
  const double* __restrict__ up = xp + P::NX * ph.state_len;
  const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
  double acc[NA > 0 ? NA : 1];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.0;
  for (int c = 0; c < ncx; ++c) {
    const double v = Vx[a * ncx + c];
#pragma unroll
    for (int k = 0; k < NA; ++k)
      if (P::D_ar(C, k) < P::NX) acc[k] += v * err_x<P>(ph, xp, s, P::D_ar(C, k), iv.lm + c, back_slot);
  }
  for (int c = 0; c < K; ++c) {
    const double v = Vu[a * K + c];
#pragma unroll
    for (int k = 0; k < NA; ++k)
      if (P::D_ar(C, k) >= P::NX) acc[k] += v * up[(P::D_ar(C, k) - P::NX) * ph.L_m + iv.lm + c];
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) arg[P::D_ar(C, k)] = acc[k];
  const double tau = A.errdb[iv.tau_off + a];
  arg[P::NX + P::NU] = (tau - 0.5) * dt + mt;
#pragma unroll
  for (int i = 0; i < P::NS; ++i) arg[P::NX + P::NU + 1 + i] = s[i];
}
// rows of both sides of one chunk of states at augmented row a (fs: the chunk's dynamics values, row length `ld`)
template <class P, int C>
__device__ __forceinline__ void err_rows(const PkArgs& A, const PkPhase& ph, const PkErrIv& iv, const double* s, double dt,
                                         int a, const double* __restrict__ fs, int ld) {
  constexpr int I0 = P::D_c0(C), CN = P::D_cn(C);
  const int K = iv.K, ncx = K + 1 - P::SCHEME, na = K + 1, nr = K + 1 - P::SCHEME;
  const double* __restrict__ Tm = A.errdb + iv.tab_off + na * ncx + na * K;
  const double* __restrict__ Im = Tm + nr * ncx;
  const double* __restrict__ xp = A.x + ph.x_off;
  const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
  double tx[CN], itf[CN];
#pragma unroll
  for (int e = 0; e < CN; ++e) tx[e] = itf[e] = 0.0;
  for (int c = 0; c < ncx; ++c) {
    const double v = Tm[a * ncx + c];
#pragma unroll
    for (int e = 0; e < CN; ++e) tx[e] += v * err_x<P>(ph, xp, s, I0 + e, iv.lm + c, back_slot);
  }
  for (int c = 0; c < na; ++c) {
    const double v = Im[a * na + c] * iv.width * 0.5;
#pragma unroll
    for (int e = 0; e < CN; ++e) itf[e] += v * fs[e * ld + c];
  }
#pragma unroll
  for (int e = 0; e < CN; ++e) {
    const int64_t pos = iv.out_off + (int64_t)(I0 + e) * iv.rows + iv.row0 + a;
    put(&A.o_errT[pos], tx[e]);
    put(&A.o_errI[pos], itf[e] * dt);
  }
}
template <class P, int C = 0>
__device__ __forceinline__ void err_passes_wide(const PkArgs& A, const PkPhase& ph, const PkErrIv& iv, const double* s,
                                                double dt, double mt, bool valid, int a, double* __restrict__ fs) {
  if constexpr (C < P::D_NG) {
    constexpr int CN = P::D_cn(C);
    if (valid) {
      double arg[P::NARG], o[CN];
      err_args<P, C>(A, ph, iv, s, dt, mt, a, arg);
      P::mid_dyn_g(Grp<C>{}, arg, o);
#pragma unroll
      for (int e = 0; e < CN; ++e) fs[e * PK_WAVE + a] = o[e];
    }
    wave_lds_sync();       // (a wave stages for itself only)
    if (valid && a < iv.K + 1 - P::SCHEME) err_rows<P, C>(A, ph, iv, s, dt, a, fs, PK_WAVE);
    wave_lds_sync();       // (the next pass overwrites the rows)
    err_passes_wide<P, C + 1>(A, ph, iv, s, dt, mt, valid, a, fs);
  }
}
template <class P, int C = 0>
__device__ __forceinline__ void err_big_passes_wide(const PkArgs& A, const PkPhase& ph, const PkErrIv& iv, const double* s,
                                                    double dt, double mt, double* __restrict__ fs, int ld) {
  if constexpr (C < P::D_NG) {
    constexpr int CN = P::D_cn(C);
    const int na = iv.K + 1, nr = iv.K + 1 - P::SCHEME;
    for (int a = (int)threadIdx.x; a < na; a += PK_BLOCK) {
      double arg[P::NARG], o[CN];
      err_args<P, C>(A, ph, iv, s, dt, mt, a, arg);
      P::mid_dyn_g(Grp<C>{}, arg, o);
#pragma unroll
      for (int e = 0; e < CN; ++e) fs[e * ld + a] = o[e];
    }
    __syncthreads();
    for (int a = (int)threadIdx.x; a < nr; a += PK_BLOCK) err_rows<P, C>(A, ph, iv, s, dt, a, fs, ld);
    __syncthreads();
    err_big_passes_wide<P, C + 1>(A, ph, iv, s, dt, mt, fs, ld);
  }
}

template <class P>
__device__ __forceinline__ void interval_err(const PkArgs& A, int first, int cnt, double* __restrict__ lds, int lane) {
  const PkPhase& ph = A.ph[P::INDEX];
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  if constexpr (P::WIDE) {
    const int W0 = A.erriv[first].K + 1, jj0 = lane / W0, a0 = lane - jj0 * W0;
    const bool valid0 = jj0 < cnt;
    const PkErrIv iv0 = A.erriv[first + (valid0 ? jj0 : 0)];
    return err_passes_wide<P>(A, ph, iv0, s, dt, mt, valid0, a0, lds + jj0 * W0);
  }
  const int K = A.erriv[first].K;                       // wave-uniform (all intervals of a group share phase and K)
  const int W = K + 1;
  const int jj = lane / W, a = lane - jj * W;           // interval of the group, augmented node / row / slot within it
  const bool valid = jj < cnt;
  const PkErrIv iv = A.erriv[first + (valid ? jj : 0)];
  const int ncx = K + 1 - P::SCHEME, na = K + 1, nr = K + 1 - P::SCHEME;
  const double* __restrict__ Vx = A.errdb + iv.tab_off;
  const double* __restrict__ Vu = Vx + na * ncx;
  const double* __restrict__ Tm = Vu + na * K;
  const double* __restrict__ Im = Tm + nr * ncx;
  const double* __restrict__ xp = A.x + ph.x_off;
  const double* __restrict__ up = xp + P::NX * ph.state_len;
  const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
  double* __restrict__ xs = lds + jj * W;               // this interval's columns of the wave's [row][64] arrays
  double* __restrict__ us = lds + P::NX * PK_WAVE + jj * W;
  double* __restrict__ fs = lds + (P::NX + P::NU) * PK_WAVE + jj * W;
  if (valid && a < ncx) {
    const int slot = iv.lm + a;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) {
      double v = xp[i * ph.state_len + slot];
      if (slot == 0) v = P::front_value(i, v, s);
      if (slot == back_slot) v = P::back_value(i, v, s);
      xs[i * PK_WAVE + a] = v;
    }
  }
  if (valid && a < K) {
#pragma unroll
    for (int i = 0; i < P::NU; ++i) us[i * PK_WAVE + a] = up[i * ph.L_m + iv.lm + a];
  }
  wave_lds_sync();       // (a wave stages for itself only)
  if (valid) {           // (a < na always)
    double arg[P::NARG], o[P::G_NOUT];
#pragma unroll
    for (int i = 0; i < P::NX; ++i) arg[i] = 0.0;
#pragma unroll
    for (int i = 0; i < P::NU; ++i) arg[P::NX + i] = 0.0;
    for (int c = 0; c < ncx; ++c) {
      const double v = Vx[a * ncx + c];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) arg[i] += v * xs[i * PK_WAVE + c];
    }
    for (int c = 0; c < K; ++c) {
      const double v = Vu[a * K + c];
#pragma unroll
      for (int i = 0; i < P::NU; ++i) arg[P::NX + i] += v * us[i * PK_WAVE + c];
    }
    const double tau = A.errdb[iv.tau_off + a];
    arg[P::NX + P::NU] = (tau - 0.5) * dt + mt;
#pragma unroll
    for (int i = 0; i < P::NS; ++i) arg[P::NX + P::NU + 1 + i] = s[i];
    P::mid_g(arg, o);
#pragma unroll
    for (int i = 0; i < P::NX; ++i) fs[i * PK_WAVE + a] = o[i];
  }
  wave_lds_sync();       // (a wave stages for itself only)
  if (valid && a < nr) {
    double tx[P::NX], itf[P::NX];
#pragma unroll
    for (int i = 0; i < P::NX; ++i) tx[i] = itf[i] = 0.0;
    for (int c = 0; c < ncx; ++c) {
      const double v = Tm[a * ncx + c];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) tx[i] += v * xs[i * PK_WAVE + c];
    }
    for (int c = 0; c < na; ++c) {
      const double v = Im[a * na + c] * iv.width * 0.5;
#pragma unroll
      for (int i = 0; i < P::NX; ++i) itf[i] += v * fs[i * PK_WAVE + c];
    }
#pragma unroll
    for (int i = 0; i < P::NX; ++i) {
      const int64_t pos = iv.out_off + (int64_t)i * iv.rows + iv.row0 + a;
      put(&A.o_errT[pos], tx[i]);
      put(&A.o_errI[pos], itf[i] * dt);
    }
  }
}

// The same for ONE interval with more augmented nodes than a wave has lanes (K >= 64): all 256 threads of the
// workgroup walk the K + 1 nodes / rows; the staged rows are PK_ERR_BIG_LDS_ROW doubles long in LDS, or -- more than that
// many augmented nodes -- PkArgs.big_row doubles in the interval's slot of the global staging buffer.  Same arithmetic,
// same operation order per node as interval_err.
#define PK_ERR_BIG_LDS_ROW 264
template <class P>
__device__ __forceinline__ void interval_err_big(const PkArgs& A, int first, double* __restrict__ lds0) {
  const PkPhase& ph = A.ph[P::INDEX];
  double s[PK_NS], dt, mt;
  phase_scalars<P>(A, ph, s, dt, mt);
  const PkErrIv iv = A.erriv[first];
  const int K = iv.K, t = (int)threadIdx.x;
  const int ncx = K + 1 - P::SCHEME, na = K + 1, nr = K + 1 - P::SCHEME;
  const bool in_lds = na <= PK_ERR_BIG_LDS_ROW;
  const int PK_ERR_BIG_ROW = in_lds ? PK_ERR_BIG_LDS_ROW : A.big_row;
  double* __restrict__ lds = in_lds ? lds0 : A.big_stage + (size_t)iv.stage * (size_t)A.big_slot;
  if constexpr (P::WIDE) return err_big_passes_wide<P>(A, ph, iv, s, dt, mt, lds, PK_ERR_BIG_ROW);
  const double* __restrict__ Vx = A.errdb + iv.tab_off;
  const double* __restrict__ Vu = Vx + na * ncx;
  const double* __restrict__ Tm = Vu + na * K;
  const double* __restrict__ Im = Tm + nr * ncx;
  const double* __restrict__ xp = A.x + ph.x_off;
  const double* __restrict__ up = xp + P::NX * ph.state_len;
  const int back_slot = P::SCHEME ? ph.L_m - 1 : ph.L_m;
  double* __restrict__ xs = lds;
  double* __restrict__ us = lds + P::NX * PK_ERR_BIG_ROW;
  double* __restrict__ fs = lds + (P::NX + P::NU) * PK_ERR_BIG_ROW;
  for (int a = t; a < ncx; a += PK_BLOCK) {
    const int slot = iv.lm + a;
#pragma unroll
    for (int i = 0; i < P::NX; ++i) {
      double v = xp[i * ph.state_len + slot];
      if (slot == 0) v = P::front_value(i, v, s);
      if (slot == back_slot) v = P::back_value(i, v, s);
      xs[i * PK_ERR_BIG_ROW + a] = v;
    }
  }
  for (int a = t; a < K; a += PK_BLOCK) {
#pragma unroll
    for (int i = 0; i < P::NU; ++i) us[i * PK_ERR_BIG_ROW + a] = up[i * ph.L_m + iv.lm + a];
  }
  __syncthreads();
#if PK_BIG_MFMA
  // the four products of the interval -- interpolation to the K + 1 augmented nodes, V_x x and V_u u, and the two sides of
  // the collocation equation, T_aug x and (I_aug d/2) f -- on the fp64 matrix cores; the interpolated values pass through
  // LDS ([variable][node], behind the staged rows) to the thread that evaluates the dynamics at the node
  const int wave = t >> 6, lane = t & 63;
  double* __restrict__ xa = lds + (2 * P::NX + P::NU) * PK_ERR_BIG_ROW;      // [NX + NU][PK_ERR_BIG_ROW]
  for (int rb = wave; rb < (na + 15) / 16; rb += PK_WAVES_PER_BLOCK) {
    for (int cb = 0; cb < (P::NX + 15) / 16; ++cb)
      mfma_rows(Vx, ncx, na, ncx, 1.0, xs, PK_ERR_BIG_ROW, P::NX, rb, cb, lane,
                [&](int a, int i, double v) { xa[i * PK_ERR_BIG_ROW + a] = v; });
    for (int cb = 0; cb < (P::NU + 15) / 16; ++cb)
      mfma_rows(Vu, K, na, K, 1.0, us, PK_ERR_BIG_ROW, P::NU, rb, cb, lane,
                [&](int a, int i, double v) { xa[(P::NX + i) * PK_ERR_BIG_ROW + a] = v; });
  }
  __syncthreads();
  for (int a = t; a < na; a += PK_BLOCK) {
    double arg[P::NARG], o[P::G_NOUT];
#pragma unroll
    for (int i = 0; i < P::NX + P::NU; ++i) arg[i] = xa[i * PK_ERR_BIG_ROW + a];
    const double tau = A.errdb[iv.tau_off + a];
    arg[P::NX + P::NU] = (tau - 0.5) * dt + mt;
#pragma unroll
    for (int i = 0; i < P::NS; ++i) arg[P::NX + P::NU + 1 + i] = s[i];
    P::mid_g(arg, o);
#pragma unroll
    for (int i = 0; i < P::NX; ++i) fs[i * PK_ERR_BIG_ROW + a] = o[i];
  }
  __syncthreads();
  for (int rb = wave; rb < (nr + 15) / 16; rb += PK_WAVES_PER_BLOCK)
    for (int cb = 0; cb < (P::NX + 15) / 16; ++cb) {
      mfma_rows(Tm, ncx, nr, ncx, 1.0, xs, PK_ERR_BIG_ROW, P::NX, rb, cb, lane, [&](int a, int i, double v) {
        put(&A.o_errT[iv.out_off + (int64_t)i * iv.rows + iv.row0 + a], v);
      });
      mfma_rows(Im, na, nr, na, iv.width * 0.5, fs, PK_ERR_BIG_ROW, P::NX, rb, cb, lane, [&](int a, int i, double v) {
        put(&A.o_errI[iv.out_off + (int64_t)i * iv.rows + iv.row0 + a], v * dt);
      });
    }
}
#else
  for (int a = t; a < na; a += PK_BLOCK) {
    double arg[P::NARG], o[P::G_NOUT];
#pragma unroll
    for (int i = 0; i < P::NX; ++i) arg[i] = 0.0;
#pragma unroll
    for (int i = 0; i < P::NU; ++i) arg[P::NX + i] = 0.0;
    for (int c = 0; c < ncx; ++c) {
      const double v = Vx[a * ncx + c];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) arg[i] += v * xs[i * PK_ERR_BIG_ROW + c];
    }
    for (int c = 0; c < K; ++c) {
      const double v = Vu[a * K + c];
#pragma unroll
      for (int i = 0; i < P::NU; ++i) arg[P::NX + i] += v * us[i * PK_ERR_BIG_ROW + c];
    }
    const double tau = A.errdb[iv.tau_off + a];
    arg[P::NX + P::NU] = (tau - 0.5) * dt + mt;
#pragma unroll
    for (int i = 0; i < P::NS; ++i) arg[P::NX + P::NU + 1 + i] = s[i];
    P::mid_g(arg, o);
#pragma unroll
    for (int i = 0; i < P::NX; ++i) fs[i * PK_ERR_BIG_ROW + a] = o[i];
  }
  __syncthreads();
  for (int a = t; a < nr; a += PK_BLOCK) {
    double tx[P::NX], itf[P::NX];
#pragma unroll
    for (int i = 0; i < P::NX; ++i) tx[i] = itf[i] = 0.0;
    for (int c = 0; c < ncx; ++c) {
      const double v = Tm[a * ncx + c];
#pragma unroll
      for (int i = 0; i < P::NX; ++i) tx[i] += v * xs[i * PK_ERR_BIG_ROW + c];
    }
    for (int c = 0; c < na; ++c) {
      const double v = Im[a * na + c] * iv.width * 0.5;
#pragma unroll
      for (int i = 0; i < P::NX; ++i) itf[i] += v * fs[i * PK_ERR_BIG_ROW + c];
    }
#pragma unroll
    for (int i = 0; i < P::NX; ++i) {
      const int64_t pos = iv.out_off + (int64_t)i * iv.rows + iv.row0 + a;
      put(&A.o_errT[pos], tx[i]);
      put(&A.o_errI[pos], itf[i] * dt);
    }
  }
}
#endif

template <class Gen>
__device__ __forceinline__ void kernel_err(const PkArgs& A) {
  extern __shared__ double pk_lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  {   // an interval with more augmented nodes than a wave has lanes is the first group of a block of its own
    const int g0 = (int)blockIdx.x * PK_WAVES_PER_BLOCK;
    if (g0 < A.n_erriv && A.errgrp[2 * g0 + 1] == 1) {
      const int first0 = A.errgrp[2 * g0];
      if (A.erriv[first0].K + 1 > PK_WAVE) return Gen::interval_err_big(A.erriv[first0].phase, A, first0, pk_lds);
    }
  }
  const int g = (int)blockIdx.x * PK_WAVES_PER_BLOCK + wave;
  if (g >= A.n_erriv) return;
  // a workgroup never mixes phases (the host pads every phase to a multiple of 4 groups with count 0)
  const int first = A.errgrp[2 * g], cnt = A.errgrp[2 * g + 1];
  if (cnt <= 0) return;
  Gen::interval_err(A.erriv[first].phase, A, first, cnt, pk_lds + wave * Gen::LDS_E, lane);
}

// Triplet list -> CSR values on device (hand J / H to a GPU linear solver without a
// PCIe round trip).  The (row, col) sort, the duplicate runs and the permutation are computed once per mesh
// on the host (pockit_amd/csr.py); duplicates are summed in triplet order (deterministic).
__device__ __forceinline__ void kernel_csr(const PkArgs& A) {
  if (A.csr_seg == nullptr) {          // no (row, col) repeats: a pure permutation
    const int stride = (int)gridDim.x * PK_BLOCK;
    for (int p = (int)blockIdx.x * PK_BLOCK + (int)threadIdx.x; p < A.n_csr; p += stride) A.csr_out[p] = A.csr_in[A.csr_perm[p]];
    return;
  }
  // Repeated entries: the runs of the permutation are stored per slice of 256 consecutive CSR entries, transposed --
  // csr_perm[off[b] + k * 256 + t] is the k-th triplet of entry 256 b + t (-1 beyond its count), off = csr_seg -- so that
  // the 256 threads of a slice read the index array and (neighbouring entries being neighbouring nodes) mostly the triplets
  // coalesced, four independent gathers in flight per thread; sums in triplet order, as before (bit-identical).  The first
  // form -- one thread walking perm[seg[p] .. seg[p + 1]) -- took 53 us for the 7.16 M Hessian triplets of 40k nodes.
  const int nblk = (A.n_csr + PK_BLOCK - 1) / PK_BLOCK;
  for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int lo = A.csr_seg[b], width = (A.csr_seg[b + 1] - lo) / PK_BLOCK, p = b * PK_BLOCK + (int)threadIdx.x;
    const int32_t* __restrict__ idx = A.csr_perm + lo + threadIdx.x;
    double acc = 0.0;
    int k = 0;
    for (; k + 4 <= width; k += 4) {
      const int i0 = idx[(k + 0) * PK_BLOCK], i1 = idx[(k + 1) * PK_BLOCK], i2 = idx[(k + 2) * PK_BLOCK], i3 = idx[(k + 3) * PK_BLOCK];
      const double v0 = i0 >= 0 ? A.csr_in[i0] : 0.0, v1 = i1 >= 0 ? A.csr_in[i1] : 0.0;
      const double v2 = i2 >= 0 ? A.csr_in[i2] : 0.0, v3 = i3 >= 0 ? A.csr_in[i3] : 0.0;
      if (i0 >= 0) acc += v0;
      if (i1 >= 0) acc += v1;
      if (i2 >= 0) acc += v2;
      if (i3 >= 0) acc += v3;
    }
    for (; k < width; ++k) {
      const int i0 = idx[k * PK_BLOCK];
      if (i0 >= 0) acc += A.csr_in[i0];
    }
    if (p < A.n_csr) A.csr_out[p] = acc;
  }
}

// One workgroup per outer-product block (generic, table driven; O(n^2) outputs exist only for
// objectives / system constraints that are nonlinear in the integrals -- small problems in practice).
__device__ __forceinline__ void kernel_outer(const PkArgs& A) {
  __shared__ double red[PK_WAVES_PER_BLOCK];
  __shared__ double sums[2];
  for (int b = blockIdx.x; b < A.n_outer; b += gridDim.x) {
    const PkOuter d = A.outer[b];
    const double* __restrict__ GA = A.o_aux + d.offA;
    const double* __restrict__ GB = A.o_aux + d.offB;
    const double m = A.o_aux[d.offM];
    double* __restrict__ out = A.o_hess + d.pos;
    if (!(d.flags & 1)) {
      const int tot = d.lenA * d.lenB;
      for (int t = threadIdx.x; t < tot; t += PK_BLOCK) {
        const int i = t / d.lenB, j = t - i * d.lenB;
        out[t] = GA[i] * GB[j] * m;
      }
      continue;
    }
    for (int which = 0; which < 2; ++which) {          // collapsed runs: deterministic block sums
      const bool need = which == 0 ? (d.flags & 2) : (d.flags & 4);
      if (!need) continue;                              // (uniform across the workgroup)
      const double* __restrict__ G = which == 0 ? GA : GB;
      const int len = which == 0 ? d.lenA : d.lenB;
      double v = 0.0;
      for (int t = threadIdx.x; t < len; t += PK_BLOCK) v += G[t];
      v = wave_sum(v);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
      __syncthreads();
      if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < PK_WAVES_PER_BLOCK; ++w) tot += red[w];
        sums[which] = tot;
      }
    }
    __syncthreads();
    const int n = (d.flags & 2) ? 1 : d.lenA;
    const int ntri = n * (n + 1) / 2;
    for (int t = threadIdx.x; t < ntri; t += PK_BLOCK) {
      int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= t) ++i;
      while (i * (i + 1) / 2 > t) --i;
      const int j = t - i * (i + 1) / 2;
      const double ai = (d.flags & 2) ? sums[0] : GA[i], aj = (d.flags & 2) ? sums[0] : GA[j];
      const double bi = (d.flags & 4) ? sums[1] : GB[i], bj = (d.flags & 4) ? sums[1] : GB[j];
      out[t] = ai * bj * m;
      if (d.flags & 8) out[ntri + t] = bi * aj * m;
    }
    __syncthreads();
  }
}

template <class Gen>
__device__ __forceinline__ void kernel_xall(const PkArgs& A) {
  if (PK_DIAG(1024)) return;                                   // diagnostic switches (POCKIT_AMD_DEBUG_FLAGS)
  if (PK_IS_EDGE_BLOCK() && PK_DIAG(2048)) return;
  if (PK_IS_EDGE_BLOCK()) return edge_block<Gen>(A, 0, true, A.items, A.n_items);
  if (A.flags & 32) {   // split launch: workgroups 2b (Jacobian) and 2b + 1 (values) share tile block b
    const int slot = pk::xcd_tile_block((int)blockIdx.x, 1, (int)gridDim.x);
    const int blk = slot >> 1;
    PK_TILE_PROLOGUE_AT();
#ifdef PK_BIG
    PkTile tl0;
    const bool big = big_block(A.tile, A.n_tiles, blk, tl0);
#else
    constexpr bool big = false;
#endif
    if (!(slot & 1)) {
#ifdef PK_BIG
      if (big) return Gen::bigx2(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, -1);
#endif
      Gen::tile_xall2(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, -1);
      return;
    }
#ifdef PK_BIG
    if (big) Gen::bigx1(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, -1);
#endif
    if (!big) Gen::tile_xall1(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, -1);
    publish_block_partials(A.partial, wint, blk);
    publish_block_partials(A.partial2, wgrad, blk);
    return;
  }
  PK_TILE_PROLOGUE(1);
#ifdef PK_BIG
  PkTile tl0;
  if (big_block(A.tile, A.n_tiles, blk, tl0)) Gen::bigx0(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, -1);
  else
#endif
  Gen::tile_xall(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, -1);
  publish_block_partials(A.partial, wint, blk);
  publish_block_partials(A.partial2, wgrad, blk);
}

// ============================================================================================
// pk_cycle: the whole NLP-callback cycle (f, grad f, g, J, H on one x, lambda, sigma) in ONE launch.
// A 12k-node cycle moves 15 MB (2 us at HBM peak) while every launch costs a fixed 2-5 us and the kernels of a
// stream do not overlap, so the cycle is the union of pk_xall's and pk_hess's workgroups instead of their sequence:
//   workgroup 0  boundary nodes / system level of g and J          (edge_block mode 0)
//   workgroup 1  boundary nodes / system level of H                (edge_block mode 1)
//   workgroup 2  the sums over all nodes (integrals, f, shared gradient slots), fed by the tile workgroups of
//                this same launch through the hand-off slots (fin_body<HANDOFF>)
//   the rest     tile workgroups of four waves = four consecutive tiles, ONE role per workgroup; the workgroups of a
//                tile block follow each other ([Jacobian, values, Hessian], or [x-part, Hessian] when the x-part is
//                not split) so that every XCD gets a contiguous range of tiles of every output array
// Every wave runs exactly what it runs in pk_xall / pk_hess, they just run at the same time.
// ============================================================================================
// pre: the four values a tile wave needs before it can ask for its tile record (tile list, its length, the launch flags,
// the grid size), passed as leading scalar kernel arguments: with kernarg preloading (gfx940+,
// -amdgpu-kernarg-preload-count) they arrive in SGPRs with the wave, so the record load is not queued behind a first
// round trip to the kernarg segment.
// COMPACT = false: pk_cycle, the reference layouts only.  COMPACT = true: pk_cyclec, the same launch with the roles of the
// compact layouts selectable by the flags -- a kernel of its own, because the compact Hessian's per-node code raises the
// register count of whatever kernel holds it (humanoid: 166 -> 177 VGPRs = 3 -> 2 waves per SIMD, +3.5 % on the 40k-node cycle
// when it sat in pk_cycle itself; profiles/r04_c_ab_compact_roles.txt).
template <class Gen, bool COMPACT>
__device__ __forceinline__ void kernel_cycle(const PkTile* pre_tile, int pre_n_tiles, int pre_flags, int pre_grid,
                                             const PkArgs& A) {
  // The waves read their PkArgs fields where they use them (PK_DEFINE_KERNELS).  The kernarg segment is new with every
  // launch: every 64-byte line of it is a scalar-cache miss the first time a wave of the CU asks, and read one after
  // the other those misses would follow each other down the wave's serial chain.  One dword of every line is requested
  // here, all at once; the requests are collected together with the tile-record load (one round trip for all of them).
  // Without this the lazy form loses 3-6 % on the one-phase benchmarks (profiles/r02_ka_ab.txt).
  constexpr int ka_bytes = (int)offsetof(PkArgs, ph) + PK_NPHASE_DIM * (int)sizeof(PkPhase);
  constexpr int ka_lines = (ka_bytes + 63) / 64 + 1;     // (+1: the PkArgs do not start on a line boundary)
  int ka_v[ka_lines];
  {
    const int PK_CONST_AS* ka = (const int PK_CONST_AS*)(uintptr_t)&A;
#pragma unroll
    for (int o = 0; o < ka_lines; ++o) ka_v[o] = ka[(o * 64 < ka_bytes - 4 ? o * 64 : ka_bytes - 4) / 4];
  }
#define PK_KA_COLLECT()                                                        \
  {                                                                            \
    int ka_acc = 0;                                                            \
    _Pragma("unroll") for (int o = 0; o < ka_lines; ++o) ka_acc |= ka_v[o];    \
    asm volatile("" ::"s"(ka_acc));                                            \
  }
  if (blockIdx.x < 2 && PK_DIAG(2048)) return;     // diagnostic switches (POCKIT_AMD_DEBUG_FLAGS): no boundary work,
  if (blockIdx.x == 2 && PK_DIAG(65536)) return;   // no finalize workgroup (the hand-off slots then stay filled)
  if (blockIdx.x < 3) {
    PK_KA_COLLECT();
    const int rec = A.n_tiles * 3 + (int)blockIdx.x;
    (void)rec;
    PK_MARK_AT(rec, 0);
#ifdef PK_TRACE
    if (A.trace != nullptr && threadIdx.x == 0) A.trace[(size_t)rec * 16 + 14] = __builtin_readcyclecounter();
#endif
    // (flags bit 9 / bit 8: the launch serves the compact Jacobian / Hessian layout -- the host passes that layout's items)
    if (blockIdx.x == 0) edge_block<Gen>(A, (COMPACT && (pre_flags & 512)) ? 4 : 0, true, A.items, A.n_items);
    else if (blockIdx.x == 1) {
      if (!(pre_flags & 128)) edge_block<Gen>(A, (COMPACT && (pre_flags & 256)) ? 3 : 1, false, A.items2, A.n_items2);
    }
    else fin_handoff<Gen>(A);
#ifdef PK_TRACE
    __builtin_amdgcn_s_waitcnt(0);
#endif
    PK_MARK_AT(rec, 9);
#ifdef PK_TRACE
    if (A.trace != nullptr && threadIdx.x == 0) A.trace[(size_t)rec * 16 + 15] = __builtin_readcyclecounter();
#endif
    return;
  }
#ifdef PK_TRACE
  const unsigned long long pk_t_entry = __builtin_amdgcn_s_memrealtime();      // (before the tile record is asked for: mark 14)
#endif
  const int slot = pk::xcd_tile_block((int)blockIdx.x, 3, pre_grid);
  if constexpr (Gen::GROUPED) {
    // A model evaluated in groups: the workgroups of a tile block are [Jacobian pass 0 .. NJ-1 | values | dynamics pass 0 ..
    // ND-1 (a WIDE model's chunks of states, tile_dyn_pick) | Hessian pass 0 .. NH-1]  (NJ / ND / NH: the most passes any
    // phase has; a phase with fewer leaves the surplus workgroups at once).
    constexpr int NJ = Gen::J_NGMAX, ND = Gen::D_NGMAX, NH = Gen::H_NGMAX, PER = 1 + NJ + ND + NH;
    const int blk = slot / PER, sub = slot - blk * PER;
    if ((pre_flags & 128) && sub > NJ + ND) return;       // (x-only launch: no Hessian)
    const bool ch = COMPACT && (pre_flags & 256) != 0, cj = COMPACT && (pre_flags & 512) != 0;
    // (the compact roles are pass-parallel too: the role's workgroups share its passes round robin -- tile_jacc_part,
    //  tile_hessc_part -- a compact layout may have more or fewer passes than the reference layout's role has workgroups)
    PK_TILE_PROLOGUE_FROM(pre_tile, pre_n_tiles);
    PK_KA_COLLECT();
#ifdef PK_BIG
    PkTile tl0;
    if (big_block(pre_tile, pre_n_tiles, blk, tl0)) {     // (a workgroup-wide interval: its roles walk their passes themselves)
      if (sub == 0) Gen::bigx2(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, -1);
      else if (sub == NJ) Gen::bigx1(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, blk);
      else if (sub == NJ + ND + 1) Gen::bigh(tl0.phase, A, tl0, PK_STAGE(A));
      return;
    }
#endif
    if (sub < NJ) {
      if (cj) Gen::tile_jaccp(tl.phase, sub, NJ, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, lane);
      else Gen::tile_jacg(tl.phase, sub, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, lane);
    } else if (sub == NJ) {
      if (cj) Gen::tile_xall1cp(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, blk);
      else Gen::tile_xall1p(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, blk);
    } else if (sub <= NJ + ND) {
      Gen::tile_dyn(tl.phase, sub - NJ - 1, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, lane);
    } else {
      if (ch) Gen::tile_hesscp(tl.phase, sub - NJ - ND - 1, NH, A, tl, PK_STAGE(A) + wave * Gen::LDS_G, lane);
      else Gen::tile_hessg(tl.phase, sub - NJ - ND - 1, A, tl, PK_STAGE(A) + wave * Gen::LDS_H, lane);
    }
    return;
  }
  const bool split = (pre_flags & 32) != 0;
  const int blk = split ? slot / 3 : slot >> 1;       // the tile block; its workgroups follow each other in dispatch order
  const int sub = slot - blk * (split ? 3 : 2);       // split: 0 Jacobian, 1 values, 2 Hessian; else 0 x-part, 1 Hessian
  // (bit 7 of the flags: an x-only launch -- f, grad f, g, J of a new iterate whose multipliers are not known yet, the host
  //  shim's pk_prepare_x; the Hessian workgroups of the grid leave at once)
  if ((pre_flags & 128) && sub == (split ? 2 : 1)) return;
  PK_TILE_PROLOGUE_FROM(pre_tile, pre_n_tiles);
  PK_KA_COLLECT();
#undef PK_KA_COLLECT
#ifdef PK_TRACE
  if (A.trace != nullptr && lane == 0 && tl.pad >= 0) {      // role of the record: 0 values / whole tile, 1 Jacobian, 2 Hessian
    const int role = sub == (split ? 2 : 1) ? 2 : (split && sub == 0 ? 1 : 0);
    A.trace[(size_t)(tl.pad * 3 + role) * 16 + 14] = pk_t_entry;
  }
#endif
#ifdef PK_BIG
  PkTile tl0;
  if (big_block(pre_tile, pre_n_tiles, blk, tl0)) {
    if (sub == (split ? 2 : 1)) Gen::bigh(tl0.phase, A, tl0, PK_STAGE(A));
    else if (!split) Gen::bigx0(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, blk);
    else if (sub == 0) Gen::bigx2(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, -1);
    else Gen::bigx1(tl0.phase, A, tl0, PK_STAGE(A), wint, wgrad, blk);
    return;
  }
#endif
  // The compact layouts ride in the same launch (flags bit 8: Hessian, bit 9: Jacobian; the host sets them only for meshes
  // without workgroup-wide intervals): the Hessian workgroups run the per-node compact kernel body (tile_hessc), the
  // Jacobian role runs tile_jacc, the values role leaves the translation entries to it -- a compact cycle is ONE launch too.
  const bool ch = COMPACT && (pre_flags & 256) != 0, cj = COMPACT && (pre_flags & 512) != 0;
  if (sub == (split ? 2 : 1)) {
    if (ch) Gen::tile_hessc(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_G, wint, wgrad, lane);
    else Gen::tile_hess(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_H, wint, wgrad, lane);
  } else if (!split) {
    if (cj) {      // (unsplit x-part: values, then the compact Jacobian, in the same wave and the same LDS rows)
      Gen::tile_xall1c(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, blk);
      pk::wave_lds_sync();
      Gen::tile_jacc(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint, wgrad, lane);
    } else {
      Gen::tile_xall(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, blk);
    }
  } else if (sub == 0) {
    if (cj) Gen::tile_jacc(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint, wgrad, lane);
    else Gen::tile_xall2(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, -1);
  } else {
    if (cj) Gen::tile_xall1c(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, blk);
    else Gen::tile_xall1(tl.phase, A, tl, PK_STAGE(A) + wave * Gen::LDS_X, wint + wave * PK_NRED, wgrad + wave * PK_NRED, lane, blk);
  }
}

// The reductions over all workgroups, by ONE workgroup (all 256 threads must call it).
// flags bit 3: I_k = dt * sum of partials -> Ibuf; bit 0: f = F_o(I, s) -> o_f (systembase.py:592-605);
// bit 4: the gradient slots shared by all nodes (systembase.py:654-657).
template <class Gen>
__device__ __forceinline__ void fin_body(const PkArgs& A) {
  __shared__ double red[PK_WAVES_PER_BLOCK];
  __shared__ double tot[PK_NPHASE_DIM * PK_NRED];
  __shared__ double dts[PK_NPHASE_DIM];
  if ((int)threadIdx.x < PK_NPHASE) dts[threadIdx.x] = Gen::phase_dt(threadIdx.x, A);   // one phase per thread
  double* __restrict__ gsh = A.o_gshared ? A.o_gshared : A.o_grad;     // (see PkArgs.o_gshared)
  if (A.flags & 16)                                                                       // dead / shared slots start at 0
    for (int z = threadIdx.x; z < A.n_gz; z += PK_BLOCK) gsh[A.ib[A.gz_off + z]] = 0.0;
  if (A.flags & 8) {
    for (int n = 0; n < Gen::N_INT; ++n) {
      const int k = Gen::int_phase(n);
      const double sum = block_sum_partials(A, A.partial, k, Gen::int_slot(n), red);
      if (threadIdx.x == 0) A.Ibuf[Gen::int_global(n)] = sum * dts[k];
    }
  }
  if (A.flags & 16) {
    for (int k = 0; k < PK_NPHASE; ++k)
      for (int r = 0; r < Gen::gr_nr(k); ++r) {
        const double v = block_sum_partials(A, A.partial2, k, r, red);
        if (threadIdx.x == 0) tot[k * PK_NRED + r] = v;
      }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const PkSys sy{A.x + A.l_s, A.Ibuf, A.sigma, A.lam};
  if (A.flags & 1) put_f(A.o_f, Gen::sys_objective(sy));
  if (A.flags & 16) {
    for (int k = 0; k < PK_NPHASE; ++k)
      for (int r = 0; r < Gen::gr_nr(k); ++r) gsh[A.ib[A.ph[k].red_off + r]] += tot[k * PK_NRED + r];
    if (!(A.flags & 2)) {
      double gs[PK_NS];
      Gen::sys_grad_static(sy, gs);
      for (int i = 0; i < A.n_s; ++i) gsh[A.l_s + i] += gs[i];
    }
  }
}

// ============================================================================================
// The ONLY exchange a sharded cycle needs -- the sums over all nodes.  Every rank evaluates its share of the mesh
// intervals into its own HBM (pk_cycle on the shard's tiles); what couples the shards is a handful of doubles: the
// integrals (-> f) and the gradient entries of t0 / tf / static parameters.  Each rank posts its partial vector
// [integrals | shared gradient slots] into every peer's mailbox (peer-mapped fine-grained device memory, stores over
// xGMI), raises a flag (the cycle number), waits for the flags of all peers, and adds the vectors IN RANK ORDER (every
// rank gets bit-identical sums) -- one workgroup, no collective library call, no host round trip.  It runs INSIDE the
// cycle's launch (pk_cycle's finalize workgroup, flags bit 6: a sharded cycle is ONE launch per GPU) or as a launch of
// its own behind it (pk_xchg).
// Mailbox of a rank: [parity 2][sender world][xc_stride] words, word 0 of a sender's slot = flag, data from word 1; behind
// them PK_XC_STATE words of this rank's own state (word 0: cycles exchanged so far, word 1: exchanges that timed out).
// Two parities: a rank that is one cycle ahead posts into the other half (it cannot be two ahead: it needs every
// peer's flag of the cycle in between).  The poll is bounded (then the sums read NaN).
// ============================================================================================
#define PK_XC_CAP 512                                   // doubles of a partial vector (host checks)
#ifndef PK_XC_POLL_LIMIT
#define PK_XC_POLL_LIMIT (1 << 24)                      // poll rounds for a peer's flag (~0.5 us each: of the order of ten seconds --
#endif                                                  // a peer may be held up by a module load or a page fault), then NaN + status
__device__ __forceinline__ void sys_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long sys_load(unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// mine[0 .. n_small) (LDS, this rank's partial vector, written by the calling workgroup and visible: the caller has
// synchronized) -> total[0 .. n_small) (LDS, the sums over all ranks); all PK_BLOCK threads call it
__device__ __forceinline__ void exchange_partials(const PkArgs& A, const double* mine, double* total, int n_small) {
  __shared__ int late;
  __shared__ unsigned long long epoch_sh;
  const int t = threadIdx.x, W = A.xc_world, me = A.xc_rank;
  // The cycle number lives in DEVICE memory -- word 0 of the state block behind this rank's own mailbox -- and is advanced
  // by the exchange itself: the launch arguments of a sharded cycle are then the same for every cycle and a batch of them
  // can be replayed as a hipGraph (a host-side count, round 2, was a kernel argument that changed with every launch).
  // xc_epoch > 0 overrides it (a caller that numbers the cycles itself).  Word 1 counts exchanges that gave up waiting.
  unsigned long long* state = A.xc_box[me] + 2 * (size_t)W * A.xc_stride;
  if (t == 0) {
    late = 0;
    epoch_sh = A.xc_epoch > 0 ? (unsigned long long)A.xc_epoch : sys_load(state) + 1ull;
  }
  __syncthreads();
  const unsigned long long epoch = epoch_sh;
  const size_t half = (size_t)(epoch & 1ull) * W * A.xc_stride;
  for (int i = t; i < n_small; i += PK_BLOCK) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(mine[i]);
    for (int q = 0; q < W; ++q)
      if (q != me) sys_store(A.xc_box[q] + half + (size_t)me * A.xc_stride + 1 + i, b);
  }
  __threadfence_system();                               // my data before my flags, for every observer
  __syncthreads();
  if (t < W && t != me) sys_store(A.xc_box[t] + half + (size_t)me * A.xc_stride, epoch);
  if (t < W && t != me) {                               // one polling thread per peer
    unsigned long long* flag = A.xc_box[me] + half + (size_t)t * A.xc_stride;
    long tries = 0;
    const long xlimit = A.poll_limit > 0 ? (long)A.poll_limit : (long)PK_XC_POLL_LIMIT;
    while (sys_load(flag) != epoch && ++tries < xlimit) __builtin_amdgcn_s_sleep(PK_POLL_SLEEP);
    if (tries >= xlimit) late = 1;
  }
  __threadfence_system();
  __syncthreads();
  for (int i = t; i < n_small; i += PK_BLOCK) {
    double sum = 0.0;
    for (int q = 0; q < W; ++q)                         // rank order: the same additions on every rank
      sum += q == me ? mine[i]
                     : __longlong_as_double((long long)sys_load(A.xc_box[me] + half + (size_t)q * A.xc_stride + 1 + i));
    total[i] = late ? __longlong_as_double(0x7FF8000000000000ll) : sum;
  }
  if (t == 0) {
    if (A.xc_epoch <= 0) sys_store(state, epoch);       // (read again by the next launch of this stream)
    if (late) sys_store(state + 1, sys_load(state + 1) + 1ull);      // the host reads it: pk_exchange_status
    if (late && A.status != nullptr) {
      __hip_atomic_fetch_add(A.status + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
  }
  __syncthreads();
}

// pk_cycle's finalize workgroup: the same sums, in the same fixed shape (thread t adds the partials of workgroups
// t, t + 256, ... in ascending order, DPP wave tree, waves 0..3 in order) and the same order of additions per
// gradient slot as fin_body -- the results are bit-identical -- but fed through the hand-off slots of the launch it
// runs in, and arranged so that its own serial chain is ONE memory round trip after the last partial sum arrives:
// everything independent of the sums (phase durations, slot indices) is fetched first, the slots of ALL rows are
// polled together (a `row` = one integrand or one shared gradient slot of one phase; the rows are compile-time
// constants of the model), sums and slot values are combined in LDS and every output is stored once (no
// read-modify-write of global memory).
template <class Gen>
__device__ __forceinline__ void fin_handoff(const PkArgs& A) {
  constexpr int NR = Gen::N_ROWS > 0 ? Gen::N_ROWS : 1;
  static_assert(NR <= PK_BLOCK, "one thread per row");
  __shared__ double red[NR * PK_WAVES_PER_BLOCK];
  __shared__ double tot[NR];
  __shared__ double Ish[PK_NI];
  __shared__ double gsh[PK_NS];
  __shared__ double dts[PK_NPHASE_DIM];
  __shared__ double ssh[PK_NS];                     // static parameters, fetched before the wait (F_o(I, s) reads them)
  __shared__ int ridx[NR];
  const int t = threadIdx.x, wave = t >> 6;
  // where the shared gradient slots go: fetched NOW (a kernel argument of its own cache line: asked for after the wait
  // it put a scalar-cache miss on the finalize chain -- the two-phase rocket lost 7 %)
  double* __restrict__ gout = A.o_gshared ? A.o_gshared : A.o_grad;
  asm volatile("" : "+s"(gout));
  if (t < PK_NPHASE) dts[t] = Gen::phase_dt(t, A);
  for (int i = t; i < PK_NI; i += PK_BLOCK) Ish[i] = 0.0;      // (integrals no system function refers to stay 0)
  for (int i = t; i < A.n_s && i < PK_NS; i += PK_BLOCK) ssh[i] = A.x[A.l_s + i];
  const int gz0 = t < A.n_gz ? A.ib[A.gz_off + t] : -1;
#pragma unroll
  for (int row = 0; row < Gen::N_ROWS; ++row)
    if (t == row) ridx[row] = Gen::row_arr(row) ? A.ib[A.ph[Gen::row_phase(row)].red_off + Gen::row_slot(row)] : -1;
  constexpr int per = PK_WAVES_PER_BLOCK;              // a partial sum per tile block (four tiles)
  double acc[NR];
  int most = 0;
#pragma unroll
  for (int row = 0; row < Gen::N_ROWS; ++row) {
    acc[row] = 0.0;
    most = max(most, (A.ph[Gen::row_phase(row)].tile_hi - A.ph[Gen::row_phase(row)].tile_lo) / per);
  }
  PK_MARK_AT(A.n_tiles * 3 + 2, 1);
  for (int j = t; j < most + t; j += PK_BLOCK) {        // (uniform trip count; j - t = 0, 256, ...)
    unsigned long long bits[NR];
    unsigned long long* slot[NR];
#pragma unroll
    for (int row = 0; row < Gen::N_ROWS; ++row) {
      const PkPhase& ph = A.ph[Gen::row_phase(row)];
      const int b = ph.tile_lo / per + j;
      slot[row] = b < ph.tile_hi / per ? (Gen::row_arr(row) ? A.cpart2 : A.cpart) + (size_t)b * PK_NRED + Gen::row_slot(row)
                                       : nullptr;
    }
    // one poll round = the slots of ALL rows in flight together (one memory round trip), repeated until none is empty
    const int limit = A.poll_limit > 0 ? A.poll_limit : PK_POLL_LIMIT;
    bool empty = true;
    for (int tries = 0; tries < limit && empty; ++tries) {
      empty = false;
#pragma unroll
      for (int row = 0; row < Gen::N_ROWS; ++row) bits[row] = slot[row] ? handoff_peek(slot[row]) : 0ull;
#pragma unroll
      for (int row = 0; row < Gen::N_ROWS; ++row) empty |= bits[row] == PK_EMPTY;
      if (empty) __builtin_amdgcn_s_sleep(PK_POLL_SLEEP);
    }
    if (empty && A.status != nullptr) {     // gave up: the sums of this launch are NaN -- tell the host, ahead of f
      __hip_atomic_fetch_add(A.status, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
    }
#pragma unroll
    for (int row = 0; row < Gen::N_ROWS; ++row)
      if (slot[row]) {
        handoff_clear(slot[row]);
        acc[row] += __longlong_as_double((long long)bits[row]);      // (PK_EMPTY after a timed-out poll: a NaN)
      }
  }
  PK_MARK_AT(A.n_tiles * 3 + 2, 2);
#pragma unroll
  for (int row = 0; row < Gen::N_ROWS; ++row) acc[row] = wave_sum(acc[row]);      // (all trees first: see wave_sums_to)
  if ((t & 63) == 0) {
#pragma unroll
    for (int row = 0; row < Gen::N_ROWS; ++row) red[row * PK_WAVES_PER_BLOCK + wave] = acc[row];
  }
  __syncthreads();
  if (t < Gen::N_ROWS) {      // thread = row (a chain of N_ROWS one-thread branches was 1 us of the launch's tail for 13 rows)
    double v = 0.0;
#pragma unroll
    for (int w = 0; w < PK_WAVES_PER_BLOCK; ++w) v += red[t * PK_WAVES_PER_BLOCK + w];
    tot[t] = v;
    if (!Gen::row_arr(t)) {                               // integrand row n = row: I_k = dt * sum   (phasebase.py:997-1006)
      const double Ik = v * dts[Gen::row_phase(t)];
      Ish[Gen::int_global(t)] = Ik;
      A.Ibuf[Gen::int_global(t)] = Ik;
    }
  }
  __syncthreads();
  const PkSys sy{ssh, Ish, A.sigma, A.lam};
#ifdef PK_SHARDED
  const bool xchg = (A.flags & 64) != 0;                  // sharded cycle: the sums over the RANKS, inside this launch
#else
  constexpr bool xchg = false;                            // (code objects of single-GPU evaluators carry no exchange code)
#endif
  if (t == 0 && !xchg) put_f(A.o_f, Gen::sys_objective(sy)); // systembase.py:592-605
  if (t == 64) {
#pragma unroll
    for (int i = 0; i < PK_NS; ++i) gsh[i] = 0.0;
    if (!(A.flags & 2)) Gen::sys_grad_static(sy, gsh);      // (sharded: the primary shard adds the direct dependence)
  }
  __syncthreads();
  // gradient slots no tile writes (end slots, t0 / tf, static parameters): 0 + the sums of the rows that land on the
  // slot, phase by phase, + the objective's direct dependence on a static parameter   (systembase.py:654-657)
  // (the finalize workgroup stages no tile: the exchange's two vectors live in the launch's dynamic LDS, which the
  // runtime sizes to at least 2 PK_XC_CAP doubles for a sharded cycle -- static arrays would cost EVERY workgroup of
  // the launch 8 KB of LDS and the humanoid its second workgroup per CU)
  extern __shared__ double pk_lds[];
  double* __restrict__ xmine = pk_lds;
  double* __restrict__ xtotal = pk_lds + PK_XC_CAP;
  for (int z = t; z < A.n_gz; z += PK_BLOCK) {
    const int idx = z == t ? gz0 : A.ib[A.gz_off + z];
    double v = 0.0;
#pragma unroll
    for (int row = 0; row < Gen::N_ROWS; ++row)
      if (Gen::row_arr(row) && ridx[row] == idx) v += tot[row];
    if (idx >= A.l_s && idx < A.l_s + A.n_s) v += gsh[idx - A.l_s];
    if (xchg) xmine[PK_NI + z] = v;
    else gout[idx] = v;
  }
  if (!xchg) return;                                      // (wave-uniform)
  // this shard's partial vector [integrals | the slots above, in their order]: PK_NI + n_gz <= PK_XC_CAP (host checks)
  for (int i = t; i < PK_NI; i += PK_BLOCK) xmine[i] = Ish[i];
  __syncthreads();
  exchange_partials(A, xmine, xtotal, PK_NI + A.n_gz);
  for (int i = t; i < PK_NI; i += PK_BLOCK) A.Ibuf[i] = xtotal[i];
  for (int z = t; z < A.n_gz; z += PK_BLOCK) gout[z == t ? gz0 : A.ib[A.gz_off + z]] = xtotal[PK_NI + z];
  if (t == 0) {
    const PkSys syt{ssh, xtotal, A.sigma, A.lam};
    put_f(A.o_f, Gen::sys_objective(syt));                // f on the global integrals, on every rank
  }
}

// pk_xchg: the exchange as a launch of its own, behind the shard's pk_cycle on the same stream (A/B form of the in-launch
// exchange; also what a caller uses that runs the callbacks one by one).  Partial vector: [integrals | the gradient
// slots xc_idx lists].
template <class Gen>
__device__ __forceinline__ void kernel_xchg(const PkArgs& A) {
  __shared__ double mine[PK_XC_CAP];
  __shared__ double total[PK_XC_CAP];
  const int t = threadIdx.x;
  const int n_I = PK_NI, n_small = n_I + A.xc_nsh;
  double* __restrict__ gsh = A.o_gshared ? A.o_gshared : A.o_grad;
  for (int i = t; i < n_small; i += PK_BLOCK) mine[i] = i < n_I ? A.Ibuf[i] : gsh[A.xc_idx[i - n_I]];
  __syncthreads();
  exchange_partials(A, mine, total, n_small);
  for (int i = t; i < n_small; i += PK_BLOCK) {
    if (i < n_I) A.Ibuf[i] = total[i];
    else gsh[A.xc_idx[i - n_I]] = total[i];
  }
  if (t == 0 && (A.flags & 1)) {
    const PkSys sy{A.x + A.l_s, total, A.sigma, A.lam};
    put_f(A.o_f, Gen::sys_objective(sy));               // systembase.py:592-605, on the global integrals
  }
}

// pk_runs: dst[dst_off + i] = src[src_off + i] for every chunk (src_off, dst_off, len) of the table -- the pack /
// unpack passes of the RCCL gather / all-gather reassembly (A/B forms of the exchange): a rank's owned output positions
// are a few dozen contiguous runs, cut into chunks of at most 16 Ki doubles by the host, one workgroup per chunk.
__device__ __forceinline__ void kernel_runs(const PkArgs& A) {
  typedef double pk_d2 __attribute__((ext_vector_type(2)));
  for (int c = blockIdx.x; c < A.rc_n; c += gridDim.x) {
    const int64_t so = A.rc_table[3 * c], dofs = A.rc_table[3 * c + 1], len = A.rc_table[3 * c + 2];
    const double* __restrict__ src = A.rc_src + so;
    double* __restrict__ dst = A.rc_dst + dofs;
    if ((((uintptr_t)src ^ (uintptr_t)dst) & 8) == 0) {      // congruent modulo 16 bytes: 16 bytes per lane (the host-landed
      const int64_t head = ((uintptr_t)src >> 3) & 1;         // sharded cycle stores these runs over PCIe)
      const int64_t pairs = (len - (head < len ? head : len)) >> 1;
      const pk_d2* __restrict__ s2 = reinterpret_cast<const pk_d2*>(src + head);
      pk_d2* __restrict__ d2 = reinterpret_cast<pk_d2*>(dst + head);
      for (int64_t i = threadIdx.x; i < pairs; i += PK_BLOCK) d2[i] = s2[i];
      if (threadIdx.x == 0) {
        if (head && len > 0) dst[0] = src[0];
        if (head + 2 * pairs < len) dst[len - 1] = src[len - 1];
      }
      continue;
    }
    for (int64_t i = threadIdx.x; i < len; i += PK_BLOCK) dst[i] = src[i];
  }
}

template <class Gen>
__device__ __forceinline__ void kernel_fin(const PkArgs& A) {
  fin_body<Gen>(A);
}

}  // namespace pk

/* pk_cycle's PkArgs are read from the kernarg segment where a wave uses them, not en bloc on entry.  As a by-value
   argument every field any path of the kernel touches is loaded in the entry block and stays live (100 SGPRs, spills to VGPR
   lanes, scalar waits in front of the tile-record load); read lazily a tile wave loads what its phase and role need (60
   SGPRs).  Two-phase rocket: 158k -> 193k cycles/s.  &A of a by-value kernel argument IS its place in the kernarg segment
   (constant address space). */
#define PK_DEFINE_CYCLE(GEN, NAME, COMPACT)                                                                       \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void NAME(const PkTile* pre_tile, int32_t pre_n_tiles,        \
                                                              int32_t pre_flags, int32_t pre_grid, PkArgs A) {     \
    const PkArgs PK_CONST_AS* A_ = (const PkArgs PK_CONST_AS*)(                                                    \
        (const char PK_CONST_AS*)__builtin_amdgcn_kernarg_segment_ptr() + PK_CYCLE_ARGS_OFFSET);                   \
    pk::kernel_cycle<GEN, COMPACT>(pre_tile, pre_n_tiles, pre_flags, pre_grid, *(const PkArgs*)A_);                  \
  }

#define PK_DEFINE_KERNELS(GEN)                                                                         \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_int(PkArgs A) { pk::kernel_int<GEN>(A); }   \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_fin(PkArgs A) { pk::kernel_fin<GEN>(A); }   \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_g(PkArgs A) { pk::kernel_g<GEN>(A); }       \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_grad(PkArgs A) { pk::kernel_grad<GEN>(A); } \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_jac(PkArgs A) { pk::kernel_jac<GEN>(A); }   \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_hess(PkArgs A) { pk::kernel_hess<GEN>(A); } \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_xall(PkArgs A) { pk::kernel_xall<GEN>(A); } \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_aux(PkArgs A) { pk::kernel_aux<GEN>(A); }   \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_outer(PkArgs A) { pk::kernel_outer(A); }     \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_hessc(PkArgs A) { pk::kernel_hessc<GEN>(A); } \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_jacc(PkArgs A) { pk::kernel_jacc<GEN>(A); }   \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_err(PkArgs A) { pk::kernel_err<GEN>(A); }     \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_csr(PkArgs A) { pk::kernel_csr(A); }             \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_xchg(PkArgs A) { pk::kernel_xchg<GEN>(A); }     \
  extern "C" __global__ __launch_bounds__(PK_BLOCK) void pk_runs(PkArgs A) { pk::kernel_runs(A); }           \
  PK_DEFINE_CYCLE(GEN, pk_cycle, false)                                                                   \
  PK_DEFINE_CYCLE(GEN, pk_cyclec, true)

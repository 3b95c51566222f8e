// pk_abi.h -- plain-data structures shared by the host runtime (pk_runtime.cpp) and the device
// code (pk_kernels.hip.h + generated model code).  Everything is POD; offsets index into three
// device "blob" arrays (int32 ib[], double db[], int64 lb[]) uploaded once per (model, mesh).
#pragma once
#include <stdint.h>

#define PK_WAVE 64
#define PK_WAVES_PER_BLOCK 4     // (1, 2 and 8 were measured slower: profiles/r04_z_waves_per_workgroup.txt)
#define PK_BLOCK (PK_WAVE * PK_WAVES_PER_BLOCK)

// One phase of the problem on its mesh.
struct PkPhase {
  int32_t scheme;      // 0 = LGR, 1 = LGL
  int32_t n_x, n_u, n_c;
  int32_t L_m;         // middle-stage nodes
  int32_t L_d;         // defect rows per state
  int32_t state_len;   // L_m + 1 (LGR) or L_m (LGL)
  int32_t L;           // length of the phase block of x (states, controls, t0, tf)
  int32_t x_off;       // start of the phase block in x
  int32_t g_off;       // first defect row in g / lambda
  int32_t path_off;    // first path-constraint row in g / lambda
  int32_t mid_lo, mid_hi;   // middle node range [mid_lo, mid_hi)
  int32_t tile_lo, tile_hi; // tiles of this phase; both multiples of PK_WAVES_PER_BLOCK (padded with empty tiles)
  int32_t tau_off;     // db: node positions tau[L_m] in [0,1]
  int32_t w_off;       // db: quadrature weights w[L_m]
  int32_t width_off;   // db: interval widths d[N]
  int32_t jseg_off;    // lb: base offset in J of every Jacobian segment of the phase (I then N)
  int32_t jt_off;      // lb: base offset in J of the constant translation piece of every state
  int32_t hseg_off;    // lb: base offset in H of every Hessian segment of the phase (I then N)
  int32_t red_off;     // ib: NLP index of every gradient reduction slot of the phase
  int32_t aseg_off;    // lb: base offset in the auxiliary buffer of every auxiliary segment of the phase
  int32_t hcseg_off;   // lb: base offset in the compact Hessian of every compact segment of the phase
  int32_t ivK_off;     // ib: points per interval K[N]
  int32_t ivfull_off;  // ib: db offset of the dense R x K integration block of every interval
  int32_t ivld_off;    // ib: first defect row (within a state) of every interval
  int32_t n_int;       // number of mesh intervals N
  int32_t jcseg_off;   // lb: base offset in the compact Jacobian of every compact segment of the phase (I, then D, then N)
  int32_t jct_off;     // lb: base offset in the compact Jacobian of the translation piece of every state
};

// A run of `nj` consecutive intervals of one kind handled by one wavefront (<= 64 nodes).
struct PkTile {
  int32_t phase;
  int32_t j0, nj;      // first interval, number of intervals
  int32_t kid;         // kind with front/back columns dropped (Jacobian / Hessian)
  int32_t kidf;        // full kind (constraint values)
  int32_t q0;          // first node
  int32_t r0;          // first defect row (within a state)
  int32_t offI;        // position of the tile inside every I-expanded segment
  int32_t offT;        // position of the tile inside every translation piece
  int32_t K;           // points per interval
  int32_t last;        // 1 if the tile ends the phase
  // copy of the two kinds' table locations (saves a dependent load per tile)
  int32_t nnzI, nnzT;  // entries per interval of kind `kid`
  int32_t irc_off;     // ib: (r, c) pairs of kind `kid`
  int32_t iv_off;      // db: values of kind `kid`
  int32_t tv_off;      // db: translation values of kind `kid`
  int32_t full_off;    // db: dense R x K block of kind `kidf`
  int32_t pad;         // (device: index of the tile, set by the kernel prologue)
  // floor(p / d) = umulhi(p, magic) for p < 2^16 with magic = floor((2^32 - 1) / d) + 1: no integer division on
  // the device for the three per-tile divisors
  uint32_t magicI;     // d = nnzI   (entries per interval of the integration pattern)
  uint32_t magicR;     // d = R      (defect rows per interval)
  uint32_t magicT;     // d = nnzT   (translation entries per interval)
  int32_t stage;       // (device: staging slot of an interval with more than 256 points, set by pk_set_problem)
};

// Entry tables of one interval pattern (unit width).
struct PkKind {
  int32_t K, R;        // points, defect rows
  int32_t nnzI, nnzT;  // entries of the integration / translation block after dropping columns
  int32_t irc_off;     // ib: (r, c) pairs of the integration entries, row-major
  int32_t iv_off;      // db: values of those entries
  int32_t tv_off;      // db: values (+1/-1) of the translation entries
  int32_t full_off;    // db: dense R x K integration block
};

// out[pos] = coef * E[eid] * (lam >= 0 ? lambda[lam] : 1)   (boundary nodes, system level)
struct PkItem {
  int64_t pos;
  double coef;
  int32_t eid;
  int32_t lam;
};

// One outer-product block of a system-level Hessian (objective / system constraints nonlinear in the
// integrals; reference: easyderiv.py:323-355,393-430).  A, B: runs of quadrature-weighted gradient entries
// in the auxiliary buffer; M: location of the scalar multiplier.
//   kron:  out[pos + i*lenB + j] = A[i] * B[j] * m
//   tril:  A' = collapseA ? {sum A} : A (same for B); for (i >= j) in row-major lower-triangular order:
//          out[pos + t] = A'[i] * B'[j] * m, and if `second` out[pos + ntri + t] = B'[i] * A'[j] * m
struct PkOuter {
  int64_t pos;
  int32_t offA, lenA, offB, lenB, offM;
  int32_t flags;       // bit 0 tril, bit 1 collapseA, bit 2 collapseB, bit 3 second
  int32_t count, pad;
};

// One mesh interval of the error-estimation pass (pk_err): the interval is re-collocated with K + 1
// points (reference: phasebase.py:1339-1372).  Offsets index the error-table blob `errdb`:
//   tab_off -> [V_x (K+1) x ncx | V_u (K+1) x K | T nr x ncx | I nr x (K+1)]   (row-major, unit interval)
//   tau_off -> position in [0, 1] of the K + 1 augmented nodes of this interval
// with ncx = K + 1, nr = K + 1 (LGR) or ncx = nr = K (LGL).
struct PkErrIv {
  int32_t phase;
  int32_t K;
  int32_t lm;          // first node of the interval (state / control slot)
  int32_t row0;        // first output row (within a state)
  int32_t tab_off;
  int32_t tau_off;
  int32_t rows;        // output rows per state of the phase
  int32_t stage;       // (device: staging slot of an interval with more than 263 augmented nodes, set by the library)
  int64_t out_off;     // start of the phase in the two output arrays ([n_x][rows] each)
  double width;        // interval width (fraction of the phase)
};

// Phases a code object holds BY VALUE in its kernel arguments (PkArgs.ph, last member).  The reference puts no limit on the
// number of phases (systembase.py:148-187); here the code generator raises the constant for a model with more than 8
// (codegen.py emits the #define in front of this header) and the library passes offsetof(PkArgs, ph) + that many records, up
// to PK_HOST_MAX_PHASES = 15 KB of records -- this stack takes 32 KB of kernel arguments (tools/kernarg_probe.hip, measured).
#ifndef PK_MAX_PHASES
#define PK_MAX_PHASES 8
#endif
#define PK_HOST_MAX_PHASES 128
#define PK_CYCLE_ARGS_OFFSET 24   // pk_cycle: bytes of leading scalar kernel arguments in front of its PkArgs
#define PK_MAX_RANKS 64       // ranks of one sharded NLP (pk_xchg: one polling thread per peer)
#define PK_XC_STATE 16        // state words behind the 2 x world x stride words of a rank's mailbox (cycle count, time-outs)

// pk_cycle's in-launch hand-off: a slot of cpart / cpart2 is either PK_EMPTY (a quiet-NaN pattern no arithmetic
// produces) or the value a tile workgroup published during the current launch.
#define PK_EMPTY 0x7FF8C0DEC0DEC0DEull

struct PkArgs {
  const double* x;        // NLP variables (device)
  const double* lam;      // constraint multipliers (device; Hessian only)
  double* o_f;            // outputs (device); a launch writes only the ones its kernel produces
  double* o_grad;         // [n]
  double* o_g;            // [m]
  double* o_jac;          // [nnz_J]
  double* o_hess;         // [nnz_H]
  double* o_aux;          // auxiliary buffer (integral gradient entries x w, multipliers); outer-product path only
  const PkOuter* outer;
  double sigma;           // objective factor (Hessian only)
  const PkPhase* phase;
  const PkTile* tile;
  const PkKind* kind;
  const PkItem* items;
  const PkItem* items2;   // pk_cycle only: the Hessian's boundary / system items (items = the Jacobian's)
  const int32_t* ib;
  const double* db;
  const int64_t* lb;
  double* Ibuf;           // integrals I_k (pre-pass result)
  double* partial;        // [workgroups][PK_NRED] per-workgroup partial sums of the integrands
  double* partial2;       // [workgroups][PK_NRED] per-workgroup partial sums of the shared gradient slots
  unsigned long long* cpart;   // pk_cycle only: the same two arrays as 64-bit patterns, handed from the tile workgroups
  unsigned long long* cpart2;  // to the finalize workgroup INSIDE one launch (every slot holds PK_EMPTY between launches)
  const PkErrIv* erriv;   // mesh error estimation (pk_err only)
  const int32_t* errgrp;  // (first record, count) of the run of intervals every wavefront of pk_err handles
  const double* errdb;
  double* o_errT;         // T_aug x      per phase [n_x][rows]
  double* o_errI;         // dt I_aug f   per phase [n_x][rows]
  const double* csr_in;   // triplet values -> CSR values (pk_csr only): out[p] = sum_{q in [seg[p], seg[p+1])} in[perm[q]]
  const int32_t* csr_seg; // nullptr when no (row, col) repeats: out[p] = in[perm[p]]
  const int32_t* csr_perm;
  double* csr_out;
  unsigned long long* trace;   // developer tracing (models generated with POCKIT_AMD_TRACE=1): [tile][16] s_memtime marks
  double* o_gshared;      // where the gradient slots shared by all nodes go (NULL: o_grad) -- a shard whose o_grad is
                          // another GPU's buffer keeps its partial sums of those slots local
  // pk_xchg: exchange of the shard's partial sums [integrals | shared gradient slots] through peer-mapped mailboxes
  unsigned long long* const* xc_box;   // [world] base of every rank's mailbox as mapped in this process (own entry: own)
  const int32_t* xc_idx;  // NLP index of every shared gradient slot
  // pk_runs: copy of contiguous runs, table of (src offset, dst offset, length) per chunk
  const int64_t* rc_table;
  const double* rc_src;
  double* rc_dst;
  // staging rows of intervals too long for the workgroup's LDS (more than 256 points): slot s of sub-slot u starts at
  // big_stage + (4 s + u) * big_slot doubles, rows of big_row doubles   (u: 0 x-part / values, 1 Jacobian role, 2 Hessian;
  // pk_err: its own buffer, one sub-slot per interval)
  double* big_stage;
  // Status words in pinned host memory (system-scope atomics): [0] in-launch hand-offs of pk_cycle's partial sums that gave
  // up waiting (the sums of that launch read NaN), [1] exchanges between the ranks that gave up (pk_xchg / in-launch).  The
  // library compares them with what it has seen after every wait and turns a change into error 97 (pk_runtime.cpp).
  unsigned long long* status;
  int32_t n_tiles, n_items;
  int32_t n_items2, poll_limit;   // poll_limit > 0: poll rounds before a hand-off gives up (tests; 0: PK_POLL_LIMIT)
  int32_t n_phase, n;
  int32_t l_s, n_s, n_sys, m;
  int32_t gz_off, n_gz;   // ib: gradient slots the finalize kernel zero-fills
  int32_t flags;          // bit 0: pk_fin writes f; bit 1: secondary shard (no system-level / boundary work);
                          // bit 3: pk_fin reduces the integrals into Ibuf; bit 4: pk_fin reduces the gradient slots
  int32_t n_outer;
  int32_t n_erriv;        // (pk_err: number of wave groups)
  int32_t n_csr;          // CSR entries
  int32_t xc_world, xc_rank, xc_epoch, xc_nsh;   // ranks, this rank, cycle number (> 0: given by the host; 0: counted on the
                                                 // device, in the mailbox's state block), number of shared gradient slots
  int32_t xc_stride, rc_n;                       // mailbox words per sender (multiple of 16); chunks of the run table
  int32_t big_row, big_slot;
  PkPhase ph[PK_MAX_PHASES];   // the phases by value (kernarg segment): no dependent global load
};

// Device-side model description and the element-wise log-density interpreter.
//
// The reference compiles `Model.logp` + `pytensor.grad` into one function
// (`ValueGradFunction`, pymc/model/core.py:142-305).  Here the model spec that crosses
// the C ABI (include/nuts_mi355.h) is "compiled" on the host into flat tables:
//
//   * per value variable, the list of factor arguments it appears in ("contributions"),
//     so that the reverse-mode gradient of element i is a GATHER over a fixed list in a
//     fixed order (deterministic, no floating-point atomics);
//   * size-1 variables that broadcast against a larger factor get a "broadcast term":
//     the thread that owns the factor element accumulates d logp / d (that scalar) and
//     the per-workgroup partials are summed in workgroup order by the control kernel;
//   * "deferred" elements (every size-1 variable and the mu/sigma hyper-parameters of
//     the hierarchical-logit node) need a cross-workgroup reduction before their
//     gradient is known; the control kernel finishes them.
#pragma once
#include "device_math.h"
#include "nuts_mi355.h"

#define MAX_BTERMS 8        // broadcast terms per model
#define MAX_DERIVED 4       // derived vectors (NUTS_D_DERIVED factors) per model
#define MAX_FACTOR_BT 6     // broadcast operands per factor
#define MAX_DEFERRED 256    // deferred elements per model (one control-kernel thread each)
#define LOGIT_MAXD 8
#define SEG_MAIN_MAX 10     // fixed-slot segment layout: at most this many main segments per group (else pointer tables)

struct VarDev {  // nuts_var + what the spec compiler derived
  int32_t offset, size, transform, deferred;
  double lower, upper;
  // peephole: an untransformed variable whose only factor is its own Normal(mu, sigma) prior with constant
  // parameters (the non-centred `z ~ N(0, 1)` block of a hierarchical model) is evaluated in closed form:
  // logp_i = -(x - mu)^2 / (2 sigma^2) - np_lognorm ;  d/dx = -(x - mu) / sigma^2
  int32_t normal_prior;
  int32_t def_base;   // deferred variables: position of element 0 in the model's deferred list
  double np_mu, np_inv_var, np_lognorm;
};

// arg == -1: the variable occurs somewhere in a factor WITH an expression program (differentiated through the program);
// arg == -2: the variable is GATHERED into the factor (NUTS_OP_GATHER): `dist` / `pad` are the offsets in ModelDev.csr of the
//            row pointers [size + 1] and of the factor-element list of the inverse index, never an owner
struct Contrib {  // "variable k is operand `slot` (0=a,1=b,2=c) of argument `arg` of factor f"
  int32_t f;
  int16_t arg, slot;
  int32_t owner;  // 1: this contribution also accounts for the factor's logp and broadcast terms
  // pre-decoded form of the common case "the variable itself is argument `arg`, every other argument is a
  // constant" (priors with fixed parameters): the interpreter evaluates it without touching the factor table
  int32_t fast;   // 0 general; 1 constants folded (p[]); 2 = 1 + Normal / HalfNormal closed form (p[2] = 1/sigma, p[3] = log sigma)
  int32_t dist, pad;
  double konst;
  double p[4];
};

// Gathered adjoints (round 6).  A variable that enters a factor through NUTS_OP_GATHER operands used to have its gradient formed by the
// PARAMETER element: one forward + reverse sweep of the factor's program per (element of the variable, factor element that indexes it)
// -- a weight matrix broadcast over the rows of a likelihood (a neural network, a softmax regression: every coefficient is read by
// every row) cost (number of coefficients) x (rows) sweeps per gradient, executed one parameter at a time.  Now every element of
// such a factor is swept ONCE, ahead of everything else of the leaf (k_gsweep, kernels.h; a phase of the single-workgroup kernel):
// the sweep leaves, per (variable, index vector) pair of the factor -- a "slot" --, d logp_element / d (that gather) in
// ModelDev.adj[slot.adj_off + element]; the parameter element then adds up the entries its inverse index lists, in index order
// (the same numbers in the same order as before).
#define MAX_GSLOTS 64       // (variable, index vector) pairs per factor served this way; a factor with more keeps the old path
struct GSlot { int32_t var, did; int64_t adj_off; };
struct GSweepFactor { int32_t f, slot0, n_slots, elem0; int32_t orphan, pad; };   // orphan: no variable owns the factor -- the sweep also accounts its log-density and its scalars' adjoints (k_gsweep)
// A variable element that MANY factor elements index (a regression coefficient read by every row: an inverse-index list of 10^5
// entries) is not added up by the one thread that owns the element: a workgroup of its own totals the list first (k_gadj_reduce, a
// phase of the single-workgroup kernel) -- thread t adds entries t, t + 256, ... in order, the threads' partials are combined in the
// fixed order of `block_sum` -- and the owner reads one number.
#define GADJ_LONG 512       // lists from this length on
#define GADJ_CHUNK 4096     // entries of a long list one workgroup totals (16 per thread); a list of 10^5 entries is 25 workgroups' work
struct GLong { int64_t adj_off; int32_t lst_off, len; int32_t nchunk, pad; };   // one CHUNK of a long list; nchunk (on a list's first chunk): how many follow each other   // slots [slot0, slot0 + n_slots) of ModelDev's slot table; elem0: first of its elements in the sweep's numbering

struct FactorBT {  // broadcast (size-1 variable) operands of a factor whose size is > 1
  int32_t n, pad;
  struct { int16_t arg, slot; int32_t bterm; } e[MAX_FACTOR_BT];
};

struct RowsDev {  // hierarchical Bernoulli-logit node (rows sorted by group); see rows_kernel.h
  int64_t N, Npad;
  int32_t D, G;
  const double* Xt;        // [n_spans][D][span] span-tiled copy of X: one contiguous 8*D*span-byte block per wave-iteration
  const int8_t* y;         // [Npad]
  const int64_t* gptr;     // [G+1] first row of each group
  int32_t off_mu, off_sigma, off_z, sigma_tr;
  int32_t var_mu, var_sigma, var_z, pad;
  int32_t def_mu, def_sigma;   // positions of mu[0] / sigma[0] in the deferred list
  // fixed-slot segment layout (segK > 0): seg_part is [G][segK][D]; slots 0 .. segK-3 are the group's main segments in
  // wave order (unused ones stay zero), slot segK-2 the mixed span in which the group starts, slot segK-1 the mixed
  // span in which it only ends.  A z element's segment addresses then follow from its index alone, and kernel B can
  // issue the loads with its very first ones instead of after a round trip through the segment pointers.
  int32_t segK, seg_pad;
  int64_t n_spans;         // Npad / span
  int32_t n_waves, n_seg;  // waves in the row-streaming launch; total (wave, group) segments
  const int32_t* run_ptr;  // [n_waves+1] runs of main wave w = [run_ptr[w], run_ptr[w+1])
  const int4* runs;        // {first span, number of spans, group, segment slot}: consecutive spans lying entirely inside one group
  const int32_t* gseg_ptr; // [G+1] main segments of group g = [gseg_ptr[g], gseg_ptr[g+1])
  double* seg_part;        // [n_seg][D]  d logp / d beta_g partial of each (wave, group) run
  // mixed spans (contain a group boundary or padding rows): one wave each, extra workgroups of the same launch
  int32_t n_mixed, n_mixed_seg;
  const int64_t* mixed_span;     // [n_mixed] span index
  const int32_t* mixed_g0;       // [n_mixed] group of the span's first row
  const int32_t* mixed_seg_base; // [n_mixed]
  const int32_t* gmix_ptr;       // [G+1] mixed segments of group g
  double* mixed_part;            // [n_mixed_seg][D]
  double* wave_lp;         // [n_waves + n_mixed]   log-likelihood partial of each wave
  // ---- group-aligned row pass (rows_ga_kernel.h): workgroup g streams exactly the rows of group g ----
  // Xt / y are then tiled PER GROUP (group g owns tiles [ga_tile0[g], ga_tile0[g+1]), the last one zero-padded), the
  // span tables above are not built, and there is no O(n) kernel between two row passes: the workgroup that streamed a
  // group holds that group's complete d logp / d beta and finishes its D z elements itself.
  int32_t ga, ga_w;              // active; waves per workgroup
  int32_t ga_dx, ga_gpw;         // stored columns per tile: D, or D - 1 when column 0 of X is identically 1 (not stored);
                                 // ga_gpw > 0: group-BLOCK pass (rows_gb_kernel.h), a workgroup owns ga_gpw consecutive groups
  int32_t ga_nblk, ga_bsz;       // second-level reduction: blocks of ga_bsz consecutive groups (ga_bsz <= 64)
  int32_t ga_T_uni, ga_flags;    // > 0: every group has this many tiles (and ga_ng_uni rows): geometry without table look-ups
  int64_t ga_ng_uni;
  const int32_t* ga_tile0;       // [G+1] tiles of group g = ga_tile0[g+1] - ga_tile0[g]
  // X / y are laid out per CHUNK = the tiles one wave streams (group g, wave w): chunk (g, w) starts at element ga_coff[g W + w]
  // of Xt (its y at that offset / D).  Chunk starts are skewed by an odd multiple of 256 B: with a power-of-two group stride
  // (C2-L: 32 tiles x 8 KiB = 256 KiB) the G W lock-stepped streams would sit on the same memory channels all the time.
  const int64_t* ga_coff;        // [G W]
  int64_t ga_cstride_uni;        // uniform geometry: ga_coff[c] = c * ga_cstride_uni (no table look-up)
  unsigned* ga_ticket;           // [ga_nblk] arrivals of the block's groups in the current launch (self-resetting)
  double* ga_part;               // [G][PART_STRIDE] per-group partial record (written write-through, read by the block's last arriver)
  double* ga_bpart;              // [2][ga_nrec][PART_STRIDE] block partials (+ the records of the auxiliary workgroups), double-buffered by launch parity
  // Auxiliary workgroups (rows_aux.h): when the model is more than the closed forms below -- other hyper-prior families, further
  // scalar or vector variables with their own factors -- ga_naux extra workgroups of the same launch run the element-wise
  // interpreter over every element that is not a z element (ga_auxel of them) and leave one record each behind the ga_nblk block
  // partials: ga_nrec = ga_nblk + ga_naux records per launch.  ga_naux == 0: the closed forms (the benchmark's model).
  int32_t ga_naux, ga_nrec, ga_auxel, ga_pad2;
  // closed forms of what the interpreter would evaluate for this model (checked by the spec compiler):
  double z_np_mu, z_np_inv_var, z_np_lognorm;   // z ~ Normal(mu0, s0) untransformed
  double mu_c[3];                               // mu ~ Normal(p1, .): {p1, 1/sigma, log sigma}
  double sg_c[2];                               // sigma ~ HalfNormal(.): {1/sigma, log sigma}
};

struct MvnDev {
  int32_t k, off;
  const double* mu;    // [k]
  const double* prec;  // [k][k]
  double konst;        // -k/2 log(2 pi) - logdet
  double* rowq;        // [k] delta_i * (P delta)_i
  double* gdense;      // [n] -(P delta)
  // "cholesky" solver: W = chol(cov)^-1 and its transpose as full row-major matrices (zeros in the other triangle), scratch
  const double* winv; const double* winv_t;
  double* wy;          // [3][k]: delta, y = W delta, P delta = W^T y
  // row-aligned pass (kernels.h, k_mvn_aligned): the model IS the MvNormal node, so the workgroup that owns rows [bR, bR + R) also
  // finishes those elements of the leapfrog and leaves one record for the control work
  int32_t aligned, al_nwg;   // rows per workgroup (0: off), workgroups
  double* al_part;           // [2][al_nwg][MVA_RS] per-workgroup records (record-major, compact), double-buffered by launch parity
};

// dense node 3: Normal mixture over observed rows (mixture_kernel.h)
#define MIX_MAXK 16
struct MixDev {
  int64_t N;
  int32_t K, off_mu, off_sigma, off_w;   // element offsets of the parameter variables (off_sigma / off_w < 0: the constants below)
  int32_t tr_sigma, nwg;
  // Dirichlet weights under the simplex transform (w_simplex != 0): the variable at off_w is the transformed value y of K - 1
  // elements, w = softmax([y, -sum(y)]); the node adds sum((alpha - 1) log w) + w_konst - K logsumexp([y, -sum(y)])
  // (Dirichlet.logp + SimplexTransform.log_jac_det; w_konst = gammaln(sum(alpha)) - sum(gammaln(alpha)) + log(K))
  int32_t w_simplex, pad_;
  double alpha[MIX_MAXK], w_konst;
  const double* y;        // [N]
  const double* assign;   // [N] assignments in the data pool (float-coded integers), nullptr: marginal form
  double sigma_c[MIX_MAXK], logw_c[MIX_MAXK];
  double* part;           // [nwg][3 MIX_MAXK + 1] per-workgroup sums {R_k, A_k, B_k, logp}
  double* gdense;         // [n] the node's gradient w.r.t. the constrained values (zero outside its parameters)
  double* lp;             // the node's logp
};

// dense node 4: generalised linear model rows (glm_kernel.h)
struct GlmDev {
  int64_t N;
  int32_t P, Ppad;          // covariates; width of the register layout = 2 lpr ch >= P (also the length of a workgroup's record)
  int32_t xstride, xpad_;   // stored row length in doubles: P rounded up to even (16-byte rows), NOT Ppad -- a row's last chunks
                            // read on into the next row (whose bytes the pass reads anyway; beta is 0 there and the gradient slots
                            // beyond P are dropped), so the pass moves 8 N P bytes whatever the layout's width
  int32_t family, lpr, ch;  // NUTS_GLM_*; lanes per row (a power of two); 16-byte chunks per lane: Ppad = 2 lpr ch
  int32_t off_beta, off_icpt, off_sigma, tr_sigma, nwg;   // element offsets (off_icpt / off_sigma < 0: none / the constant)
  // beta a DERIVED vector (NUTS_D_DERIVED, off_beta < 0): its values, written by k_derive before the pass, and where the node leaves
  // d logp / d beta for the interpreter to carry on to the variables of the expression (both in the model's data pool)
  const double* beta_buf; double* beta_seed;
  double sigma_c;           // constant sigma (Normal family without a sigma variable)
  double konst;             // parameter-free part of the log-likelihood (Poisson: -sum_i factln(y_i))
  const double* X;          // [N][Ppad]
  const double* Xt;         // [P][N] (small nodes only: what the single-workgroup kernel reads, small_kernel.h; else nullptr)
  const double* y;          // [N]
  double* part;             // [nwg][Ppad + 4] per-workgroup sums: d/dbeta[Ppad], d/dintercept, d/dsigma, logp, (pad)
  double* gdense;           // [n] the node's gradient w.r.t. the constrained values (zero outside its parameters)
  double* lp;               // the node's logp
};

// The adjoint sweep with its operands RESOLVED (k_gsweep_fast, kernels.h; round 6).  Measured with a phase clock
// (tools/sweep_ticks.py): an interpreted instruction of the generic sweep costs ~2 700 cycles forwards and ~3 500 backwards, and what
// it waits for is its LEAF operands -- a gather is an LDS look-up of the data reference, a load of the index, then the loads of the
// position, each after the other (500 - 700 cycles per trip to the L2), again in the reverse sweep.  Here the host lists every
// distinct leaf of a swept factor once (SwLeaf: what to load from where, already resolved to pointers), the kernel fetches ALL of
// them up front with the loads of four leaves in flight together, and the two sweeps run on LDS only: operands are results of
// earlier instructions, constants, or entries of the leaf array; the adjoint of a leaf is one LDS cell that the end of the sweep
// hands to its slot (a gather / a predictor column); a scalar's adjoint goes straight to its broadcast accumulator, push by push, as in
// the generic sweep.  Same arithmetic in the same order.
#define SW_LEAF 6              // operand kind of the rewritten programs: entry `ref` of the element's leaf array
#define SWL_DATA 0             // value = ptr[bcast ? 0 : li]                      (data vector, predictor column)
#define SWL_GATHER 1           // value = transform(q'[voff + (int) ptr[li]])       (var[idx[li]])
#define SWL_VAR 2              // value = transform(q'[voff + (bcast ? 0 : li)])    (a scalar that broadcasts, an element-aligned vector)
// (layouts the kernel reads with wide scalar loads: a leaf's first 32 bytes are what every leaf needs, a factor's first 64 its header)
struct alignas(16) SwLeaf {
  int32_t kind, bcast, voff, transform;
  const double* ptr;           // SWL_DATA: the vector; SWL_GATHER: the index vector (as doubles); SWL_VAR: unused
  int64_t pad;
  double lower, upper;
};
struct alignas(64) SwFactor {
  int32_t f, elem0, size, orphan;
  int32_t leaf0, n_leaves, n2, instr0;   // leaves [leaf0, leaf0 + n_leaves); the first n2 need the position (GATHER / VAR)
  int32_t n_instr, nargs, dist, slot0;   // slot0: first of the factor's n_slots entries in the table of adjoint offsets
  int32_t n_slots, blk0;                 // blk0: the first of the factor's 64-element blocks in the launch's numbering
  double konst;
  nuts_operand arg[4][3];                // the factor's arguments (a, b, c) with their operands rewritten
};
static_assert(sizeof(SwLeaf) == 48 && sizeof(nuts_instr) == 64 && sizeof(nuts_operand) == 16 && sizeof(SwFactor) == 256, "the scalar-driven sweep reads these with wide loads");

// dense node 5: linear predictors read by the factors through NUTS_OP_LIN operands (lin_kernel.h)
#define LIN_MAXUSE 4     // factors that may read one predictor column
#define LIN_CHUNK 2048   // rows per workgroup of the transposed mat-vec (8 per thread)
struct LinCol {
  int32_t transform, n_use;        // transform of the coefficient variable (its constrained value is the coefficient); -1: a derived vector
  double lower, upper;
  int64_t adj_off[LIN_MAXUSE];     // where the sweep of each reading factor leaves d logp / d eta of this column (ModelDev.adj)
  int32_t use_size[LIN_MAXUSE];    // elements of that factor (N, or anything when the predictor has one row and broadcasts)
};
struct LinDev {
  int64_t N;
  int32_t P, K, nchunk, pad;
  const double* Xt;                // [P][N]: X transposed -- a thread per row reads it coalesced, forwards and backwards
  double* eta;                     // [K][N] the predictors at this leaf's position
  double* part;                    // N > 1: [K][P][nchunk] partial sums of X^T adj over row chunks; N == 1: [K] total adjoint of the column
  const int32_t* coef;             // [K][P] where coefficient p of column k lives: element of the raveled vector, or (derived) index into pool
  LinCol col[NUTS_LIN_MAXK];
};
// one entry per coefficient that receives a gradient from the node: target address and the partial sums that make it up, in order
struct LinTarget { double* dst; int32_t src0, n_src; };
struct LinSrc { int32_t lin, k, p, pad; };

// per-workgroup partial record written by the vector kernel, summed (in workgroup order) by the control kernel
#define PART_LP 0
#define PART_BT 1
#define PART_DMU (PART_BT + MAX_BTERMS)
#define PART_DSG (PART_DMU + LOGIT_MAXD)
#define PART_DOT (PART_DSG + LOGIT_MAXD)

struct ModelDev {
  int32_t n, n_vars, n_factors, n_data;
  int32_t n_bterms, n_orphans, n_deferred, nblk;   // (n_orphans: all of them -- the single-workgroup kernel walks every one)
  const double* pool;         // data vectors of the spec
  const int32_t* orphans;     // [n_orphans] factors without an owning variable
  const int32_t* deferred_g;  // [n_deferred][2] (element, variable): global copy of the program's list
  int has_logit, has_mvn;
  RowsDev lg;
  MvnDev mv;
  int has_mix, has_glm;
  MixDev mix;
  GlmDev glm;
  double* part;               // [nblk][part_stride]
  int32_t part_stride, prog_bytes;
  // "lean" control path (see kernels.h): the only deferred elements are the hierarchical-logit node's mu / sigma, so
  // kernel B evaluates everything of them that does not need the cross-workgroup sums and leaves
  // {d logp/dx local part, dx/dq, dlog|J|/dq, p_half} per deferred element here
  const int32_t* csr;         // inverse indices of gathered variables: per (factor, variable) [size + 1] row pointers, then the factor elements
  double* def_loc;            // [2][MAX_DEFERRED][4] (second copy: the group-aligned row pass double-buffers by launch parity)
  int32_t lean_ok, lean_pad;
  // derived vectors (NUTS_D_DERIVED): factor ids and the offset of each one's values in `pool` (its seed follows the values)
  int32_t n_derived, pool_len;   // (pool_len: doubles in `pool`, derived vectors included)
  int32_t derived_f[MAX_DERIVED];
  int64_t derived_off[MAX_DERIVED];
  // the tables above packed into one blob (kernels copy it into LDS: the interpreter then never waits on HBM)
  const char* prog;
  int32_t po_vars, po_cptr, po_contrib, po_factors, po_fbt, po_btvar, po_data, po_deferred, po_instrs, po_pad;
  long long* ticks;           // [64] phase timestamps of the last B / C launch (only written in -DNUTS_KTIMING builds)
  int32_t tick_j;             // restrict the timestamps to leaf j of a doubling (-1: every leaf)
  // A failed PARAMETER check kills the reference's whole factor -- `check_parameters` reduces its conditions with `all`
  // (dist_math.py:68-74, logprob/utils.py:209-225), so where it fails the gradient of EVERY element of that factor is 0.
  // The sampler never looks at a gradient behind logp = -inf; `nuts_model_logp_grad` (the ValueGradFunction call) does:
  // its pass records the factors with a failed check here (mode 1) and, if there are any, a second pass treats those
  // factors as dead for all their elements (mode 2).  Mode 0 (every pass inside a transition): element-wise, nothing recorded.
  int32_t fdead_mode;
  int32_t* fdead;             // [n_factors]
  // gathered adjoints (see GSlot): the factors swept ahead of the leaf's other work, their slots, the adjoint pool
  int32_t n_gsf, n_gs_elems;  // factors; their elements in all
  int32_t po_gsf, po_gslot;   // tables in the program blob
  // Swept factors WITHOUT an owning variable: kernel B used to walk their elements a second time (another forward + reverse sweep per
  // element) for nothing but the factor's log-density and the adjoints of the scalars that broadcast into it.  k_gsweep now leaves
  // both, per workgroup, in gs_part [n_gs_blocks][1 + MAX_BTERMS]; kernel B's workgroup 0 adds the records up in order and walks only
  // orphans [0, n_orphans_b) of the list (the swept ones are sorted behind them).
  int32_t n_gs_blocks, n_orphans_b;
  double* gs_part;
  // resolved-operand sweep (SwFactor above): tables in one global blob the kernel copies into LDS; sw_rows = LDS doubles per thread
  int32_t n_swf, sw_bytes, sw_rows, sw_po_leaf, sw_po_instr, sw_po_slot;
  int32_t sw_max_leaves, sw_max_slots, sw_max_instr, sw_blocks;
  const char* sw_blob;
  int32_t gs_lds_rows, gs_lds_bytes;   // > 0: k_gsweep_lds -- rows (doubles per thread) of the sweep's LDS block: 2 x (longest program) + (most slots); the launch's dynamic LDS in all
  double* adj;
  int32_t n_glong, glong_pad; // long inverse-index lists (GLong): entries, in global memory next to their totals
  const GLong* glong;
  double* adj_red;            // [n_glong]
  // dense node 5 (LinDev): the predictors' tables live in global memory; lin_gdense [n]: the node's gradient w.r.t. the constrained
  // values (zero outside the coefficients), the protocol of the other dense nodes
  int32_t n_lins, n_lin_targets;
  const LinDev* lins;
  const LinTarget* lin_targets;
  const LinSrc* lin_srcs;
  double* lin_gdense;
};

#ifdef NUTS_KTIMING
__device__ __forceinline__ long long tick_now() {   // drains outstanding memory ops first: true phase boundaries
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return (long long)t;
}
#define TICK(md, cond, slot) do { if (cond) (md).ticks[slot] = tick_now(); } while (0)
#else
#define TICK(md, cond, slot) do { } while (0)
#endif

// Agent-scope accesses for data that crosses workgroups INSIDE one launch (group-aligned row pass, rows_ga_kernel.h /
// rows_ga_tree.h): write-through store (visible to every XCD's L2) and an L1-bypassing load.  Both sides of a hand-off use
// them (MI355X guide: "sc1 stores AND sc1 loads"); the flag / counter that orders the hand-off is stored after
// `s_waitcnt vmcnt(0)`.
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
template <bool AGENT>
__device__ __forceinline__ double ld_maybe_agent(const double* p) { return AGENT ? ld_agent(p) : *p; }

// sum_{s in [s0, s1)} base[s * stride], in index order, with up to 8 loads in flight at a time
// (AGENT: the records were written by other workgroups of THIS launch)
template <bool AGENT = false, int BATCH = 8>
__device__ __forceinline__ double sum_strided(const double* base, int stride, int s0, int s1) {
  // BATCH loads are in flight together, the additions are in index order whatever BATCH is (same bits).  A round of loads of
  // records another XCD wrote costs ~1300 cycles: callers that sum dozens of records per thread ask for a batch that covers them.
  double acc = 0.0;
  for (int s = s0; s < s1; s += BATCH) {
    double v[BATCH];
    // unconditional loads from a clamped index (a predicated load would sit in its own branch and serialise)
#pragma unroll
    for (int u = 0; u < BATCH; ++u) v[u] = ld_maybe_agent<AGENT>(base + (int64_t)min(s + u, s1 - 1) * stride);
#pragma unroll
    for (int u = 0; u < BATCH; ++u) acc += (s + u < s1) ? v[u] : 0.0;
  }
  return acc;
}

#define CTL_CHUNKS 8   // the per-workgroup partials are summed in CTL_CHUNKS contiguous chunks, then the chunks in order

// gradient of a deferred element once the cross-workgroup sum S of its share is known (one explicit form, used by
// every kernel that finishes such an element, so that all of them produce the same bits)
__device__ __forceinline__ double deferred_finish(double gx_local, double S, double dxdq, double dj) {
  return fma(gx_local + S, dxdq, dj);
}

#define PROG_LDS_MAX 12288

// The interpreter's view of the model tables (LDS copy when it fits, the global blob otherwise).
struct Prog {
  const VarDev* vars;
  const nuts_instr* instrs;   // expression programs of the factors (nuts_factor.instr_off / n_instr)
  const int32_t* csr;         // inverse indices of the gathered variables (NUTS_OP_GATHER): global memory, see Contrib
  const int32_t* var_cptr;
  const Contrib* contrib;
  const nuts_factor* factors;
  const FactorBT* fbt;
  const int32_t* bterm_var;
  const nuts_data_ref* data;
  const int32_t* deferred;   // [n_deferred][2] = (element, variable)
  const double* pool;
  int n_vars;
  int fdead_mode;            // see ModelDev
  int32_t* fdead;
  const GSweepFactor* gsf;   // gathered adjoints (see GSlot)
  const GSlot* gslot;
  double* adj;
  const double* adj_red;
  const GLong* glong;
  int n_gsf;
  const LinDev* lins;        // linear predictors (NUTS_OP_LIN operands)
  long long* ticks;          // (lab builds, -DNUTS_KTIMING: ModelDev.ticks)
};

__device__ __forceinline__ Prog prog_view(const ModelDev& md, const char* base) {
  Prog pg;
  pg.vars = reinterpret_cast<const VarDev*>(base + md.po_vars);
  pg.var_cptr = reinterpret_cast<const int32_t*>(base + md.po_cptr);
  pg.contrib = reinterpret_cast<const Contrib*>(base + md.po_contrib);
  pg.factors = reinterpret_cast<const nuts_factor*>(base + md.po_factors);
  pg.fbt = reinterpret_cast<const FactorBT*>(base + md.po_fbt);
  pg.bterm_var = reinterpret_cast<const int32_t*>(base + md.po_btvar);
  pg.data = reinterpret_cast<const nuts_data_ref*>(base + md.po_data);
  pg.deferred = reinterpret_cast<const int32_t*>(base + md.po_deferred);
  pg.instrs = reinterpret_cast<const nuts_instr*>(base + md.po_instrs);
  pg.csr = md.csr;
  pg.pool = md.pool;
  pg.n_vars = md.n_vars;
  pg.fdead_mode = md.fdead_mode; pg.fdead = md.fdead;
  pg.gsf = reinterpret_cast<const GSweepFactor*>(base + md.po_gsf);
  pg.gslot = reinterpret_cast<const GSlot*>(base + md.po_gslot);
  pg.adj = md.adj; pg.adj_red = md.adj_red; pg.glong = md.glong; pg.n_gsf = md.n_gsf;
  pg.lins = md.lins;
  pg.ticks = md.ticks;
  return pg;
}

// Cooperative copy of the program blob into LDS (all threads of a 256-thread workgroup; ends with a barrier).
// Split in two so that the global loads are in flight together with the caller's own first loads.
struct ProgRegs { uint4 r[PROG_LDS_MAX / 16 / 256]; };
__device__ __forceinline__ void prog_issue(const ModelDev& md, ProgRegs& pr) {
  const int n16 = (md.prog_bytes + 15) >> 4;
#pragma unroll
  for (int u = 0; u < PROG_LDS_MAX / 16 / 256; ++u)   // unconditional loads from a clamped index
    pr.r[u] = reinterpret_cast<const uint4*>(md.prog)[min((int)threadIdx.x + u * 256, n16 - 1)];
}
__device__ __forceinline__ Prog load_prog(const ModelDev& md, char* s_prog, const ProgRegs& pr) {
  const char* base = md.prog;
  if (md.prog_bytes <= PROG_LDS_MAX) {
    const int n16 = (md.prog_bytes + 15) >> 4;
#pragma unroll
    for (int u = 0; u < PROG_LDS_MAX / 16 / 256; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < n16) reinterpret_cast<uint4*>(s_prog)[i] = pr.r[u];
    }
    __syncthreads();
    base = s_prog;
  }
  return prog_view(md, base);
}

// How a kernel obtains the position vector it evaluates the model at.  In "composed" mode the
// first half of the leapfrog (integration.py:118-127) is recomputed on the fly from the source
// slot of the trajectory arena, so no kernel has to materialise q' before the row-streaming pass.
struct QView {
  const double* q;
  const double* p;
  const double* g;
  const double* var;
  double eps, half;
  int composed;
  // Auxiliary workgroups of the one-launch row passes (rows_aux.h) only: q' of the model's DEFERRED elements, by position in the
  // deferred list.  Those workgroups run next to the control work of the previous leaf, which is still writing the source state's
  // gradient and momentum of exactly these elements, so they compose them from the previous launch's records themselves and the
  // interpreter reads them here instead of through the arena.  nullptr everywhere else (folded away at compile time).
  const double* defq = nullptr;
  __device__ __forceinline__ double p_half(int64_t i) const { return fma(half, g[i], p[i]); }
  __device__ __forceinline__ double at(int64_t i) const {
    if (!composed) return q[i];
    const double ph = fma(half, g[i], p[i]);
    const double v = var[i] * ph;
    return fma(eps, v, q[i]);
  }
};

// value transforms: forward value only / full (x, dx/dq, log|J|, dlog|J|/dq)
// (pymc/logprob/transforms.py:880-891, 1017-1088)
__device__ __forceinline__ double transform_x(const VarDev& v, double qi) {
  switch (v.transform) {
    case NUTS_TR_LOG: return exp(qi);
    case NUTS_TR_LOGODDS: return sigmoid_d(qi);
    case NUTS_TR_INTERVAL: { double s = sigmoid_d(qi); return s * v.upper + (1.0 - s) * v.lower; }
    default: return qi;
  }
}

__device__ __forceinline__ void transform_full(const VarDev& v, double qi, double& x, double& dx, double& lj, double& dj) {
  switch (v.transform) {
    case NUTS_TR_LOG:
      x = exp(qi); dx = x; lj = qi; dj = 1.0;
      break;
    case NUTS_TR_LOGODDS: {
      double s = sigmoid_d(qi);
      x = s; dx = s * (1.0 - s); lj = -softplus_d(-qi) - softplus_d(qi); dj = 1.0 - 2.0 * s;
    } break;
    case NUTS_TR_INTERVAL: {
      double s = sigmoid_d(qi);
      x = s * v.upper + (1.0 - s) * v.lower;
      dx = (v.upper - v.lower) * s * (1.0 - s);
      lj = log(v.upper - v.lower) - 2.0 * softplus_d(-qi) - qi;
      dj = 1.0 - 2.0 * s;
    } break;
    default:
      x = qi; dx = 1.0; lj = 0.0; dj = 0.0;
  }
}

__device__ __forceinline__ int find_var(const Prog& pg, int i) {
  int k = 0;
  while (k + 1 < pg.n_vars && i >= pg.vars[k].offset + pg.vars[k].size) ++k;
  return k;
}

// `own_var` / `own_x`: the calling thread's own element (same local index li) -- its constrained value is
// already in a register, everything else is recomposed through the view.
// DEFQ: the caller is an auxiliary workgroup of a one-launch row pass (rows_aux.h) and qv.defq holds q' of the deferred elements
// (a template parameter, not a run-time test of the pointer: the test alone cost k_small_draw 16 registers and 1.3 KB of scratch)
// OL: a non-identity transform is applied by a CALL (transform_x_ol) instead of inline.  A factor evaluated through the general path
// expands twelve operands, each with its own copy of exp / sigmoid / their divisions: 160 instructions per operand, 18 000 per
// instance of the model evaluation in the single-workgroup kernel -- 42 000 instructions (340 KB) of which a leaf executes a few
// hundred scattered over all of it, every one fetched from L2 (the instruction cache holds 64 KB).  The latency-bound caller takes
// the call; the throughput-bound kernels keep the inline form.
__device__ __noinline__ double transform_x_ol(int transform, double lower, double upper, double qi) {
  switch (transform) {
    case NUTS_TR_LOG: return exp(qi);
    case NUTS_TR_LOGODDS: return sigmoid_d(qi);
    case NUTS_TR_INTERVAL: { double s = sigmoid_d(qi); return s * upper + (1.0 - s) * lower; }
    default: return qi;
  }
}
template <bool OL>
__device__ __forceinline__ double transform_x_sel(const VarDev& v, double qi) {
  if constexpr (OL) {
    if (v.transform != NUTS_TR_LOG && v.transform != NUTS_TR_LOGODDS && v.transform != NUTS_TR_INTERVAL) return qi;
    return transform_x_ol(v.transform, v.lower, v.upper, qi);
  } else {
    return transform_x(v, qi);
  }
}

// LIN: the kernel may meet NUTS_OP_LIN operands (a model with linear predictors always runs the instantiations that carry the
// expression-program interpreter; the lean ones compile the branch out)
template <bool DEFQ = false, bool OL = false, bool LIN = true>
__device__ __forceinline__ double op_value(const nuts_operand& o, int li, const Prog& pg, const QView& qv, int own_var,
                                           double own_x) {
  if (o.kind == NUTS_OP_CONST) return o.c;
  if (o.kind == NUTS_OP_DATA) {
    const nuts_data_ref r = pg.data[o.ref];
    return pg.pool[r.offset + (r.size > 1 ? li : 0)];
  }
  if (o.kind == NUTS_OP_GATHER) {   // var[idx[li]]
    const nuts_data_ref r = pg.data[(int)o.c];
    const VarDev v = pg.vars[o.ref];
    return transform_x_sel<OL>(v, qv.at(v.offset + (int)pg.pool[r.offset + li]));
  }
  if constexpr (LIN) if (o.kind == NUTS_OP_LIN) {      // eta_k[li], written by k_lin_fwd ahead of this kernel (one row: broadcast)
    const LinDev& L = pg.lins[o.ref];
    return L.eta[(int64_t)(int)o.c * L.N + (L.N > 1 ? li : 0)];
  }
  if (o.ref == own_var) return own_x;
  const VarDev v = pg.vars[o.ref];
  if constexpr (DEFQ) { if (v.deferred) return transform_x_sel<OL>(v, qv.defq[v.def_base + (v.size > 1 ? li : 0)]); }
  return transform_x_sel<OL>(v, qv.at(v.offset + (v.size > 1 ? li : 0)));
}

// log-density of one element and its partials w.r.t. each argument.
// (not inlined: one copy of the 18-way switch and its libm expansions per kernel keeps kernels B and C small
// enough for the instruction cache -- they are launched once per leapfrog between two passes of kernel A)
// `pdead` is set when a PARAMETER check failed (the reference's `check_parameters`; the support checks on the value are plain
// element-wise switches there): the caller decides what that means for the factor's other elements (factor_kill below).
// Arguments and results travel BY VALUE (registers under the device calling convention): with pointer parameters the caller's a[4],
// d[4] and the flag had to live in scratch -- 80 to 144 B in every kernel that evaluates factors.
struct DistOut { double lp, d0, d1, d2, d3; int dead; };
__device__ __forceinline__ DistOut dist_eval_body(int dist, double konst, double a0, double a1, double a2, double a3) {
  const double a[4] = {a0, a1, a2, a3};
  double d[4];
  int pdead_v = 0;
  int* const pdead = &pdead_v;
  const double NINF = -INFINITY;
  const double LOG_SQRT_2PI = 0.91893853320467274178;
  const double LOG_SQRT_2_OVER_PI = -0.22579135264472743236;
  const double LOG_PI = 1.14472988584940017414;
  const double LOG_2 = 0.69314718055994530942;
  double lp = 0.0;
  bool dead = false;
#define KILL_UNLESS(cond) if (!(cond)) { lp = NINF; dead = true; }                  /* support of the value: element-wise switch */
#define KILL_PARAM(cond) if (!(cond)) { lp = NINF; dead = true; *pdead = 1; }     /* check_parameters(...) */
  d[0] = d[1] = d[2] = d[3] = 0.0;
  switch (dist) {
    case NUTS_D_NORMAL: {  // continuous.py:526-532
      double sg = a[2], z = (a[0] - a[1]) / sg;
      lp = -0.5 * z * z - LOG_SQRT_2PI - log(sg);
      KILL_PARAM(sg > 0)
      d[0] = -z / sg; d[1] = z / sg; d[2] = (z * z - 1.0) / sg;
    } break;
    case NUTS_D_HALFNORMAL: {  // continuous.py:909-916
      double sg = a[1], z = a[0] / sg;
      lp = -0.5 * z * z + LOG_SQRT_2_OVER_PI - log(sg);
      KILL_UNLESS(a[0] >= 0)
      KILL_PARAM(sg > 0)
      d[0] = -z / sg; d[1] = (z * z - 1.0) / sg;
    } break;
    case NUTS_D_CAUCHY: {  // continuous.py:2287-2293
      double be = a[2], z = (a[0] - a[1]) / be;
      lp = -LOG_PI - log(be) - log1p(z * z);
      KILL_PARAM(be > 0)
      double w = 2.0 * z / (1.0 + z * z);
      d[0] = -w / be; d[1] = w / be; d[2] = (-1.0 + w * z) / be;
    } break;
    case NUTS_D_HALFCAUCHY: {  // continuous.py:2383-2390
      double be = a[1], z = a[0] / be;
      lp = LOG_2 - LOG_PI - log(be) - log1p(z * z);
      KILL_UNLESS(a[0] >= 0)
      KILL_PARAM(be > 0)
      double w = 2.0 * z / (1.0 + z * z);
      d[0] = -w / be; d[1] = (-1.0 + w * z) / be;
    } break;
    case NUTS_D_STUDENTT: {  // continuous.py:1935-1950 (nu constant)
      double nu = a[1], sg = a[3], z = (a[0] - a[2]) / sg;
      lp = konst - log(sg) - (nu + 1.0) / 2.0 * log1p(z * z / nu);
      KILL_PARAM(sg > 0)
      double w = (nu + 1.0) * z / (nu + z * z);
      d[0] = -w / sg; d[2] = w / sg; d[3] = (-1.0 + w * z) / sg;
    } break;
    case NUTS_D_BETA: {  // continuous.py:1248-1262 (alpha, beta constant)
      double v = a[0], al = a[1], be = a[2];
      lp = (al == 1.0 ? 0.0 : (al - 1.0) * log(v)) + (be == 1.0 ? 0.0 : (be - 1.0) * log1p(-v)) + konst;
      d[0] = (al == 1.0 ? 0.0 : (al - 1.0) / v) - (be == 1.0 ? 0.0 : (be - 1.0) / (1.0 - v));
      KILL_UNLESS(v >= 0 && v <= 1)
    } break;
    case NUTS_D_EXPONENTIAL: {  // continuous.py:1478-1486 (mu = 1/lam)
      double v = a[0], lam = a[1];
      lp = log(lam) - v * lam;
      KILL_UNLESS(v >= 0)
      KILL_PARAM(lam > 0)
      d[0] = -lam; d[1] = 1.0 / lam - v;
    } break;
    case NUTS_D_UNIFORM: {  // continuous.py:309-321
      double v = a[0], lo = a[1], hi = a[2];
      lp = -log(hi - lo);
      KILL_UNLESS(v >= lo && v <= hi)
      KILL_PARAM(lo <= hi)
    } break;
    case NUTS_D_BERNOULLI_LOGIT: {  // discrete.py:351-352,362-374
      double y = a[0], eta = a[1];
      lp = (y != 0.0) ? -softplus_d(-eta) : -softplus_d(eta);
      KILL_UNLESS(y >= 0 && y <= 1)
      d[1] = y - sigmoid_d(eta);
    } break;
    case NUTS_D_LOGNORMAL: {  // continuous.py:1807-1819
      double v = a[0], sg = a[2], lv = log(v), z = (lv - a[1]) / sg;
      lp = -0.5 * z * z - LOG_SQRT_2PI - log(sg) - lv;
      KILL_UNLESS(v > 0)
      KILL_PARAM(sg > 0)
      d[0] = (-z / sg - 1.0) / v; d[1] = z / sg; d[2] = (z * z - 1.0) / sg;
    } break;
    case NUTS_D_BERNOULLI: {  // discrete.py:362-374
      double y = a[0], p = a[1];
      lp = (y != 0.0) ? log(p) : log1p(-p);
      d[1] = (y != 0.0) ? 1.0 / p : -1.0 / (1.0 - p);
      KILL_UNLESS(y >= 0 && y <= 1)
      KILL_PARAM(p >= 0 && p <= 1)
    } break;
    case NUTS_D_TRUNCNORMAL: {  // continuous.py:720-746; bounds constant (lower = a[3], upper = konst)
      const double v = a[0], mu = a[1], sg = a[2], lo = a[3], hi = konst;
      const double z = (v - mu) / sg;
      const bool lb = lo > -INFINITY, ub = hi < INFINITY;
      const double za = (lo - mu) / sg, zb = (hi - mu) / sg;   // standardised bounds
      double norm = 0.0;
      if (lb && ub) {   // log_diff_normal_cdf(mu, sigma, upper, lower), dist_math.py:145-183
        const double x = zb / 1.4142135623730951, y = za / 1.4142135623730951;
        double t;
        if (y > 0) t = -y * y + log(erfcx(y) - exp(y * y - x * x) * erfcx(x));
        else if (x < 0) t = -x * x + log(erfcx(-x) - exp(x * x - y * y) * erfcx(-y));
        else t = log(erf(x) - erf(y));
        norm = log(0.5) + t;
      } else if (lb) {  // normal_lccdf(mu, sigma, lower), dist_math.py:136-142
        norm = za > 1.0 ? log(erfcx(za / 1.4142135623730951) / 2.0) - za * za / 2.0 : log1p(-erfc(-za / 1.4142135623730951) / 2.0);
      } else if (ub) {  // normal_lcdf(mu, sigma, upper), dist_math.py:126-133
        norm = zb < -1.0 ? log(erfcx(-zb / 1.4142135623730951) / 2.0) - zb * zb / 2.0 : log1p(-erfc(zb / 1.4142135623730951) / 2.0);
      }
      lp = -0.5 * z * z - LOG_SQRT_2PI - log(sg) - norm;
      KILL_PARAM(sg > 0)
      if (lb) KILL_UNLESS(!(v < lo))
      if (ub) KILL_UNLESS(!(v > hi))
      if (lb && ub) KILL_PARAM(lo <= hi)
      // d norm / d mu = (phi(za) - phi(zb)) / (sigma Z), d norm / d sigma = (za phi(za) - zb phi(zb)) / (sigma Z),
      // with phi / Z taken in log space through `norm` = log Z
      const double ra = lb ? exp(-0.5 * za * za - LOG_SQRT_2PI - norm) : 0.0;
      const double rb = ub ? exp(-0.5 * zb * zb - LOG_SQRT_2PI - norm) : 0.0;
      d[0] = -z / sg;
      d[1] = z / sg - (ra - rb) / sg;
      d[2] = (z * z - 1.0) / sg - ((lb ? za * ra : 0.0) - (ub ? zb * rb : 0.0)) / sg;
    } break;
    case NUTS_D_BINOMIAL: {  // discrete.py:141-154; logpow(x, m) of dist_math.py:92-107
      const double y = a[0], nn = a[1], p = a[2], m2 = nn - y;
      const double lx = log(p), l1 = log1p(-p);
      const bool z1 = lx == NINF && y <= 0, z2 = l1 == NINF && m2 <= 0;
      const double t1 = z1 ? (y == 0 ? 0.0 : NINF) : y * lx;
      const double t2 = z2 ? (m2 == 0 ? 0.0 : NINF) : m2 * l1;
      lp = a[3] + t1 + t2;
      d[2] = (z1 ? 0.0 : y / p) - (z2 ? 0.0 : m2 / (1.0 - p));
      KILL_UNLESS(!(y < 0 || y > nn))
      KILL_PARAM(nn >= 0)
      KILL_PARAM(p >= 0 && p <= 1)
    } break;
    case NUTS_D_GAMMA: {  // continuous.py:2512-2521 (alpha constant; the reference goes through scale = 1/beta)
      const double v = a[0], al = a[1], be = 1.0 / (1.0 / a[2]);
      const double lb = log(be), lv = log(v), m2 = al - 1.0;
      const double t1 = (lb == NINF && al <= 0) ? (al == 0 ? 0.0 : NINF) : al * lb;   // logpow(beta, alpha)
      const bool z2 = lv == NINF && m2 <= 0;
      const double t2 = z2 ? (m2 == 0 ? 0.0 : NINF) : m2 * lv;                           // logpow(value, alpha - 1)
      lp = konst + t1 - be * v + t2;
      d[0] = -be + (z2 ? 0.0 : m2 / v);
      d[2] = al / be - v;
      KILL_UNLESS(v >= 0)
      KILL_PARAM(al > 0)
      KILL_PARAM(be > 0)
    } break;
    case NUTS_D_INVGAMMA: {  // continuous.py:2631-2639 (alpha constant)
      const double v = a[0], al = a[1], be = a[2];
      const double lb = log(be), lv = log(v), m2 = -al - 1.0;
      const double t1 = (lb == NINF && al <= 0) ? (al == 0 ? 0.0 : NINF) : al * lb;
      const double t2 = (lv == NINF && m2 <= 0) ? (m2 == 0 ? 0.0 : NINF) : m2 * lv;
      lp = konst + t1 - be / v + t2;
      d[0] = be / (v * v) + m2 / v;
      d[2] = al / be - 1.0 / v;
      KILL_UNLESS(v >= 0)
      KILL_PARAM(al > 0)
      KILL_PARAM(be > 0)
    } break;
    case NUTS_D_LAPLACE: {  // continuous.py:1570-1576
      const double r = a[0] - a[1], b = a[2];
      const double sg = r > 0 ? 1.0 : (r < 0 ? -1.0 : 0.0);
      lp = -log(2.0 * b) - fabs(r) / b;
      d[0] = -sg / b; d[1] = sg / b; d[2] = -1.0 / b + fabs(r) / (b * b);
      KILL_PARAM(b > 0)
    } break;
    case NUTS_D_POISSON: {  // discrete.py:581-597; factln(y) arrives as data
      const double y = a[0], mu = a[1];
      const double lm = log(mu);
      const bool z = lm == NINF && y <= 0;
      lp = (z ? (y == 0 ? 0.0 : NINF) : y * lm) - a[2] - mu;
      d[1] = (z ? 0.0 : y / mu) - 1.0;
      if (mu == 0 && y == 0) lp = 0.0;
      KILL_UNLESS(!(y < 0))
      KILL_PARAM(mu >= 0)
    } break;
    case NUTS_D_POTENTIAL: {  // pm.Potential: the term itself is the log-density contribution
      lp = a[0];
      d[0] = 1.0;
    } break;
    case NUTS_D_DERIVED: {  // a derived vector (include/nuts_mi355.h): no density of its own; d logp / d element = the seed the
      lp = 0.0;             // dense node that reads the vector left in the factor's third argument
      d[0] = a[2];
    } break;
    default: lp = NAN;
  }
  // every support / parameter check is a `switch(cond, logp, -inf)` in the reference graph
  // (dist_math.py:50-74, logprob/utils.py:209-225): its gradient is 0 where the check fails.
  if (dead) d[0] = d[1] = d[2] = d[3] = 0.0;
#undef KILL_UNLESS
#undef KILL_PARAM
  return DistOut{lp, d[0], d[1], d[2], d[3], pdead_v};
}
__device__ __noinline__ DistOut dist_eval_v(int dist, double konst, double a0, double a1, double a2, double a3) {
  return dist_eval_body(dist, konst, a0, a1, a2, a3);
}
// ... for a caller whose lanes all evaluate the SAME distribution (the scalar-driven sweep, kernels.h k_gsweep_fast): the code arrives in
// a vector register like every argument of a call; read back into a scalar one, the 19-way switch is scalar compares and branches
__device__ __noinline__ DistOut dist_eval_u(int dist, double konst, double a0, double a1, double a2, double a3) {
  return dist_eval_body(__builtin_amdgcn_readfirstlane(dist), konst, a0, a1, a2, a3);
}
__device__ __forceinline__ double dist_eval(int dist, double konst, const double* a, double* d, int* pdead) {
  const DistOut o = dist_eval_v(dist, konst, a[0], a[1], a[2], a[3]);
  d[0] = o.d0; d[1] = o.d1; d[2] = o.d2; d[3] = o.d3;
  if (o.dead) *pdead = 1;
  return o.lp;
}
__device__ __forceinline__ double dist_eval_uniform(int dist, double konst, const double* a, double* d, int* pdead) {
  const DistOut o = dist_eval_u(dist, konst, a[0], a[1], a[2], a[3]);
  d[0] = o.d0; d[1] = o.d1; d[2] = o.d2; d[3] = o.d3;
  if (o.dead) *pdead = 1;
  return o.lp;
}

// Evaluate element `li` of factor `f`: returns its logp, fills d[k] (partials w.r.t. argument k)
// and the b / c operand values of every argument (needed for the chain rule through a + b*c).
template <bool DEFQ = false, bool OL = false, bool LIN = true>
__device__ __forceinline__ double factor_eval(const Prog& pg, const QView& qv, const nuts_factor& f, int li, int own_var,
                                              double own_x, double* d, double* bv, double* cv, int* pdead) {
  double a[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k < f.nargs) {
      const nuts_term& t = f.arg[k];
      const double av = op_value<DEFQ, OL, LIN>(t.a, li, pg, qv, own_var, own_x);
      bv[k] = op_value<DEFQ, OL, LIN>(t.b, li, pg, qv, own_var, own_x);
      cv[k] = op_value<DEFQ, OL, LIN>(t.c, li, pg, qv, own_var, own_x);
      a[k] = av + bv[k] * cv[k];
    } else {
      a[k] = 0.0; bv[k] = cv[k] = 0.0;
    }
  }
  return dist_eval(f.dist, f.konst, a, d, pdead);
}

// what a failed parameter check of factor `f` means for the element just evaluated (ModelDev.fdead_mode)
__device__ __forceinline__ void factor_kill(const Prog& pg, int f, int pdead, double& lpf, double* d) {
  if (pg.fdead_mode == 0) return;
  if (pg.fdead_mode == 1) { if (pdead) pg.fdead[f] = 1; return; }   // (every writer stores the same value)
  if (pg.fdead[f]) { lpf = -INFINITY; d[0] = d[1] = d[2] = d[3] = 0.0; }
}

// ---- expression programs (include/nuts_mi355.h): a factor whose arguments are not plain terms --------------------------------
// Element `li` of a factor WITH a program, in reverse mode: one forward sweep over the instructions (values `tv`), the factor's
// density and its partials w.r.t. the arguments, then ONE reverse sweep that carries the adjoints `ta` back through the
// instructions.  What arrives at a leaf operand that is a variable goes
//   * into `gwrt` when the variable is `wrt` (the calling thread's own element of it: element-aligned with the factor, or the
//     element a NUTS_OP_GATHER operand picked for this factor element), and
//   * when `want_bt` (the caller owns the factor's log-density), into the LDS accumulator of the broadcast term of every scalar
//     variable that occurs in the program --
// so a factor costs one forward and one reverse sweep however many variables it mentions (round 4: one forward-tangent pass per
// variable).  An adjoint that is exactly zero is not propagated: the unselected branch of a `switch` stays out of the gradient
// even where its own partial is infinite.
// (not inlined, and its two scratch arrays are indexed dynamically: only models that carry a program pay for them)
struct ProgFwd { double a[4], bv[4], cv[4]; int pdead; };

__device__ __forceinline__ double prog_op_value(int op, double k, double x, double y, double z) {
  switch (op) {
    case NUTS_E_ADD: return x + y;
    case NUTS_E_SUB: return x - y;
    case NUTS_E_MUL: return x * y;
    case NUTS_E_DIV: return x / y;
    case NUTS_E_NEG: return -x;
    case NUTS_E_EXP: return exp(x);
    case NUTS_E_LOG: return log(x);
    case NUTS_E_LOG1P: return log1p(x);
    case NUTS_E_SIGMOID: return sigmoid_d(x);
    case NUTS_E_SOFTPLUS: return softplus_d(x);
    case NUTS_E_SQRT: return sqrt(x);
    case NUTS_E_SQR: return x * x;
    case NUTS_E_RECIPROCAL: return 1.0 / x;
    case NUTS_E_TANH: return tanh(x);
    case NUTS_E_ABS: return fabs(x);
    case NUTS_E_POWC: return pow(x, k);
    case NUTS_E_GT: return x > y ? 1.0 : 0.0;
    case NUTS_E_GE: return x >= y ? 1.0 : 0.0;
    case NUTS_E_LT: return x < y ? 1.0 : 0.0;
    case NUTS_E_LE: return x <= y ? 1.0 : 0.0;
    case NUTS_E_EQ: return x == y ? 1.0 : 0.0;
    case NUTS_E_NEQ: return x != y ? 1.0 : 0.0;
    case NUTS_E_AND: return (x != 0.0 && y != 0.0) ? 1.0 : 0.0;
    case NUTS_E_OR: return (x != 0.0 || y != 0.0) ? 1.0 : 0.0;
    case NUTS_E_NOT: return x != 0.0 ? 0.0 : 1.0;
    case NUTS_E_SWITCH: return x != 0.0 ? y : z;
    case NUTS_E_GAMMALN: return lgamma(x);
    case NUTS_E_ERF: return erf(x);
    case NUTS_E_ERFC: return erfc(x);
    case NUTS_E_ERFCX: return erfcx(x);
    case NUTS_E_LOG1MEXP: return log1mexp_d(x);
    case NUTS_E_EXPM1: return expm1(x);
    case NUTS_E_SIGN: return x > 0 ? 1.0 : (x < 0 ? -1.0 : (x == 0 ? 0.0 : x));
    case NUTS_E_MAXIMUM: return (x != x || y != y) ? NAN : (x > y ? x : y);
    case NUTS_E_MINIMUM: return (x != x || y != y) ? NAN : (x < y ? x : y);
    case NUTS_E_POW: return pow(x, y);
    case NUTS_E_FLOOR: return floor(x);
    case NUTS_E_CEIL: return ceil(x);
    case NUTS_E_SIN: return sin(x);
    case NUTS_E_COS: return cos(x);
    case NUTS_E_ARCTAN: return atan(x);
    case NUTS_E_LOGADDEXP: return logaddexp_d(x, y);
    case NUTS_E_CLIP: return x < y ? y : (x > z ? z : x);
    case NUTS_E_CHECK: return x;
    case NUTS_E_LOG2: return log2(x);
    case NUTS_E_LOG10: return log10(x);
    case NUTS_E_DIGAMMA: return digamma_d(x);
    default: return NAN;
  }
}

// The reverse rules: what the operands of instruction `op` receive for the adjoint g of its result v (x, y, z: the operands' values,
// read only where a rule needs them) -- one copy, shared by the generic sweep and the resolved-operand sweep (k_gsweep_fast).
__device__ __forceinline__ void prog_op_adjoint(int op, double kconst, double g, double v, double x, double y, double z, double& gx, double& gy,
                                                double& gz, bool& px, bool& py, bool& pz) {
  gx = 0.0; gy = 0.0; gz = 0.0;
  px = false; py = false; pz = false;
  switch (op) {
    case NUTS_E_ADD: gx = g; gy = g; px = py = true; break;
    case NUTS_E_SUB: gx = g; gy = -g; px = py = true; break;
    case NUTS_E_MUL: gx = g * y; gy = g * x; px = py = true; break;
    case NUTS_E_DIV: gx = g / y; gy = -g * v / y; px = py = true; break;
    case NUTS_E_NEG: gx = -g; px = true; break;
    case NUTS_E_EXP: gx = g * v; px = true; break;
    case NUTS_E_LOG: gx = g / x; px = true; break;
    case NUTS_E_LOG1P: gx = g / (1.0 + x); px = true; break;
    case NUTS_E_SIGMOID: gx = g * v * (1.0 - v); px = true; break;
    case NUTS_E_SOFTPLUS: gx = g * sigmoid_d(x); px = true; break;
    case NUTS_E_SQRT: gx = g * 0.5 / v; px = true; break;
    case NUTS_E_SQR: gx = g * 2.0 * x; px = true; break;
    case NUTS_E_RECIPROCAL: gx = -g * v * v; px = true; break;
    case NUTS_E_TANH: gx = g * (1.0 - v * v); px = true; break;
    case NUTS_E_ABS: gx = g * (x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0)); px = true; break;
    case NUTS_E_POWC: gx = g * kconst * pow(x, kconst - 1.0); px = true; break;
    case NUTS_E_SWITCH: if (x != 0.0) { gy = g; py = true; } else { gz = g; pz = true; } break;
    case NUTS_E_GAMMALN: gx = g * digamma_d(x); px = true; break;
    case NUTS_E_ERF: gx = g * 1.1283791670955126 * exp(-x * x); px = true; break;
    case NUTS_E_ERFC: gx = -g * 1.1283791670955126 * exp(-x * x); px = true; break;
    case NUTS_E_ERFCX: gx = g * (2.0 * x * v - 1.1283791670955126); px = true; break;
    case NUTS_E_LOG1MEXP: gx = -g / expm1(-x); px = true; break;
    case NUTS_E_EXPM1: gx = g * (v + 1.0); px = true; break;
    case NUTS_E_MAXIMUM: case NUTS_E_MINIMUM: gx = g; gy = g; px = v == x; py = v == y; break;
    case NUTS_E_POW: gx = g * y * pow(x, y - 1.0); px = true; if (x != 0.0) { gy = g * v * log(x); py = true; } break;
    case NUTS_E_SIN: gx = g * cos(x); px = true; break;
    case NUTS_E_COS: gx = -g * sin(x); px = true; break;
    case NUTS_E_ARCTAN: gx = g / (1.0 + x * x); px = true; break;
    case NUTS_E_LOGADDEXP: gx = g * sigmoid_d(x - y); gy = g * sigmoid_d(y - x); px = py = true; break;
    case NUTS_E_CLIP: if (x < y) { gy = g; py = true; } else if (x > z) { gz = g; pz = true; } else { gx = g; px = true; } break;
    case NUTS_E_CHECK: gx = g; px = true; break;
    case NUTS_E_LOG2: gx = g / (x * 0.6931471805599453094); px = true; break;
    case NUTS_E_LOG10: gx = g / (x * 2.3025850929940456840); px = true; break;
    case NUTS_E_DIGAMMA: gx = g * trigamma_d(x); px = true; break;
    default: break;   // comparisons, logic, sign, floor, ceil: piecewise constant
  }
}

// Where a sweep keeps the instructions' values and adjoints: a thread's own arrays (scratch memory: every kernel but one) or, for the
// adjoint sweep's kernel (k_gsweep_lds), a column of an LDS block -- element i of thread t at base[i * stride + t], conflict-free.
// A sweep is a chain of dependent reads and writes of these arrays; in scratch each is a trip to the L2 (500+ cycles), in LDS ~100.
struct LdsVec {
  double* base; int stride;
  __device__ __forceinline__ double& operator[](int i) const { return base[i * stride]; }
};

// forward sweep: tv[i] for every instruction, then the factor's arguments
// (selects, not d[arg]: a run-time index into the four-entry arrays put all of them in scratch -- 144 B in every kernel that evaluates
// factors)
// (bit masks rather than a select chain: selects between loads of one array are folded back into an indexed load)
__device__ __forceinline__ double pick4(const double* v, int i) {
  unsigned long long r = 0ull;
#pragma unroll
  for (int k = 0; k < 4; ++k) r |= (unsigned long long)__double_as_longlong(v[k]) & (0ull - (unsigned long long)(i == k || (k == 3 && i > 3)));
  return __longlong_as_double((long long)r);
}
// (Code size is what this interpreter is bound by: with the operand fetch -- five operand kinds, three transforms -- inlined at every
// use, one instantiation was 330 KB of code against a 64 KB instruction cache, and an interpreted instruction cost ~4 000 cycles of
// instruction fetch.  The fetch and the adjoint's `push` are therefore inlined at a handful of sites only: loops stay rolled, the
// reverse rules compute what to push and ONE tail pushes it, transforms are calls.)
template <typename TV>
__device__ __forceinline__ void prog_forward(const Prog& pg, const QView& qv, const nuts_factor& f, int li, int own_var, double own_x,
                                             TV tv, ProgFwd& o) {
  auto val = [&](const nuts_operand& q) { return q.kind == NUTS_OP_TMP ? tv[q.ref] : op_value<false, true>(q, li, pg, qv, own_var, own_x); };
  const nuts_instr* ins = pg.instrs + f.instr_off;
  o.pdead = 0;
#pragma unroll 1
  for (int i = 0; i < f.n_instr; ++i) {
    const nuts_instr& I = ins[i];
    const int op = I.op;
    const double x = val(I.x);
    const bool has_y = op <= NUTS_E_DIV || (op >= NUTS_E_GT && op <= NUTS_E_OR) || op == NUTS_E_SWITCH || op == NUTS_E_MAXIMUM || op == NUTS_E_MINIMUM ||
                       op == NUTS_E_POW || op == NUTS_E_LOGADDEXP || op == NUTS_E_CLIP || op == NUTS_E_CHECK;
    const double y = has_y ? val(I.y) : 0.0;
    const double z = (op == NUTS_E_SWITCH || op == NUTS_E_CLIP) ? val(I.z) : 0.0;
    if (op == NUTS_E_CHECK && y == 0.0) o.pdead = 1;
    tv[i] = prog_op_value(op, I.k, x, y, z);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { o.a[k] = 0.0; o.bv[k] = o.cv[k] = 0.0; }
#pragma unroll 1
  for (int k = 0; k < f.nargs; ++k) {
    const nuts_term& tm = f.arg[k];
    const double b = val(tm.b), c = val(tm.c);
    const double a = val(tm.a) + b * c;
    // (selects, not o.a[k]: a run-time index would put the three arrays in scratch)
    o.a[0] = k == 0 ? a : o.a[0]; o.a[1] = k == 1 ? a : o.a[1]; o.a[2] = k == 2 ? a : o.a[2]; o.a[3] = k == 3 ? a : o.a[3];
    o.bv[0] = k == 0 ? b : o.bv[0]; o.bv[1] = k == 1 ? b : o.bv[1]; o.bv[2] = k == 2 ? b : o.bv[2]; o.bv[3] = k == 3 ? b : o.bv[3];
    o.cv[0] = k == 0 ? c : o.cv[0]; o.cv[1] = k == 1 ? c : o.cv[1]; o.cv[2] = k == 2 ? c : o.cv[2]; o.cv[3] = k == 3 ? c : o.cv[3];
  }
}

// value of argument 0 of element `li` (a NUTS_D_DERIVED factor's vector element)
__device__ __noinline__ double factor_arg0_value(const Prog& pg, const QView& qv, const nuts_factor& f, int li) {
  double tv[NUTS_MAX_FACTOR_INSTR];
  ProgFwd o;
  prog_forward(pg, qv, f, li, -1, 0.0, tv, o);
  return o.a[0];
}

// `gs` / `ngs` (the sweep of k_gsweep only): the factor's slots; the adjoint of every gather operand is added up per slot and
// stored at adj[slot.adj_off + li] -- nothing else of the sweep is kept.
template <typename TV>
__device__ __forceinline__ double factor_prog_rev_t(const Prog& pg, const QView& qv, const nuts_factor& f, int fi, int li, int own_var, double own_x,
                                                    int wrt, int want_bt, double* s_bacc, int bstride, double* gwrt_out, int wrt_did,
                                                    const GSlot* gs, int ngs, TV tv, TV ta, TV gadj) {
#ifdef NUTS_KTIMING
  // phases of ONE element's sweep (k_gsweep*, element 0): ticks[32 ..] += {forward, density, arguments' adjoints, reverse, stores}; [38]: sweeps
  const bool tk_ = gs && li == 0 && blockIdx.x == 0 && threadIdx.x == 0;
  long long tk0_ = tk_ ? tick_now() : 0;
#define SWEEP_TICK(slot) do { if (tk_) { const long long t_ = tick_now(); pg.ticks[slot] += t_ - tk0_; tk0_ = t_; } } while (0)
#else
#define SWEEP_TICK(slot) do { } while (0)
#endif
  if (gs) for (int sl = 0; sl < ngs; ++sl) gadj[sl] = 0.0;
  ProgFwd o;
  prog_forward(pg, qv, f, li, own_var, own_x, tv, o);
  SWEEP_TICK(32);
  double d[4];
  int pdead = o.pdead;
  double lp = dist_eval(f.dist, f.konst, o.a, d, &pdead);
  if (o.pdead) { lp = -INFINITY; d[0] = d[1] = d[2] = d[3] = 0.0; }   // a failed NUTS_E_CHECK: as a failed check inside a density
  factor_kill(pg, fi, pdead, lp, d);
  SWEEP_TICK(33);
  double gw = 0.0;
  const FactorBT& bt = pg.fbt[fi];
  auto val = [&](const nuts_operand& q) { return q.kind == NUTS_OP_TMP ? tv[q.ref] : op_value<false, true>(q, li, pg, qv, own_var, own_x); };
  auto push = [&](const nuts_operand& q, double g) {
    if (q.kind == NUTS_OP_TMP) { ta[q.ref] += g; return; }
    if (q.kind != NUTS_OP_VAR && q.kind != NUTS_OP_GATHER && q.kind != NUTS_OP_LIN) return;
    if (gs) {
      // (a predictor column is a slot like a gathered variable: var = -1 - predictor, did = column)
      if (q.kind == NUTS_OP_GATHER || q.kind == NUTS_OP_LIN) {
        const int sv = q.kind == NUTS_OP_LIN ? -1 - q.ref : q.ref;
        for (int sl = 0; sl < ngs; ++sl)
          if (gs[sl].var == sv && gs[sl].did == (int)q.c) { gadj[sl] += g; break; }
      } else if (want_bt) {   // (the sweep owns this factor's log-density: the scalars that broadcast into it are credited here)
        for (int b = 0; b < bt.n; ++b)
          if (pg.bterm_var[bt.e[b].bterm] == q.ref) { s_bacc[bt.e[b].bterm * bstride] += g; break; }
      }
      return;
    }
    if (q.kind == NUTS_OP_LIN) return;   // (outside the sweep: the predictor's adjoint is the sweep's business)
    // `wrt_did` < 0: the caller owns an element-aligned (or scalar) occurrence of `wrt`: its direct operands; >= 0: it stands for the
    // gather of `wrt` through index vector `wrt_did`: those operands only (the others belong to other contributions of the variable)
    if (q.ref == wrt && (wrt_did < 0 ? q.kind == NUTS_OP_VAR : (q.kind == NUTS_OP_GATHER && (int)q.c == wrt_did))) { gw += g; return; }
    if (want_bt && q.kind == NUTS_OP_VAR)
      for (int b = 0; b < bt.n; ++b)
        if (pg.bterm_var[bt.e[b].bterm] == q.ref) { s_bacc[bt.e[b].bterm * bstride] += g; break; }
  };
  for (int i = 0; i < f.n_instr; ++i) ta[i] = 0.0;
#pragma unroll 1
  for (int k = 0; k < f.nargs; ++k) {
    const double dk = pick4(d, k);
    if (dk != 0.0) {
      const nuts_term& tm = f.arg[k];
      push(tm.a, dk); push(tm.b, dk * pick4(o.cv, k)); push(tm.c, dk * pick4(o.bv, k));
    }
  }
  SWEEP_TICK(34);
  const nuts_instr* ins = pg.instrs + f.instr_off;
  // operands whose VALUE a reverse rule reads (bit i: opcode i)
  constexpr unsigned long long NEED_X =
      (1ull << NUTS_E_MUL) | (1ull << NUTS_E_LOG) | (1ull << NUTS_E_LOG1P) | (1ull << NUTS_E_SOFTPLUS) | (1ull << NUTS_E_SQR) | (1ull << NUTS_E_ABS) |
      (1ull << NUTS_E_POWC) | (1ull << NUTS_E_SWITCH) | (1ull << NUTS_E_GAMMALN) | (1ull << NUTS_E_ERF) | (1ull << NUTS_E_ERFC) | (1ull << NUTS_E_ERFCX) |
      (1ull << NUTS_E_LOG1MEXP) | (1ull << NUTS_E_MAXIMUM) | (1ull << NUTS_E_MINIMUM) | (1ull << NUTS_E_POW) | (1ull << NUTS_E_SIN) | (1ull << NUTS_E_COS) |
      (1ull << NUTS_E_ARCTAN) | (1ull << NUTS_E_LOGADDEXP) | (1ull << NUTS_E_CLIP) | (1ull << NUTS_E_LOG2) | (1ull << NUTS_E_LOG10) | (1ull << NUTS_E_DIGAMMA);
  constexpr unsigned long long NEED_Y = (1ull << NUTS_E_MUL) | (1ull << NUTS_E_DIV) | (1ull << NUTS_E_MAXIMUM) | (1ull << NUTS_E_MINIMUM) | (1ull << NUTS_E_POW) |
                                        (1ull << NUTS_E_LOGADDEXP) | (1ull << NUTS_E_CLIP);
#pragma unroll 1
  for (int i = f.n_instr - 1; i >= 0; --i) {
    const double g = ta[i];
    if (g == 0.0) continue;
    const nuts_instr& I = ins[i];
    const double v = tv[i];
    const int op = I.op;
    const double x = ((NEED_X >> op) & 1ull) ? val(I.x) : 0.0;
    const double y = ((NEED_Y >> op) & 1ull) ? val(I.y) : 0.0;
    const double z = op == NUTS_E_CLIP ? val(I.z) : 0.0;
    double gx, gy, gz;
    bool px, py, pz;
    prog_op_adjoint(op, I.k, g, v, x, y, z, gx, gy, gz, px, py, pz);
    if (px) push(I.x, gx);
    if (py) push(I.y, gy);
    if (pz) push(I.z, gz);
  }
  SWEEP_TICK(35);
  if (gs) for (int sl = 0; sl < ngs; ++sl) pg.adj[gs[sl].adj_off + li] = gadj[sl];
  SWEEP_TICK(36);
#ifdef NUTS_KTIMING
  if (tk_) pg.ticks[38] += 1;
#endif
#undef SWEEP_TICK
  *gwrt_out = gw;
  return lp;
}
__device__ __noinline__ double factor_prog_rev(const Prog& pg, const QView& qv, const nuts_factor& f, int fi, int li, int own_var, double own_x,
                                               int wrt, int want_bt, double* s_bacc, int bstride, double* gwrt_out, int wrt_did = -1,
                                               const GSlot* gs = nullptr, int ngs = 0) {
  double tv[NUTS_MAX_FACTOR_INSTR], ta[NUTS_MAX_FACTOR_INSTR];
  double gadj[MAX_GSLOTS];
  return factor_prog_rev_t<double*>(pg, qv, f, fi, li, own_var, own_x, wrt, want_bt, s_bacc, bstride, gwrt_out, wrt_did, gs, ngs, tv, ta, gadj);
}
// ... the adjoint sweep with its arrays in LDS (k_gsweep_lds: `tv`, `ta`, `gadj` are this thread's columns)
__device__ __noinline__ double factor_sweep_lds(const Prog& pg, const QView& qv, const nuts_factor& f, int fi, int li, int want_bt, double* s_bacc,
                                                int bstride, const GSlot* gs, int ngs, LdsVec tv, LdsVec ta, LdsVec gadj) {
  double gw;
  return factor_prog_rev_t<LdsVec>(pg, qv, f, fi, li, -1, 0.0, -1, want_bt, s_bacc, bstride, &gw, -1, gs, ngs, tv, ta, gadj);
}

// total of one long inverse-index list (whole workgroup; `sm`: blockDim.x / 64 doubles); the result is valid in thread 0
__device__ __forceinline__ double gadj_long_total(const ModelDev& md, const GLong& L, double* sm) {
  const int32_t* lst = md.csr + L.lst_off;
  const double* adj = md.adj + L.adj_off;
  double acc = 0.0;
  const int nt = (int)blockDim.x;
  for (int t0 = threadIdx.x; t0 < L.len; t0 += 8 * nt) {   // (eight loads in flight per thread, added in index order)
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = adj[lst[min(t0 + u * nt, L.len - 1)]];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (t0 + u * nt < L.len) ? v[u] : 0.0;
  }
  return block_sum<false>(acc, sm);
}

// element `e` (in the sweep's numbering) of the factors with gathered adjoints: one forward + reverse sweep, the slots' adjoints stored
// ACCOUNT (k_gsweep): returns the element's log-density and credits the factor's broadcast scalars (`s_bacc`) when no variable owns
// the factor; the single-workgroup kernel keeps walking its orphans itself and passes false
template <bool ACCOUNT = false>
__device__ __forceinline__ double gsweep_element(const Prog& pg, const QView& qv, int e, double* s_bacc = nullptr, int bstride = 0) {
  int t = 0;
  while (t + 1 < pg.n_gsf && e >= pg.gsf[t + 1].elem0) ++t;
  const GSweepFactor sf = pg.gsf[t];
  double gw;
  const int acct = ACCOUNT && sf.orphan;
  const double lp = factor_prog_rev(pg, qv, pg.factors[sf.f], sf.f, e - sf.elem0, -1, 0.0, -1, acct, s_bacc, bstride, &gw, -1, pg.gslot + sf.slot0, sf.n_slots);
  return acct ? lp : 0.0;
}

template <bool ACCOUNT>
__device__ __forceinline__ double gsweep_element_lds(const Prog& pg, const QView& qv, int e, double* s_bacc, int bstride, LdsVec tv, LdsVec ta, LdsVec gadj) {
  int t = 0;
  while (t + 1 < pg.n_gsf && e >= pg.gsf[t + 1].elem0) ++t;
  const GSweepFactor sf = pg.gsf[t];
  const int acct = ACCOUNT && sf.orphan;
  const double lp = factor_sweep_lds(pg, qv, pg.factors[sf.f], sf.f, e - sf.elem0, acct, s_bacc, bstride, pg.gslot + sf.slot0, sf.n_slots, tv, ta, gadj);
  return acct ? lp : 0.0;
}

__device__ __forceinline__ double dot4(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__device__ __forceinline__ double slot_grad(const double* d, const double* bv, const double* cv, int arg, int slot) {
  const double dd = pick4(d, arg);
  return slot == 0 ? dd : (slot == 1 ? dd * pick4(cv, arg) : dd * pick4(bv, arg));
}

// Reverse-mode gather for element i (local index li) of variable k: d logp / d x_i from every factor the
// variable appears in, plus (for owning contributions) the factor's logp and its broadcast terms.
// `s_bacc` is the calling thread's column of the LDS broadcast accumulators ([MAX_BTERMS][blockDim]).
// PROG = false: the model carries neither an expression program nor a gathered operand (the host picks the instantiation,
// nuts_model::has_prog) -- the call into the out-of-line interpreter and everything it keeps alive across the call are compiled
// out: with it k_small_draw<1024> spilled 357 registers and ran 59 us per leapfrog at n = 1002 instead of 24.
template <bool PROG = true, bool DEFQ = false, bool OL = false>
__device__ __forceinline__ void gather_element(const Prog& pg, const QView& qv, int k, int li, double x, double& gx, double& lp,
                                               double* s_bacc, int bstride) {
  for (int c = pg.var_cptr[k]; c < pg.var_cptr[k + 1]; ++c) {
    const Contrib cb = pg.contrib[c];
    if (cb.fast == 2) {   // value ~ Normal(p[1], sigma) or HalfNormal(sigma): no division, no log
      const bool half = cb.dist == NUTS_D_HALFNORMAL;
      const double r = half ? x : x - cb.p[1];
      const double z = r * cb.p[2];
      double lpf = -0.5 * z * z - cb.p[3] + (half ? -0.22579135264472743236 : -0.91893853320467274178);
      double g = -z * cb.p[2];
      if (half && !(x >= 0)) { lpf = -INFINITY; g = 0.0; }
      gx += g;
      if (cb.owner) lp += lpf;
      continue;
    }
    if (cb.fast) {
      double a[4] = {cb.p[0], cb.p[1], cb.p[2], cb.p[3]}, d[4];
      a[0] = cb.arg == 0 ? x : a[0]; a[1] = cb.arg == 1 ? x : a[1]; a[2] = cb.arg == 2 ? x : a[2]; a[3] = cb.arg == 3 ? x : a[3];
      int pdead = 0;
      double lpf = dist_eval(cb.dist, cb.konst, a, d, &pdead);
      factor_kill(pg, cb.f, pdead, lpf, d);
      gx += cb.arg == 0 ? d[0] : (cb.arg == 1 ? d[1] : (cb.arg == 2 ? d[2] : d[3]));
      if (cb.owner) lp += lpf;
      continue;
    }
    const nuts_factor& f = pg.factors[cb.f];
    if constexpr (PROG) {
    if (cb.arg == -2) {    // gathered into the factor: every factor element that indexes this element, in index order
      const int32_t* ptr = pg.csr + cb.dist;
      const int32_t* lst = pg.csr + cb.pad;
      if (cb.p[2] != 0.0) {   // the factor's elements have been swept (k_gsweep): their adjoints of this (variable, index vector) pair
        if (cb.p[3] >= 0.0) {   // ... and some of this variable's lists are long: totalled by k_gadj_reduce (csr: per element, the entry or -1)
          const int lg_ = (pg.csr + (int64_t)cb.p[3])[li];
          if (lg_ >= 0) { gx += sum_strided(pg.adj_red + lg_, 1, 0, pg.glong[lg_].nchunk); continue; }   // (the chunks' totals, in order)
        }
        const double* adj = pg.adj + (int64_t)cb.p[1];
        const int t1 = ptr[li + 1];
        for (int t = ptr[li]; t < t1; t += 8) {   // (eight loads in flight, added in index order)
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = adj[lst[min(t + u, t1 - 1)]];
#pragma unroll
          for (int u = 0; u < 8; ++u) gx += (t + u < t1) ? v[u] : 0.0;
        }
        continue;
      }
      for (int t = ptr[li]; t < ptr[li + 1]; ++t) {
        double gw;
        factor_prog_rev(pg, qv, f, cb.f, lst[t], -1, 0.0, k, 0, s_bacc, bstride, &gw, (int)cb.p[0]);
        gx += gw;
      }
      continue;
    }
    if (f.n_instr > 0) {   // expression program: one forward + one reverse sweep; the owner also collects the factor's logp and
      double gw;           // the adjoints of the scalars that broadcast into it
      const double lpf = factor_prog_rev(pg, qv, f, cb.f, li, k, x, k, cb.owner, s_bacc, bstride, &gw);
      gx += gw;
      if (cb.owner) lp += lpf;
      continue;
    }
    }
    double d[4], bv[4], cv[4];
    int pdead = 0;
    double lpf = factor_eval<DEFQ, OL, PROG>(pg, qv, f, li, k, x, d, bv, cv, &pdead);
    factor_kill(pg, cb.f, pdead, lpf, d);
    gx += slot_grad(d, bv, cv, cb.arg, cb.slot);
    if (cb.owner) {
      lp += lpf;
      const FactorBT& bt = pg.fbt[cb.f];
      for (int b = 0; b < bt.n; ++b) s_bacc[bt.e[b].bterm * bstride] += slot_grad(d, bv, cv, bt.e[b].arg, bt.e[b].slot);
    }
  }
}

// One element of a factor WITHOUT an owning variable (only scalars and data): its logp and the broadcast terms of its scalars.
// Shared by the orphan loops of kernel B (kernels.h) and of the single-workgroup kernel (small_kernel.h).
template <bool PROG = true, bool DEFQ = false, bool OL = false>
__device__ __forceinline__ double orphan_element(const Prog& pg, const QView& qv, int fi, int li, double* s_bacc, int bstride) {
  const nuts_factor& f = pg.factors[fi];
  const FactorBT& bt = pg.fbt[fi];
  if constexpr (PROG) if (f.n_instr > 0 || f.pad) {   // (pad != 0: the spec compiler marks factors with gathered operands; they take the general evaluator)
    double gw;
    return factor_prog_rev(pg, qv, f, fi, li, -1, 0.0, -1, 1, s_bacc, bstride, &gw);
  }
  double dv[4], bv[4], cv[4];
  int pdead = 0;
  double lpo = factor_eval<DEFQ, OL, PROG>(pg, qv, f, li, -1, 0.0, dv, bv, cv, &pdead);
  factor_kill(pg, fi, pdead, lpo, dv);
  for (int b = 0; b < bt.n; ++b) s_bacc[bt.e[b].bterm * bstride] += slot_grad(dv, bv, cv, bt.e[b].arg, bt.e[b].slot);
  return lpo;
}

// Group-aligned row pass for the hierarchical Bernoulli-logit node: ONE launch per leapfrog step.
//
// The span-partitioned pass (rows_kernel.h) balances rows over waves regardless of the group structure, which leaves
// the gradient of a group's z elements scattered over several workgroups; an O(n) kernel (kernel B, kernels.h) has to
// run between two row passes to combine them, kick the momenta and form the tree's dot products -- 10 us of pure
// latency per leapfrog on the benchmark shape (20 % of the step).  When the groups are large and balanced (C2-L: 1248
// groups x 4000 rows) the pass is partitioned BY GROUP instead:
//
//   * workgroup g (W waves) streams exactly the rows of group g.  X is tiled per group ([tiles][D][SPAN], last tile of a
//     group zero-padded and masked), so no tile is shared between workgroups and there are no "mixed" spans; all G
//     workgroups are resident at once (W is chosen so that G W waves fit the chip).
//   * after its last tile the workgroup holds the complete d logp / d beta_g (W wave partials combined through LDS in
//     wave order) and finishes the group's D z elements itself: gradient, second half kick p' = p_half + eps/2 g',
//     v' = M^-1 p', q' store, and the tree-merge dot products of `leaf_post` (kernels.h) restricted to its D elements.
//   * what must cross workgroups is small: per group one record {logp, d/dmu[D], d/dsigma[D], dots}.  Records are
//     written write-through (agent-scope stores), the workgroup takes a ticket on its block's arrival counter (blocks of
//     ga_bsz consecutive groups), and the block's last arriver sums the block's records in group order into a block
//     partial.  No floating-point atomics, no dependence on arrival order: the sums are in fixed order, the ticket only
//     decides WHO does them.  (Hand-off form: MI355X guide, "sc1 payload -> s_waitcnt vmcnt(0) -> ticket; consumer reads
//     with agent-scope loads".)
//   * the next leaf's launch starts from the ga_nblk block partials exactly as it used to start from kernel B's
//     per-workgroup partials: every wave finishes mu' / sigma' in its prologue (rows_hyper_fold_elem), workgroup 0
//     runs the control work of the previous leaf (control_lean) -- both read the block partials of the PREVIOUS launch's
//     parity, this launch writes the other one.
//
// Arena writes of a speculative leaf (one that starts while its predecessor's control work decides to stop the tree) go
// to the slot of a state nobody will read; arrival counters reset themselves (last arriver) and, because the `aborted`
// flag can flip while a launch is in flight, once more at the end of every draw (k_draw_finish).
#pragma once
#include "rows_kernel.h"

// the local part of a hyper-parameter element: what kernel B evaluates through the interpreter on the lean path
// (transform_full + its own prior), here in closed form (the spec compiler only selects this path for mu ~ Normal,
// sigma ~ HalfNormal with constant parameters)
__device__ __forceinline__ void ga_def_local(const RowsDev& R, bool is_mu, double qn, double& gx, double& dxdq, double& dj, double& lp) {
  if (is_mu) {
    const double r = qn - R.mu_c[0];
    const double z = r * R.mu_c[1];
    lp = -0.5 * z * z - R.mu_c[2] + (-0.91893853320467274178);
    gx = -z * R.mu_c[1]; dxdq = 1.0; dj = 0.0;
  } else {
    double x, lj;
    if (R.sigma_tr == NUTS_TR_LOG) { x = exp(qn); dxdq = x; lj = qn; dj = 1.0; }   // transforms.py:880-891
    else { x = qn; dxdq = 1.0; lj = 0.0; dj = 0.0; }
    const double z = x * R.sg_c[0];
    double lpf = -0.5 * z * z - R.sg_c[1] + (-0.22579135264472743236);
    double g = -z * R.sg_c[0];
    if (!(x >= 0)) { lpf = -INFINITY; g = 0.0; }                                   // continuous.py:909-916 support check
    gx = g; lp = lj + lpf;
  }
}

// X tile at element offset `xoff` of Xt ([D][SPAN] doubles; its y bytes at xoff / D): lane l holds rows RPL*l .. RPL*l+RPL-1
template <int D, int RPL>
__device__ __forceinline__ void ga_load(const RowsDev& R, int64_t xoff, int lane, double (&x)[D][RPL], uint32_t& ybits) {
  constexpr int SPAN = WAVE * RPL;
  const double* tile = R.Xt + xoff + lane * RPL;
  const int8_t* yp = R.y + xoff / D + lane * RPL;
  if (RPL == 2) ybits = *reinterpret_cast<const uint16_t*>(yp);
  else ybits = *reinterpret_cast<const uint32_t*>(yp);
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int k = 0; k < RPL; k += 2) {
      const double2 a = *reinterpret_cast<const double2*>(tile + d * SPAN + k);
      x[d][k] = a.x; x[d][k + 1] = a.y;
    }
  }
}

// ---- hand-counted tile loads (D = 8, two rows per lane) ----------------------------------------------------------------------
// hipcc cannot keep a tile in flight across the loop: it drains every outstanding load (`s_waitcnt vmcnt(0)`) at the loop header
// before it lets the next request go (the destination registers of the buffers double as its address temporaries), so the
// stream ran with ONE tile per wave in flight however deep the source-level pipeline was.  The nine loads of a tile are
// therefore issued from one asm statement (scalar base + per-lane offset, no vector address arithmetic) and waited for with a
// counted `vmcnt(N)`, N = 9 x (tiles requested after the one needed).  Form (ii) of the guide's inline-asm rules: "=&v" loads,
// and a wait statement that names every destination "+v" ahead of the first consumer.
typedef double ga_v2d __attribute__((ext_vector_type(2)));
struct GaTileRegs { ga_v2d c[8]; uint32_t y; };

__device__ __forceinline__ void ga_issue8(const double* base /* wave-uniform: first element of the tile */, const int8_t* ybase,
                                          uint32_t voff16, uint32_t voff2, GaTileRegs& t) {
  const double* base2 = base + 512;   // columns 4 .. 7 (the immediate offset field is 13 bits)
  asm volatile(
      "s_nop 4\n\t"
      "global_load_dwordx4 %0, %9, %10\n\t"
      "global_load_dwordx4 %1, %9, %10 offset:1024\n\t"
      "global_load_dwordx4 %2, %9, %10 offset:2048\n\t"
      "global_load_dwordx4 %3, %9, %10 offset:3072\n\t"
      "global_load_dwordx4 %4, %9, %11\n\t"
      "global_load_dwordx4 %5, %9, %11 offset:1024\n\t"
      "global_load_dwordx4 %6, %9, %11 offset:2048\n\t"
      "global_load_dwordx4 %7, %9, %11 offset:3072\n\t"
      "global_load_ushort %8, %12, %13"
      : "=&v"(t.c[0]), "=&v"(t.c[1]), "=&v"(t.c[2]), "=&v"(t.c[3]), "=&v"(t.c[4]), "=&v"(t.c[5]), "=&v"(t.c[6]), "=&v"(t.c[7]), "=&v"(t.y)
      : "v"(voff16), "s"(base), "s"(base2), "v"(voff2), "s"(ybase)
      : "memory");
}

template <int N>
__device__ __forceinline__ void ga_wait8(GaTileRegs& t) {   // at most N loads may still be outstanding; `t` is complete afterwards
  asm volatile("s_waitcnt vmcnt(%9)"
               : "+v"(t.c[0]), "+v"(t.c[1]), "+v"(t.c[2]), "+v"(t.c[3]), "+v"(t.c[4]), "+v"(t.c[5]), "+v"(t.c[6]), "+v"(t.c[7]), "+v"(t.y)
               : "i"(N));
}

// ---- the same with SEVEN stored columns: column 0 of X is identically 1 (the intercept) and is not stored ----
// Detected when the model is built (engine.hip): the tiles are [7][SPAN], 7 KiB instead of 8, and the kernel multiplies by the
// literal 1.0 -- fma(1, beta_0, 0) and fma(r, 1, acc) are what the stored column would have produced, bit for bit.  The row
// pass moves 57 B per row instead of 65 (the ALGORITHMIC figure of SURVEY 8d stays 69: bench.py prices both).
struct GaTileRegs7 { ga_v2d c[7]; uint32_t y; };

__device__ __forceinline__ void ga_issue8(const double* base, const int8_t* ybase, uint32_t voff16, uint32_t voff2, GaTileRegs7& t) {
  const double* base2 = base + 512;   // stored columns 4 .. 6
  asm volatile(
      "s_nop 4\n\t"
      "global_load_dwordx4 %0, %8, %9\n\t"
      "global_load_dwordx4 %1, %8, %9 offset:1024\n\t"
      "global_load_dwordx4 %2, %8, %9 offset:2048\n\t"
      "global_load_dwordx4 %3, %8, %9 offset:3072\n\t"
      "global_load_dwordx4 %4, %8, %10\n\t"
      "global_load_dwordx4 %5, %8, %10 offset:1024\n\t"
      "global_load_dwordx4 %6, %8, %10 offset:2048\n\t"
      "global_load_ushort %7, %11, %12"
      : "=&v"(t.c[0]), "=&v"(t.c[1]), "=&v"(t.c[2]), "=&v"(t.c[3]), "=&v"(t.c[4]), "=&v"(t.c[5]), "=&v"(t.c[6]), "=&v"(t.y)
      : "v"(voff16), "s"(base), "s"(base2), "v"(voff2), "s"(ybase)
      : "memory");
}

template <int N>
__device__ __forceinline__ void ga_wait8(GaTileRegs7& t) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(t.c[0]), "+v"(t.c[1]), "+v"(t.c[2]), "+v"(t.c[3]), "+v"(t.c[4]), "+v"(t.c[5]), "+v"(t.c[6]), "+v"(t.y)
               : "i"(N));
}

// tile registers -> the [8][2] operand of ga_tile
__device__ __forceinline__ void ga_unpack(const GaTileRegs& t, double (&xx)[8][2]) {
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) { xx[dd][0] = t.c[dd].x; xx[dd][1] = t.c[dd].y; }
}
__device__ __forceinline__ void ga_unpack(const GaTileRegs7& t, double (&xx)[8][2]) {
  xx[0][0] = 1.0; xx[0][1] = 1.0;
#pragma unroll
  for (int dd = 1; dd < 8; ++dd) { xx[dd][0] = t.c[dd - 1].x; xx[dd][1] = t.c[dd - 1].y; }
}
template <int DX> struct GaTileSel { typedef GaTileRegs type; };
template <> struct GaTileSel<7> { typedef GaTileRegs7 type; };

// one tile of SPAN = 64 RPL rows: forward (eta, log-lik) + backward (d/dbeta) in registers
// ILV: the rows of a lane may be interleaved by the scheduler (rows_gal_kernel.h: a dependent fp64 fma issues every ~24 cycles, and
// a wave that holds ONE tile has the registers for two rows' temporaries; per row the operations and their order are the same)
template <int D, int RPL, bool ILV = false>
__device__ __forceinline__ void ga_tile(const double (&x)[D][RPL], uint32_t yb, const double (&beta)[D], int nvalid, int lane,
                                        double (&acc)[D], double& lp) {
  if constexpr (ILV && RPL == 2) {
    double eta[2] = {0.0, 0.0}, yk[2], l[2], rr[2];
#pragma unroll
    for (int d = 0; d < D; ++d) { eta[0] = fma(x[d][0], beta[d], eta[0]); eta[1] = fma(x[d][1], beta[d], eta[1]); }
    yk[0] = (double)(yb & 0xffu); yk[1] = (double)((yb >> 8) & 0xffu);
    logit_row2(eta, yk, l, rr);
    // (row 0's contributions enter the lane's sums before row 1's, as in the loop below)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool in = lane * 2 + k < nvalid;
      lp += in ? l[k] : 0.0;
      rr[k] = in ? rr[k] : 0.0;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { acc[d] = fma(rr[0], x[d][0], acc[d]); acc[d] = fma(rr[1], x[d][1], acc[d]); }
    return;
  }
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    double eta = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) eta = fma(x[d][k], beta[d], eta);
    double l, rr;
    logit_row(eta, (double)((yb >> (8 * k)) & 0xffu), l, rr);
    const bool in = lane * RPL + k < nvalid;   // (only the zero-padded last tile of a group has rows to mask)
    lp += in ? l : 0.0;
    rr = in ? rr : 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = fma(rr, x[d][k], acc[d]);
    // rows are finished one after the other: interleaving the two rows of a lane would double the live temporaries of the
    // exp / log1p / reciprocal sequences, and the registers are better spent on tiles in flight
    if constexpr (!ILV) __builtin_amdgcn_sched_barrier(0);
  }
}

#define GA_MAXW 4
#define GA_MAXCHUNK 8      // the block reduce sums chunks of 8 consecutive groups (ga_bsz <= 64)
#define GA_F_NOTAIL 1      // ga_flags, TIMING EXPERIMENTS ONLY: stop after the stream (results are wrong)

// OCC: waves per SIMD the register budget is sized for (4: 128 VGPRs, 3: 168); PIPE: tiles in flight per wave (2 or 3).
// G workgroups of W waves must all be resident: 4 OCC waves per CU >= W (G / CUs).
struct GaArgs;
template <int D>
__device__ __forceinline__ void ga_tail(const GaArgs& T, int g, double (&s_acc)[4][2][D + 1], double (&s_red)[NDOT],
                                        double (&s_cp)[8][PART_STRIDE], int (&s_info)[4], double (&s_keep)[5][WAVE]);

// The arguments travel as ONE struct: the tail of the kernel reads them from an LDS copy of the kernarg segment (made while the
// first tile is in flight), so that nothing the tail needs -- a few dozen pointers and scalars -- has to stay in scalar
// registers across the streaming loop.  (They did: 100+ SGPRs, spilled into VGPR lanes, spilled on into scratch; and a scratch
// reload inside the loop waits, in order, for every tile load issued before it -- the prefetch was gone.)
struct GaArgs {
  ModelDev md; ArenaDev A; EvalIO io;
  int j, rev, fold, par, d, max_depth;
  double Emax;
  HostStatus* st;
  // `fold` bit 0: workgroup 0 does control work -- of leaf (cio, cj, cd), publishing sequence number cseq when it is the last
  // leaf of a doubling.  Inside a doubling that is leaf j - 1 of the same doubling; the FIRST leaf of a doubling queued by the
  // host's look-ahead carries the control work of the previous doubling's last leaf (which would otherwise be a launch of its
  // own, 10-35 us between two row passes).  `fold` bit 1: the source state of this leaf is the leaf of the previous launch, so
  // mu' / sigma' come from that launch's block partials (always so inside a doubling; across doublings only when the tree keeps
  // growing on the same side -- otherwise the source is an older edge state whose entries are complete in the arena).
  EvalIO cio;
  int cj, cd, cseq, cpad;
};
#define GA_FOLD_CTL 1
#define GA_FOLD_SRC 2
#include "rows_aux.h"

// mu', sigma' of this leaf: lane l gets q' (`hval0`) and p_half (`hph0`) of hyper-parameter element l mod 2D.  GA_FOLD_SRC: the source
// state is the leaf of the previous launch, whose mu / sigma gradient is finished here from that launch's records.
template <int D>
__device__ __forceinline__ void ga_hyper(const ModelDev& md, const QView& qv, int fold, int par, int lane, double& hval0, double& hph0) {
  const RowsDev& R = md.lg;
  if (fold & GA_FOLD_SRC) {
    const LeanSrc prev = lean_src(md, par ^ 1);
    rows_hyper_fold_elem<D>(R, prev.part, prev.stride, prev.nblk, prev.def_loc, qv, lane, hval0, hph0);
  } else {
    const int e = lane % (2 * D);
    const int i = e < D ? R.off_mu + e : R.off_sigma + (e - D);
    if (qv.composed) { hph0 = qv.p_half(i); hval0 = fma(qv.eps, qv.var[i] * hph0, qv.q[i]); }
    else { hph0 = 0.0; hval0 = qv.q[i]; }
  }
}

// DX: stored columns of X per tile (D, or D - 1 = 7 when column 0 is identically 1; only with D = 8)
template <int D, int RPL, int OCC, int PIPE, int DX = D>
__global__ __launch_bounds__(64 * GA_MAXW, OCC) void k_rows_ga(GaArgs a) {
  constexpr int SPAN = WAVE * RPL;
  static_assert(DX == D || (D == 8 && DX == 7), "only the intercept column of an 8-column model is elided");
  typedef typename GaTileSel<DX>::type Tile;
  constexpr int LOADS = DX + 1;   // loads per tile: the stored columns + y
  const ModelDev& md = a.md;
  const ArenaDev& A = a.A;
  const EvalIO& io = a.io;
  const int j = a.j, rev = a.rev, fold = a.fold, par = a.par, d = a.d, max_depth = a.max_depth;
  const double Emax = a.Emax;
  HostStatus* const st = a.st;
  const RowsDev& R = md.lg;
  int b = (int)blockIdx.x;
  if (fold & GA_FOLD_CTL) {   // workgroup 0: control work, from the previous launch's block partials
    if (b == 0) { control_lean(md, A, a.cio, a.cj, a.cd, Emax, max_depth, st, a.cseq, lean_src(md, par ^ 1)); return; }
    --b;
  }
  const int g = b;
  // (the wave index is wave-uniform, but only `readfirstlane` tells the compiler: without it every tile address is per-lane
  // 64-bit VGPR arithmetic, and the streaming loop runs out of registers)
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6), W = (int)blockDim.x >> 6;
  Leaf lf; QView qv;
  const int aborted = load_aborted(io, A);
  resolve_leaf(io, A, j, lf, qv);
  const bool leaf = io.mode != MODE_PLAIN;
  __shared__ double s_acc[GA_MAXW][2][D + 1];   // [wave][first / second half of its tiles][d/dbeta, log-lik]
  __shared__ double s_red[NDOT];
  __shared__ double s_cp[GA_MAXCHUNK][PART_STRIDE];
  __shared__ int s_info[4];                      // {this workgroup is its block's last arriver, m, last}
  __shared__ double s_keep[5][WAVE];             // wave 0's per-lane prologue values the tail needs again (kept out of the stream's registers)
  __shared__ __attribute__((aligned(16))) char s_args[(sizeof(GaArgs) + 15) / 16 * 16];
  __shared__ __attribute__((aligned(16))) char s_auxprog[GA_AUX_PROG_LDS];   // auxiliary workgroups only (28 KB per workgroup in all: five per CU still fit)
  // workgroups behind the G groups: the auxiliary workgroups (rows_aux.h).  They take mu', sigma' of this leaf from the same
  // prologue arithmetic as everybody else and leave BEFORE any tile is requested: the call below may save registers around it,
  // and a register with a hand-counted load in flight must never be stored (tests/test_abi.py guards that).
  if (b >= R.G) {
    static_assert(GA_AUX_SCRATCH_DOUBLES(GA_MAXW) <= GA_MAXCHUNK * PART_STRIDE, "auxiliary scratch does not fit the chunk buffer");
    double hv, hp;
    ga_hyper<D>(md, qv, fold, par, lane, hv, hp);
    const int aux_id = b - R.G;
    // (LDS lent from the block reduce's chunk buffer, which only a block's last arriver uses)
    ga_aux<OCC>((const GaArgs*)__builtin_amdgcn_kernarg_segment_ptr(), aux_id, hv, hp, &s_cp[0][0], GA_MAXW, s_auxprog,
                R.ga_bpart + ((int64_t)par * R.ga_nrec + R.ga_nblk + aux_id) * PART_STRIDE, 1);
    return;
  }

  // ---- geometry of this wave's stream (no memory access when every group has the same number of rows) ----
  int T; int64_t ng, cbase;
  if (R.ga_T_uni > 0) { T = R.ga_T_uni; ng = R.ga_ng_uni; cbase = (int64_t)(g * W + w) * R.ga_cstride_uni; }
  else {
    T = __builtin_amdgcn_readfirstlane(R.ga_tile0[g + 1] - R.ga_tile0[g]);
    ng = __builtin_amdgcn_readfirstlane((int)(R.gptr[g + 1] - R.gptr[g]));
    const int64_t cb = R.ga_coff[g * W + w];
    cbase = ((int64_t)__builtin_amdgcn_readfirstlane((int)(cb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(cb & 0xffffffffll));
  }
  // The wave's tiles (its chunk) are streamed in two halves whose order alternates between launches (the tail of the previous
  // pass is still in the 256 MiB Infinity Cache); each half is summed in fixed tile order, wave-reduced and parked in its own
  // LDS slot, so the result does not depend on which half was streamed first.
  const int c0 = (int)((int64_t)w * T / W), c2 = (int)((int64_t)(w + 1) * T / W);   // this wave's tiles of the group: [c0, c2)
  const int n = c2 - c0;
  const int nA = (n + 1) / 2;                                   // tiles of the first half (chunk-local [0, nA)), second [nA, n)
  const int nsw = rev ? n - nA : nA;                            // position in the sequence where the second-streamed half starts
  constexpr int64_t TS = (int64_t)DX * SPAN;                    // doubles per tile
  auto local_at = [&](int i) { return rev ? (i < nsw ? nA + i : i - nsw) : i; };
  auto tile_at = [&](int i) { return cbase + (int64_t)local_at(i) * TS; };
  const int l_last = (c2 == T) ? n - 1 : -1;                    // chunk-local index of the group's last (zero-padded) tile, if it is here
  const int n_last = (int)(ng - (int64_t)(T - 1) * SPAN);     // its valid rows
  // PIPE tiles in flight per wave; the first ones are requested before anything else, so HBM is busy during the prologue
  double xa[D][RPL], xb[D][RPL], xc[PIPE == 3 ? D : 1][RPL];
  uint32_t ya = 0, yb = 0, yc = 0;
  Tile ta, tb, tc;
  // (a wave without tiles requests the chunk's first tile slot anyway: the layout keeps one tile of slack behind every chunk)
  if constexpr (D == 8 && RPL == 2) {
    const uint32_t voff16 = (uint32_t)lane * 16u, voff2 = (uint32_t)lane * 2u;
    const int64_t o0 = tile_at(0), o1 = tile_at(min(1, max(n - 1, 0)));
    ga_issue8(R.Xt + o0, R.y + o0 / DX, voff16, voff2, ta);
    ga_issue8(R.Xt + o1, R.y + o1 / DX, voff16, voff2, tb);
  } else {
    ga_load<D, RPL>(R, tile_at(0), lane, xa, ya);
    if (PIPE == 3) ga_load<D, RPL>(R, tile_at(min(1, max(n - 1, 0))), lane, xb, yb);
  }
  if (lane == 0) {
#pragma unroll
    for (int dd = 0; dd <= D; ++dd) { s_acc[w][0][dd] = 0.0; s_acc[w][1][dd] = 0.0; }
  }
  {
    const uint4* ka = (const uint4*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int t = tid; t < (int)((sizeof(GaArgs) + 15) / 16); t += (int)blockDim.x) reinterpret_cast<uint4*>(s_args)[t] = ka[t];
  }
  __syncthreads();   // (every wave is still at the top of the kernel: this costs nothing, and the copy is readable from here on)

  // ---- prologue: mu', sigma' of this leaf (every wave), z' of this group ----
  double hval0, hph0;   // lane l: q' and p_half of hyper-parameter element l mod 2D
  ga_hyper<D>(md, qv, fold, par, lane, hval0, hph0);
  const int dl = lane % D;
  const int iz = R.off_z + g * D + dl;
  double beta[D];
  {
    double zq, zph;
    if (qv.composed) { zph = fma(qv.half, qv.g[iz], qv.p[iz]); zq = fma(qv.eps, qv.var[iz] * zph, qv.q[iz]); }
    else { zph = 0.0; zq = qv.q[iz]; }
    const double m_lane = __shfl(hval0, dl);
    const double sraw = __shfl(hval0, D + dl);
    const double s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sraw) : sraw;
    const double bl = fma(s_lane, zq, m_lane);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) beta[dd] = readlane_d(bl, dd);
    if (w == 0) { s_keep[0][lane] = hval0; s_keep[1][lane] = hph0; s_keep[2][lane] = zq; s_keep[3][lane] = zph; s_keep[4][lane] = s_lane; }
  }
  if (aborted) return;   // (a workgroup that sees the flag takes no ticket: see the header)

  // ---- the stream ----
  {
    double acc[D], lp = 0.0;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) acc[dd] = 0.0;
    int half = rev ? 1 : 0;   // which half of the wave's range the current accumulators belong to
    auto flush = [&]() {
#pragma unroll
      for (int dd = 0; dd < D; ++dd) {
        const double sum = wave_sum(acc[dd]);
        if (lane == 0) s_acc[w][half][dd] = sum;
        acc[dd] = 0.0;
      }
      const double sum = wave_sum(lp);
      if (lane == 0) s_acc[w][half][D] = sum;
      lp = 0.0;
      half ^= 1;
    };
#define GA_STAGE(X, Y, I)                                                                   \
    {                                                                                        \
      if ((I) == nsw) flush();                                                               \
      ga_tile<D, RPL>(X, Y, beta, local_at(I) == l_last ? n_last : SPAN, lane, acc, lp);     \
    }
    const int nm1 = n - 1;
    if constexpr (D == 8 && RPL == 2) {
      // hand-counted loads: PIPE tiles in flight per wave; the requests past the end re-read the wave's last tile (an L2 hit) so that
      // every stage has exactly 9 (PIPE - 1) younger loads behind the tile it waits for
      const uint32_t voff16 = (uint32_t)lane * 16u, voff2 = (uint32_t)lane * 2u;
      auto issue = [&](int i, Tile& t) {
        const int64_t off = tile_at(min(i, nm1));
        ga_issue8(R.Xt + off, R.y + off / DX, voff16, voff2, t);
      };
#define GA_ASTAGE(T, I)                                                                      \
      {                                                                                      \
        ga_wait8<LOADS * (PIPE - 1)>(T);                                                     \
        if ((I) == nsw) flush();                                                             \
        double xx[8][2];                                                                     \
        ga_unpack(T, xx);                                                                    \
        ga_tile<8, 2>(xx, T.y, beta, local_at(I) == l_last ? n_last : SPAN, lane, acc, lp); \
      }
      if constexpr (PIPE == 3) {
        issue(2, tc);
        for (int i = 0; i < n; i += 3) {
          GA_ASTAGE(ta, i)
          if (i + 1 >= n) break;
          issue(i + 3, ta);
          GA_ASTAGE(tb, i + 1)
          if (i + 2 >= n) break;
          issue(i + 4, tb);
          GA_ASTAGE(tc, i + 2)
          issue(i + 5, tc);
        }
        ga_wait8<0>(ta); ga_wait8<0>(tb); ga_wait8<0>(tc);
      } else {
        for (int i = 0; i < n; i += 2) {
          GA_ASTAGE(ta, i)
          if (i + 1 >= n) break;
          issue(i + 2, ta);
          GA_ASTAGE(tb, i + 1)
          issue(i + 3, tb);
        }
        ga_wait8<0>(ta); ga_wait8<0>(tb);
      }
#undef GA_ASTAGE
    } else {
    // compiler-counted loads.  The prefetch of tile i + PIPE - 1 is UNCONDITIONAL (past the end it re-requests the wave's last tile):
    // a load inside an `if` makes hipcc drain every outstanding load (`s_waitcnt vmcnt(0)`) at the join.
    if constexpr (PIPE == 3) {
      for (int i = 0; i < n; i += 3) {
        ga_load<D, RPL>(R, tile_at(min(i + 2, nm1)), lane, xc, yc);
        GA_STAGE(xa, ya, i)
        if (i + 1 >= n) break;
        ga_load<D, RPL>(R, tile_at(min(i + 3, nm1)), lane, xa, ya);
        GA_STAGE(xb, yb, i + 1)
        if (i + 2 >= n) break;
        ga_load<D, RPL>(R, tile_at(min(i + 4, nm1)), lane, xb, yb);
        GA_STAGE(xc, yc, i + 2)
      }
    } else {
      for (int i = 0; i < n; i += 2) {
        ga_load<D, RPL>(R, tile_at(min(i + 1, nm1)), lane, xb, yb);
        GA_STAGE(xa, ya, i)
        if (i + 1 >= n) break;
        ga_load<D, RPL>(R, tile_at(min(i + 2, nm1)), lane, xa, ya);
        GA_STAGE(xb, yb, i + 1)
      }
    }
    }
#undef GA_STAGE
    if (n > 0) flush();
  }

  ga_tail<D>(*reinterpret_cast<const GaArgs*>(s_args), g, s_acc, s_red, s_cp, s_info, s_keep);
}

// Everything after the stream; `T` is the LDS copy of the kernel arguments.
template <int D>
__device__ __forceinline__ void ga_tail(const GaArgs& T, int g, double (&s_acc)[GA_MAXW][2][D + 1], double (&s_red)[NDOT],
                                        double (&s_cp)[GA_MAXCHUNK][PART_STRIDE], int (&s_info)[4], double (&s_keep)[5][WAVE]) {
  const ModelDev& md = T.md;
  const ArenaDev& A = T.A;
  const EvalIO& io = T.io;
  const RowsDev& R = md.lg;
  const int j = T.j, par = T.par, d = T.d;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6, W = (int)blockDim.x >> 6;
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  const bool leaf = io.mode != MODE_PLAIN;
  const bool tree = io.mode == MODE_TREE;
  const int dl = lane % D;
  const int iz = R.off_z + g * D + dl;
  // the operands of the first merge levels belong to earlier leaves: wave 0 requests them as soon as its own stream is done,
  // before it waits for the other waves, so they are in flight during the combine
  MergePrefetch mpf;
  if (w == 0 && tree && !(R.ga_flags & GA_F_NOTAIL)) merge_prefetch(A, lf, j, iz, mpf);
  __syncthreads();
  if (R.ga_flags & GA_F_NOTAIL) return;

  if (w == 0) {
    // ---- wave 0: the group's D z elements (lane = coordinate) ----
    const double hval = s_keep[0][lane], hph = s_keep[1][lane], zq = s_keep[2][lane], zph = s_keep[3][lane], s_lane = s_keep[4][lane];
    double db = 0.0, lpg = 0.0;
    for (int ww = 0; ww < W; ++ww) { db += s_acc[ww][0][dl] + s_acc[ww][1][dl]; lpg += s_acc[ww][0][D] + s_acc[ww][1][D]; }
    const bool zact = lane < D;
    int idx[1] = {iz};
    bool act[1] = {zact};
    double grad[1] = {0.0}, ph[1] = {zph};
    {
      const double r = zq - R.z_np_mu;                       // z ~ Normal(mu0, s0) in closed form (continuous.py:526-532)
      const double gx = -r * R.z_np_inv_var;
      const double lpz = -0.5 * r * r * R.z_np_inv_var - R.z_np_lognorm;
      grad[0] = gx + s_lane * db;                            // d/dz = prior + sigma_d * d/dbeta_d
      lpg += wave_sum(zact ? lpz : 0.0);
      if (zact) {
        if (leaf) { A.G[lf.d_o + iz] = grad[0]; A.Q[lf.d_o + iz] = zq; }
        else io.grad[iz] = grad[0];
      }
    }
    // the hyper-parameter elements' local parts + their q' (one workgroup does it for the launch; with auxiliary workgroups, they do)
    if (g == 0 && R.ga_naux == 0) {
      const int e = lane;
      const bool hact = e < 2 * D, is_mu = e < D;
      double gx, dxdq, dj, lpd;
      ga_def_local(R, is_mu, hval, gx, dxdq, dj, lpd);
      lpg += wave_sum(hact ? lpd : 0.0);
      if (hact) {
        const int dd = is_mu ? e : e - D;
        const int slot = (is_mu ? R.def_mu : R.def_sigma) + dd;
        double2* loc = reinterpret_cast<double2*>(md.def_loc + (int64_t)par * 4 * MAX_DEFERRED) + 2 * slot;
        loc[0] = make_double2(gx, dxdq);
        loc[1] = make_double2(dj, hph);
        if (leaf) A.Q[lf.d_o + (is_mu ? R.off_mu : R.off_sigma) + dd] = hval;
      }
    }
    int m = 0; bool last = false;
    if (leaf) leaf_post<1>(A, lf, j, d, tree, idx, act, grad, ph, s_red, 1, m, last, tree ? &mpf : nullptr);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- this group's record, write-through ----
    double* rec = R.ga_part + (int64_t)g * PART_STRIDE;
    if (lane == 0) st_agent(rec + PART_LP, lpg);
    if (zact) { st_agent(rec + PART_DMU + lane, db); st_agent(rec + PART_DSG + lane, db * zq); }
    if (leaf) {
      for (int k = lane; k < NDOT; k += WAVE)
        if (dot_needed(k, m, last)) st_agent(rec + PART_DOT + k, s_red[k]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has left this CU before the ticket is taken

    // ---- ticket: the block's last arriver sums the block's records in group order ----
    const int blk = g / R.ga_bsz;
    const int cnt = min(R.G, (blk + 1) * R.ga_bsz) - blk * R.ga_bsz;
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&R.ga_ticket[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    const int is_last = (int)old + 1 == cnt;
    if (is_last && lane == 0) __hip_atomic_store(&R.ga_ticket[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) { s_info[0] = is_last; s_info[1] = m; s_info[2] = last ? 1 : 0; }
  }
  __syncthreads();
  if (!s_info[0]) return;

  // ---- the last arriver's workgroup: block partial = sum of the block's records, chunks of 8 groups, chunks in order ----
  {
    const int m = s_info[1];
    const bool last = s_info[2] != 0;
    const int blk = g / R.ga_bsz, g0 = blk * R.ga_bsz;
    const int cnt = min(R.G, g0 + R.ga_bsz) - g0;
    const int nch = (cnt + 7) / 8;
    const int nn = 1 + 2 * D + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
    auto need_slot = [&](int q) {
      if (q < 1) return PART_LP;
      if (q < 1 + D) return PART_DMU + (q - 1);
      if (q < 1 + 2 * D) return PART_DSG + (q - 1 - D);
      if (q < 1 + 2 * D + 1 + 6 * m) return PART_DOT + (q - 1 - 2 * D);
      return PART_DOT + DOT_TOP + (q - 1 - 2 * D - 1 - 6 * m);
    };
    const int NT = (int)blockDim.x;
    for (int p = tid; p < nn * nch; p += NT) {
      const int c = p / nn, k = need_slot(p - c * nn);
      const int gg0 = c * 8, gcnt = min(8, cnt - gg0);
      const double* src = R.ga_part + (int64_t)(g0 + gg0) * PART_STRIDE + k;
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld_agent(src + (int64_t)min(u, gcnt - 1) * PART_STRIDE);
      double sum = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += (u < gcnt) ? v[u] : 0.0;
      s_cp[c][k] = sum;
    }
    __syncthreads();
    double* bp = R.ga_bpart + ((int64_t)par * R.ga_nrec + blk) * PART_STRIDE;
    for (int q = tid; q < nn; q += NT) {
      const int k = need_slot(q);
      double sum = 0.0;
      for (int c = 0; c < nch; ++c) sum += s_cp[c][k];
      bp[k] = sum;
    }
  }
}

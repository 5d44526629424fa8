// Lockstep chains of ONE hierarchical-logit model on the group-aligned row pass (the benchmark's model, BASELINE configs[1]: eight
// chains; the reference runs chains as independent processes, pymc/sampling/mcmc.py:1385-1500, sampling/parallel.py:477-589, and
// vmaps them on an accelerator, sampling/jax.py:341-348).
//
// `k_rows_ga` (rows_ga_kernel.h) is bound by the bytes of X: 99.6 % of what a leapfrog reads is the design matrix, which every
// chain of a rank reads identically.  `k_rows_ga_multi<NC, OCC, DX>` is the same launch for NC chains that stand at a tree leaf at
// the same time (engine.hip, the chain group: each chain's host thread deposits its launch, the last one to arrive submits them
// together): workgroup g streams the tiles of group g ONCE and evaluates every row under NC coefficient vectors (beta_g of each
// chain in scalar registers, NC sets of gradient accumulators), then finishes the group's D z elements of every chain -- second
// kick, v', merge dot products, the group's record, the ticket on the chain's own arrival counters -- exactly as the single-chain
// kernel's tail does, one wave per chain.  Workgroups 0 .. GAM_MAXC-1 carry the control work each chain folds into its launch (a
// chain's always in workgroup `slot`, i.e. on the same XCD).
//
// A chain's numbers do not depend on its company: every fma of a row, every wave sum (the exchange tree of mvn_multi_kernel.h
// reproduces `wave_sum`'s association order), every record and block partial is formed from the chain's own operands in the
// single-chain kernel's order, so a chain in a group is BITWISE the chain run alone (tests/test_gpu_chain_group.py).
//
// Chains keep their own arena, control block, uniform stream, status words, records, block partials, tickets and local parts
// (`ga_part`, `ga_bpart`, `ga_ticket`, `def_loc` of the chain's own model handle): nothing is shared but the read-only node data
// (X, y, the closed-form priors) and the in-order stream the group submits to.
//
// The pass is then bound by the vector pipe, not by HBM: ~73 fp64 operations per row and chain (logit_row) against 57 bytes per
// row -- one chain needs ~15 us of arithmetic for a 45 us stream, four chains are arithmetic-bound.
#pragma once
#include "mvn_multi_kernel.h"   // (wave_sum_many)

#define GAM_MAXC 8   // control workgroups of a launch = places in a rows group (round 6: eight, rows_gal_kernel.h; THIS kernel carries at most GAM_MAXNC chains)
#define GAM_MAXNC 4

struct GaLeafArgs {   // one chain's arguments of k_rows_ga (GaArgs without the model)
  ArenaDev A;
  EvalIO io, cio;
  double Emax;
  HostStatus* st;
  double* ga_part;       // [G][PART_STRIDE]           this chain's per-group records
  double* ga_bpart;      // [2][ga_nrec][PART_STRIDE]  its block partials, double-buffered by ITS launch parity
  unsigned* ga_ticket;   // [ga_nblk]                  its arrival counters
  double* def_loc;       // [2][MAX_DEFERRED][4]       local parts of its hyper-parameter elements
  int j, fold, par, d, max_depth, cj, cd, cseq;
  int slot, pad;         // the chain's place in its group: its control work always runs in workgroup `slot`
};

template <int NC>
struct GaMultiArgs {
  GaLeafArgs c[NC];
  int rev, pad;          // traversal order of a wave's two half-ranges in THIS launch (the result does not depend on it)
};

__device__ __forceinline__ LeanSrc gam_src(const RowsDev& R, const GaLeafArgs& L, int par) {
  return LeanSrc{L.ga_bpart + (int64_t)par * R.ga_nrec * PART_STRIDE, PART_STRIDE, R.ga_nrec, L.def_loc + (int64_t)par * 4 * MAX_DEFERRED};
}

// ga_hyper (rows_ga_kernel.h) on a chain's own records
template <int D>
__device__ __forceinline__ void gam_hyper(const RowsDev& R, const GaLeafArgs& L, const QView& qv, int lane, double& hval0, double& hph0) {
  if (L.fold & GA_FOLD_SRC) {
    const LeanSrc prev = gam_src(R, L, L.par ^ 1);
    rows_hyper_fold_elem<D>(R, prev.part, prev.stride, prev.nblk, prev.def_loc, qv, lane, hval0, hph0);
  } else {
    const int e = lane % (2 * D);
    const int i = e < D ? R.off_mu + e : R.off_sigma + (e - D);
    if (qv.composed) { hph0 = qv.p_half(i); hval0 = fma(qv.eps, qv.var[i] * hph0, qv.q[i]); }
    else { hph0 = 0.0; hval0 = qv.q[i]; }
  }
}

// The tail wave of chain `L` in workgroup g: the "wave 0" part of ga_tail on this chain's sums and buffers.
template <int D, bool PF>
__device__ __forceinline__ void gam_tail_wave(const ModelDev& md, const GaLeafArgs& L, int g, int W, const double (&s_acc)[GA_MAXW][2][D + 1],
                                              double* s_red, int* s_info, const double (&s_keep)[5][WAVE], const MergePrefetch& mpf) {
  const RowsDev& R = md.lg;
  const ArenaDev& A = L.A;
  const int lane = threadIdx.x & (WAVE - 1);
  const int j = L.j, par = L.par, d = L.d;
  Leaf lf; QView qv;
  resolve_leaf(L.io, A, j, lf, qv);
  const int dl = lane % D;
  const int iz = R.off_z + g * D + dl;
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
    if (zact) { A.G[lf.d_o + iz] = grad[0]; A.Q[lf.d_o + iz] = zq; }
  }
  if (g == 0) {   // the hyper-parameter elements' local parts + their q' (one workgroup does it for the launch)
    const int e = lane;
    const bool hact = e < 2 * D, is_mu = e < D;
    double gx, dxdq, dj, lpd;
    ga_def_local(R, is_mu, hval, gx, dxdq, dj, lpd);
    lpg += wave_sum(hact ? lpd : 0.0);
    if (hact) {
      const int dd = is_mu ? e : e - D;
      const int slot = (is_mu ? R.def_mu : R.def_sigma) + dd;
      double2* loc = reinterpret_cast<double2*>(L.def_loc + (int64_t)par * 4 * MAX_DEFERRED) + 2 * slot;
      loc[0] = make_double2(gx, dxdq);
      loc[1] = make_double2(dj, hph);
      A.Q[lf.d_o + (is_mu ? R.off_mu : R.off_sigma) + dd] = hval;
    }
  }
  int m = 0; bool last = false;
  leaf_post<1>(A, lf, j, d, true, idx, act, grad, ph, s_red, 1, m, last, PF ? &mpf : nullptr, 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- this group's record, write-through ----
  double* rec = L.ga_part + (int64_t)g * PART_STRIDE;
  if (lane == 0) st_agent(rec + PART_LP, lpg);
  if (zact) { st_agent(rec + PART_DMU + lane, db); st_agent(rec + PART_DSG + lane, db * zq); }
  for (int k = lane; k < NDOT; k += WAVE)
    if (dot_needed(k, m, last)) st_agent(rec + PART_DOT + k, s_red[k]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has left this CU before the ticket is taken

  // ---- ticket: the block's last arriver sums the block's records in group order ----
  const int blk = g / R.ga_bsz;
  const int cnt = min(R.G, (blk + 1) * R.ga_bsz) - blk * R.ga_bsz;
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(&L.ga_ticket[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  const int is_last = (int)old + 1 == cnt;
  if (is_last && lane == 0) __hip_atomic_store(&L.ga_ticket[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) { s_info[0] = is_last; s_info[1] = m; s_info[2] = last ? 1 : 0; }
}

// The last arriver's workgroup: block partial of chain `L` = sum of the block's records, chunks of 8 groups, chunks in order
// (every thread of the workgroup; ends with the stores, the caller separates two chains with a barrier)
template <int D>
__device__ __forceinline__ void gam_block_partial(const RowsDev& R, const GaLeafArgs& L, int g, const int* s_info, double (&s_cp)[GA_MAXCHUNK][PART_STRIDE]) {
  const int tid = threadIdx.x;
  const int m = s_info[1];
  const bool last = s_info[2] != 0;
  const int blk = g / R.ga_bsz, g0 = blk * R.ga_bsz;
  const int cnt = min(R.G, g0 + R.ga_bsz) - g0;
  const int nch = (cnt + 7) / 8;
  const int nn = 1 + 2 * D + 1 + 6 * m + (last ? 6 : 0);
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
    const double* src = L.ga_part + (int64_t)(g0 + gg0) * PART_STRIDE + k;
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_agent(src + (int64_t)min(u, gcnt - 1) * PART_STRIDE);
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += (u < gcnt) ? v[u] : 0.0;
    s_cp[c][k] = sum;
  }
  __syncthreads();
  double* bp = L.ga_bpart + ((int64_t)L.par * R.ga_nrec + blk) * PART_STRIDE;
  for (int q = tid; q < nn; q += NT) {
    const int k = need_slot(q);
    double sum = 0.0;
    for (int c = 0; c < nch; ++c) sum += s_cp[c][k];
    bp[k] = sum;
  }
}

// One tile into registers with ORDINARY loads (the single-chain kernel's hand-counted `global_load_dwordx4` + `vmcnt(N)` keep two
// tiles in flight across the loop, but a register with such a load pending must never be spilled -- tests/test_abi.py -- and NC
// sets of accumulators leave the allocator no such guarantee; here a spill costs time, never correctness, and a launch bound by its
// arithmetic with twelve waves per CU does not need the deeper prefetch: the other waves cover a wave's wait)
template <int DX>
__device__ __forceinline__ void gam_load(const double* base /* first element of the tile */, const int8_t* ybase, int lane,
                                         typename GaTileSel<DX>::type& t) {
  const ga_v2d* p = reinterpret_cast<const ga_v2d*>(base) + lane;
#pragma unroll
  for (int k = 0; k < DX; ++k) t.c[k] = p[k * WAVE];
  t.y = *reinterpret_cast<const uint16_t*>(ybase + 2 * lane);
}

// D = 8 covariates, two rows per lane, two tile buffers per wave; DX stored columns (7: the intercept column is not stored).  Grid: GAM_MAXC control workgroups + G group workgroups; block: the W waves of the
// model's layout (a wave's chunk of tiles is fixed when the model is created).
template <int NC, int OCC, int DX>
__global__ __launch_bounds__(64 * GA_MAXW, OCC) void k_rows_ga_multi(ModelDev md, GaMultiArgs<NC> ma) {
  constexpr int D = 8, RPL = 2, SPAN = WAVE * RPL, PIPE = 2;
  typedef typename GaTileSel<DX>::type Tile;
  const RowsDev& R = md.lg;
  if ((int)blockIdx.x < GAM_MAXC) {   // control workgroups
    int ci = -1;
#pragma unroll
    for (int c = 0; c < NC; ++c) ci = (int)blockIdx.x == ma.c[c].slot ? c : ci;
    if (ci < 0) return;
    const GaLeafArgs& L = ma.c[ci];
    if (L.fold & GA_FOLD_CTL) control_lean(md, L.A, L.cio, L.cj, L.cd, L.Emax, L.max_depth, L.st, L.cseq, gam_src(R, L, L.par ^ 1));
    return;
  }
  const int g = (int)blockIdx.x - GAM_MAXC;
  const int rev = ma.rev;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6), W = (int)blockDim.x >> 6;
  __shared__ double s_acc[NC][GA_MAXW][2][D + 1];   // [chain][wave][first / second half of its tiles][d/dbeta, log-lik]
  __shared__ double s_red[NC][NDOT];
  __shared__ double s_cp[GA_MAXCHUNK][PART_STRIDE];
  __shared__ int s_info[NC][4];
  __shared__ double s_keep[NC][5][WAVE];            // the tail wave's per-lane prologue values of each chain
  __shared__ double s_beta[NC][D];                  // beta_g of each chain (written by the chain's wave, read by all into scalar registers)
  __shared__ __attribute__((aligned(16))) char s_args[(sizeof(GaMultiArgs<NC>) + 15) / 16 * 16];

  // ---- geometry of this wave's stream (as k_rows_ga) ----
  int T; int64_t ng, cbase;
  if (R.ga_T_uni > 0) { T = R.ga_T_uni; ng = R.ga_ng_uni; cbase = (int64_t)(g * W + w) * R.ga_cstride_uni; }
  else {
    T = __builtin_amdgcn_readfirstlane(R.ga_tile0[g + 1] - R.ga_tile0[g]);
    ng = __builtin_amdgcn_readfirstlane((int)(R.gptr[g + 1] - R.gptr[g]));
    const int64_t cb = R.ga_coff[g * W + w];
    cbase = ((int64_t)__builtin_amdgcn_readfirstlane((int)(cb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(cb & 0xffffffffll));
  }
  const int c0 = (int)((int64_t)w * T / W), c2 = (int)((int64_t)(w + 1) * T / W);
  const int n = c2 - c0;
  const int nA = (n + 1) / 2;
  const int nsw = rev ? n - nA : nA;
  constexpr int64_t TS = (int64_t)DX * SPAN;
  auto local_at = [&](int i) { return rev ? (i < nsw ? nA + i : i - nsw) : i; };
  auto tile_at = [&](int i) { return cbase + (int64_t)local_at(i) * TS; };
  const int l_last = (c2 == T) ? n - 1 : -1;
  const int n_last = (int)(ng - (int64_t)(T - 1) * SPAN);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int dd = 0; dd <= D; ++dd) { s_acc[c][w][0][dd] = 0.0; s_acc[c][w][1][dd] = 0.0; }
  }
  {   // the chains' arguments -> LDS: the tail reads them from there (nothing of them stays in scalar registers across the stream)
    static_assert(alignof(GaMultiArgs<NC>) == 8 && sizeof(GaMultiArgs<NC>) % 8 == 0, "kernarg layout: the second argument follows the model at the next multiple of 8");
    const uint2* ka = (const uint2*)__builtin_amdgcn_kernarg_segment_ptr() + (sizeof(ModelDev) + 7) / 8;
    for (int t = tid; t < (int)(sizeof(GaMultiArgs<NC>) / 8); t += (int)blockDim.x) reinterpret_cast<uint2*>(s_args)[t] = ka[t];
  }

  // ---- prologue: chain c's by wave c mod W (the wave that will finish the chain): mu', sigma' of its leaf, z' of this group,
  // beta_g -> LDS.  Unlike the single-chain kernel no tile is in flight yet: NC prologues with hand-counted loads pending would
  // leave the register allocator no room (it spilled), and a launch that is bound by its arithmetic does not miss the overlap.
  int dead = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) dead |= load_aborted(ma.c[c].io, ma.c[c].A) ? (1 << c) : 0;
  if (dead == (1 << NC) - 1) return;   // every chain's tree has ended: the launch drains (no tickets)
  for (int c = w; c < NC; c += W) {
    const GaLeafArgs& L = ma.c[c];
    Leaf lf; QView qv;
    resolve_leaf(L.io, L.A, L.j, lf, qv);
    double hval0, hph0;
    gam_hyper<D>(R, L, qv, lane, hval0, hph0);
    const int dl = lane % D;
    const int iz = R.off_z + g * D + dl;
    double zq, zph;
    if (qv.composed) { zph = fma(qv.half, qv.g[iz], qv.p[iz]); zq = fma(qv.eps, qv.var[iz] * zph, qv.q[iz]); }
    else { zph = 0.0; zq = qv.q[iz]; }
    const double m_lane = __shfl(hval0, dl);
    const double sraw = __shfl(hval0, D + dl);
    const double s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sraw) : sraw;
    const double bl = fma(s_lane, zq, m_lane);
    if (lane < D) s_beta[c][lane] = bl;
    s_keep[c][0][lane] = hval0; s_keep[c][1][lane] = hph0; s_keep[c][2][lane] = zq; s_keep[c][3][lane] = zph; s_keep[c][4][lane] = s_lane;
  }
  __syncthreads();
  // Two tile buffers per wave (one being evaluated, one requested) up to three chains; with FOUR sets of accumulators the second
  // buffer is what pushed the allocator into spilling inside the loop (107 us per launch against 80 for three chains, rocprofv3
  // r05c): one buffer then, the wave's next tile is requested when the current one has been evaluated -- at ~1.3 us of arithmetic
  // per tile and twelve waves per CU the other waves cover the wait.
  constexpr bool TWO = NC <= 3;
  Tile ta, tb;
  {
    const int64_t o0 = tile_at(0), o1 = tile_at(min(1, max(n - 1, 0)));
    gam_load<DX>(R.Xt + o0, R.y + o0 / DX, lane, ta);
    if constexpr (TWO) gam_load<DX>(R.Xt + o1, R.y + o1 / DX, lane, tb);
  }
  double beta[NC][D];   // wave-uniform: scalar registers
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int dd = 0; dd < D; ++dd) beta[c][dd] = readlane_d(s_beta[c][dd], 0);

  // ---- the stream: every tile once, every row under NC coefficient vectors ----
  {
    double acc[NC][D], lp[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      lp[c] = 0.0;
#pragma unroll
      for (int dd = 0; dd < D; ++dd) acc[c][dd] = 0.0;
    }
    int half = rev ? 1 : 0;
    auto flush = [&]() {
      // NC (D + 1) wave sums at once, each in wave_sum's association order (mvn_multi_kernel.h, wave_sum_many)
      double flat[NC * D];
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int dd = 0; dd < D; ++dd) flat[c * D + dd] = acc[c][dd];
      int idx;
      const double t = wave_sum_many<NC * D>(flat, lane, idx);
      constexpr int NVP = NC * D <= 8 ? 8 : NC * D <= 16 ? 16 : 32;
      if (lane < NVP && idx < NC * D) s_acc[idx / D][w][half][idx % D] = t;
      int idx2;
      const double t2 = wave_sum_many<NC>(lp, lane, idx2);
      constexpr int NVP2 = NC <= 1 ? 1 : NC <= 2 ? 2 : 4;
      if (lane < NVP2 && idx2 < NC) s_acc[idx2][w][half][D] = t2;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        lp[c] = 0.0;
#pragma unroll
        for (int dd = 0; dd < D; ++dd) acc[c][dd] = 0.0;
      }
      half ^= 1;
    };
    const int nm1 = n - 1;
    auto issue = [&](int i, Tile& t) {   // (unconditional: past the end it re-reads the wave's last tile)
      const int64_t off = tile_at(min(i, nm1));
      gam_load<DX>(R.Xt + off, R.y + off / DX, lane, t);
    };
#define GAM_STAGE(TT, I)                                                                     \
    {                                                                                        \
      if ((I) == nsw) flush();                                                               \
      double xx[8][2];                                                                       \
      ga_unpack(TT, xx);                                                                     \
      const uint32_t yy = TT.y;                                                              \
      const int nv = local_at(I) == l_last ? n_last : SPAN;                                  \
      /* (two chains and more: the rows of a lane side by side, logit_row2 -- the launch is bound by dependent fp64 issue, not HBM) */ \
      _Pragma("unroll") for (int c = 0; c < NC; ++c) ga_tile<8, 2, (NC == 2)>(xx, yy, beta[c], nv, lane, acc[c], lp[c]); \
    }
    if constexpr (TWO) {
      for (int i = 0; i < n; i += 2) {
        GAM_STAGE(ta, i)
        if (i + 1 >= n) break;
        issue(i + 2, ta);
        GAM_STAGE(tb, i + 1)
        issue(i + 3, tb);
      }
    } else {
      for (int i = 0; i < n; ++i) {
        GAM_STAGE(ta, i)
        issue(i + 1, ta);
      }
    }
#undef GAM_STAGE
    if (n > 0) flush();
  }

  // ---- tails: chain c is finished by wave c mod W; then the block partials of the chains this workgroup arrived last for ----
  const GaMultiArgs<NC>& Tm = *reinterpret_cast<const GaMultiArgs<NC>*>(s_args);
  MergePrefetch mpf;
  // (the operands of the first merge levels belong to earlier leaves: requested -- for the first chain a wave finishes -- before the
  // wave waits for the others)
  if (w < NC && !((dead >> w) & 1)) {
    const GaLeafArgs& L = Tm.c[w];
    Leaf lf; QView qv;
    resolve_leaf(L.io, L.A, L.j, lf, qv);
    merge_prefetch(L.A, lf, L.j, R.off_z + g * D + lane % D, mpf);
  }
  __syncthreads();
  for (int c = w; c < NC; c += W) {
    if ((dead >> c) & 1) { if (lane == 0) s_info[c][0] = 0; continue; }
    if (c == w) gam_tail_wave<D, true>(md, Tm.c[c], g, W, s_acc[c], s_red[c], s_info[c], s_keep[c], mpf);
    else gam_tail_wave<D, false>(md, Tm.c[c], g, W, s_acc[c], s_red[c], s_info[c], s_keep[c], mpf);
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (!s_info[c][0]) continue;            // (workgroup-uniform)
    gam_block_partial<D>(R, Tm.c[c], g, s_info[c], s_cp);
    __syncthreads();
  }
}

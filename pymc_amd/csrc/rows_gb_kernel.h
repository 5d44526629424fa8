// Group-BLOCK row pass for the hierarchical Bernoulli-logit node with SMALL groups: still ONE launch per leapfrog step, and
// nothing crosses workgroups inside the launch.
//
// The group-aligned pass (rows_ga_kernel.h) gives every group a workgroup and hands the per-group records to the group's block
// through write-through stores, an arrival ticket and a last arriver that sums the block -- a chain of cross-XCD round trips
// that a 57 us stream hides (C2-L) and a cache-resident model does not: at C2-S (1248 groups x 80 rows, one tile each) that
// launch takes 24.5 us, as long as the two launches of the general path together (profiles/r03b_c2s_lab.txt).  Here:
//
//   * workgroup b (W waves) owns the GPW consecutive groups [b GPW, (b + 1) GPW): wave w streams groups b GPW + w, + W, ... whole
//     (a group has a handful of tiles), so d logp / d beta_g is complete in ONE wave and the wave finishes its group's D z
//     elements itself -- gradient, second half kick, v' = M^-1 p', q' store, the tree-merge dot products of `leaf_post`;
//   * the per-group records stay in LDS; after a barrier the workgroup sums them in group order into its block partial and
//     writes it with plain stores.  The kernel boundary publishes it: the next launch starts from the ga_nblk = ceil(G / GPW)
//     block partials exactly as the group-aligned pass starts from its 32 (rows_hyper_fold_elem in every wave's prologue,
//     control_lean in workgroup 0), so the host side -- folded control, also across doublings, look-ahead, draw_many -- is
//     the group-aligned pass's, unchanged.
//
// Same arithmetic per row and per element as the group-aligned pass (logit_row, ga_tile, ga_def_local, leaf_post); what differs
// is the association of the cross-group sums (GPW groups, then CTL_CHUNKS chunks of block partials) -- fixed, so results are
// bit-reproducible run to run.  Selected when the model is exactly what the group-aligned pass evaluates in closed form and
// the groups are too small for it (engine.hip); ragged and empty groups are fine (tile counts come from ga_tile0).
#pragma once
#include "rows_ga_kernel.h"

// ga_flags, TIMING EXPERIMENTS ONLY, compiled in by -DNUTS_KNOCKOUT (tools/build_ticks.sh knockout, tools/gb_knockout.py; results are
// wrong with any of them set)
#define GB_F_EMPTY 2        // every workgroup returns at once: the launch's fixed cost
#define GB_F_NOSUMS 4       // the prologue does not total the block partials
#define GB_F_NOSTREAM 8     // no tiles
#define GB_F_NOPOST 16      // no leaf_post
#define GB_F_NOBPART 32     // no block partial
#define GB_F_NOCTL 64       // the control workgroup returns at once
#define GB_F_PROLOGUE 128   // row workgroups return after the prologue
#ifdef NUTS_KNOCKOUT
#define GB_XF(bit) ((R.ga_flags & (bit)) != 0)
#else
#define GB_XF(bit) false
#endif
#ifndef GB_W
#define GB_W 8            // waves per workgroup (8 x 8 groups: half the block partials of 4 x 4 -- 10.9 vs 11.8 us per launch at C2-S)
#endif
#define GB_MAXGPW 16      // groups per workgroup (LDS records)

// one tile ([DX][SPAN] doubles at element offset xoff, y bytes at xoff / DX) into the [D][2] operand of ga_tile
template <int D, int DX>
__device__ __forceinline__ void gb_load(const RowsDev& R, int64_t xoff, int lane, double (&x)[D][2], uint32_t& ybits) {
  constexpr int SPAN = WAVE * 2;
  const double* tile = R.Xt + xoff + lane * 2;
  ybits = *reinterpret_cast<const uint16_t*>(R.y + xoff / DX + lane * 2);
  if (DX != D) { x[0][0] = 1.0; x[0][1] = 1.0; }   // the intercept column is not stored (rows_ga_kernel.h, GaTileRegs7)
#pragma unroll
  for (int c = 0; c < DX; ++c) {
    const double2 a = *reinterpret_cast<const double2*>(tile + c * SPAN);
    x[c + (D - DX)][0] = a.x; x[c + (D - DX)][1] = a.y;
  }
}

// The launch's body for workgroup `b_in` of one chain `a` of model `md`.  `karg`: the kernel's own argument block in a launch of the
// chain's own (the auxiliary workgroups re-read it, rows_aux.h), nullptr for the merged launch of a chain group
// (rows_gb_multi_kernel.h: b_in == 0 is then ALWAYS the chain's control slot, and there are no auxiliary workgroups).
// What one chain brings to a launch (everything of GaArgs but the model); in the merged launch of a chain group the model is the
// base member's and `ga_bpart` / `def_loc` -- fields of the MODEL in a launch of its own -- are the chain's.
struct GbChain {
  const ArenaDev& A; const EvalIO& io; const EvalIO& cio;
  int j, fold, par, d, max_depth, cj, cd, cseq;
  double Emax; HostStatus* st;
  double* ga_bpart; double* def_loc;
};

template <int D, int DX>
__device__ __forceinline__ void gb_body(const ModelDev& md, const GbChain& a, int b_in, const GaArgs* karg) {
  constexpr int SPAN = WAVE * 2;
  constexpr int64_t TS = (int64_t)DX * SPAN;
  const ArenaDev& A = a.A;
  const EvalIO& io = a.io;
  const RowsDev& R = md.lg;
  const int j = a.j, fold = a.fold, par = a.par, d = a.d;
  // (lean_src(md, par) on the chain's own records: slot-major block partials [2][PART_STRIDE][npad], local parts [2][MAX_DEFERRED][4])
  auto src_of = [&](int pp) {
    const int npad_ = (R.ga_nrec + WAVE - 1) / WAVE * WAVE;
    return LeanSrc{a.ga_bpart + (int64_t)pp * PART_STRIDE * npad_, 1, R.ga_nrec, a.def_loc + (int64_t)pp * 4 * MAX_DEFERRED, npad_};
  };
  int b = b_in;
  if (GB_XF(GB_F_EMPTY)) return;
  if ((fold & GA_FOLD_CTL) || !karg) {   // workgroup 0: control work, from the previous launch's block partials
    if (b == 0 && (GB_XF(GB_F_NOCTL) || !(fold & GA_FOLD_CTL))) return;
    if (b == 0) { control_lean<false, 8, true>(md, A, a.cio, a.cj, a.cd, a.Emax, a.max_depth, a.st, a.cseq, src_of(par ^ 1), GB_W * WAVE > VEC_THREADS ? VEC_THREADS : 0); return; }
    --b;
  }
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int GPW = R.ga_gpw;
  const int g0 = b * GPW, cnt = min(R.G, g0 + GPW) - g0;
  Leaf lf; QView qv;
  const int aborted = load_aborted(io, A);
  resolve_leaf(io, A, j, lf, qv);
  const bool leaf = io.mode != MODE_PLAIN, tree = io.mode == MODE_TREE;
  __shared__ double s_rec[GB_MAXGPW][PART_STRIDE];   // per-group records {logp, d/dmu[D], d/dsigma[D], dots}
  __shared__ double s_red[GB_W][NDOT];               // a wave's dot products of the group it is finishing
  __shared__ int s_ml[2];
  __shared__ __attribute__((aligned(16))) char s_auxprog[GA_AUX_PROG_LDS];   // auxiliary workgroups only (rows_aux.h)
  const bool tk = b == (int)(R.ga_nblk / 2) && tid == 0 && (md.tick_j < 0 || j == md.tick_j);   // NUTS_KTIMING builds only
  TICK(md, tk, 0);

  // geometry of a group: first tile offset, tiles, rows (no table look-up when every group has the same number of rows)
  auto geom = [&](int g, int& T, int& ng, int64_t& cbase) {
    if (R.ga_T_uni > 0) { T = R.ga_T_uni; ng = (int)R.ga_ng_uni; cbase = (int64_t)g * R.ga_cstride_uni; }
    else {
      T = __builtin_amdgcn_readfirstlane(R.ga_tile0[g + 1] - R.ga_tile0[g]);
      ng = __builtin_amdgcn_readfirstlane((int)(R.gptr[g + 1] - R.gptr[g]));
      const int64_t cb = R.ga_coff[g];
      cbase = ((int64_t)__builtin_amdgcn_readfirstlane((int)(cb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(cb & 0xffffffffll));
    }
  };
  // the first tile of this wave's first group is requested before anything else (a wave without a group reads group g0's)
  double xa[D][2];
  uint32_t ya;
  int T0, ng0; int64_t cb0;
  geom(min(g0 + w, R.G - 1), T0, ng0, cb0);
  gb_load<D, DX>(R, cb0, lane, xa, ya);

  // ---- prologue: mu', sigma' of this leaf.  Summing the ga_nblk block partials is a matter of load LATENCY and of the number of
  // cache lines touched (records another XCD wrote: ~1300 cycles per round of loads; record-major with sum_strided's eight-load
  // rounds one wave took 6.4 us of a 19 us launch at C2-S, and the control workgroup 4 us -- profiles/r03c, r03d), so the block
  // partials are stored SLOT-major and a slot is totalled by one wave: a handful of coalesced loads, all in flight, then the
  // wave's fixed DPP tree (slot_sum, kernels.h).  control_lean totals the same way: the gradient the control work stores for
  // these elements and the one composed here are the same bits. ----
  __shared__ double s_hS[2 * D];
  const int he = lane % (2 * D);
  const bool h_mu = he < D;
  const int hi = (h_mu ? R.off_mu : R.off_sigma) + (h_mu ? he : he - D);
  double2 hl01 = make_double2(0.0, 1.0), hl23 = make_double2(0.0, 0.0);
  double hq = 0.0, hv = 0.0;
  if (fold & GA_FOLD_SRC) {
    const LeanSrc prev = src_of(par ^ 1);
    const int slot = (h_mu ? R.def_mu : R.def_sigma) + (h_mu ? he : he - D);
    hl01 = reinterpret_cast<const double2*>(prev.def_loc)[2 * slot];
    hl23 = reinterpret_cast<const double2*>(prev.def_loc)[2 * slot + 1];
    hq = qv.q[hi]; hv = qv.var[hi];
    if (GB_XF(GB_F_NOSUMS)) { if (tid < 2 * D) s_hS[tid] = 0.0; } else {
    // wave w totals hyper slots w, w + GB_W, ... (slot-major block partials: coalesced wave loads, all in flight)
    constexpr int PERW = (2 * D + GB_W - 1) / GB_W;
    double v[PERW][SLOT_SUM_MAXR];
#pragma unroll
    for (int u = 0; u < PERW; ++u) {
      const int pe = min(w + u * GB_W, 2 * D - 1);
      const int k = pe < D ? PART_DMU + pe : PART_DSG + (pe - D);
      slot_sum_issue(prev.part + (int64_t)k * prev.slot_major, prev.slot_major, lane, v[u]);
    }
#pragma unroll
    for (int u = 0; u < PERW; ++u) {
      const double tot = slot_sum_finish(v[u], prev.slot_major, lane);
      if (w + u * GB_W < 2 * D && lane == 0) s_hS[w + u * GB_W] = tot;
    }
    }
  }
  // z' of this wave's first group: its loads do not depend on the hyper-parameters
  double zq_first = 0.0, zph_first = 0.0;
  {
    const int izf = R.off_z + min(g0 + w, R.G - 1) * D + (lane % D);
    if (qv.composed) { zph_first = fma(qv.half, qv.g[izf], qv.p[izf]); zq_first = fma(qv.eps, qv.var[izf] * zph_first, qv.q[izf]); }
    else zq_first = qv.q[izf];
  }
  __syncthreads();
  double hval0, hph0;   // lane l: q' and p_half of hyper-parameter element l mod 2D
  if (fold & GA_FOLD_SRC) {
    const double S = s_hS[he];
    const double g = deferred_finish(hl01.x, S, hl01.y, hl23.x);
    const double p_src = fma(qv.half, g, hl23.y);                      // p' of the previous leaf (integration.py:131)
    hph0 = fma(qv.half, g, p_src);                                     // this leaf's p_half
    hval0 = fma(qv.eps, hv * hph0, hq);                                // this leaf's q'
  } else {
    if (qv.composed) { hph0 = qv.p_half(hi); hval0 = fma(qv.eps, qv.var[hi] * hph0, qv.q[hi]); }
    else { hph0 = 0.0; hval0 = qv.q[hi]; }
  }
  const int dl = lane % D;
  const double m_lane = __shfl(hval0, dl);
  const double sraw = __shfl(hval0, D + dl);
  const double s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sraw) : sraw;
  if (aborted) return;
  if (b >= R.ga_nblk) {   // auxiliary workgroup (rows_aux.h): everything of the model that is not a z element
    static_assert(GA_AUX_SCRATCH_DOUBLES(GB_W) <= GB_MAXGPW * PART_STRIDE, "auxiliary scratch does not fit the record buffer");
    const int aux_id = b - R.ga_nblk;
    const int npad = (R.ga_nrec + WAVE - 1) / WAVE * WAVE;
    if (karg) ga_aux<0>(karg, aux_id, hval0, hph0, &s_rec[0][0], GB_W, s_auxprog,
                        a.ga_bpart + (int64_t)par * PART_STRIDE * npad + (R.ga_nblk + aux_id), npad);
    return;
  }
  if (GB_XF(GB_F_PROLOGUE)) { if (m_lane + s_lane == 12345.678) A.Q[lf.d_o] = 0.0; return; }
  int m = 0; bool last = false;
  TICK(md, tk, 1);

  for (int gl = w; gl < cnt; gl += GB_W) {
    const int g = g0 + gl;
    int T, ng; int64_t cbase;
    geom(g, T, ng, cbase);
    const int iz = R.off_z + g * D + dl;
    // z' of this group and beta_g
    double zq = zq_first, zph = zph_first;
    if (gl != w) {
      if (qv.composed) { zph = fma(qv.half, qv.g[iz], qv.p[iz]); zq = fma(qv.eps, qv.var[iz] * zph, qv.q[iz]); }
      else { zph = 0.0; zq = qv.q[iz]; }
    }
    MergePrefetch mpf;
    if (tree) merge_prefetch(A, lf, j, iz, mpf);   // operands of the first merge levels: in flight during the stream
    double beta[D];
    {
      const double bl = fma(s_lane, zq, m_lane);
#pragma unroll
      for (int dd = 0; dd < D; ++dd) beta[dd] = readlane_d(bl, dd);
    }
    TICK(md, tk && gl == w, 2);
    // ---- the stream: the group's tiles in order, one tile ahead ----
    double acc[D], lp = 0.0;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) acc[dd] = 0.0;
    const int n_last = ng - (T - 1) * SPAN;
    if (gl != w) gb_load<D, DX>(R, cbase, lane, xa, ya);
    for (int t = 0; t < (GB_XF(GB_F_NOSTREAM) ? 0 : T); ++t) {
      double xb[D][2];
      uint32_t yb;
      gb_load<D, DX>(R, cbase + (int64_t)min(t + 1, T - 1) * TS, lane, xb, yb);   // unconditional prefetch (re-reads the last tile at the end)
      ga_tile<D, 2>(xa, ya, beta, t == T - 1 ? n_last : SPAN, lane, acc, lp);
#pragma unroll
      for (int dd = 0; dd < D; ++dd) { xa[dd][0] = xb[dd][0]; xa[dd][1] = xb[dd][1]; }
      ya = yb;
    }
    TICK(md, tk && gl == w, 3);
    // ---- the group's D z elements (lane = coordinate) ----
    double db = 0.0;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) { const double sum = wave_sum(acc[dd]); db = (dl == dd) ? sum : db; }
    double lpg = wave_sum(lp);
    const bool zact = lane < D;
    int idx[1] = {iz};
    bool act[1] = {zact};
    double grad[1] = {0.0}, ph[1] = {zph};
    {
      const double r = zq - R.z_np_mu;                       // z ~ Normal(mu0, s0) in closed form (continuous.py:526-532)
      const double gx = -r * R.z_np_inv_var;
      const double lpz = -0.5 * r * r * R.z_np_inv_var - R.z_np_lognorm;
      grad[0] = gx + s_lane * db;                            // d/dz = prior + sigma_d * d/dbeta_d
      lpg += wave_sum8(zact ? lpz : 0.0);
      if (zact) {
        if (leaf) { A.G[lf.d_o + iz] = grad[0]; A.Q[lf.d_o + iz] = zq; }
        else io.grad[iz] = grad[0];
      }
    }
    if (g == 0 && R.ga_naux == 0) {   // the hyper-parameter elements' local parts + their q' (one wave does it for the launch; with auxiliary workgroups, they do)
      const int e = lane;
      const bool hact = e < 2 * D, is_mu = e < D;
      double gx, dxdq, dj, lpd;
      ga_def_local(R, is_mu, hval0, gx, dxdq, dj, lpd);
      lpg += wave_sum(hact ? lpd : 0.0);
      if (hact) {
        const int dd = is_mu ? e : e - D;
        const int slot = (is_mu ? R.def_mu : R.def_sigma) + dd;
        double2* loc = reinterpret_cast<double2*>(a.def_loc + (int64_t)par * 4 * MAX_DEFERRED) + 2 * slot;
        loc[0] = make_double2(gx, dxdq);
        loc[1] = make_double2(dj, hph0);
        if (leaf) A.Q[lf.d_o + (is_mu ? R.off_mu : R.off_sigma) + dd] = hval0;
      }
    }
    TICK(md, tk && gl == w, 4);
    if (leaf && !GB_XF(GB_F_NOPOST))
      leaf_post<1, false, D <= 8>(A, lf, j, d, tree, idx, act, grad, ph, &s_red[w][0], 1, m, last, tree ? &mpf : nullptr, 0);
    TICK(md, tk && gl == w, 5);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double* rec = s_rec[gl];
    if (lane == 0) rec[PART_LP] = lpg;
    if (zact) { rec[PART_DMU + lane] = db; rec[PART_DSG + lane] = db * zq; }
    if (leaf) {
      for (int k = lane; k < NDOT; k += WAVE)
        if (dot_needed(k, m, last)) rec[PART_DOT + k] = s_red[w][k];
    }
  }
  TICK(md, tk, 6);
  if (tid == 0 && leaf && tree) { int mm = 0; while (((j >> mm) & 1) && mm < d) ++mm; s_ml[0] = mm; s_ml[1] = (j + 1 == (1 << d)); }
  __syncthreads();

  // ---- block partial: the workgroup's records summed in group order ----
  if (!GB_XF(GB_F_NOBPART)) {
    const int mm = (leaf && tree) ? s_ml[0] : 0;
    const bool ll = (leaf && tree) ? s_ml[1] != 0 : false;
    const int nn = 1 + 2 * D + (leaf ? 1 + 6 * mm + (ll ? 6 : 0) : 0);
    const int npad = (R.ga_nrec + WAVE - 1) / WAVE * WAVE;            // slot-major: bp[k * npad + b] (lean_src)
    double* bp = a.ga_bpart + (int64_t)par * PART_STRIDE * npad + b;
    for (int q = tid; q < nn; q += (int)blockDim.x) {
      int k;
      if (q < 1) k = PART_LP;
      else if (q < 1 + D) k = PART_DMU + (q - 1);
      else if (q < 1 + 2 * D) k = PART_DSG + (q - 1 - D);
      else if (q < 1 + 2 * D + 1 + 6 * mm) k = PART_DOT + (q - 1 - 2 * D);
      else k = PART_DOT + DOT_TOP + (q - 1 - 2 * D - 1 - 6 * mm);
      double sum = 0.0;
      for (int gl = 0; gl < cnt; ++gl) sum += s_rec[gl][k];
      bp[(int64_t)k * npad] = sum;
    }
  }
  TICK(md, tk, 7);
}

template <int D, int DX = D>
__global__ __launch_bounds__(64 * GB_W) void k_rows_gb(GaArgs a) {
  const GbChain ch{a.A, a.io, a.cio, a.j, a.fold, a.par, a.d, a.max_depth, a.cj, a.cd, a.cseq, a.Emax, a.st, a.md.lg.ga_bpart, a.md.def_loc};
  gb_body<D, DX>(a.md, ch, (int)blockIdx.x, (const GaArgs*)__builtin_amdgcn_kernarg_segment_ptr());
}

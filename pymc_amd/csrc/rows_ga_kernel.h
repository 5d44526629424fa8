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

__device__ __forceinline__ void st_agent(double* p, double v) {   // write-through store (visible to every XCD's L2)
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

// one tile of SPAN = 64 RPL rows: forward (eta, log-lik) + backward (d/dbeta) in registers
template <int D, int RPL>
__device__ __forceinline__ void ga_tile(const double (&x)[D][RPL], uint32_t yb, const double (&beta)[D], int nvalid, int lane,
                                        double (&acc)[D], double& lp) {
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
  }
}

#define GA_MAXW 4

template <int D, int RPL>
__global__ __launch_bounds__(64 * GA_MAXW, 4) void k_rows_ga(ModelDev md, ArenaDev A, EvalIO io, int j, int rev, int fold, int par, int d,
                                                             double Emax, int max_depth, HostStatus* st) {
  constexpr int SPAN = WAVE * RPL;
  const RowsDev& R = md.lg;
  int b = (int)blockIdx.x;
  if (fold) {   // workgroup 0: the control work of the previous leaf, from the previous launch's block partials
    if (b == 0) { control_lean(md, A, io, j - 1, d, Emax, max_depth, st, 0, lean_src(md, par ^ 1)); return; }
    --b;
  }
  const int g = b;
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6, W = (int)blockDim.x >> 6;
  Leaf lf; QView qv;
  const int aborted = load_aborted(io, A);
  resolve_leaf(io, A, j, lf, qv);
  const bool leaf = io.mode != MODE_PLAIN;
  __shared__ double s_acc[GA_MAXW][2][D + 1];   // [wave][first / second half of its tiles][d/dbeta, log-lik]
  __shared__ double s_red[NDOT];

  // ---- prologue: mu', sigma' of this leaf (every wave), z' of this group ----
  const int t0 = R.ga_tile0[g], T = R.ga_tile0[g + 1] - t0;
  const int64_t ng = R.gptr[g + 1] - R.gptr[g];
  double hval, hph;   // lane l: q' and p_half of hyper-parameter element l mod 2D
  if (fold) {
    const LeanSrc prev = lean_src(md, par ^ 1);
    rows_hyper_fold_elem<D>(R, prev.part, prev.stride, prev.nblk, prev.def_loc, qv, lane, hval, hph);
  } else {
    const int e = lane % (2 * D);
    const int i = e < D ? R.off_mu + e : R.off_sigma + (e - D);
    if (qv.composed) { hph = qv.p_half(i); hval = fma(qv.eps, qv.var[i] * hph, qv.q[i]); }
    else { hph = 0.0; hval = qv.q[i]; }
  }
  const int dl = lane % D;
  const int iz = R.off_z + g * D + dl;
  double zq, zph;
  if (qv.composed) { zph = fma(qv.half, qv.g[iz], qv.p[iz]); zq = fma(qv.eps, qv.var[iz] * zph, qv.q[iz]); }
  else { zph = 0.0; zq = qv.q[iz]; }
  const double m_lane = __shfl(hval, dl);
  const double sraw = __shfl(hval, D + dl);
  const double s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sraw) : sraw;
  double beta[D];
  {
    const double bl = fma(s_lane, zq, m_lane);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) beta[dd] = readlane_d(bl, dd);
  }
  if (aborted) return;   // (a workgroup that sees the flag takes no ticket: see the header)

  // ---- the stream: this wave's share of the group's tiles, in two halves whose order alternates between launches ----
  // (the tail of the previous pass is still in the 256 MiB Infinity Cache; each half is summed in fixed tile order,
  // wave-reduced, and parked in its own LDS slot, so the result does not depend on which half was streamed first)
  const int a0 = t0 + (int)((int64_t)w * T / W), a2 = t0 + (int)((int64_t)(w + 1) * T / W);
  const int a1 = a0 + (a2 - a0 + 1) / 2;
  const int t_last = t0 + T - 1;
  const int n_last = (int)(ng - (int64_t)(T - 1) * SPAN);   // valid rows of the group's last (zero-padded) tile
  for (int h = 0; h < 2; ++h) {
    const int second = ((h == 0) == (rev != 0)) ? 1 : 0;   // rev: second half first
    const int s0 = second ? a1 : a0, s1 = second ? a2 : a1;
    double acc[D], lp = 0.0;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) acc[dd] = 0.0;
    for (int t = s0; t < s1; ++t) {
      double x[D][RPL];
      uint32_t yb;
      rows_load<D, RPL>(R, t, lane, x, yb);
      ga_tile<D, RPL>(x, yb, beta, t == t_last ? n_last : SPAN, lane, acc, lp);
    }
#pragma unroll
    for (int dd = 0; dd < D; ++dd) {
      const double sum = wave_sum(acc[dd]);
      if (lane == 0) s_acc[w][second][dd] = sum;
    }
    const double sum = wave_sum(lp);
    if (lane == 0) s_acc[w][second][D] = sum;
  }

  // ---- the operands of the first merge levels belong to earlier leaves: in flight during the combine ----
  MergePrefetch mpf;
  const bool tree = io.mode == MODE_TREE;
  if (w == 0 && tree) merge_prefetch(A, lf, j, iz, mpf);

  __syncthreads();
  if (w != 0) return;

  // ---- wave 0: the group's D z elements (lane = coordinate) ----
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
  // the hyper-parameter elements' local parts + their q' (one workgroup does it for the launch)
  if (g == 0) {
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
  const int g0 = blk * R.ga_bsz, g1 = min(R.G, g0 + R.ga_bsz);
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(&R.ga_ticket[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  if ((int)old + 1 != g1 - g0) return;
  if (lane == 0) __hip_atomic_store(&R.ga_ticket[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int nn = 1 + 2 * D + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
  double* bp = R.ga_bpart + ((int64_t)par * R.ga_nblk + blk) * PART_STRIDE;
  for (int q = lane; q < nn; q += WAVE) {
    int k;
    if (q < 1) k = PART_LP;
    else if (q < 1 + D) k = PART_DMU + (q - 1);
    else if (q < 1 + 2 * D) k = PART_DSG + (q - 1 - D);
    else if (q < 1 + 2 * D + 1 + 6 * m) k = PART_DOT + (q - 1 - 2 * D);
    else k = PART_DOT + DOT_TOP + (q - 1 - 2 * D - 1 - 6 * m);
    const double* src = R.ga_part + (int64_t)g0 * PART_STRIDE + k;
    double s = 0.0;
    for (int gg = 0; gg < g1 - g0; gg += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld_agent(src + (int64_t)min(gg + u, g1 - g0 - 1) * PART_STRIDE);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (gg + u < g1 - g0) ? v[u] : 0.0;
    }
    bp[k] = s;
  }
}

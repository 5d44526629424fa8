// Lockstep chains of ONE MvNormal model through the MATRIX CORES: up to 16 chains per launch (BASELINE configs[2]: "exercises MFMA
// path"; the reference vmaps chains on an accelerator, pymc/sampling/jax.py:341-348; the node is
// pymc/distributions/multivariate.py:158-185 quaddist_matrix / quaddist_chol with a constant precision).
//
// `k_mvn_aligned_multi` (mvn_multi_kernel.h) keeps NC sets of R partial sums per lane: at four chains it is at the register
// budget and its pass costs 13 us (3.3 us per chain).  With 5 .. 16 chains standing at a leaf at the same time the mat-vec is a
// small GEMM, Y[16 rows][chains] = P[16 rows][K] D[K][chains], D = q_c - mu -- `v_mfma_f64_16x16x4_f64`, one 16 x 16 accumulator
// tile per wave whatever the number of chains (tools/mfma_matvec_bench.hip, profiles/r02k_mfma_matvec.txt: 9.6 us for 16 chains as
// two launches; fused here).
//
// Workgroup b owns rows [16 b, 16 b + 16) (the model's layout: NUTS_MVN_ALIGNED = 16, so that the single-chain launches of the
// same chains -- start states, draws outside a group -- file their records under the same workgroups).  Its 16 waves split the K
// columns in steps of 16 (wave w: steps w, w + 16, ...); operand layout of the instruction: A lane l = P[row0 + (l & 15)][kk + 4 (l >> 4) + i],
// B lane l = D[kk + 4 (l >> 4) + i][chain l & 15], i = 0 .. 3 over four MFMAs; accumulator register t of lane l =
// Y[(l >> 4) + 4 t][l & 15].  Every lane reads 32 contiguous bytes of its row of P and of its chain's q per step.  The waves'
// tiles meet in LDS; wave c then finishes chain c's 16 elements (second kick, v', merge dot products, first half of the next
// leaf, the workgroup's record) with the tail of the other row-aligned kernels (mvm_tail_core), its operands requested before the
// stream began.  Workgroups 0 .. 15 carry the control work a chain folds into its launch (the chain in place `slot`).
//
// The sums of a row are formed in another order than in `k_mvn_aligned` (four k per MFMA, sixteen column slices): a chain in a
// wide group is NOT bitwise the chain alone -- it is held to the oracle (log-density 1e-10, the sampler's integers) like any
// other kernel (tests/test_gpu_chain_group.py).  The chains' arguments exceed the 4 KB a kernel takes by value (16 x 360 B): they
// are read from a ring in pinned host memory the submitting thread has just filled, as kernel arguments are.
#pragma once

#define MFM_MAXC 16
#define MFM_R 16
#define MFM_WAVES 16
typedef double mfm_v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(MFM_WAVES* WAVE, 1) void k_mvn_mfma_multi(ModelDev md, const MvaLeafArgs* __restrict__ args, int nc) {
  const MvnDev& mv = md.mv;
  if ((int)blockIdx.x < MFM_MAXC) {
    if (threadIdx.x >= VEC_THREADS) return;
    int ci = -1;
    for (int c = 0; c < nc; ++c) ci = (int)blockIdx.x == args[c].slot ? c : ci;
    if (ci < 0) return;
    const MvaLeafArgs& L = args[ci];
    if (L.fold) mva_control(md, L.A, L.cio, L.cj, L.cd, L.Emax, L.max_depth, L.st, L.cseq, L.par ^ 1, VEC_THREADS, L.al_part);
    return;
  }
  constexpr int R = MFM_R;
  const int b = (int)blockIdx.x - MFM_MAXC;
  __shared__ double s_part[MFM_WAVES][R][MFM_MAXC + 1];   // (+ 1: the tail's column reads fall on different banks)
  __shared__ double s_red[MFM_MAXC][NDOT];
  __shared__ int s_dead;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = mv.k, row0 = b * R;
  // ---- which chains are alive; the position vector each column of B reads ----
  const int chain = lane & 15, ks = lane >> 4;
  const double* qc = nullptr;
  {
    bool alive = false;
    if (lane < MFM_MAXC && lane < nc) {
      const MvaLeafArgs& Lc = args[lane];
      alive = !load_aborted(Lc.io, Lc.A);
      if (alive) { Leaf l2; QView q2; resolve_leaf(Lc.io, Lc.A, Lc.j, l2, q2); qc = q2.q; }
    }
    const unsigned long long m = __ballot(alive);
    if (m == 0ull) return;             // every chain's tree has ended: the launch drains
    qc = reinterpret_cast<const double*>(__shfl((unsigned long long)reinterpret_cast<uintptr_t>(qc), chain, WAVE));
    if (tid == 0) s_dead = (int)(~m & 0xffffull);
  }
  // ---- the tail of chain w: request what is known at launch ----
  Leaf lf; QView qv; MergePrefetch mpf;
  double phv = 0.0, qr = 0.0, mur = 0.0, var_r = 0.0;
  const int my = min(row0 + min(lane, R - 1), K - 1);
  bool mine_alive = false;
  if (w < nc) {
    const MvaLeafArgs& L = args[w];
    mine_alive = !load_aborted(L.io, L.A);
    if (mine_alive) {
      resolve_leaf(L.io, L.A, L.j, lf, qv);
      const bool leaf = L.io.mode != MODE_PLAIN, tree = L.io.mode == MODE_TREE;
      if (leaf) { phv = L.A.P[lf.d_o + my]; var_r = L.A.var[my]; }
      qr = qv.q[my]; mur = mv.mu[my];
      if (tree) merge_prefetch(L.A, lf, L.j, my, mpf);
    }
  }
  // ---- the stream: this wave's column steps through the matrix cores ----
  mfm_v4d acc = {0.0, 0.0, 0.0, 0.0};
  {
    const double* pr = mv.prec + (int64_t)min(row0 + (lane & 15), K - 1) * K + 4 * ks;
    const double* mub = mv.mu + 4 * ks;
    const double* qb = qc ? qc + 4 * ks : nullptr;
    const int nsteps = K / 16;         // (K is a multiple of 16: checked when the group is formed)
#pragma unroll 4
    for (int s = w; s < nsteps; s += MFM_WAVES) {
      const int kk = 16 * s;
      const double2 a01 = *reinterpret_cast<const double2*>(pr + kk);
      const double2 a23 = *reinterpret_cast<const double2*>(pr + kk + 2);
      double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
      if (qb) {
        const double2 q01 = *reinterpret_cast<const double2*>(qb + kk), q23 = *reinterpret_cast<const double2*>(qb + kk + 2);
        const double2 m01 = *reinterpret_cast<const double2*>(mub + kk), m23 = *reinterpret_cast<const double2*>(mub + kk + 2);
        b0 = q01.x - m01.x; b1 = q01.y - m01.y; b2 = q23.x - m23.x; b3 = q23.y - m23.y;
      }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b3, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) s_part[w][ks + 4 * t][chain] = acc[t];
  __syncthreads();
  if (w >= nc || !mine_alive) return;
  // ---- wave c = w finishes chain c: lane r < 16 holds row row0 + r ----
  double t = 0.0;
  {
    const int r = min(lane, R - 1);
#pragma unroll
    for (int ww = 0; ww < MFM_WAVES; ++ww) t += s_part[ww][r][w];
  }
  mvm_tail_core<R>(mv, args[w], b, t, s_red[w], lf, qv, phv, var_r, qr, mur, mpf);
}

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
// Workgroup b owns rows [8 b, 8 b + 8) (the model's layout of `k_mvn_aligned<8>`, so that the single-chain launches of the same
// chains -- start states, draws outside a group -- file their records under the same workgroups, and every CU of the chip
// streams its share of P; the tile's rows 8 .. 15 are zeros).  Its 16 waves split the K columns in steps of 16 (wave w: steps
// w, w + 16, ...); operand layout of the instruction: A lane l = P[row0 + (l & 15)][kk + 4 (l >> 4) + i],
// B lane l = D[kk + 4 (l >> 4) + i][chain l & 15], i = 0 .. 3 over four MFMAs; accumulator register t of lane l =
// Y[(l >> 4) + 4 t][l & 15].  Every lane reads 32 contiguous bytes of its row of P and of its chain's q per step.  The waves'
// tiles meet in LDS; wave c then finishes chain c's 16 elements (second kick, v', merge dot products, first half of the next
// leaf, the workgroup's record) with the tail of the other row-aligned kernels (mvm_tail_core), its operands requested before the
// stream began.  Workgroups 0 .. 15 carry the control work a chain folds into its launch (the chain in place `slot`).
//
// The sums of a row are formed in another order than in `k_mvn_aligned` (four k per MFMA, sixteen column slices): a chain in a
// wide group is NOT bitwise the chain alone -- it is held to the oracle (log-density 1e-10, the sampler's integers) like any
// other kernel (tests/test_gpu_chain_group.py).
#pragma once

#define MFM_MAXC 16
#define MFM_R 8
#define MFM_WAVES 16
typedef double mfm_v4d __attribute__((ext_vector_type(4)));

// What a chain hands to a merged launch, in two parts.  Sixteen MvaLeafArgs (360 B each) exceed the 4 KB a kernel takes by value.
// Read from a block in (pinned host or LDS-staged) memory instead, every use of them in the tail -- the arena's pointers between
// two stores -- became a load the compiler could not keep in scalar registers: 7 us of a 28 us launch (tools/wide_ticks.py).  So:
// what changes from launch to launch travels by value (208 B per chain), what a chain keeps for its whole life (the arena, its
// records, its status words) lies in device memory behind a `const __restrict__` pointer -- scalar loads out of L2.
struct MfmChainConst {
  ArenaDev A;            // (uniforms / log_uniforms: see MfmLeaf)
  double Emax;
  double* al_part;
};
struct MfmLeaf {
  EvalIO io, cio;
  const double* uniforms;       // the arena's two pointers that move from batch to batch of draws (nuts_chain_draw_many)
  const double* log_uniforms;
  HostStatus* st;               // (only the launch that carries a doubling's last control work has one)
  Ctl* ctl;                     // = A.ctl: the `aborted` flag is looked at first, one dependent load less from here
  int j, fold, d, max_depth, par, cj, cd, cseq;
  int slot, pad;
};
struct MfmArgs {
  MfmLeaf c[MFM_MAXC];
  int nc, pad;
};

__device__ __forceinline__ MvaLeafArgs mfm_leaf(const MfmLeaf& l, const MfmChainConst& k) {
  MvaLeafArgs L;
  L.A = k.A; L.A.uniforms = l.uniforms; L.A.log_uniforms = l.log_uniforms;
  L.io = l.io; L.cio = l.cio; L.Emax = k.Emax; L.st = l.st; L.al_part = k.al_part;
  L.j = l.j; L.fold = l.fold; L.d = l.d; L.max_depth = l.max_depth; L.par = l.par; L.cj = l.cj; L.cd = l.cd; L.cseq = l.cseq;
  L.slot = l.slot; L.pad = 0;
  return L;
}

// D[k][chain] = q_chain[k] - mu[k], packed 16 chains to a row (128 B): the B operand of the matrix instruction, read by EVERY row
// workgroup.  Taken straight from the chains' arenas (16 vectors, 32 bytes of each per lane and step) the reads cost the launch
// 1.3 us per chain (12 us + 1.3 us x chains, rocprofv3 r05n: twice the requests of the P rows themselves); packed once by this
// kernel, a wave's step reads 2 KB of consecutive memory.  Grid: K / 16 workgroups of 256 threads (thread = (k, chain)).
__global__ __launch_bounds__(256) void k_mfm_pack(MvnDev mv, const MfmChainConst* __restrict__ konst, MfmArgs ma, double* __restrict__ dpack) {
  const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int k = i >> 4, c = i & 15;
  if (k >= mv.k) return;
  // (every thread resolves ITS chain: the sixteen `aborted` flags are behind two dependent loads each -- looked at one chain after
  // the other by every thread, they were 16 us of an 18 us kernel)
  double v = 0.0;
  if (c < ma.nc) {
    const MfmLeaf lc = ma.c[c];
    const MfmChainConst* kc = konst + lc.slot;
    const int dead = lc.io.mode == MODE_TREE ? __hip_atomic_load(&lc.ctl->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (!dead) {
      ArenaDev A;
      A.S = kc->A.S; A.n = kc->A.n; A.Q = kc->A.Q; A.P = kc->A.P; A.G = kc->A.G; A.var = kc->A.var;
      Leaf l2; QView q2;
      resolve_leaf(lc.io, A, lc.j, l2, q2);
      v = q2.q[k] - mv.mu[k];
    }
  }
  dpack[i] = v;
}

// Grid: MFM_MAXC control workgroups + al_nwg row workgroups of 8 rows (the layout of `k_mvn_aligned<8>`: every CU streams);
// block: 16 waves.  The 16 x 16 tile carries 8 real rows: lanes whose row index is 8 .. 15 load nothing and multiply zeros
// (the matrix cores are idle most of the launch either way).
__global__ __launch_bounds__(MFM_WAVES* WAVE, 1) void k_mvn_mfma_multi(MvnDev mv, const ModelDev* __restrict__ mdp, const MfmChainConst* __restrict__ konst, MfmArgs ma,
                                                                       const double* __restrict__ dpack) {
  const int nc = ma.nc;
#ifdef NUTS_KTIMING
  struct { long long* ticks; } md{mdp->ticks};
  const bool tk = (int)blockIdx.x == MFM_MAXC + mv.al_nwg / 2 && threadIdx.x == 0;   // (lab build, tools/wide_ticks.py)
#endif
  TICK(md, tk, 30);
  if ((int)blockIdx.x < MFM_MAXC) {
    if (threadIdx.x >= VEC_THREADS) return;
    int ci = -1;
#pragma unroll
    for (int c = 0; c < MFM_MAXC; ++c) ci = (c < nc && (int)blockIdx.x == ma.c[c].slot) ? c : ci;
    if (ci < 0) return;
    if (!ma.c[ci].fold) return;
    const MvaLeafArgs L = mfm_leaf(ma.c[ci], konst[ma.c[ci].slot]);
    TICK(md, blockIdx.x == 0 && threadIdx.x == 0, 38);
    // (the model itself -- 1.3 KB the control code reads a few words of -- also lies in device memory: it does not fit next to the
    // chains' arguments)
    mva_control(*mdp, L.A, L.cio, L.cj, L.cd, L.Emax, L.max_depth, L.st, L.cseq, L.par ^ 1, VEC_THREADS, L.al_part);
    TICK(md, blockIdx.x == 0 && threadIdx.x == 0, 39);
    return;
  }
  constexpr int R = MFM_R;
  const int b = (int)blockIdx.x - MFM_MAXC;
  __shared__ double s_part[MFM_WAVES][R][MFM_MAXC + 1];   // (+ 1: the tail's column reads fall on different banks)
  __shared__ double s_red[MFM_MAXC][NDOT];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = mv.k, row0 = b * R;
  // ---- which chains are alive; the position vector each column of B reads ----
  const int chain = lane & 15, ks = lane >> 4;
  TICK(md, tk, 31);
  // ---- the tail of chain w: request what is known at launch ----
  Leaf lf; QView qv; MergePrefetch mpf;
  double phv = 0.0, qr = 0.0, mur = 0.0, var_r = 0.0;
  const int my = min(row0 + min(lane, R - 1), K - 1);
  bool mine_alive = false;
  MvaLeafArgs L;
  if (w < nc) {
    const int dead = ma.c[w].io.mode == MODE_TREE ? __hip_atomic_load(&ma.c[w].ctl->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    L = mfm_leaf(ma.c[w], konst[ma.c[w].slot]);   // (wave-uniform: scalar registers)
    mine_alive = !dead;
    if (mine_alive) {
      resolve_leaf(L.io, L.A, L.j, lf, qv);
      const bool leaf = L.io.mode != MODE_PLAIN, tree = L.io.mode == MODE_TREE;
      if (leaf) { phv = L.A.P[lf.d_o + my]; var_r = L.A.var[my]; }
      qr = qv.q[my]; mur = mv.mu[my];
      if (tree) merge_prefetch(L.A, lf, L.j, my, mpf);
    }
  }
  TICK(md, tk, 32);
  // ---- the stream: this wave's column steps through the matrix cores ----
  mfm_v4d acc = {0.0, 0.0, 0.0, 0.0};
  {
    const bool arow = (lane & 15) < R;
    const double* pr = mv.prec + (int64_t)min(row0 + (lane & (R - 1)), K - 1) * K + 4 * ks;
    const double* db = dpack + (int64_t)(4 * ks) * MFM_MAXC + chain;
    const int nsteps = K / 16;         // (K is a multiple of 16: checked when the group is formed)
#pragma unroll 4
    for (int s = w; s < nsteps; s += MFM_WAVES) {
      const int kk = 16 * s;
      double2 a01 = {0.0, 0.0}, a23 = {0.0, 0.0};
      if (arow) { a01 = *reinterpret_cast<const double2*>(pr + kk); a23 = *reinterpret_cast<const double2*>(pr + kk + 2); }
      const double* dk = db + (int64_t)kk * MFM_MAXC;
      const double b0 = dk[0], b1 = dk[MFM_MAXC], b2 = dk[2 * MFM_MAXC], b3 = dk[3 * MFM_MAXC];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b3, acc, 0, 0, 0);
    }
  }
  TICK(md, tk, 33);
  // accumulator register t of lane l = Y[(l >> 4) + 4 t][l & 15]: rows 0 .. 7 are t = 0, 1
  s_part[w][ks][chain] = acc[0];
  s_part[w][ks + 4][chain] = acc[1];
  __syncthreads();
  TICK(md, tk, 34);
  if (w >= nc || !mine_alive) return;
  // ---- wave c = w finishes chain c: lane r < 8 holds row row0 + r ----
  double t = 0.0;
  {
    const int r = min(lane, R - 1);
#pragma unroll
    for (int ww = 0; ww < MFM_WAVES; ++ww) t += s_part[ww][r][w];
  }
  mvm_tail_core<R>(mv, L, b, t, s_red[w], lf, qv, phv, var_r, qr, mur, mpf);
  TICK(md, tk, 35);
}

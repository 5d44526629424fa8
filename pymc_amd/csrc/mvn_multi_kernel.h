// Lockstep chains of ONE MvNormal model on one GPU (BASELINE configs[2]: four chains; the reference runs chains as independent
// processes, pymc/sampling/mcmc.py:1385-1500, and vmaps them on an accelerator, sampling/jax.py:341-348).
//
// A chain's leapfrog on the row-aligned pass (kernels.h, k_mvn_aligned) streams the precision matrix P -- 33.5 MB at k = 2048 --
// against ONE right-hand side; four chains in four processes stream it four times.  `k_mvn_aligned_multi<R, NC>` is the same
// launch for NC chains that happen to stand at a tree leaf at the same time (engine.hip, the chain group: each chain's host thread
// deposits its launch and the last one to arrive submits them together): the streaming waves of a workgroup read its R rows of P
// once and keep NC sets of partial sums; one wave PER CHAIN then finishes that chain's R elements (second kick, v', merge dot
// products, first half of the next leaf, the workgroup's record) exactly as the single-chain kernel's tail wave does;
// workgroups 0 .. NC-1 carry the control work each chain folds into its launch.  A chain's numbers do not depend on its company:
// every fma of its mat-vec, every wave sum and every record is formed from its own operands in the single-chain kernel's order,
// so a chain in a group is BITWISE the chain run alone (tests/test_gpu_chain_group.py), whatever the interleaving.
//
// Chains keep their own arena, control block, uniform stream, status words and records (`al_part`): nothing is shared but the
// read-only node data (P, mu) and the in-order stream the group submits to.
#pragma once

#define MVM_MAXC 4
#ifndef MVM_OCC
#define MVM_OCC 4            // waves per SIMD the register budget allows: 4 = two workgroups of 4 + NC waves per CU
#endif

struct MvaLeafArgs {   // one chain's arguments of k_mvn_aligned
  ArenaDev A;
  EvalIO io, cio;
  double Emax;
  HostStatus* st;
  double* al_part;
  int j, fold, d, max_depth, par, cj, cd, cseq;
  int slot, pad;   // the chain's place in its group: its control work always runs in workgroup `slot`, i.e. on the same XCD, whose L2
                   // then keeps the chain's control block, uniforms and status words from launch to launch (the order in which
                   // chains arrive at a launch changes; with the control work following it, every launch began with misses)
};

template <int NC>
struct MvaMultiArgs {
  MvaLeafArgs c[NC];
};

// NV sums over the wave at once, each in `wave_sum`'s association order -- the balanced pairwise tree over the lanes, (0 + 1),
// (2 + 3), pairs of pairs, ... (device_math.h: the row_shr / row_bcast steps build exactly that tree, and a + b is b + a bit for
// bit) -- for the price of one: at step s lanes l and l ^ 2^s, which hold the same half-reduced values, each keep one half of
// them and hand the other half over, so the exchanges shrink 16, 8, 4, ... instead of NV full trees of six dependent steps (the
// 32 sums of four chains cost more than the mat-vec's loads that way; measured).  Returns the total of value `idx` (out) in
// lanes l < NVP; values NV .. NVP - 1 are padding.
// the value lane l ^ DIST holds: inside a quad on the DPP network (VALU speed), across quads through the LDS crossbar
template <int DIST>
__device__ __forceinline__ double lane_xor(double x) {
  if constexpr (DIST == 1) return dpp_move_d<0xB1, 0xf>(x);        // quad_perm:[1,0,3,2]
  else if constexpr (DIST == 2) return dpp_move_d<0x4E, 0xf>(x);   // quad_perm:[2,3,0,1]
  else return __shfl_xor(x, DIST, WAVE);
}

template <int NV, int N, int S>
__device__ __forceinline__ void wave_sum_many_step(double (&v)[NV], int lane, int& idx) {
  if constexpr (N > 1) {
    constexpr int half = N >> 1;
    const bool up = (lane >> S) & 1;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const double keep = up ? v[i + half] : v[i], send = up ? v[i] : v[i + half];
      v[i] = keep + lane_xor<(1 << S)>(send);
    }
    idx += up ? half : 0;
    wave_sum_many_step<NV, half, S + 1>(v, lane, idx);
  } else if constexpr ((1 << S) < WAVE) {
    v[0] += lane_xor<(1 << S)>(v[0]);
    wave_sum_many_step<NV, 1, S + 1>(v, lane, idx);
  }
}

template <int NV>
__device__ __forceinline__ double wave_sum_many(const double (&in)[NV], int lane, int& idx) {
  constexpr int NVP = NV <= 1 ? 1 : NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : 32;
  static_assert(NV <= 32, "one value per lane pair at the most");
  double v[NVP];
#pragma unroll
  for (int i = 0; i < NVP; ++i) v[i] = i < NV ? in[i] : 0.0;
  idx = 0;
  wave_sum_many_step<NVP, NVP, 0>(v, lane, idx);
  return v[0];
}

// the tail wave of chain `L` in workgroup b (lane = element row0 + lane): k_mvn_aligned's, on this chain's sums
// (`t`: this lane's row of P (q - mu), summed over the workgroup -- by whatever streamed the rows)
template <int R>
__device__ __forceinline__ void mvm_tail_core(const MvnDev& mv, const MvaLeafArgs& L, int b, double t, double* s_red,
                                              const Leaf& lf, const QView& qv, double phv, double var_r, double qr, double mur,
                                              const MergePrefetch& mpf) {
  const ArenaDev& A = L.A;
  const EvalIO& io = L.io;
  const int lane = threadIdx.x & (WAVE - 1);
  const int K = mv.k, row0 = b * R, j = L.j, d = L.d;
  const bool leaf = io.mode != MODE_PLAIN, tree = io.mode == MODE_TREE;
  const int my = min(row0 + min(lane, R - 1), K - 1);
  const bool a0 = lane < R && row0 + lane < K;
  int idx[1] = {my};
  bool act[1] = {a0};
  double grad[1] = {-t}, ph[1] = {phv};
  if (a0) {
    if (leaf) A.G[lf.d_o + my] = -t;
    else io.grad[my] = -t;
  }
  int m = 0; bool last = false;
  if (leaf) leaf_post<1, false, R <= 8>(A, lf, j, d, tree, idx, act, grad, ph, s_red, 1, m, last, tree ? &mpf : nullptr, 0);   // -> s_red[k]
  if (leaf && io.pre_next == 3 && a0) {
    const int e = lf.dir > 0 ? lf.left : lf.right;
    const int64_t eo = slot_off(A, e), no = slot_off(A, e - lf.dir);
    const double eps2 = -lf.eps, half2 = 0.5 * eps2;
    const double ph2 = fma(half2, A.G[eo + my], A.P[eo + my]);
    A.P[no + my] = ph2;
    A.Q[no + my] = fma(eps2, var_r * ph2, A.Q[eo + my]);
  } else if (leaf && io.pre_next && a0) {
    const int64_t no = slot_off(A, lf.t + lf.dir);
    const double p = fma(lf.half, -t, phv);
    const double phn = fma(lf.half, -t, p);
    A.P[no + my] = phn;
    A.Q[no + my] = fma(lf.eps, var_r * phn, qr);
  }
  const double lp = (R <= 8 ? wave_sum8(a0 ? -0.5 * (qr - mur) * t : 0.0) : wave_sum(a0 ? -0.5 * (qr - mur) * t : 0.0));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int nwg = mv.al_nwg;
  double* rec = L.al_part + ((int64_t)L.par * nwg + b) * MVA_RS;
  const int nn = 1 + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
  for (int qq = lane; qq < nn; qq += WAVE) {
    const int k = qq < 2 + 6 * m ? max(qq - 1, 0) : DOT_TOP + (qq - 2 - 6 * m);
    rec[qq] = qq == 0 ? lp : s_red[k];
  }
}

template <int R>
__device__ __forceinline__ void mvm_tail(const MvnDev& mv, const MvaLeafArgs& L, int b, const double (*s_w)[MVN_BLOCK / WAVE], double* s_red,
                                         const Leaf& lf, const QView& qv, double phv, double var_r, double qr, double mur,
                                         const MergePrefetch& mpf) {
  const int lane = threadIdx.x & (WAVE - 1);
  double t = 0.0;
#pragma unroll
  for (int ww = 0; ww < MVN_BLOCK / WAVE; ++ww) t += s_w[min(lane, R - 1)][ww];
  mvm_tail_core<R>(mv, L, b, t, s_red, lf, qv, phv, var_r, qr, mur, mpf);
}

// Grid: MVM_MAXC control workgroups (a chain's in workgroup `slot`; the others leave at once) + al_nwg row workgroups, so that the rows of
// workgroup b stay on one XCD whatever NC is.  Block: four streaming waves + NC tail waves.
//
// Register budget: 128 per lane, so that TWO workgroups fit a CU -- the launch has 4 workgroups more than the chip has CUs, and at
// one workgroup per CU (190 registers, four column steps in flight) the last four row workgroups started when the first control
// workgroup had finished: 18 us instead of 13 for four chains (measured, profiles/LAB.md).  Measured and dropped as well: eight
// streaming waves per workgroup, each with half the rows (twice the loads in flight, half the sums per lane, q read twice) --
// 16.8 us.
#define MVM_THREADS(NC) (MVN_BLOCK + (NC) * WAVE)
template <int R, int NC>
__global__ __launch_bounds__(MVM_THREADS(NC), MVM_OCC) void k_mvn_aligned_multi(ModelDev md, MvaMultiArgs<NC> ma) {
  const MvnDev& mv = md.mv;
  if (blockIdx.x < MVM_MAXC) {
    if (threadIdx.x >= VEC_THREADS) return;
    // (ONE copy of the control code and of the tail code below, indexed by the chain: with a copy per chain the four-chain kernel
    // was 150 KB of instructions, four kernels of the kind alternate, and every launch began with instruction-cache misses)
    int ci = -1;
#pragma unroll
    for (int c = 0; c < NC; ++c) ci = (int)blockIdx.x == ma.c[c].slot ? c : ci;
    if (ci < 0) return;
    const MvaLeafArgs& L = ma.c[ci];
    TICK(md, NC == MVM_MAXC && ci == 0 && threadIdx.x == 0, 24);   // (NUTS_KTIMING builds: tools/lockstep_ticks.py)
    if (L.fold) mva_control(md, L.A, L.cio, L.cj, L.cd, L.Emax, L.max_depth, L.st, L.cseq, L.par ^ 1, VEC_THREADS, L.al_part);
    TICK(md, NC == MVM_MAXC && ci == 0 && threadIdx.x == 0, 25);
    return;
  }
  const int b = (int)blockIdx.x - MVM_MAXC;
  constexpr int TW = MVN_BLOCK / WAVE;
  __shared__ double s_w[NC][R][TW];
  __shared__ double s_red[NC][NDOT];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  const int K = mv.k, row0 = b * R;
  const double* __restrict__ mu = mv.mu;
  const bool tk = NC == MVM_MAXC && b == mv.al_nwg / 2 && lane == 0;
  TICK(md, tk && w == 0, 16);
  TICK(md, tk && w == TW, 20);
  int dead = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) dead |= load_aborted(ma.c[c].io, ma.c[c].A) ? (1 << c) : 0;
  if (dead == (1 << NC) - 1) return;   // every chain's tree has ended: the launch drains
  // ---- tail waves: request what is known at launch (as the single-chain kernel's fifth wave does), wait for the sums ----
  Leaf lf; QView qv; MergePrefetch mpf;
  double phv = 0.0, qr = 0.0, mur = 0.0, var_r = 0.0;
  const int my = min(row0 + min(lane, R - 1), K - 1);
  const int tc = max(__builtin_amdgcn_readfirstlane(w) - TW, 0);   // tail waves: the chain this wave finishes
  if (w >= TW) {
    if ((dead >> tc) & 1) return;   // (this chain drains; a finished wave does not hold the barrier)
    const MvaLeafArgs& L = ma.c[tc];
    resolve_leaf(L.io, L.A, L.j, lf, qv);
    const bool leaf = L.io.mode != MODE_PLAIN, tree = L.io.mode == MODE_TREE;
    if (leaf) { phv = L.A.P[lf.d_o + my]; var_r = L.A.var[my]; }
    qr = qv.q[my]; mur = mu[my];
    if (tree) merge_prefetch(L.A, lf, L.j, my, mpf);
  } else {
    // (addresses = a wave-uniform base + ONE 32-bit byte offset per lane: the loads take the base from scalar registers, and the
    // thirteen address pairs per column step stay out of the vector registers the sums need)
    const char* q[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      Leaf l2; QView q2;
      resolve_leaf(ma.c[c].io, ma.c[c].A, ma.c[c].j, l2, q2);
      q[c] = reinterpret_cast<const char*>(q2.q);
    }
    const char* pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = reinterpret_cast<const char*>(mv.prec + (int64_t)min(row0 + r, K - 1) * K);
    const char* mub = reinterpret_cast<const char*>(mu);
    double s[NC][R];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) s[c][r] = 0.0;
    const unsigned k2b = (unsigned)(K & ~1) * 8u;
#ifdef NUTS_KTIMING
    int it = 0;
#endif
#pragma unroll 2
    for (unsigned off = 16u * (unsigned)tid; off < k2b; off += 16u * MVN_BLOCK) {
      const double2 m2 = *reinterpret_cast<const double2*>(mub + off);
      double d0[NC], d1[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
#ifdef MVM_KO_Q   // (lab build, tools/lockstep_ticks.py: every chain multiplies chain 0's position -- wrong numbers, one q stream)
        const double2 qq = *reinterpret_cast<const double2*>(q[0] + off);
        d0[c] = qq.x - m2.x + c; d1[c] = qq.y - m2.y + c;
#else
        const double2 qq = *reinterpret_cast<const double2*>(q[c] + off);
        d0[c] = qq.x - m2.x; d1[c] = qq.y - m2.y;
#endif
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double2 p = *reinterpret_cast<const double2*>(pr[r] + off);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          s[c][r] = fma(p.x, d0[c], s[c][r]);
          s[c][r] = fma(p.y, d1[c], s[c][r]);
        }
      }
#ifdef NUTS_KTIMING
      TICK(md, tk && w == 0 && it < 4, 26 + it);   // (column steps one by one: slots 26 .. 29)
      ++it;
#endif
    }
    if (tid == 0 && (K & 1)) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const double dl = reinterpret_cast<const double*>(q[c])[K - 1] - mu[K - 1];
#pragma unroll
        for (int r = 0; r < R; ++r) s[c][r] = fma(reinterpret_cast<const double*>(pr[r])[K - 1], dl, s[c][r]);
      }
    }
    TICK(md, tk && w == 0, 17);
    double flat[NC * R];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) flat[c * R + r] = s[c][r];
    int idx;
    const double t = wave_sum_many<NC * R>(flat, lane, idx);
    constexpr int NVP = NC * R <= 2 ? 2 : NC * R <= 4 ? 4 : NC * R <= 8 ? 8 : NC * R <= 16 ? 16 : 32;
    if (lane < NVP && idx < NC * R) s_w[idx / R][idx % R][w] = t;
  }
  TICK(md, tk && w == 0, 18);
  TICK(md, tk && w == TW, 21);
  __syncthreads();
  TICK(md, tk && w == 0, 19);
  if (w < TW) return;
  TICK(md, tk && w == TW, 22);
  mvm_tail<R>(mv, ma.c[tc], b, s_w[tc], s_red[tc], lf, qv, phv, var_r, qr, mur, mpf);
  TICK(md, tk && w == TW, 23);
}

// Persistent tree kernel for the group-aligned row pass: ONE launch per NUTS transition.
//
// k_rows_ga (rows_ga_kernel.h) is launched once per leapfrog step; what separates two steps there is a kernel boundary: the
// tail of step j (combine, kick, dot products, records, tickets, block partials), the drain, the dispatch of the next grid, and
// its prologue (block partials -> mu', sigma' -> beta) -- 11-13 us during which HBM carries nothing but the first two tiles per
// wave.  All G + 1 workgroups of that launch are resident at once anyway (that is how the grid is sized), so the leaf loop of
// `_build_subtree` and the doubling loop of `NUTS._hamiltonian_step` (nuts.py:204-225, 394-476) can run INSIDE the launch:
//
//   * workgroup g + 1 streams group g for every leaf of the tree.  When a wave has consumed its last tile of leaf L it requests
//     its first two tiles of leaf L + 1 at once (X does not depend on the state), so HBM stays busy while wave 0 of the workgroup
//     runs the tail of leaf L and while everybody waits for the other groups; wave 0 requests its own two tiles after its tail
//     (in-order `vmcnt`: anything it waited for in the tail would wait for the tiles first).
//   * the kernel boundary becomes a counter: the last arriver of each block publishes the block partial write-through and then
//     increments `ga_sync[GA_SYNC_DONE]`; leaf L + 1 starts in a workgroup when the counter has reached (L + 1) ga_nblk.  One wave
//     per workgroup polls (relaxed agent-scope load + s_sleep), the others wait at the workgroup barrier.  Every wait is BOUNDED
//     (constant 100 MHz clock): if the workgroups were ever not co-resident the launch ends with ST_TIMEOUT instead of hanging.
//   * workgroup 0 is the control workgroup of the folded schedule (kernels.h, control_lean), now persistent: it consumes the
//     block partials of leaf L while the row workgroups stream leaf L + 1, writes its results back (agent-scope release) and
//     publishes `ga_sync[GA_SYNC_CTL] = L + 1`.  Row workgroups only ever see its `aborted` flag, exactly as the folded launches
//     do: a leaf that starts while the control work of its predecessor decides to stop writes into slots nobody reads.
//     The double-buffered records (block partials, local parts of mu / sigma) of leaf L are overwritten by leaf L + 2, so leaf
//     L + 2 does not start before the control work of leaf L is finished (same poll, second word).
//   * doublings follow each other inside the launch.  The direction of doubling d + 1 is uniforms[2^(d+1) + d] whenever it is
//     needed at all (a completed doubling consumes a fixed number of uniforms; the host look-ahead of run_tree relies on the same
//     fact), so the row workgroups compute the geometry themselves; beyond `spec_depth` (as deep as the previous draw's tree
//     went) they first wait for the verdict of the control workgroup, so that the usual end of a tree costs no speculative pass.
//     The control workgroup checks the prediction against the control block and publishes the status word of the whole tree.
//
// Cross-workgroup data inside the launch: write-through (`st_agent`) stores read with L1-bypassing (`ld_agent`) loads, the
// counter incremented after `s_waitcnt vmcnt(0)` (MI355X guide, "sc1 stores and sc1 loads both sides").  The control workgroup's
// arena writes (P, V, G of the mu / sigma elements) are plain stores followed by ONE agent-scope release per leaf; the only
// reader in another workgroup is the first leaf of a doubling that continues from the OTHER end of the trajectory, which does
// an agent-scope acquire before it reads them (all other leaves take mu / sigma from the previous leaf's records).
// Everything else a workgroup reads back was written by itself (its group's z elements).
//
// The sums are the sums of k_rows_ga, in the same order: draws are bitwise those of the launch-per-leaf schedule
// (tests/test_gpu_benchmark_shapes.py::test_tree_kernel_is_a_pure_rescheduling).
#pragma once
#include "rows_ga_kernel.h"

struct GaTreeArgs {
  ModelDev md; ArenaDev A;
  int max_depth, spec_depth, rev0, par0, first_dir, seq, alternate;
  int opts;            // GA_TREE_* experiment / diagnostic switches (0 in production)
  double Emax, eps_abs;
  HostStatus* st;
  long long timeout;   // ticks of the constant 100 MHz clock one wait may last
  long long* dbg;      // diagnostics (NUTS_GA_TREE_DBG): [G][8] per-workgroup timestamps of ONE leaf, or nullptr
  int dbg_leaf, dbg_pad;
};

#define GA_TREE_EARLY0 1      // wave 0 also requests its first tiles before it polls (they are in flight during its prologue)
#define GA_TREE_REVMAP 2
#define GA_TREE_PHASE_G 4     // experiment: odd groups stream their halves in the opposite order (half the waves hit the
#define GA_TREE_PHASE_W 8     // Infinity Cache while the other half miss) / odd waves do
#define GA_TREE_TICK_SHIFT 8  // opts >> 8 = 1 + first leaf whose phase timestamps go to md.ticks (8 leaves x 8 stamps, workgroup G / 2)

__device__ __forceinline__ long long ga_clock() { return (long long)wall_clock64(); }

// Wait until `done` block partials have been published and the control workgroup has finished `ctl` leaves.
// Whole-wave call (every lane loads the same words).  0: reached; 1: the tree was stopped; 2: timed out.
__device__ __forceinline__ int ga_wait(const unsigned* sync, unsigned need_done, unsigned need_ctl, const int* aborted, long long timeout) {
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(sync);   // {GA_SYNC_DONE, GA_SYNC_CTL}
  const long long t0 = ga_clock();
  for (;;) {
    const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ab = aborted ? __hip_atomic_load(aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (ab) return 1;   // (the flag is written back before the progress word that follows it)
    if ((int)((unsigned)v - need_done) >= 0 && (int)((unsigned)(v >> 32) - need_ctl) >= 0) return 0;
    if (ga_clock() - t0 > timeout) return 2;
    __builtin_amdgcn_s_sleep(4);
  }
}

__device__ __forceinline__ void ga_fail(const ArenaDev& A, unsigned who) {   // a wait timed out: stop everybody
  __hip_atomic_store(&A.ga_sync[GA_SYNC_ERR], who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&A.ctl->aborted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// values that come back from LDS / memory / a function argument are wave-uniform, but only `readfirstlane` tells the compiler
__device__ __forceinline__ int ga_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t ga_uni64(int64_t v) {
  return ((int64_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
}
template <typename P>
__device__ __forceinline__ P* ga_uni_ptr(P* p) { return reinterpret_cast<P*>(ga_uni64(reinterpret_cast<int64_t>(p))); }

__device__ __forceinline__ EvalIO ga_tree_io(int dir, int edge, int left, int right, double eps_abs) {
  EvalIO io{};
  io.mode = MODE_TREE; io.lean = 1;
  io.dir = dir; io.edge = edge; io.left = left; io.right = right; io.eps = dir > 0 ? eps_abs : -eps_abs;
  return io;
}

// direction of doubling d >= 1 of a tree whose earlier doublings ran to completion (nuts.py:215 on the pre-drawn stream)
__device__ __forceinline__ int ga_tree_dir(const double* uniforms, int d) { return uniforms[(1 << d) + d - 1] < 0.5 ? 1 : -1; }

// ---- the control workgroup -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ga_tree_control(const GaTreeArgs& a, int (&s_flag)[2]) {
  const ModelDev& md = a.md;
  const ArenaDev& A = a.A;
  const int tid = threadIdx.x;
  const int nblk = md.lg.ga_nblk;
  int dir = a.first_dir, edge = 0, left = 0, right = 0;
  int L = 0;
  if (A.ctl->aborted) {   // bad initial energy (k_draw_ctl_start): no leaf will run
    if (tid == 0) publish_status(A.ctl, a.st, a.seq);
    return;
  }
  for (int d = 0; d < a.max_depth; ++d) {
    if (d > 0) {
      // geometry of this doubling: the control block is authoritative (tree_decide -> ctl_next_direction); the row workgroups
      // predicted it from the uniform stream
      const Ctl* c = A.ctl;
      const int cdir = c->dir;
      left = c->left; right = c->right; edge = c->edge;
      if (cdir != ga_tree_dir(A.uniforms, d)) {   // cannot happen (see the header); never let a wrong trajectory through
        if (tid == 0) { ga_fail(A, 3u); Ctl* cw = A.ctl; cw->aborted = 1; cw->bad_energy = 1; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); publish_status(cw, a.st, a.seq); }
        return;
      }
      dir = cdir;
    }
    const EvalIO io = ga_tree_io(dir, edge, left, right, a.eps_abs);
    const int nleaf = 1 << d;
    for (int j = 0; j < nleaf; ++j, ++L) {
      const int par = a.par0 ^ ((L + 1) & 1);
      if (tid < WAVE) {
        // (`aborted` can only be set under this workgroup by a row workgroup whose own wait timed out)
        const int r = ga_wait(A.ga_sync, (unsigned)(L + 1) * (unsigned)nblk, 0u, &A.ctl->aborted, a.timeout);
        if (tid == 0) s_flag[0] = r;
      }
      __syncthreads();
      if (s_flag[0]) {
        if (tid == 0) {
          ga_fail(A, 2u);
          const unsigned flags = ST_ABORTED | ST_BAD_ENERGY | ST_TIMEOUT;
          __hip_atomic_store(&a.st->word[a.seq & (ST_SLOTS - 1)], ((unsigned long long)(unsigned)a.seq << 32) | flags, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
      control_lean<true>(md, A, io, j, d, a.Emax, a.max_depth, (HostStatus*)nullptr, 0, lean_src(md, par));
      __syncthreads();
      // results out (P, V, G, PS, PSUM of the mu / sigma elements, E, LOGP, the control block), then the progress word
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&A.ga_sync[GA_SYNC_CTL], (unsigned)(L + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const Ctl* c = A.ctl;
      const bool stop = c->aborted != 0 || (j + 1 == nleaf && d + 1 >= a.max_depth);
      if (stop) {
        if (tid == 0) publish_status(c, a.st, a.seq);
        return;
      }
      __syncthreads();   // (s_flag is rewritten by the next wait)
    }
  }
}

// ---- one leaf's tail in a row workgroup (wave 0): as ga_tail, with write-through stores for what other workgroups read ----
template <int D>
__device__ __forceinline__ void ga_tree_tail(const GaTreeArgs& T, const EvalIO& io, int g, int j, int d, int par, const MergePrefetch& mpf,
                                             double (&s_acc)[GA_MAXW][2][D + 1], double (&s_red)[NDOT], int (&s_info)[4],
                                             double (&s_keep)[5][WAVE]) {
  const ModelDev& md = T.md;
  const ArenaDev& A = T.A;
  const RowsDev& R = md.lg;
  const int lane = threadIdx.x & (WAVE - 1), W = (int)blockDim.x >> 6;
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
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
  if (g == 0) {   // the hyper-parameter elements' local parts + their q': read by every workgroup at the next leaf
    const int e = lane;
    const bool hact = e < 2 * D, is_mu = e < D;
    double gx, dxdq, dj, lpd;
    ga_def_local(R, is_mu, hval, gx, dxdq, dj, lpd);
    lpg += wave_sum(hact ? lpd : 0.0);
    if (hact) {
      const int dd = is_mu ? e : e - D;
      const int slot = (is_mu ? R.def_mu : R.def_sigma) + dd;
      double* loc = md.def_loc + (int64_t)par * 4 * MAX_DEFERRED + 4 * slot;
      st_agent(loc, gx); st_agent(loc + 1, dxdq); st_agent(loc + 2, dj); st_agent(loc + 3, hph);
      st_agent(A.Q + lf.d_o + (is_mu ? R.off_mu : R.off_sigma) + dd, hval);
    }
  }
  int m = 0; bool last = false;
  leaf_post<1>(A, lf, j, d, true, idx, act, grad, ph, s_red, 1, m, last, &mpf);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- this group's record, write-through ----
  double* rec = R.ga_part + (int64_t)g * PART_STRIDE;
  if (lane == 0) st_agent(rec + PART_LP, lpg);
  if (zact) { st_agent(rec + PART_DMU + lane, db); st_agent(rec + PART_DSG + lane, db * zq); }
  for (int k = lane; k < NDOT; k += WAVE)
    if (dot_needed(k, m, last)) st_agent(rec + PART_DOT + k, s_red[k]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has left this CU before the ticket is taken
  const int blk = g / R.ga_bsz;
  const int cnt = min(R.G, (blk + 1) * R.ga_bsz) - blk * R.ga_bsz;
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(&R.ga_ticket[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  const int is_last = (int)old + 1 == cnt;
  if (is_last && lane == 0) __hip_atomic_store(&R.ga_ticket[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) { s_info[0] = is_last; s_info[1] = m; s_info[2] = last ? 1 : 0; }
}

// ---- the block's last arriver: block partial = sum of the block's records (chunks of 8 groups, chunks in order), then the
// progress counter.  Whole workgroup. ----
template <int D>
__device__ __forceinline__ void ga_tree_block_reduce(const GaTreeArgs& T, int g, int par, double (&s_cp)[GA_MAXCHUNK][PART_STRIDE],
                                                     const int (&s_info)[4]) {
  const RowsDev& R = T.md.lg;
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
    st_agent(bp + k, sum);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its part of the block partial has left the CU
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(&T.A.ga_sync[GA_SYNC_DONE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// Register budget.  The streaming loop needs every vector register four waves per SIMD allow (two tiles in flight = 66, the
// accumulators, exp / log1p), and a spill there is worse than slow: a reload waits, in order, for every tile requested before it,
// and a tile register saved before its counted wait is saved before its load has landed.  Two rules keep the loop spill-free:
//   * nothing heavy runs while tiles are in flight.  Waves 1 .. W-1 request their first two tiles of the leaf and go straight to
//     the workgroup barrier; wave 0 polls for the previous leaf, finishes mu' / sigma' for the whole workgroup (one wave's loads
//     instead of three), passes them through LDS, and requests ITS tiles after the barrier.  The two cases are exclusive
//     branches, so the tile registers are dead in wave 0's branch.
//   * the control workgroup's code is a separate (`noinline`) function: it is called once per launch by one workgroup, and its
//     interpreter-sized register needs stay out of this function's allocation.
// (the kernarg-segment builtin folds to a null pointer outside a kernel: the pointer is an argument)
__device__ __noinline__ void ga_tree_control_nf(const GaTreeArgs* args) {
  __shared__ int s_flag[2];
  ga_tree_control(*args, s_flag);
}

// grid = G + 1 workgroups of W waves (workgroup 0: control), all resident; D = 8, two rows per lane, two tiles in flight per wave
template <int OCC, int DX = 8>
__global__ __launch_bounds__(64 * GA_MAXW, OCC) void k_tree_ga(GaTreeArgs a) {
  constexpr int D = 8, RPL = 2, SPAN = WAVE * RPL;
  typedef typename GaTileSel<DX>::type Tile;
  constexpr int LOADS = DX + 1;
  if (blockIdx.x == 0) { ga_tree_control_nf((const GaTreeArgs*)__builtin_amdgcn_kernarg_segment_ptr()); return; }
  __shared__ double s_acc[GA_MAXW][2][D + 1];   // [wave][first / second half of its tiles][d/dbeta, log-lik]
  __shared__ double s_red[NDOT];
  __shared__ double s_cp[GA_MAXCHUNK][PART_STRIDE];
  __shared__ int s_info[4];                      // {this workgroup is its block's last arriver, m, last}
  __shared__ int s_flag[2];
  __shared__ double s_hyp[2][2 * D];             // q' and p_half of mu[0..D), sigma[0..D) at this leaf (wave 0 -> everybody)
  __shared__ double s_keep[5][WAVE];             // wave 0's per-lane prologue values the tail needs again
  __shared__ __attribute__((aligned(16))) char s_args[(sizeof(GaTreeArgs) + 15) / 16 * 16];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6), W = (int)blockDim.x >> 6;
  // (GA_TREE_REVMAP, diagnostics: workgroup b streams group G - b instead of b - 1 -- does a slow workgroup follow its place in
  // the dispatch order or the address of its rows?)
  const int g = (a.opts & GA_TREE_REVMAP) ? a.md.lg.G - (int)blockIdx.x : (int)blockIdx.x - 1;
  // the tail reads the arguments from an LDS copy of the kernarg segment (rows_ga_kernel.h: nothing it needs has to stay in
  // scalar registers across the streaming loop)
  {
    const uint4* ka = (const uint4*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int t = tid; t < (int)((sizeof(GaTreeArgs) + 15) / 16); t += (int)blockDim.x) reinterpret_cast<uint4*>(s_args)[t] = ka[t];
  }
  __syncthreads();
  const RowsDev& R = a.md.lg;
  const ArenaDev& A = a.A;
  // ---- geometry of this wave's stream: constant over the launch (no memory access when every group has the same number of rows) ----
  int T; int64_t ng, cbase;
  if (R.ga_T_uni > 0) { T = R.ga_T_uni; ng = R.ga_ng_uni; cbase = (int64_t)(g * W + w) * R.ga_cstride_uni; }
  else {
    T = ga_uni(R.ga_tile0[g + 1] - R.ga_tile0[g]);
    ng = ga_uni((int)(R.gptr[g + 1] - R.gptr[g]));
    cbase = ga_uni64(R.ga_coff[g * W + w]);
  }
  const double* const Xw = R.Xt + cbase;        // this wave's chunk
  const int8_t* const yw = R.y + cbase / DX;
  const int c0 = (int)((int64_t)w * T / W), c2 = (int)((int64_t)(w + 1) * T / W);
  const int n = c2 - c0;
  const int nA = (n + 1) / 2;
  constexpr int64_t TS = (int64_t)DX * SPAN;
  const int l_last = (c2 == T) ? n - 1 : -1;
  const int n_last = (int)(ng - (int64_t)(T - 1) * SPAN);
  const int nm1 = max(n - 1, 0);
  const uint32_t voff16 = (uint32_t)lane * 16u, voff2 = (uint32_t)lane * 2u;
  const int max_depth = a.max_depth, spec_depth = a.spec_depth, par0 = a.par0, alternate = a.alternate;
  const int early0 = a.opts & GA_TREE_EARLY0;
  const int tick0 = (a.opts >> GA_TREE_TICK_SHIFT) - 1;   // < 0: no timestamps
  const bool ticker = tick0 >= 0 && g == R.G / 2 && tid == 0;
#define GA_TICK(K) do { if (ticker && L >= tick0 && L < tick0 + 8) a.md.ticks[(L - tick0) * 8 + (K)] = ga_clock(); } while (0)
  // per-workgroup timeline of leaf `dbg_leaf`: {top, beta ready w0, stream end w0, beta ready w1, stream end w1, tail end, hw id, leaf end}
  long long* const dbg = a.dbg;
  const int dbg_leaf = a.dbg_leaf;
#define GA_DBG(K, COND) do { if (dbg && L == dbg_leaf && lane == 0 && (COND)) dbg[(int64_t)g * 8 + (K)] = ga_clock(); } while (0)
  int dir = a.first_dir, edge = 0, left = 0, right = 0;
  int rev = alternate ? (a.rev0 ^ 1) : 0;
  if (a.opts & GA_TREE_PHASE_G) rev ^= g & 1;
  if (a.opts & GA_TREE_PHASE_W) rev ^= w & 1;
  int L = 0;
  bool src_prev = false;   // the source state of the leaf is the previous leaf of this launch
  for (int d = 0; d < max_depth; ++d) {
    const int nleaf = 1 << d;
    for (int j = 0; j < nleaf; ++j, ++L) {
      const int par = par0 ^ ((L + 1) & 1);
      const int nsw = rev ? n - nA : nA;            // position in the sequence where the second-streamed half starts
      // the order of a wave's tiles alternates between leaves (the tail of the previous pass is still in the Infinity Cache)
      auto local_at = [&](int i) { return rev ? (i < nsw ? nA + i : i - nsw) : i; };
      auto issue = [&](int i, Tile& t) {
        const int64_t off = (int64_t)local_at(min(i, nm1)) * TS;
        ga_issue8(Xw + off, yw + off / DX, voff16, voff2, t);
      };
      const EvalIO io = ga_tree_io(dir, edge, left, right, a.eps_abs);
      Leaf lf; QView qv;
      resolve_leaf(io, A, j, lf, qv);
      Tile ta, tb;
      GA_TICK(0);
      GA_DBG(0, w == 0);
      if (w != 0 || early0) {
        // X does not depend on the state: the first two tiles are requested before the wait for the other groups
        issue(0, ta);
        issue(1, tb);
      }
      if (w != 0) {
        if (lane == 0) {
#pragma unroll
          for (int dd = 0; dd <= D; ++dd) { s_acc[w][0][dd] = 0.0; s_acc[w][1][dd] = 0.0; }
        }
        __syncthreads();   // (B)
      } else {
        // ---- wave 0: wait for the previous leaf (its block partials; the control work of the leaf before it), then mu', sigma' ----
        int r = 0;
        if (L > 0) r = ga_wait(A.ga_sync, (unsigned)L * (unsigned)R.ga_nblk, (unsigned)max(L - 1, 0), &A.ctl->aborted, a.timeout);
        else r = __hip_atomic_load(&A.ctl->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        if (r == 2) ga_fail(A, 1u);
        GA_TICK(1);
        if (lane == 0) {
          s_flag[0] = r;
#pragma unroll
          for (int dd = 0; dd <= D; ++dd) { s_acc[0][0][dd] = 0.0; s_acc[0][1][dd] = 0.0; }
        }
        if (r == 0) {
          double hval0, hph0;   // lane l: q' and p_half of hyper-parameter element l mod 2D
          if (src_prev) {
            const LeanSrc prev = lean_src(a.md, par ^ 1);
            rows_hyper_fold_elem<D, true>(R, prev.part, prev.stride, prev.nblk, prev.def_loc, qv, lane, hval0, hph0);
          } else {
            // the source is an edge state of an earlier doubling (or the start state): its mu / sigma entries are in the arena,
            // P and G written by the control workgroup (released leaves ago) -- acquire before reading them
            if (L > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int e = lane % (2 * D);
            const int i = e < D ? R.off_mu + e : R.off_sigma + (e - D);
            hph0 = qv.p_half(i); hval0 = fma(qv.eps, qv.var[i] * hph0, qv.q[i]);
          }
          if (lane < 2 * D) { s_hyp[0][lane] = hval0; s_hyp[1][lane] = hph0; }
        }
        __syncthreads();   // (B)
        GA_TICK(2);
        if (!early0) {
          issue(0, ta);
          issue(1, tb);
        }
      }
      if (s_flag[0]) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
      // ---- z' of this group -> beta (every wave) ----
      double beta[D];
      {
        const int dl = lane % D;
        const int iz = R.off_z + g * D + dl;
        const double zph = fma(qv.half, qv.g[iz], qv.p[iz]);
        const double zq = fma(qv.eps, qv.var[iz] * zph, qv.q[iz]);
        const double m_lane = s_hyp[0][dl];
        const double sraw = s_hyp[0][D + dl];
        const double s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sraw) : sraw;
        const double bl = fma(s_lane, zq, m_lane);
#pragma unroll
        for (int dd = 0; dd < D; ++dd) beta[dd] = readlane_d(bl, dd);
        if (w == 0) {
          s_keep[0][lane] = s_hyp[0][lane % (2 * D)]; s_keep[1][lane] = s_hyp[1][lane % (2 * D)];
          s_keep[2][lane] = zq; s_keep[3][lane] = zph; s_keep[4][lane] = s_lane;
        }
      }
      GA_TICK(3);
      GA_DBG(1, w == 0); GA_DBG(3, w == 1);
      // ---- the stream: two tiles in flight per wave, hand-counted (rows_ga_kernel.h); the requests past the end re-read the
      // wave's last tile (an L2 hit) so that every stage has exactly 9 younger loads behind the tile it waits for ----
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
#define GA_TSTAGE(TR, I)                                                                     \
        {                                                                                    \
          ga_wait8<LOADS>(TR);                                                               \
          if ((I) == nsw) flush();                                                           \
          double xx[8][2];                                                                   \
          ga_unpack(TR, xx);                                                                 \
          ga_tile<8, 2>(xx, TR.y, beta, local_at(I) == l_last ? n_last : SPAN, lane, acc, lp); \
        }
        for (int i = 0; i < n; i += 2) {
          GA_TSTAGE(ta, i)
          if (i + 1 >= n) break;
          issue(i + 2, ta);
          GA_TSTAGE(tb, i + 1)
          issue(i + 3, tb);
        }
        ga_wait8<0>(ta); ga_wait8<0>(tb);
#undef GA_TSTAGE
        if (n > 0) flush();
      }
      // ---- back half: wave 0 finishes the group's z elements and publishes the group's record; the block's last arriver sums
      // the block and moves the progress counter ----
      GA_TICK(4);
      GA_DBG(2, w == 0); GA_DBG(4, w == 1);
      const GaTreeArgs& Targs = *reinterpret_cast<const GaTreeArgs*>(s_args);
      if (w == 0) {
        // the operands of the first merge levels belong to earlier leaves: requested before the wait for the other waves
        MergePrefetch mpf;
        merge_prefetch(Targs.A, lf, j, Targs.md.lg.off_z + g * D + lane % D, mpf);
        __syncthreads();   // (A) the wave partials are in LDS
        GA_TICK(5);
        ga_tree_tail<D>(Targs, io, g, j, d, par, mpf, s_acc, s_red, s_info, s_keep);
        GA_TICK(6);
        GA_DBG(5, true);
        if (dbg && L == dbg_leaf && lane == 0)
          dbg[(int64_t)g * 8 + 6] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
      } else {
        __syncthreads();   // (A)
      }
      __syncthreads();     // (C) wave 0 has taken the ticket
      if (s_info[0]) ga_tree_block_reduce<D>(Targs, g, par, s_cp, s_info);
      GA_TICK(7);
      GA_DBG(7, w == 0);
      if (alternate) rev ^= 1;
      src_prev = true;
    }
    // ---- doubling boundary: the finished subtree's far end is the new edge state on its side (nuts.py:347-362) ----
    if (dir > 0) right += nleaf; else left -= nleaf;
    if (d + 1 >= max_depth) break;
    if (d + 1 > spec_depth) {
      // beyond the depth of the previous draw's tree this tree most likely ends here: wait for the verdict on its last leaf
      if (w == 0) {
        const int r = ga_wait(A.ga_sync, 0u, (unsigned)L, &A.ctl->aborted, a.timeout);
        if (r == 2) ga_fail(A, 1u);
        if (lane == 0) s_flag[1] = r;
      }
      __syncthreads();
      if (s_flag[1]) return;
    }
    const int ndir = ga_tree_dir(A.uniforms, d + 1);
    src_prev = ndir == dir;   // continuing on the same side: the new edge state is the leaf just finished
    dir = ndir;
    edge = dir > 0 ? right : left;
  }
#undef GA_TICK
#undef GA_DBG
}

// Latency regime: whole NUTS draws in ONE launch of ONE workgroup.
//
// For tiny models (SURVEY.md section 7 "small-n latency": eight schools has n = 10 / 26) a leapfrog is a few
// hundred flops; three launches per leapfrog plus a host round trip per doubling cost two orders of magnitude more
// than the arithmetic.  When the model fits one workgroup (n <= 1024, element-wise factors only, diagonal mass
// matrix) the whole transition -- momentum refresh, start state, every doubling of the tree, proposal gather -- runs
// inside this kernel: one thread per parameter, `__syncthreads()` instead of kernel boundaries, the control block in
// LDS for the whole draw.  The arithmetic is the same device code the three-kernel pipeline uses (gather_element,
// leaf_post, tree_decide), in the same order, so the results are identical; only the summation of the partial dot
// products changes (one workgroup instead of per-workgroup partials), which is why the two paths are not bitwise
// interchangeable within one chain.
//
// Multi-draw loop.  After tuning nothing on the host changes between draws (fixed step size, frozen
// mass matrix; `step_adapt.update` and `potential.update` return immediately when `tune` is false,
// step_sizes.py:66-68, quadpotential.py:335-337), so `n_draws` consecutive transitions run in one launch: draw k+1
// starts from draw k's proposal (position, gradient and logp are already in the arena), takes row k+1 of the
// pre-drawn momentum normals and continues in the SAME pre-drawn uniform stream where draw k stopped -- exactly the
// values `step.rng.random()` would have handed out.  Positions go to a device trace buffer, statistics to a DrawOut
// per draw; the host gathers both once.  The batch stops early after a divergent draw (the host wants the two
// phase-space points of a divergence, base_hmc.py:249-258, which live in the arena until the next draw overwrites
// them) and when the remaining uniforms could not cover a worst-case tree.
//
// log(u): the tree compares `log(rng.random())` (nuts.py:371,466).  The three-kernel pipeline gets those
// logarithms from the host; here they would cost more host time than the whole tree (a worst-case buffer per draw,
// of which a typical tree reads ten), so the control thread takes `log` of the uniforms it actually consumes.
#pragma once
#include "kernels.h"

struct SmallDrawArgs {
  const double* normals;   // [n_draws][n] standard normals of potential.random()
  const double* q_src;     // start-state cache (or nullptr: evaluate the model at A.Q slot 0)
  const double* g_src;
  double cached_logp;
  double step_size, Emax;
  int max_depth, n_draws;
  int n_uniforms, worst_uniforms;   // uniforms available in A.uniforms; what one worst-case tree can consume
  double* q_out;           // (q, grad) of the LAST draw's proposal: the next launch's start-state cache
  double* g_out;
  double* trace_q;         // [n_draws][n] proposals (nullptr when n_draws == 1: q_out is the only output)
  DrawOut* out;            // [n_draws]
  int* n_done;             // draws actually made (device int)
  HostStatus* st;
  int seq, lds_slots;      // 0: the tree in LDS when it fits; -1: in the global arena whatever the size (A/B); k > 0: at most k LDS slots (tests of the hand-over)
};

// NT = threads of the one workgroup (256, 512 or 1024: one thread per parameter, so n <= 1024 runs here; at 1024 threads the
// register budget is 128 and the kernel spills a little -- still 20.6 us per leapfrog against 25.3 us at n = 602 and 23.8
// against 25.2 at n = 1002, profiles/r02i_latency_regime.json)
//
// Round 5 -- the tree in LDS.  Every leaf of the launch above made six to eight dependent round trips to the arena, the data pool
// and the uniform stream in global memory (~0.7 us each out of L2): 10 us per leapfrog at n = 3, whatever the arithmetic.  For
// n <= SMALL_LDS_N the 256-thread kernel keeps them in LDS instead: the arena as S' = the largest power of two of slots that fit
// SMALL_ARENA_DBL doubles (512 slots at n = 3, 256 at n = 10, 64 at n = 26, 32 at n = 64), the pending-sibling sums, the
// potential's diagonal, the model's data pool (up to SMALL_POOL_DBL doubles) and the draw's window of uniforms.  The device
// functions of the tree (leaf_post, tree_decide, gather_element) take the arena as pointers + slot count and do not care where it
// lives.  A tree that outgrows the LDS slots (a doubling that would hold more than S' leaves) is copied out to the global arena,
// slot by slot, and goes on there; so does the tree of a draw that diverged, because the host fetches the two points of a
// divergence from the global arena (engine.hip finish_draw_host).  Same arithmetic in the same order either way: identical results.
#define SMALL_LDS_N 64
#define SMALL_ARENA_DBL 11264
#define SMALL_POOL_DBL 1024
#define SMALL_UNI_DBL 1100

// Round 6 -- the GLM node inside the launch (VERDICT r05 "missing" 6: the everyday regression).  A model whose only dense node is a GLM
// with few covariates and rows (P <= GLM_SMALL_P, N <= GLM_SMALL_N: ten covariates, a few thousand observations) took the general
// path's FOUR launches per leapfrog -- the row pass, its reduce, kernels B and C: 31.5 us per leapfrog whatever N, all of it launch
// latency (profiles/r06w_profile_glm_small.txt).  Here the workgroup evaluates the node itself between two barriers: thread t takes
// rows t, t + NT, ... two at a time from a TRANSPOSED copy of X (GlmDev.Xt: a column of 64 rows is one coalesced load; a row per
// thread from the row-major X was 64 cache lines per load instruction: 112 us per leapfrog at 10 000 x 10) (eta = intercept + x . beta' with beta' in LDS, the family's log-likelihood and r = d lp / d eta
// by glm_kernel.h's `glm_row`, acc_p += r x_p in registers), the waves' totals meet in LDS, and the parameters' threads pick their
// share up before the chain rule of their transforms -- what k_glm_reduce leaves in GlmDev.gdense on the general path.
// (pm.math.dot(X, beta) under a Normal / Bernoulli-logit / Poisson-log likelihood: pymc/math.py:56, model/core.py:213-267.)
#define GLM_SMALL_P 16
#define GLM_SMALL_N 4096

template <int NT, bool PROG>
__global__ __launch_bounds__(NT) void k_small_draw(ModelDev md, ArenaDev Ag, SmallDrawArgs a) {
  constexpr int NW = NT / WAVE;
  constexpr bool LDSV = NT <= 256;     // the variants that may keep the tree in LDS (NT = 64: one wave, n <= 64 -- every barrier and block sum of a leaf is then a wave's own business)
  __shared__ __attribute__((aligned(16))) char s_prog[PROG_LDS_MAX];
  __shared__ double s_bacc[MAX_BTERMS][NT];
  __shared__ double s_red[NDOT * NW];
  __shared__ double s_dot[NDOT];
  __shared__ double s_w[NW];
  __shared__ Ctl s_ctl;
  __shared__ int s_stop;
  __shared__ double s_arena[LDSV ? SMALL_ARENA_DBL : 1];
  __shared__ double s_ps[LDSV ? (MAX_LEVELS + 1) * SMALL_LDS_N : 1];
  __shared__ double s_diag[LDSV ? 2 * SMALL_LDS_N : 1];
  __shared__ double s_pool[LDSV ? SMALL_POOL_DBL : 1];
  __shared__ double s_uni[LDSV ? SMALL_UNI_DBL : 1];
  __shared__ double s_gbeta[NT == 256 ? GLM_SMALL_P : 1];                        // the GLM node: beta' at the position being evaluated,
  __shared__ double s_gpart[NT == 256 ? NW : 1][NT == 256 ? GLM_SMALL_P + 3 : 1];   // the waves' totals [d/dbeta (P), d/dintercept, d/dsigma, logp],
  __shared__ double s_gtot[NT == 256 ? GLM_SMALL_P + 3 : 1];                       // and the workgroup's
  const int tid = threadIdx.x;
  const int n = md.n;
  const bool mine = tid < n;
#ifdef NUTS_KTIMING   // (lab build, tools/small_ticks.py: where a draw's time goes inside the kernel -- thread 0's clock, summed over launches)
  long long tk_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tk_last = tick_now();
  const long long tk_begin = tk_last;
#define SMALL_TICK(i) do { if (tid == 0) { const long long now_ = tick_now(); tk_acc[i] += now_ - tk_last; tk_last = now_; } } while (0)
#else
#define SMALL_TICK(i) do { } while (0)
#endif
  Prog pg;
  if constexpr (NT >= 256) {
    ProgRegs pregs;
    prog_issue(md, pregs);
    pg = load_prog(md, s_prog, pregs);
  } else {   // (prog_issue / load_prog copy with 256 threads)
    const char* base = md.prog;
    if (md.prog_bytes <= PROG_LDS_MAX) {
      const int n16 = (md.prog_bytes + 15) >> 4;
      for (int i = tid; i < n16; i += NT) reinterpret_cast<uint4*>(s_prog)[i] = reinterpret_cast<const uint4*>(md.prog)[i];
      __syncthreads();
      base = s_prog;
    }
    pg = prog_view(md, base);
  }
  // ---- where the tree lives ----
  int S_lds = 0;
  if (LDSV && n <= SMALL_LDS_N && a.lds_slots >= 0) {
    S_lds = a.lds_slots > 0 ? min(Ag.S, a.lds_slots) : Ag.S;
    while (S_lds > 1 && (int64_t)S_lds * (4 * n + 2) > SMALL_ARENA_DBL) S_lds >>= 1;
    if (S_lds < 4) S_lds = 0;
  }
  ArenaDev A = Ag;
  ArenaDev Al = Ag;       // the arena in LDS (S_lds > 0)
  if (LDSV && S_lds > 0) {
    double* base = s_arena;
    Al.S = S_lds;
    Al.Q = base; Al.P = base + (int64_t)S_lds * n; Al.V = base + 2 * (int64_t)S_lds * n; Al.G = base + 3 * (int64_t)S_lds * n;
    Al.E = base + 4 * (int64_t)S_lds * n; Al.LOGP = Al.E + S_lds;
    Al.PS = s_ps; Al.PSUM = s_ps + (int64_t)MAX_LEVELS * n;
    if (mine) { s_diag[tid] = Ag.var[tid]; s_diag[SMALL_LDS_N + tid] = Ag.inv_stds[tid]; }
    Al.var = s_diag; Al.inv_stds = s_diag + SMALL_LDS_N;
    if (md.pool_len <= SMALL_POOL_DBL) {
      for (int i = tid; i < md.pool_len; i += NT) s_pool[i] = pg.pool[i];
      pg.pool = s_pool;
    }
  }
  // the live leaves [left, right] of the tree in LDS -> their slots of the global arena (every thread calls it)
  auto spill_tree = [&](int left, int right) {
    const int cnt = (right - left + 1) * n;
    for (int i = tid; i < cnt; i += NT) {
      const int t = left + i / n, e = i - (i / n) * n;
      const int64_t lo = (int64_t)(t & (S_lds - 1)) * n + e, go = (int64_t)(t & (Ag.S - 1)) * n + e;
      Ag.Q[go] = Al.Q[lo]; Ag.P[go] = Al.P[lo]; Ag.V[go] = Al.V[lo]; Ag.G[go] = Al.G[lo];
    }
    for (int t = left + tid; t <= right; t += NT) {
      Ag.E[t & (Ag.S - 1)] = Al.E[t & (S_lds - 1)];
      Ag.LOGP[t & (Ag.S - 1)] = Al.LOGP[t & (S_lds - 1)];
    }
    __threadfence_block();
    __syncthreads();
  };
  int k = 0;
  VarDev v{};
  if (mine) { k = find_var(pg, tid); v = pg.vars[k]; }
  for (int b = 0; b < md.n_bterms; ++b) s_bacc[b][tid] = 0.0;
  if (tid == 0) { s_ctl.cursor = 0; s_stop = 0; }
  __syncthreads();

  // logp and d logp / dq_i at the position seen through `qv` (this thread's coordinate is `qn`)
  auto eval_model = [&](const QView& qv, double qn, double& grad_i, double& logp) {
    double lp = 0.0, gx = 0.0, dxdq = 1.0, dj = 0.0;
    if constexpr (PROG) {
      if (md.n_gsf > 0) {   // gathered adjoints (model_dev.h GSlot): every element of those factors swept once, then the gathers below
        for (int e = tid; e < md.n_gs_elems; e += NT) gsweep_element(pg, qv, e);
        __threadfence_block();
        __syncthreads();
        for (int t = 0; t < md.n_glong; ++t) {   // long inverse-index lists: the whole workgroup totals each
          const double tot = gadj_long_total(md, md.glong[t], s_w);
          if (tid == 0) md.adj_red[t] = tot;
        }
        if (md.n_glong > 0) { __threadfence_block(); __syncthreads(); }
      }
    }
    if constexpr (NT == 256) if (md.has_glm) {   // the GLM node, evaluated by the whole workgroup (header comment; the 256-thread variant only: its
                                                  // register budget holds the rows in flight -- a model with a small GLM node and n <= 256 runs on it)
      const GlmDev& gm = md.glm;
      const int P = gm.P, FAMILY = gm.family;
      for (int p = tid; p < P; p += NT) s_gbeta[p] = qv.at(gm.off_beta + p);
      double icpt, sigma;
      glm_scalars(gm, qv, icpt, sigma);
      const double inv_sigma = 1.0 / sigma, log_sigma = FAMILY == NUTS_GLM_NORMAL ? log(sigma) : 0.0;
      __syncthreads();
      double acc[GLM_SMALL_P];
#pragma unroll
      for (int p = 0; p < GLM_SMALL_P; ++p) acc[p] = 0.0;
      double lp_acc = 0.0, r_acc = 0.0, ds_acc = 0.0;
      const int N = (int)gm.N;
      const double* __restrict__ Xt = gm.Xt;     // [P][N]: column p of 64 consecutive rows is one coalesced load
      for (int i0 = tid; i0 < N; i0 += 2 * NT) {
        const int i1 = i0 + NT;
        const bool two = i1 < N;
        const int j1 = two ? i1 : i0;
        double x0[GLM_SMALL_P], x1[GLM_SMALL_P];
#pragma unroll
        for (int p = 0; p < GLM_SMALL_P; ++p)      // (both rows' columns requested before the first is used)
          if (p < P) { x0[p] = Xt[(int64_t)p * N + i0]; x1[p] = Xt[(int64_t)p * N + j1]; }
        const double y0 = gm.y[i0], y1 = gm.y[j1];
        double e0 = 0.0, e1 = 0.0;
#pragma unroll
        for (int p = 0; p < GLM_SMALL_P; ++p)
          if (p < P) { const double b = s_gbeta[p]; e0 = fma(x0[p], b, e0); e1 = fma(x1[p], b, e1); }
        double lp0, r0, d0, lp1, r1, d1;
        glm_row(FAMILY, e0 + icpt, y0, sigma, inv_sigma, log_sigma, lp0, r0, d0);
        glm_row(FAMILY, e1 + icpt, y1, sigma, inv_sigma, log_sigma, lp1, r1, d1);
        if (!two) { lp1 = 0.0; r1 = 0.0; d1 = 0.0; }
#pragma unroll
        for (int p = 0; p < GLM_SMALL_P; ++p)
          if (p < P) { acc[p] = fma(r0, x0[p], acc[p]); acc[p] = fma(r1, x1[p], acc[p]); }
        lp_acc += lp0; lp_acc += lp1;
        r_acc += r0; r_acc += r1;
        ds_acc += d0; ds_acc += d1;
      }
      const int w = tid >> 6, lane = tid & (WAVE - 1);
#pragma unroll
      for (int p = 0; p < GLM_SMALL_P; ++p)
        if (p < P) { const double t = wave_sum(acc[p]); if (lane == 0) s_gpart[w][p] = t; }
      {
        const double t0 = wave_sum(r_acc), t1 = wave_sum(ds_acc), t2 = wave_sum(lp_acc);
        if (lane == 0) { s_gpart[w][P] = t0; s_gpart[w][P + 1] = t1; s_gpart[w][P + 2] = t2; }
      }
      __syncthreads();
      if (tid < P + 3) {
        double t = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) t += s_gpart[ww][tid];
        s_gtot[tid] = t;
      }
      __syncthreads();
    }
    if (mine) {
      double x, lj;
      if (v.normal_prior) {
        x = qn;
        const double r = x - v.np_mu;
        gx = -r * v.np_inv_var;
        lp = -0.5 * r * r * v.np_inv_var - v.np_lognorm;
      } else {
        transform_full(v, qn, x, dxdq, lj, dj);
        lp = lj;
        gather_element<PROG, false, true>(pg, qv, k, tid - v.offset, x, gx, lp, &s_bacc[0][tid], NT);
      }
      if constexpr (NT == 256) if (md.has_glm) {   // the node's gradient w.r.t. the constrained value of this thread's element (GlmDev.gdense on the general path)
        const GlmDev& gm = md.glm;
        if (tid >= gm.off_beta && tid < gm.off_beta + gm.P) gx += s_gtot[tid - gm.off_beta];
        else if (tid == gm.off_icpt) gx += s_gtot[gm.P];
        else if (tid == gm.off_sigma) gx += s_gtot[gm.P + 1];
        if (tid == 0) lp += s_gtot[gm.P + 2] + gm.konst;   // (n >= 1: thread 0 always has an element)
      }
    }
    SMALL_TICK(8);     // (thread 0's own element)
    for (int o = 0; o < md.n_orphans; ++o) {   // factors without an owning variable
      const int fi = md.orphans[o];
      const int fsize = pg.factors[fi].size;
      for (int li = tid; li < fsize; li += NT) lp += orphan_element<PROG, false, true>(pg, qv, fi, li, &s_bacc[0][tid], NT);
    }
    SMALL_TICK(9);     // (thread 0's share of the orphan factors)
    for (int b = 0; b < md.n_bterms; ++b) {   // scalars that broadcast against vector factors
      const double t = block_sum<true>(s_bacc[b][tid], s_w);
      if (mine && v.size == 1 && pg.bterm_var[b] == k) gx += t;
      s_bacc[b][tid] = 0.0;
    }
    SMALL_TICK(10);    // (the broadcast terms' sums: includes waiting for the slowest thread's element)
    grad_i = gx * dxdq + dj;
    logp = block_sum<true>(lp, s_w);
    SMALL_TICK(11);
  };

  Ag.log_uniforms = nullptr;   // log(u) is taken by the control thread (header comment)
  Al.log_uniforms = nullptr;
  bool in_lds = false;
  int done = 0;
  double q_prop = 0.0, g_prop = 0.0, logp_prop = 0.0;   // previous draw's proposal (this thread's coordinate)
  for (int it = 0; it < a.n_draws; ++it) {
    // every draw starts in LDS when the model allows; its window of the uniform stream comes along (the cursor runs on from draw to draw)
    in_lds = LDSV && S_lds > 0;
    A = in_lds ? Al : Ag;
    if (in_lds) {
      const int base = s_ctl.cursor;
      const int cnt = min(min(a.worst_uniforms + 2, SMALL_UNI_DBL), a.n_uniforms - base);
      for (int i = tid; i < cnt; i += NT) s_uni[i] = Ag.uniforms[base + i];
      if (a.worst_uniforms + 2 <= SMALL_UNI_DBL) A.uniforms = s_uni - base;   // (index = the stream's own cursor)
      if (it == 0 && !a.q_src && mine) A.Q[tid] = Ag.Q[tid];                   // (the host put q0 into slot 0 of the global arena)
      __syncthreads();
    }
    SMALL_TICK(0);    // setup (first draw) / the previous draw's hand-over
    // ---- start state (base_hmc.py:201-202): q0, its gradient and logp; p0 = z / sigma; E0 ----
    double logp0 = a.cached_logp;
    if (it > 0) {
      if (mine) { A.Q[tid] = q_prop; A.G[tid] = g_prop; }
      logp0 = logp_prop;
    } else if (a.q_src) {
      if (mine) { A.Q[tid] = a.q_src[tid]; A.G[tid] = a.g_src[tid]; }
    } else {
      QView qv;
      qv.q = A.Q; qv.p = qv.g = qv.var = nullptr; qv.eps = qv.half = 0.0; qv.composed = 0;
      double g0;
      eval_model(qv, mine ? A.Q[tid] : 0.0, g0, logp0);
      if (mine) A.G[tid] = g0;
    }
    double kin = 0.0;
    if (mine) {
      const double p = a.normals[(int64_t)it * n + tid] * A.inv_stds[tid];
      const double vv = A.var[tid] * p;
      A.P[tid] = p; A.V[tid] = vv; A.PSUM[tid] = p;
      kin = p * vv;
    }
    const double kin0 = block_sum<true>(kin, s_w);
    if (tid == 0) {
      Ctl* c = &s_ctl;
      const double E = 0.5 * kin0 - logp0;  // integration.py:72-74
      A.LOGP[0] = logp0; A.E[0] = E;
      c->E0 = E; c->log_size = 0.0; c->log_accept_sum = -INFINITY; c->max_energy_change = 0.0; c->div_dE = 0.0;
      c->n_proposals = 0; c->depth = 0; c->left = 0; c->right = 0; c->proposal = 0;   // (the uniform cursor runs on)
      c->turning = 0; c->diverging = 0; c->div_t = 0;
      c->bad_energy = !isfinite(E);
      c->aborted = c->bad_energy;
      c->eps_abs = a.step_size; c->n_leaves_total = 0;
      c->dir = 1; c->edge = 0; c->eps = a.step_size;
      if (!c->aborted && a.max_depth > 0) ctl_next_direction(c, A.uniforms);
    }
    __syncthreads();

    SMALL_TICK(1);    // start state
    // ---- the tree (nuts.py:204-225) ----
    for (int d = 0; d < a.max_depth && !s_ctl.aborted; ++d) {
      if (in_lds && (2 << d) > S_lds) {   // this doubling would hold more leaves than the LDS arena has slots: on in global memory
        spill_tree(s_ctl.left, s_ctl.right);
        const double* uni = A.uniforms;
        A = Ag; A.uniforms = uni; A.PS = Al.PS; A.PSUM = Al.PSUM; A.var = Al.var; A.inv_stds = Al.inv_stds;
        in_lds = false;
      }
      Leaf lf;
      lf.dir = s_ctl.dir; lf.edge = s_ctl.edge; lf.left = s_ctl.left; lf.right = s_ctl.right;
      lf.eps = s_ctl.eps; lf.half = 0.5 * s_ctl.eps;
      const int nleaf = 1 << d;
      for (int j = 0; j < nleaf; ++j) {
        lf.src = lf.edge + lf.dir * j;
        lf.t = lf.src + lf.dir;
        lf.so = slot_off(A, lf.src); lf.d_o = slot_off(A, lf.t);
        QView qv;
        qv.q = A.Q + lf.so; qv.p = A.P + lf.so; qv.g = A.G + lf.so; qv.var = A.var;
        qv.eps = lf.eps; qv.half = lf.half; qv.composed = 1;
        // first half of the leapfrog (integration.py:118-127), gradient at q'
        int idx[1] = {tid};
        bool act[1] = {mine};
        double grad[1] = {0.0}, ph[1] = {0.0};
        double qn = 0.0, logp;
        if (mine) {
          ph[0] = qv.p_half(tid);
          qn = qv.at(tid);
          A.Q[lf.d_o + tid] = qn;
        }
        SMALL_TICK(2);    // leaf set-up, first half of the leapfrog
        eval_model(qv, qn, grad[0], logp);
        SMALL_TICK(3);    // the model
        if (mine) A.G[lf.d_o + tid] = grad[0];
        // second half kick, v', kinetic energy and the U-turn dots of the merges this leaf completes
        int m; bool last;
        leaf_post<1>(A, lf, j, d, true, idx, act, grad, ph, s_red, NW, m, last);
        __syncthreads();
        SMALL_TICK(4);    // second kick, merge dot products
        for (int q = tid; q < NDOT; q += NT) {
          if (!dot_needed(q, m, last)) continue;
          double r = 0.0;
          for (int w = 0; w < NW; ++w) r += s_red[q * NW + w];
          s_dot[q] = r;
        }
        __syncthreads();
        if (tid == 0) {
          const int ts = lf.t & (A.S - 1);
          const double E = 0.5 * s_dot[0] - logp;  // integration.py:133-134
          A.LOGP[ts] = logp; A.E[ts] = E;
          tree_decide(&s_ctl, A, lf, s_dot, E, m, last, a.Emax, a.max_depth, uni_view_none());
        }
        __syncthreads();   // also makes this leaf's arena stores visible to the whole workgroup
        SMALL_TICK(5);    // dots combined, the tree's decision
#ifdef NUTS_KTIMING
        if (tid == 0) tk_acc[7] += 1;
#endif
        if (s_ctl.aborted) break;
      }
      if (s_ctl.depth >= a.max_depth) break;
    }

    // ---- proposal and statistics (nuts.py:478-489) ----
    const Ctl* c = &s_ctl;
    const int prop = c->proposal;
    const int64_t po = slot_off(A, prop);
    if (mine) {
      q_prop = A.Q[po + tid]; g_prop = A.G[po + tid];
      if (a.trace_q) a.trace_q[(int64_t)it * n + tid] = q_prop;
    }
    logp_prop = A.LOGP[prop & (A.S - 1)];
    if (tid == 0) {
      const int ps = prop & (A.S - 1);
      DrawOut* o = a.out + it;
      o->energy = A.E[ps]; o->logp = A.LOGP[ps]; o->E0 = c->E0;
      o->log_accept_sum = c->log_accept_sum; o->max_energy_change = c->max_energy_change; o->div_dE = c->div_dE;
      o->depth = c->depth; o->n_proposals = c->n_proposals; o->proposal = prop; o->cursor = c->cursor;
      o->turning = c->turning; o->diverging = c->diverging; o->bad_energy = c->bad_energy; o->div_t = c->div_t;
      // stop the batch: bad start energy (the host raises), a divergence (the host fetches the two points from the
      // arena), or not enough uniforms left for a worst-case tree
      s_stop = c->bad_energy || c->diverging || (c->cursor + a.worst_uniforms > a.n_uniforms);
    }
    done = it + 1;
    __syncthreads();   // every thread has read the proposal before slot 0 is overwritten; s_stop is visible
    // (the host reads the two points of the divergence; the leaf that diverged belongs to a subtree that was never merged, so it
    // lies outside [left, right])
    if (in_lds && s_ctl.diverging) spill_tree(min(s_ctl.left, s_ctl.div_t), max(s_ctl.right, s_ctl.div_t));
    if (s_stop) break;
  }
  SMALL_TICK(6);      // proposal, statistics
#ifdef NUTS_KTIMING
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) md.ticks[40 + i] += tk_acc[i];
    md.ticks[48] += tick_now() - tk_begin;
    md.ticks[49] += 1;
    for (int i = 8; i < 12; ++i) md.ticks[42 + i] += tk_acc[i];   // slots 50 .. 53: inside the model evaluation
  }
#endif
  if (mine) { a.q_out[tid] = q_prop; a.g_out[tid] = g_prop; }
  if (tid == 0) {
    if (a.n_done) *a.n_done = done;
    *Ag.ctl = s_ctl;
    if (a.st) publish_status(&s_ctl, a.st, a.seq);
  }
}

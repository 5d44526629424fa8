// Sampler kernels: leapfrog integrator, NUTS tree bookkeeping, adaptation.
//
// Device-resident restatement of
//   CpuLeapfrogIntegrator   pymc/step_methods/hmc/integration.py:68-145
//   _Tree                   pymc/step_methods/hmc/nuts.py:270-489
//   QuadPotentialDiagAdapt  pymc/step_methods/hmc/quadpotential.py:211-448
//
// Trajectory arena.  Every phase-space point of the current trajectory lives in
// HBM at slot (index_in_trajectory & (S-1)), S = 2^max_treedepth: Q,P,V,G are
// [S][n].  A trajectory occupies at most S consecutive indices containing 0, so
// the slot is unique, and tree nodes are plain integers: a proposal is the
// trajectory index of a leaf (no vector is ever copied when a proposal is
// selected), a subtree is (first leaf, last leaf, p_sum).  With 288 GB of HBM
// this costs 4*S*n*8 bytes (328 MB at n = 10 000).
//
// Recursion -> binary counter.  `_build_subtree` (nuts.py:442-476) is a
// post-order traversal; leaf j of a 2^d-leaf subtree completes `m` = (number of
// trailing one bits of j) merges, level l merging the stored left sibling of
// 2^l leaves with the just-finished right sibling.  The vector work of those
// merges (p_sum additions and the U-turn dot products) is static given j, so
// k_leaf_post computes all of it speculatively in one pass and k_leaf_ctl
// replays the scalar decisions in the reference's order, consuming the
// pre-drawn uniforms exactly as `rng.random()` is consumed.
#pragma once
#include "device_math.h"

#define MAX_LEVELS 12            // supports max_treedepth <= 11
#define NDOT (1 + 6 * (MAX_LEVELS + 1))
#define DOT_TOP (1 + 6 * MAX_LEVELS)
#define VEC_THREADS 256

struct Ctl {
  // trajectory-level (nuts.py:318-332)
  double E0;
  double log_size, log_accept_sum, max_energy_change;
  double div_dE;
  int n_proposals, depth, left, right, proposal, cursor;
  int aborted, turning, diverging, bad_energy;
  // current doubling
  int dir, edge;
  double eps;      // signed step (nuts.py:348,357)
  double eps_abs;
  // pending left siblings of the subtree under construction
  double st_ls[MAX_LEVELS];
  int st_prop[MAX_LEVELS];
  // outputs
  double prop_energy, prop_logp;
  int n_leaves_total;
  int pad;
};

struct ArenaDev {
  int n, S, nblk, ept;  // dimension, slots, vector-kernel workgroups, elements per thread
  double *Q, *P, *V, *G;   // [S][n]
  double *E, *LOGP;        // [S]
  double *PS;              // [MAX_LEVELS][n] pending-sibling p_sum (level 0 unused: read from P)
  double *PSUM;            // [n] whole-tree p_sum (nuts.py:330)
  double *dotp;            // [nblk][NDOT]
  const double *var, *inv_stds;  // diagonal potential (quadpotential.py:308-326)
  Ctl* ctl;
  const double* uniforms;
};

struct HostStatus {
  int aborted, turning, diverging, bad_energy, depth, cursor, n_proposals, proposal, dir, edge;
};

__device__ __forceinline__ void publish_status(const Ctl* c, HostStatus* st) {
  st->aborted = c->aborted; st->turning = c->turning; st->diverging = c->diverging; st->bad_energy = c->bad_energy;
  st->depth = c->depth; st->cursor = c->cursor; st->n_proposals = c->n_proposals; st->proposal = c->proposal;
  st->dir = c->dir; st->edge = c->edge;
}

__device__ __forceinline__ int64_t slot_off(const ArenaDev& A, int t) { return (int64_t)(t & (A.S - 1)) * A.n; }

// ---- start of a draw: p0 = z / sigma, v0 = var * p0, PSUM = p0 (base_hmc.py:201-202) ----
__global__ __launch_bounds__(VEC_THREADS) void k_draw_start(ArenaDev A, const double* __restrict__ q0,
                                                            const double* __restrict__ normals, int copy_q) {
  __shared__ double sm[VEC_THREADS / WAVE];
  double kin = 0.0;
  const int base = blockIdx.x * VEC_THREADS * A.ept;
  for (int e = 0; e < A.ept; ++e) {
    const int i = base + e * VEC_THREADS + threadIdx.x;
    if (i < A.n) {
      const double p = normals[i] * A.inv_stds[i];
      const double v = A.var[i] * p;
      if (copy_q) A.Q[i] = q0[i];
      A.P[i] = p; A.V[i] = v; A.PSUM[i] = p;
      kin = fma(p, v, kin);
    }
  }
  const double tot = block_sum<false>(kin, sm);
  if (threadIdx.x == 0) A.dotp[(int64_t)blockIdx.x * NDOT] = tot;
}

// direction of the next doubling: `(rng.random() < 0.5) * 2 - 1` (nuts.py:215)
__device__ __forceinline__ void ctl_next_direction(Ctl* c, const double* uniforms) {
  const double u = uniforms[c->cursor++];
  c->dir = (u < 0.5) ? 1 : -1;
  c->eps = c->dir > 0 ? c->eps_abs : -c->eps_abs;
  c->edge = c->dir > 0 ? c->right : c->left;
}

__global__ void k_draw_ctl_start(ArenaDev A, double step_size, int max_depth, HostStatus* st) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < A.nblk; ++b) s += A.dotp[(int64_t)b * NDOT];
    Ctl* c = A.ctl;
    const double logp = A.LOGP[0];
    const double E = 0.5 * s - logp;  // integration.py:72-74
    A.E[0] = E;
    c->E0 = E;
    c->log_size = 0.0;
    c->log_accept_sum = -INFINITY;
    c->max_energy_change = 0.0;
    c->div_dE = 0.0;
    c->n_proposals = 0; c->depth = 0; c->left = 0; c->right = 0; c->proposal = 0; c->cursor = 0;
    c->turning = 0; c->diverging = 0;
    c->bad_energy = !isfinite(E);
    c->aborted = c->bad_energy;
    c->eps_abs = step_size;
    c->n_leaves_total = 0;
    c->dir = 1; c->edge = 0; c->eps = step_size;
    if (!c->aborted && max_depth > 0) ctl_next_direction(c, A.uniforms);
    if (st) publish_status(c, st);
  }
}

// ---- first half of a leapfrog (integration.py:118-127) ----
//   p_half = p + eps/2 * g ; v = var * p_half ; q' = q + eps * v
__global__ __launch_bounds__(VEC_THREADS) void k_leaf_pre(ArenaDev A, int j) {
  const Ctl* c = A.ctl;
  if (c->aborted) return;
  const int src = c->edge + c->dir * j, dst = src + c->dir;
  const double eps = c->eps, half = 0.5 * eps;
  const int64_t so = slot_off(A, src), d_o = slot_off(A, dst);
  const int base = blockIdx.x * VEC_THREADS * A.ept;
  for (int e = 0; e < A.ept; ++e) {
    const int i = base + e * VEC_THREADS + threadIdx.x;
    if (i < A.n) {
      const double ph = fma(half, A.G[so + i], A.P[so + i]);
      const double v = A.var[i] * ph;
      A.P[d_o + i] = ph;
      A.Q[d_o + i] = fma(eps, v, A.Q[so + i]);
    }
  }
}

// ---- second half of the leapfrog + all vector work of the tree merges this leaf completes ----
template <int EPT>
__global__ __launch_bounds__(VEC_THREADS) void k_leaf_post(ArenaDev A, int j, int d) {
  const Ctl* c = A.ctl;
  if (c->aborted) return;
  __shared__ double sm[6][VEC_THREADS / WAVE];
  const int dir = c->dir, edge = c->edge;
  const int t = edge + dir * (j + 1);
  const double half = 0.5 * c->eps;
  const int64_t to = slot_off(A, t);
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
  double* dots = A.dotp + (int64_t)blockIdx.x * NDOT;
  const int base = blockIdx.x * VEC_THREADS * EPT;

  // p' = p_half + eps/2 g' ; v' = var p' ; K = 1/2 p'.v'   (integration.py:131-134)
  double acc[EPT], vt[EPT];
  double kin = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * VEC_THREADS + threadIdx.x;
    acc[e] = 0.0; vt[e] = 0.0;
    if (i < A.n) {
      const double p = fma(half, A.G[to + i], A.P[to + i]);
      const double v = A.var[i] * p;
      A.P[to + i] = p; A.V[to + i] = v;
      acc[e] = p; vt[e] = v;
      kin = fma(p, v, kin);
    }
  }
  {
    double s = wave_sum(kin);
    if (lane == 0) sm[0][w] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double r = 0; for (int k = 0; k < VEC_THREADS / WAVE; ++k) r += sm[0][k]; dots[0] = r; }
    __syncthreads();
  }

  // merges: level l joins leaves [j-2^(l+1)+1, j-2^l] (t1) with [j-2^l+1, j] (t2)   (nuts.py:452-463)
  int m = 0;
  while (((j >> m) & 1) && m < d) ++m;
  for (int l = 0; l < m; ++l) {
    const int t1_left = edge + dir * (j - (2 << l) + 2);
    const int t1_right = edge + dir * (j - (1 << l) + 1);
    const int t2_left = t1_right + dir;
    const int64_t o1l = slot_off(A, t1_left), o1r = slot_off(A, t1_right), o2l = slot_off(A, t2_left);
    const double* ps1 = (l == 0) ? (A.P + o1r) : (A.PS + (int64_t)l * A.n);  // a single leaf's p_sum is its p
    double dd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = base + e * VEC_THREADS + threadIdx.x;
      if (i < A.n) {
        const double s1 = ps1[i], s2 = acc[e];
        const double rho = s1 + s2;                  // tree1.p_sum + tree2.p_sum
        const double v1l = A.V[o1l + i];
        dd[0] = fma(rho, v1l, dd[0]);
        dd[1] = fma(rho, vt[e], dd[1]);
        if (l >= 1) {
          const double rho1 = s1 + A.P[o2l + i];     // tree1.p_sum + tree2.left.p
          dd[2] = fma(rho1, v1l, dd[2]);
          dd[3] = fma(rho1, A.V[o2l + i], dd[3]);
          const double rho2 = A.P[o1r + i] + s2;     // tree1.right.p + tree2.p_sum
          dd[4] = fma(rho2, A.V[o1r + i], dd[4]);
          dd[5] = fma(rho2, vt[e], dd[5]);
        }
        acc[e] = rho;
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { double s = wave_sum(dd[k]); if (lane == 0) sm[k][w] = s; }
    __syncthreads();
    if (threadIdx.x < 6) { double r = 0; for (int k = 0; k < VEC_THREADS / WAVE; ++k) r += sm[threadIdx.x][k]; dots[1 + 6 * l + threadIdx.x] = r; }
    __syncthreads();
  }

  if (j + 1 < (1 << d)) {
    // subtree not complete: park the merged p_sum as the pending left sibling of level m
    if (m >= 1) {
      double* ps = A.PS + (int64_t)m * A.n;
#pragma unroll
      for (int e = 0; e < EPT; ++e) { const int i = base + e * VEC_THREADS + threadIdx.x; if (i < A.n) ps[i] = acc[e]; }
    }
  } else {
    // subtree complete: top-level merge of `extend` (nuts.py:346-390), speculative
    const int first = edge + dir;  // first leaf of the new subtree
    int lm_begin, lm_end, rm_begin, rm_end, new_left, new_right;
    if (dir > 0) { lm_begin = c->left; lm_end = c->right; rm_begin = first; rm_end = t; new_left = c->left; new_right = t; }
    else         { lm_begin = t; lm_end = first; rm_begin = c->left; rm_end = c->right; new_left = t; new_right = c->right; }
    const int64_t onl = slot_off(A, new_left), onr = slot_off(A, new_right);
    const int64_t olb = slot_off(A, lm_begin), ole = slot_off(A, lm_end), orb = slot_off(A, rm_begin), ore = slot_off(A, rm_end);
    double dd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = base + e * VEC_THREADS + threadIdx.x;
      if (i < A.n) {
        const double old = A.PSUM[i], sub = acc[e];
        const double tot = old + sub;                       // p_sum[:] += tree.p_sum
        A.PSUM[i] = tot;
        const double lm_sum = dir > 0 ? old : sub, rm_sum = dir > 0 ? sub : old;
        dd[0] = fma(tot, A.V[onl + i], dd[0]);
        dd[1] = fma(tot, A.V[onr + i], dd[1]);
        const double r1 = lm_sum + A.P[orb + i];             // leftmost_p_sum + rightmost_begin.p
        dd[2] = fma(r1, A.V[olb + i], dd[2]);
        dd[3] = fma(r1, A.V[orb + i], dd[3]);
        const double r2 = A.P[ole + i] + rm_sum;             // leftmost_end.p + rightmost_p_sum
        dd[4] = fma(r2, A.V[ole + i], dd[4]);
        dd[5] = fma(r2, A.V[ore + i], dd[5]);
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) { double s = wave_sum(dd[k]); if (lane == 0) sm[k][w] = s; }
    __syncthreads();
    if (threadIdx.x < 6) { double r = 0; for (int k = 0; k < VEC_THREADS / WAVE; ++k) r += sm[threadIdx.x][k]; dots[DOT_TOP + threadIdx.x] = r; }
  }
}

// ---- scalar decisions of one leaf (nuts.py:394-476 and, on the last leaf, 334-392) ----
__global__ __launch_bounds__(128) void k_leaf_ctl(ArenaDev A, int j, int d, double Emax, int max_depth,
                                                  HostStatus* st) {
  Ctl* c = A.ctl;
  if (c->aborted) {
    if (threadIdx.x == 0 && st) publish_status(c, st);
    return;
  }
  __shared__ double dot[NDOT];
  int m = 0;
  while (((j >> m) & 1) && m < d) ++m;
  const bool last = (j + 1 == (1 << d));
  // fixed-order reduction of the per-workgroup partials
  for (int k = threadIdx.x; k < NDOT; k += blockDim.x) {
    const bool need = (k == 0) || (k >= 1 && k < 1 + 6 * m) || (last && k >= DOT_TOP);
    double s = 0.0;
    if (need) for (int b = 0; b < A.nblk; ++b) s += A.dotp[(int64_t)b * NDOT + k];
    dot[k] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;

  const int dir = c->dir;
  const int t = c->edge + dir * (j + 1);
  const int ts = t & (A.S - 1);
  const double logp = A.LOGP[ts];
  const double E = 0.5 * dot[0] - logp;  // integration.py:133-134
  A.E[ts] = E;
  double dE = E - c->E0;                 // nuts.py:408-410
  if (isnan(dE)) dE = INFINITY;
  c->log_accept_sum = logaddexp_d(c->log_accept_sum, dE > 0 ? -dE : 0.0);  // nuts.py:412-414
  if (fabs(dE) > fabs(c->max_energy_change)) c->max_energy_change = dE;     // nuts.py:417-418
  c->n_proposals += 1;                                                      // nuts.py:436-437
  c->n_leaves_total += 1;
  if (!(dE < Emax)) {                                                       // nuts.py:419,433-435
    c->diverging = 1; c->aborted = 1; c->div_dE = dE;
    c->depth += 1;                                                          // extend: self.depth += 1 happens regardless
  } else {
    double cur_ls = -dE;
    int cur_prop = t;
    bool turning = false;
    for (int l = 0; l < m && !turning; ++l) {
      const double* dd = &dot[1 + 6 * l];
      turning = (dd[0] <= 0) || (dd[1] <= 0);
      if (!turning && l >= 1) {
        turning = (dd[2] <= 0) || (dd[3] <= 0);
        if (!turning) turning = (dd[4] <= 0) || (dd[5] <= 0);
      }
      const double ls = logaddexp_d(c->st_ls[l], cur_ls);                   // nuts.py:464
      const double u = A.uniforms[c->cursor++];
      if (log(u) < cur_ls - ls) { /* keep tree2's proposal */ } else cur_prop = c->st_prop[l];
      cur_ls = ls;
    }
    if (turning) {
      c->turning = 1; c->aborted = 1; c->depth += 1;
    } else if (!last) {
      c->st_ls[m] = cur_ls; c->st_prop[m] = cur_prop;
    } else {
      // extend (nuts.py:365-392)
      if (dir > 0) c->right = t; else c->left = t;
      c->depth += 1;
      const double u = A.uniforms[c->cursor++];
      if (log(u) < cur_ls - c->log_size) c->proposal = cur_prop;
      c->log_size = logaddexp_d(cur_ls, c->log_size);
      const double* dd = &dot[DOT_TOP];
      bool turn = (dd[0] <= 0) || (dd[1] <= 0);
      if (!turn) turn = (dd[2] <= 0) || (dd[3] <= 0);
      if (!turn) turn = (dd[4] <= 0) || (dd[5] <= 0);
      if (turn) { c->turning = 1; c->aborted = 1; }
      else if (c->depth < max_depth) ctl_next_direction(c, A.uniforms);
    }
  }
  if (st) publish_status(c, st);
}

// ---- end of draw: gather the proposal and tree statistics (nuts.py:478-489) ----
struct DrawOut {
  double energy, logp, E0, log_accept_sum, max_energy_change, div_dE;
  int depth, n_proposals, proposal, cursor, turning, diverging, bad_energy, pad;
};

__global__ __launch_bounds__(VEC_THREADS) void k_draw_finish(ArenaDev A, double* __restrict__ q_out,
                                                             double* __restrict__ g_out, DrawOut* out) {
  const Ctl* c = A.ctl;
  const int prop = c->proposal;
  const int64_t po = slot_off(A, prop);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) {
    q_out[i] = A.Q[po + i];
    g_out[i] = A.G[po + i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int ps = prop & (A.S - 1);
    out->energy = A.E[ps]; out->logp = A.LOGP[ps]; out->E0 = c->E0;
    out->log_accept_sum = c->log_accept_sum; out->max_energy_change = c->max_energy_change; out->div_dE = c->div_dE;
    out->depth = c->depth; out->n_proposals = c->n_proposals; out->proposal = prop; out->cursor = c->cursor;
    out->turning = c->turning; out->diverging = c->diverging; out->bad_energy = c->bad_energy;
  }
}

// ---- mass-matrix adaptation (quadpotential.py:328-355, 431-437) ----
//   flags bit0: add sample to fg and bg;  bit1: var = clip(fg.m2 / fg.count)
__global__ __launch_bounds__(VEC_THREADS) void k_potential_update(int n, const double* __restrict__ x,
                                                                  double* fg_mean, double* fg_m2, double fg_count_new,
                                                                  double* bg_mean, double* bg_m2, double bg_count_new,
                                                                  double* var, double* stds, double* inv_stds, int flags) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double m2 = fg_m2[i];
    if (flags & 1) {
      const double xi = x[i];
      double mean = fg_mean[i];
      double od = xi - mean;
      mean += od / fg_count_new;
      m2 += od * (xi - mean);
      fg_mean[i] = mean; fg_m2[i] = m2;
      double bm = bg_mean[i], b2 = bg_m2[i];
      od = xi - bm;
      bm += od / bg_count_new;
      b2 += od * (xi - bm);
      bg_mean[i] = bm; bg_m2[i] = b2;
    }
    if (flags & 2) {
      double v = m2 / fg_count_new;
      v = fmin(fmax(v, 1e-12), 1e12);   // np.clip(var, 1e-12, 1e12)
      if (isnan(m2 / fg_count_new)) v = m2 / fg_count_new;
      const double s = sqrt(v);
      var[i] = v; stds[i] = s; inv_stds[i] = 1.0 / s;
    }
  }
}

// ---- plain leapfrog for HamiltonianMC / property tests: only the kinetic dot ----
__global__ __launch_bounds__(VEC_THREADS) void k_leaf_post_simple(ArenaDev A, int j) {
  const Ctl* c = A.ctl;
  if (c->aborted) return;
  __shared__ double sm[VEC_THREADS / WAVE];
  const int t = c->edge + c->dir * (j + 1);
  const double half = 0.5 * c->eps;
  const int64_t to = slot_off(A, t);
  const int base = blockIdx.x * VEC_THREADS * A.ept;
  double kin = 0.0;
  for (int e = 0; e < A.ept; ++e) {
    const int i = base + e * VEC_THREADS + threadIdx.x;
    if (i < A.n) {
      const double p = fma(half, A.G[to + i], A.P[to + i]);
      const double v = A.var[i] * p;
      A.P[to + i] = p; A.V[to + i] = v;
      kin = fma(p, v, kin);
    }
  }
  const double tot = block_sum<false>(kin, sm);
  if (threadIdx.x == 0) A.dotp[(int64_t)blockIdx.x * NDOT] = tot;
}

__global__ void k_energy_simple(ArenaDev A, int j) {
  if (threadIdx.x != 0) return;
  Ctl* c = A.ctl;
  if (c->aborted) return;
  double s = 0.0;
  for (int b = 0; b < A.nblk; ++b) s += A.dotp[(int64_t)b * NDOT];
  const int ts = (c->edge + c->dir * (j + 1)) & (A.S - 1);
  A.E[ts] = 0.5 * s - A.LOGP[ts];
}

// Model log-density + reverse-mode gradient kernels.
//
// What the reference gets from PyTensor's compiled `ValueGradFunction`
// (pymc/model/core.py:142-305: one C/Numba function returning (logp, dlogp) for
// a raveled q) is computed here by four kernel stages on the library stream:
//
//   M1 k_model_elem     one workgroup: value transforms + Jacobians
//                       (pymc/logprob/transforms.py:880-891,967-1088) and every
//                       element-wise factor of the spec, reverse-mode into gx.
//   M2 k_logit_rows     the HBM-streaming pass over the observation matrix
//                       (hierarchical Bernoulli-logit rows): fused forward
//                       (eta, log-likelihood) + backward (d/d beta_g) in ONE
//                       read of X, wave-segmented by group.
//      k_mvn_matvec     precision mat-vec for the MvNormal node.
//   M3 k_logit_groups   per-group combine of the wave segments -> d/dz, and
//                       per-workgroup partials for d/dmu, d/dsigma.
//   M4 k_model_final    chain rule through the transforms, final logp scalar.
//
// All reductions have a fixed order (no FP atomics): results are bit-reproducible.
#pragma once
#include "device_math.h"
#include "nuts_mi355.h"

struct LogitDev {
  int64_t N, Npad;       // rows, rows padded to a multiple of ROWS_PER_SPAN
  int32_t D, G;
  const double* Xt;      // [D][Npad] column-major (SoA) copy of X: coalesced 16 B/lane loads
  const int8_t* y;       // [Npad]
  const int32_t* gid;    // [Npad] (padding rows repeat the last group id; X = 0 there)
  int32_t off_mu, off_sigma, off_z, sigma_tr;
  int64_t n_spans;       // Npad / ROWS_PER_SPAN
  int32_t n_waves;       // waves in the M2 launch
  int32_t n_seg;         // total (wave, group) segments
  const int32_t* seg_base;   // [n_waves] first segment slot of each wave
  const int32_t* gseg_ptr;   // [G+1] segments of group g = [gseg_ptr[g], gseg_ptr[g+1])
  double* seg_part;      // [n_seg][D]  d logp / d beta_g partial of each segment
  double* wave_lp;       // [n_waves]   log-likelihood partial of each wave
  int32_t n_gblk;        // workgroups of M3
  double* gblk_part;     // [n_gblk][2][D]  partial sums over groups of dbeta and dbeta*z
};

struct MvnDev {
  int32_t k, off;
  const double* mu;    // [k]
  const double* prec;  // [k][k]
  double konst;        // -k/2 log(2 pi) - logdet
  double* rowq;        // [k] delta_i * (P delta)_i
};

struct ModelDev {
  int32_t n, n_vars, n_factors, n_data;
  const nuts_var* vars;
  const nuts_factor* factors;
  const nuts_data_ref* data;
  const double* pool;
  // scratch (per model; one evaluation in flight at a time)
  double* x;       // [n] constrained values
  double* dxdq;    // [n]
  double* djac;    // [n] d log|J| / dq
  double* gx;      // [n] d logp / dx from element-wise factors
  double* gdense;  // [n] d logp / dx from dense nodes
  double* lp_elem; // [1] element-wise factors + Jacobians
  int has_logit, has_mvn;
  LogitDev lg;
  MvnDev mv;
};

#define ELEM_THREADS 1024
#define ROWS_PER_LANE 4
#define ROWS_PER_SPAN (ROWS_PER_LANE * WAVE)
#define ROWS_BLOCK 256
#define LOGIT_MAXD 8

// ---------------------------------------------------------------------------
// M1: element-wise factors
// ---------------------------------------------------------------------------

__device__ __forceinline__ double op_value(const nuts_operand& o, int i, const ModelDev& md) {
  if (o.kind == NUTS_OP_CONST) return o.c;
  if (o.kind == NUTS_OP_DATA) {
    const nuts_data_ref r = md.data[o.ref];
    return md.pool[r.offset + (r.size > 1 ? i : 0)];
  }
  const nuts_var v = md.vars[o.ref];
  return md.x[v.offset + (v.size > 1 ? i : 0)];
}

__device__ __forceinline__ bool op_is_bcast_var(const nuts_operand& o, const ModelDev& md) {
  return o.kind == NUTS_OP_VAR && md.vars[o.ref].size == 1;
}

// log-density of one element and its partials w.r.t. each argument.
__device__ __forceinline__ double dist_eval(int dist, double konst, const double* a, double* d) {
  const double NINF = -INFINITY;
  const double LOG_SQRT_2PI = 0.91893853320467274178;
  const double LOG_SQRT_2_OVER_PI = -0.22579135264472743236;
  const double LOG_PI = 1.14472988584940017414;
  const double LOG_2 = 0.69314718055994530942;
  double lp = 0.0;
  bool dead = false;
#define KILL_UNLESS(cond) if (!(cond)) { lp = NINF; dead = true; }
  d[0] = d[1] = d[2] = d[3] = 0.0;
  switch (dist) {
    case NUTS_D_NORMAL: {  // continuous.py:526-532
      double sg = a[2], z = (a[0] - a[1]) / sg;
      lp = -0.5 * z * z - LOG_SQRT_2PI - log(sg);
      KILL_UNLESS(sg > 0)
      d[0] = -z / sg; d[1] = z / sg; d[2] = (z * z - 1.0) / sg;
    } break;
    case NUTS_D_HALFNORMAL: {  // continuous.py:909-916
      double sg = a[1], z = a[0] / sg;
      lp = -0.5 * z * z + LOG_SQRT_2_OVER_PI - log(sg);
      KILL_UNLESS(a[0] >= 0)
      KILL_UNLESS(sg > 0)
      d[0] = -z / sg; d[1] = (z * z - 1.0) / sg;
    } break;
    case NUTS_D_CAUCHY: {  // continuous.py:2287-2293
      double be = a[2], z = (a[0] - a[1]) / be;
      lp = -LOG_PI - log(be) - log1p(z * z);
      KILL_UNLESS(be > 0)
      double w = 2.0 * z / (1.0 + z * z);
      d[0] = -w / be; d[1] = w / be; d[2] = (-1.0 + w * z) / be;
    } break;
    case NUTS_D_HALFCAUCHY: {  // continuous.py:2383-2390
      double be = a[1], z = a[0] / be;
      lp = LOG_2 - LOG_PI - log(be) - log1p(z * z);
      KILL_UNLESS(a[0] >= 0)
      KILL_UNLESS(be > 0)
      double w = 2.0 * z / (1.0 + z * z);
      d[0] = -w / be; d[1] = (-1.0 + w * z) / be;
    } break;
    case NUTS_D_STUDENTT: {  // continuous.py:1935-1950 (nu constant)
      double nu = a[1], sg = a[3], z = (a[0] - a[2]) / sg;
      lp = konst - log(sg) - (nu + 1.0) / 2.0 * log1p(z * z / nu);
      KILL_UNLESS(sg > 0)
      double w = (nu + 1.0) * z / (nu + z * z);
      d[0] = -w / sg; d[2] = w / sg; d[3] = (-1.0 + w * z) / sg;
    } break;
    case NUTS_D_BETA: {  // continuous.py:1248-1262 (alpha, beta constant)
      double v = a[0], al = a[1], be = a[2];
      lp = (al == 1.0 ? 0.0 : (al - 1.0) * log(v)) + (be == 1.0 ? 0.0 : (be - 1.0) * log1p(-v)) + konst;
      d[0] = (al == 1.0 ? 0.0 : (al - 1.0) / v) - (be == 1.0 ? 0.0 : (be - 1.0) / (1.0 - v));
      KILL_UNLESS(v >= 0 && v <= 1)
    } break;
    case NUTS_D_EXPONENTIAL: {  // continuous.py:1478-1486 (mu = 1/lam)
      double v = a[0], lam = a[1];
      lp = log(lam) - v * lam;
      KILL_UNLESS(v >= 0)
      KILL_UNLESS(lam > 0)
      d[0] = -lam; d[1] = 1.0 / lam - v;
    } break;
    case NUTS_D_UNIFORM: {  // continuous.py:309-321
      double v = a[0], lo = a[1], hi = a[2];
      lp = -log(hi - lo);
      KILL_UNLESS(v >= lo && v <= hi)
      KILL_UNLESS(lo <= hi)
    } break;
    case NUTS_D_BERNOULLI_LOGIT: {  // discrete.py:351-352,362-374
      double y = a[0], eta = a[1];
      lp = (y != 0.0) ? -softplus_d(-eta) : -softplus_d(eta);
      KILL_UNLESS(y >= 0 && y <= 1)
      d[1] = y - sigmoid_d(eta);
    } break;
    case NUTS_D_LOGNORMAL: {  // continuous.py:1807-1819
      double v = a[0], sg = a[2], lv = log(v), z = (lv - a[1]) / sg;
      lp = -0.5 * z * z - LOG_SQRT_2PI - log(sg) - lv;
      KILL_UNLESS(v > 0)
      KILL_UNLESS(sg > 0)
      d[0] = (-z / sg - 1.0) / v; d[1] = z / sg; d[2] = (z * z - 1.0) / sg;
    } break;
    case NUTS_D_BERNOULLI: {  // discrete.py:362-374
      double y = a[0], p = a[1];
      lp = (y != 0.0) ? log(p) : log1p(-p);
      d[1] = (y != 0.0) ? 1.0 / p : -1.0 / (1.0 - p);
      KILL_UNLESS(y >= 0 && y <= 1)
      KILL_UNLESS(p >= 0 && p <= 1)
    } break;
    default: lp = NAN;
  }
  // every support / parameter check is a `switch(cond, logp, -inf)` in the reference graph
  // (dist_math.py:50-74, logprob/utils.py:209-225): its gradient is 0 where the check fails.
  if (dead) d[0] = d[1] = d[2] = d[3] = 0.0;
#undef KILL_UNLESS
  return lp;
}

__global__ __launch_bounds__(ELEM_THREADS) void k_model_elem(ModelDev md, const double* __restrict__ q,
                                                             const int* __restrict__ abort_flag) {
  if (abort_flag && *abort_flag) return;
  __shared__ double sm[ELEM_THREADS / WAVE];
  const int tid = threadIdx.x;
  double lp_local = 0.0;
  // -- transforms ----------------------------------------------------------
  for (int vi = 0; vi < md.n_vars; ++vi) {
    const nuts_var v = md.vars[vi];
    for (int i = tid; i < v.size; i += ELEM_THREADS) {
      const int k = v.offset + i;
      const double qi = q[k];
      double x, dx, lj, dj;
      switch (v.transform) {
        case NUTS_TR_LOG:  // transforms.py:880-891
          x = exp(qi); dx = x; lj = qi; dj = 1.0;
          break;
        case NUTS_TR_LOGODDS: {  // transforms.py:1076-1088
          double s = sigmoid_d(qi);
          x = s; dx = s * (1.0 - s); lj = -softplus_d(-qi) - softplus_d(qi); dj = 1.0 - 2.0 * s;
        } break;
        case NUTS_TR_INTERVAL: {  // transforms.py:1017-1073
          double s = sigmoid_d(qi);
          x = s * v.upper + (1.0 - s) * v.lower;
          dx = (v.upper - v.lower) * s * (1.0 - s);
          lj = log(v.upper - v.lower) - 2.0 * softplus_d(-qi) - qi;
          dj = 1.0 - 2.0 * s;
        } break;
        default:
          x = qi; dx = 1.0; lj = 0.0; dj = 0.0;
      }
      md.x[k] = x; md.dxdq[k] = dx; md.djac[k] = dj; md.gx[k] = 0.0;
      lp_local += lj;
    }
  }
  __syncthreads();
  // -- factors ---------------------------------------------------------------
  for (int fi = 0; fi < md.n_factors; ++fi) {
    const nuts_factor* f = &md.factors[fi];
    const int nargs = f->nargs, dist = f->dist, size = f->size;
    const double konst = f->konst;
    double bsum[12];
#pragma unroll
    for (int s = 0; s < 12; ++s) bsum[s] = 0.0;
    for (int i = tid; i < size; i += ELEM_THREADS) {
      double a[4], d[4], bv[4], cv[4];
      for (int k = 0; k < 4; ++k) {
        if (k < nargs) {
          const nuts_term& t = f->arg[k];
          double av = op_value(t.a, i, md);
          bv[k] = op_value(t.b, i, md);
          cv[k] = op_value(t.c, i, md);
          a[k] = av + bv[k] * cv[k];
        } else { a[k] = 0.0; bv[k] = cv[k] = 0.0; }
      }
      lp_local += dist_eval(dist, konst, a, d);
      for (int k = 0; k < 4; ++k) {
        if (k >= nargs) break;
        const nuts_term& t = f->arg[k];
        const double g3[3] = {d[k], d[k] * cv[k], d[k] * bv[k]};
        const nuts_operand* ops[3] = {&t.a, &t.b, &t.c};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const nuts_operand& o = *ops[s];
          if (o.kind != NUTS_OP_VAR) continue;
          const nuts_var v = md.vars[o.ref];
          if (v.size == 1) bsum[k * 3 + s] += g3[s];
          else md.gx[v.offset + i] += g3[s];  // element i of this var is touched by this thread only
        }
      }
    }
    // broadcast (size-1) variables: deterministic workgroup reduction
    for (int k = 0; k < nargs; ++k) {
      const nuts_term& t = f->arg[k];
      const nuts_operand* ops[3] = {&t.a, &t.b, &t.c};
      for (int s = 0; s < 3; ++s) {
        if (!op_is_bcast_var(*ops[s], md)) continue;  // uniform across the workgroup
        double tot = block_sum<false>(bsum[k * 3 + s], sm);
        if (tid == 0) md.gx[md.vars[ops[s]->ref].offset] += tot;
      }
    }
    __syncthreads();
  }
  double tot = block_sum<false>(lp_local, sm);
  if (tid == 0) *md.lp_elem = tot;
}

// ---------------------------------------------------------------------------
// M2: hierarchical Bernoulli-logit rows (the HBM-bound pass)
// ---------------------------------------------------------------------------
//
// Each wave owns a contiguous range of 256-row spans; in a span lane l holds rows
// 4l..4l+3, so every column read is one 16-byte load per lane, 1 KiB contiguous per
// wave-instruction.  Rows are sorted by group: a wave keeps per-lane accumulators
// of d logp / d beta_g for its current group and flushes a wave-reduced partial
// ("segment") whenever the group changes.  Segment slots are static (computed on
// the host from the group ids), so the combine order is fixed.

template <int D>
__device__ __forceinline__ void logit_flush(double (&acc)[D], double* __restrict__ seg_part, int& seg, int lane) {
#pragma unroll
  for (int d = 0; d < D; ++d) {
    double s = wave_sum(acc[d]);
    if (lane == 0) seg_part[(int64_t)seg * D + d] = s;
    acc[d] = 0.0;
  }
  ++seg;
}

template <int D>
__global__ __launch_bounds__(ROWS_BLOCK) void k_logit_rows(LogitDev lg, const double* __restrict__ q,
                                                           const int* __restrict__ abort_flag) {
  if (abort_flag && *abort_flag) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = blockIdx.x * (ROWS_BLOCK / WAVE) + (threadIdx.x >> 6);
  if (wave >= lg.n_waves) return;
  const int64_t s0 = (int64_t)wave * lg.n_spans / lg.n_waves;
  const int64_t s1 = (int64_t)(wave + 1) * lg.n_spans / lg.n_waves;
  if (s0 >= s1) {
    if (lane == 0) lg.wave_lp[wave] = 0.0;
    return;
  }
  int seg = lg.seg_base[wave];

  double mu[D], sg[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    mu[d] = q[lg.off_mu + d];
    double s = q[lg.off_sigma + d];
    sg[d] = lg.sigma_tr == NUTS_TR_LOG ? exp(s) : s;
  }
  double acc[D], beta[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { acc[d] = 0.0; beta[d] = 0.0; }
  double lp = 0.0;
  int g_cur = -1;

  for (int64_t sp = s0; sp < s1; ++sp) {
    const int64_t r0 = sp * ROWS_PER_SPAN + (int64_t)lane * ROWS_PER_LANE;
    const int4 gi = *reinterpret_cast<const int4*>(lg.gid + r0);
    const uint32_t y4 = *reinterpret_cast<const uint32_t*>(lg.y + r0);
    double x[D][ROWS_PER_LANE];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double2 a = *reinterpret_cast<const double2*>(lg.Xt + (int64_t)d * lg.Npad + r0);
      const double2 b = *reinterpret_cast<const double2*>(lg.Xt + (int64_t)d * lg.Npad + r0 + 2);
      x[d][0] = a.x; x[d][1] = a.y; x[d][2] = b.x; x[d][3] = b.y;
    }
    const int g[ROWS_PER_LANE] = {gi.x, gi.y, gi.z, gi.w};
    const int g_first = __builtin_amdgcn_readfirstlane(gi.x);
    const int g_last = __builtin_amdgcn_readlane(gi.w, WAVE - 1);

    if (g_first == g_last) {
      // ---- fast path: the whole span belongs to one group (wave-uniform beta) ----
      if (g_first != g_cur) {
        if (g_cur >= 0) logit_flush<D>(acc, lg.seg_part, seg, lane);
        g_cur = g_first;
#pragma unroll
        for (int d = 0; d < D; ++d) beta[d] = mu[d] + sg[d] * q[lg.off_z + (int64_t)g_cur * D + d];
      }
#pragma unroll
      for (int k = 0; k < ROWS_PER_LANE; ++k) {
        double eta = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) eta = fma(x[d][k], beta[d], eta);
        const bool valid = (r0 + k) < lg.N;
        const double yk = (double)((y4 >> (8 * k)) & 0xffu);
        const double e = exp(-fabs(eta));
        const double l1p = log1p(e);
        const double inv = 1.0 / (1.0 + e);
        const double sgm = eta >= 0 ? inv : e * inv;          // sigmoid(eta)
        const double spl = (eta > 0 ? eta : 0.0) + l1p;       // softplus(eta)
        const double r = valid ? (yk - sgm) : 0.0;
        lp += valid ? (yk * eta - spl) : 0.0;                 // y ? -softplus(-eta) : -softplus(eta)
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fma(r, x[d][k], acc[d]);
      }
    } else {
      // ---- mixed span: per-row beta gather, then masked accumulation group by group ----
      double rr[ROWS_PER_LANE];
#pragma unroll
      for (int k = 0; k < ROWS_PER_LANE; ++k) {
        double eta = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double b = mu[d] + sg[d] * q[lg.off_z + (int64_t)g[k] * D + d];
          eta = fma(x[d][k], b, eta);
        }
        const bool valid = (r0 + k) < lg.N;
        const double yk = (double)((y4 >> (8 * k)) & 0xffu);
        const double e = exp(-fabs(eta));
        const double l1p = log1p(e);
        const double inv = 1.0 / (1.0 + e);
        const double sgm = eta >= 0 ? inv : e * inv;
        const double spl = (eta > 0 ? eta : 0.0) + l1p;
        rr[k] = valid ? (yk - sgm) : 0.0;
        lp += valid ? (yk * eta - spl) : 0.0;
      }
      int gq = g_first;
      while (true) {
        if (gq != g_cur) {
          if (g_cur >= 0) logit_flush<D>(acc, lg.seg_part, seg, lane);
          g_cur = gq;
        }
        int nxt = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < ROWS_PER_LANE; ++k) {
          const double r = (g[k] == gq) ? rr[k] : 0.0;
#pragma unroll
          for (int d = 0; d < D; ++d) acc[d] = fma(r, x[d][k], acc[d]);
          if (g[k] > gq && g[k] < nxt) nxt = g[k];
        }
        nxt = wave_min_i(nxt);
        if (nxt == 0x7fffffff) break;
        gq = nxt;
      }
      // beta for the (possibly continuing) last group of the span
#pragma unroll
      for (int d = 0; d < D; ++d) beta[d] = mu[d] + sg[d] * q[lg.off_z + (int64_t)g_cur * D + d];
    }
  }
  if (g_cur >= 0) logit_flush<D>(acc, lg.seg_part, seg, lane);
  lp = wave_sum(lp);
  if (lane == 0) lg.wave_lp[wave] = lp;
}

// M3: per-group combine. thread = (group, d); 256 threads = 32 groups x 8.
template <int D>
__global__ __launch_bounds__(256) void k_logit_groups(LogitDev lg, ModelDev md, const double* __restrict__ q,
                                                      const int* __restrict__ abort_flag) {
  if (abort_flag && *abort_flag) return;
  constexpr int GPB = 256 / D;
  __shared__ double sh[2][256];
  const int d = threadIdx.x % D, gl = threadIdx.x / D;
  const int g = blockIdx.x * GPB + gl;
  double db = 0.0, dbz = 0.0;
  if (g < lg.G) {
    for (int s = lg.gseg_ptr[g]; s < lg.gseg_ptr[g + 1]; ++s) db += lg.seg_part[(int64_t)s * D + d];
    double sgd = q[lg.off_sigma + d];
    sgd = lg.sigma_tr == NUTS_TR_LOG ? exp(sgd) : sgd;
    const double z = q[lg.off_z + (int64_t)g * D + d];
    md.gdense[lg.off_z + (int64_t)g * D + d] = sgd * db;
    dbz = db * z;
  }
  sh[0][threadIdx.x] = db;
  sh[1][threadIdx.x] = dbz;
  __syncthreads();
  if (threadIdx.x < 2 * D) {
    const int which = threadIdx.x / D, dd = threadIdx.x % D;
    double s = 0.0;
    for (int k = 0; k < GPB; ++k) s += sh[which][k * D + dd];
    lg.gblk_part[((int64_t)blockIdx.x * 2 + which) * D + dd] = s;
  }
}

// ---------------------------------------------------------------------------
// MvNormal precision mat-vec: one wave per row of P (multivariate.py:165-185, 275-295)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mvn_matvec(MvnDev mv, ModelDev md, const double* __restrict__ q,
                                                    const int* __restrict__ abort_flag) {
  if (abort_flag && *abort_flag) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const int row = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
  if (row >= mv.k) return;
  const double* __restrict__ pr = mv.prec + (int64_t)row * mv.k;
  double s = 0.0;
  const int k2 = mv.k & ~1;
  for (int j = lane * 2; j < k2; j += 2 * WAVE) {
    const double2 p = *reinterpret_cast<const double2*>(pr + j);
    s = fma(p.x, q[mv.off + j] - mv.mu[j], s);
    s = fma(p.y, q[mv.off + j + 1] - mv.mu[j + 1], s);
  }
  if (lane == 0 && (mv.k & 1)) s = fma(pr[mv.k - 1], q[mv.off + mv.k - 1] - mv.mu[mv.k - 1], s);
  s = wave_sum(s);
  if (lane == 0) {
    md.gdense[mv.off + row] = -s;
    mv.rowq[row] = (q[mv.off + row] - mv.mu[row]) * s;
  }
}

// ---------------------------------------------------------------------------
// M4: chain rule through the value transforms + final scalar
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_model_final(ModelDev md, double* __restrict__ grad, double* __restrict__ logp,
                                                     const int* __restrict__ abort_flag) {
  if (abort_flag && *abort_flag) return;
  __shared__ double sm[256 / WAVE];
  const LogitDev& lg = md.lg;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < md.n; i += gridDim.x * blockDim.x) {
    double gd = 0.0;
    if (md.has_logit) {
      if (i >= lg.off_z && i < lg.off_z + lg.G * D) gd = md.gdense[i];
      else if (i >= lg.off_mu && i < lg.off_mu + D) {
        for (int b = 0; b < lg.n_gblk; ++b) gd += lg.gblk_part[((int64_t)b * 2 + 0) * D + (i - lg.off_mu)];
      } else if (i >= lg.off_sigma && i < lg.off_sigma + D) {
        for (int b = 0; b < lg.n_gblk; ++b) gd += lg.gblk_part[((int64_t)b * 2 + 1) * D + (i - lg.off_sigma)];
      }
    }
    if (md.has_mvn && i >= md.mv.off && i < md.mv.off + md.mv.k) gd += md.gdense[i];
    grad[i] = (md.gx[i] + gd) * md.dxdq[i] + md.djac[i];
  }
  if (blockIdx.x == 0) {
    double s = 0.0;
    if (md.has_logit)
      for (int w = threadIdx.x; w < lg.n_waves; w += blockDim.x) s += lg.wave_lp[w];
    double tot = block_sum<false>(s, sm);
    double sq = 0.0;
    if (md.has_mvn)
      for (int r = threadIdx.x; r < md.mv.k; r += blockDim.x) sq += md.mv.rowq[r];
    double totq = block_sum<false>(sq, sm);
    if (threadIdx.x == 0) {
      double r = *md.lp_elem + tot;
      if (md.has_mvn) r += md.mv.konst - 0.5 * totq;
      *logp = r;
    }
  }
}

// Dense node 5: linear predictors inside factor arguments (include/nuts_mi355.h, nuts_lin).
//
// `pm.math.dot(X, beta)` (pymc/math.py:56) with X constant data, read by ANY factor -- the location of a StudentT, the log-mean of
// a NegativeBinomial, the K columns under a softmax -- and, X with one row, a weighted sum over a long axis (`pt.sum(x)`).  What
// `pytensor.grad` does with the `Dot` node inside ValueGradFunction (model/core.py:213-267) is cut in three here:
//
//   k_lin_fwd / k_lin_fwd1   eta_k = X coef_k at this leaf's position, before anything reads it;
//   (k_gsweep, kernels.h)    one forward + reverse sweep per factor element; a predictor column is a slot of the gathered-adjoint
//                            table, so the sweep leaves d logp / d eta_k[i] in ModelDev.adj;
//   k_lin_bwd / k_lin_bwd1   X^T adj_k, as per-chunk partial sums in a fixed order;
//   k_lin_fin                every coefficient adds its partials up, in order, into lin_gdense (or the seed of a derived vector),
//                            which kernels B / C add to the element's gradient like the other dense nodes' share.
//
// X is kept TRANSPOSED ([P][N]): a thread per row reads column p of 64 consecutive rows as one coalesced load, forwards and backwards,
// whatever P is.  Nothing here is atomic: the result does not depend on the schedule.
#pragma once
#include "kernels.h"

__device__ __forceinline__ double lin_coef(const ModelDev& md, const LinDev& L, const QView& qv, int k, int p) {
  const int e = L.coef[k * L.P + p];
  const LinCol& c = L.col[k];
  if (c.transform < 0) return md.pool[e];   // an element of a derived vector (k_derive ran before this kernel)
  VarDev v{};
  v.transform = c.transform; v.lower = c.lower; v.upper = c.upper;
  return transform_x(v, qv.at(e));
}

// N > 1: a thread per row, the K columns' coefficients in LDS
template <int KT>
__global__ __launch_bounds__(256) void k_lin_fwd(ModelDev md, ArenaDev A, EvalIO io, int j, int li) {
  if (load_aborted(io, A)) return;
  extern __shared__ double s_coef[];   // [K][P]
  const LinDev& L = md.lins[li];
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  const int P = L.P, K = L.K;
  const int64_t N = L.N;
  for (int t = threadIdx.x; t < K * P; t += 256) s_coef[t] = lin_coef(md, L, qv, t / P, t % P);
  __syncthreads();
  const double* Xt = L.Xt;
  double* eta = L.eta;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    double acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.0;
    int p = 0;
    for (; p + 4 <= P; p += 4) {   // four columns of X in flight; the additions stay in column order
      const double x0 = Xt[(int64_t)p * N + i], x1 = Xt[(int64_t)(p + 1) * N + i], x2 = Xt[(int64_t)(p + 2) * N + i], x3 = Xt[(int64_t)(p + 3) * N + i];
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) {
          const double* b = s_coef + k * P + p;
          acc[k] = fma(x3, b[3], fma(x2, b[2], fma(x1, b[1], fma(x0, b[0], acc[k]))));
        }
    }
    for (; p < P; ++p) {
      const double x0 = Xt[(int64_t)p * N + i];
#pragma unroll
      for (int k = 0; k < KT; ++k) if (k < K) acc[k] = fma(x0, s_coef[k * P + p], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) if (k < K) eta[(int64_t)k * N + i] = acc[k];
  }
}

// N == 1 (a weighted sum over a long axis): one workgroup per column, eight loads in flight per thread, fixed order
__global__ __launch_bounds__(1024) void k_lin_fwd1(ModelDev md, ArenaDev A, EvalIO io, int j, int li) {
  if (load_aborted(io, A)) return;
  __shared__ double s_w[1024 / WAVE];
  const LinDev& L = md.lins[li];
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  const int k = blockIdx.x, P = L.P, nt = 1024;
  double acc = 0.0;
  for (int p0 = threadIdx.x; p0 < P; p0 += 8 * nt) {
    double x[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int p = min(p0 + u * nt, P - 1); x[u] = L.Xt[p]; b[u] = lin_coef(md, L, qv, k, p); }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = (p0 + u * nt < P) ? fma(x[u], b[u], acc) : acc;
  }
  const double t = block_sum<false>(acc, s_w);
  if (threadIdx.x == 0) L.eta[k] = t;
}

// d logp / d eta_k[i]: what the sweeps of the factors that read the column left behind
__device__ __forceinline__ double lin_adj(const ModelDev& md, const LinCol& c, int64_t i) {
  double a = md.adj[c.adj_off[0] + i];
  for (int u = 1; u < c.n_use; ++u) a += md.adj[c.adj_off[u] + i];
  return a;
}

// N > 1: workgroup (c, p) = rows [c LIN_CHUNK, (c + 1) LIN_CHUNK) of column p of X against the K adjoint vectors
template <int KT>
__global__ __launch_bounds__(256) void k_lin_bwd(ModelDev md, ArenaDev A, EvalIO io, int li) {
  if (load_aborted(io, A)) return;
  __shared__ double s_w[KT][256 / WAVE];
  const LinDev& L = md.lins[li];
  const int c = blockIdx.x, p = blockIdx.y, K = L.K;
  const int64_t N = L.N, i0 = (int64_t)c * LIN_CHUNK, i1 = min(N, i0 + LIN_CHUNK);
  const double* xr = L.Xt + (int64_t)p * N;
  double acc[KT];
  const double* ak[KT];   // the usual case, one reading factor per column: its adjoints, the address hoisted out of the row loop
  bool one_use = true;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    acc[k] = 0.0;
    ak[k] = md.adj + L.col[k < K ? k : 0].adj_off[0];
    one_use = one_use && (k >= K || L.col[k].n_use == 1);
  }
  if (one_use) {
#pragma unroll 8
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      const double x = xr[i];
#pragma unroll
      for (int k = 0; k < KT; ++k) if (k < K) acc[k] = fma(x, ak[k][i], acc[k]);
    }
  } else {
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      const double x = xr[i];
#pragma unroll
      for (int k = 0; k < KT; ++k) if (k < K) acc[k] = fma(x, lin_adj(md, L.col[k], i), acc[k]);
    }
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) s_w[k][w] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    const int k = threadIdx.x;
    double t = 0.0;
    for (int ww = 0; ww < 256 / WAVE; ++ww) t += s_w[k][ww];
    L.part[((int64_t)k * L.P + p) * L.nchunk + c] = t;
  }
}

// N == 1: the column's total adjoint = the sum over every element of every factor that read it (the predictor broadcast)
__global__ __launch_bounds__(1024) void k_lin_bwd1(ModelDev md, ArenaDev A, EvalIO io, int li) {
  if (load_aborted(io, A)) return;
  __shared__ double s_w[1024 / WAVE];
  const LinDev& L = md.lins[li];
  const int k = blockIdx.x, nt = 1024;
  const LinCol& c = L.col[k];
  double acc = 0.0;
  for (int u = 0; u < c.n_use; ++u) {
    const double* a = md.adj + c.adj_off[u];
    const int M = c.use_size[u];
    for (int i0 = threadIdx.x; i0 < M; i0 += 8 * nt) {
      double v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = a[min(i0 + t * nt, M - 1)];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += (i0 + t * nt < M) ? v[t] : 0.0;
    }
  }
  const double t = block_sum<false>(acc, s_w);
  if (threadIdx.x == 0) L.part[k] = t;
}

// every coefficient that receives a gradient from the node: its partial sums, in the order the table lists them
__global__ __launch_bounds__(256) void k_lin_fin(ModelDev md, ArenaDev A, EvalIO io) {
  if (load_aborted(io, A)) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= md.n_lin_targets) return;
  const LinTarget T = md.lin_targets[t];
  double g = 0.0;
  for (int s = T.src0; s < T.src0 + T.n_src; ++s) {
    const LinSrc S = md.lin_srcs[s];
    const LinDev& L = md.lins[S.lin];
    if (L.N > 1) g += sum_strided(L.part + ((int64_t)S.k * L.P + S.p) * L.nchunk, 1, 0, L.nchunk);
    else g += L.Xt[S.p] * L.part[S.k];
  }
  *T.dst = g;
}

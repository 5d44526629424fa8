// QuadPotentialFullAdapt on the device (pymc/step_methods/hmc/quadpotential.py:748-852 with `_WeightedCovariance`, :855-910).
//
// The reference keeps two online covariance estimators (foreground / background windows) in NumPy, and after every
// `update_window` tuning draws turns the foreground estimate into the mass matrix: cov = raw_cov / (n - 1), a LAPACK Cholesky
// (n^3 / 3), `velocity = cov p`, `random = solve_triangular(chol^T, z)`.  With the estimators on the host that is, per tuning
// draw at n = 2048: two rank-1 updates of 33.5 MB matrices, a 2.9 GFLOP factorisation on a host core and 67 MB over PCIe.
// Here everything stays in HBM:
//
//   k_fa_diffs / k_fa_rank1   `add_sample` of both estimators: old_diff, mean update, new_diff, raw_cov += new_diff old_diff^T
//                             (one pass over the two matrices, HBM-bound: 134 MB at n = 2048)
//   k_fa_cov                  cov = raw_cov / (n - 1) into the matrix `k_dense_mv` multiplies by, and its lower triangle into L
//   k_chol_panel / _update    blocked right-looking Cholesky, 64 x 64 blocks: every panel workgroup factors the diagonal block in
//                             LDS (redundantly: 87 kflop, cheaper than a launch) and solves its block against it; the trailing
//                             update A_ij -= L_ik L_jk^T is the one place on this path where the work IS a matrix product
//                             (64 x 64 x 64 per workgroup, n^3 / 3 in all) and runs on the matrix cores:
//                             v_mfma_f64_16x16x4_f64, operands staged through LDS.  A non-positive pivot raises a flag that
//                             `raise_ok` reports, as the reference reports LinAlgError (quadpotential.py:806-812, 845-847).
//   k_trsv_block              random(): L^T p = z by blocked back substitution, one launch per 64-column block (each workgroup
//                             solves the 64 x 64 triangle itself, then updates its slice of the remaining right-hand side)
//
// The device Cholesky does not round like LAPACK's, so a chain adapted here agrees with the host-adapted one to rounding, not
// bit for bit; the host estimator (`pymc_amd/quadpotential.py`, bitwise the reference's class) stays available.
#pragma once
#include "device_math.h"

#define FA_NB 64   // block size of the factorisation

// ---- add_sample of both estimators ---------------------------------------------------------------------------------------
// scratch: [4][n] = old_diff fg, new_diff fg, old_diff bg, new_diff bg
__global__ __launch_bounds__(256) void k_fa_diffs(int n, const double* __restrict__ x, double* fg_mean, double* bg_mean, double fg_cnt_new,
                                                  double bg_cnt_new, double* __restrict__ scratch) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double xi = x[i];
  double m = fg_mean[i];
  double od = xi - m;
  m += od / fg_cnt_new;
  fg_mean[i] = m;
  scratch[i] = od; scratch[n + i] = xi - m;
  m = bg_mean[i];
  od = xi - m;
  m += od / bg_cnt_new;
  bg_mean[i] = m;
  scratch[2 * n + i] = od; scratch[3 * n + i] = xi - m;
}

// raw[i][j] += new_diff[i] * old_diff[j] for both estimators; grid (ceil(n / 256), n): one row per blockIdx.y
__global__ __launch_bounds__(256) void k_fa_rank1(int n, const double* __restrict__ scratch, double* __restrict__ fg_raw,
                                                  double* __restrict__ bg_raw) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= n) return;
  const int64_t o = (int64_t)i * n + j;
  fg_raw[o] = fma(scratch[n + i], scratch[j], fg_raw[o]);
  bg_raw[o] = fma(scratch[3 * n + i], scratch[2 * n + j], bg_raw[o]);
}

// C = raw / denom (full matrix); L = its lower triangle (upper part zero); resets the factorisation's failure flag
__global__ __launch_bounds__(256) void k_fa_cov(int n, const double* __restrict__ raw, double denom, double* __restrict__ C,
                                                double* __restrict__ L, int* fail) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (i == 0 && j == 0) *fail = 0;
  if (j >= n) return;
  const int64_t o = (int64_t)i * n + j;
  const double c = raw[o] / denom;
  C[o] = c;
  L[o] = j <= i ? c : 0.0;
}

// ---- blocked Cholesky (lower), in place in L [n][n] ------------------------------------------------------------------------
// Step k: workgroup b (0 .. nb-k-1) owns block row i = k + b.  All load A_kk into LDS and factor it (the same arithmetic in
// every workgroup, so they agree bit for bit); workgroup 0 writes L_kk back, the others solve X L_kk^T = A_ik and write L_ik.
__global__ __launch_bounds__(256) void k_chol_panel(int n, double* __restrict__ L, int k, int* fail) {
  __shared__ double s_d[FA_NB][FA_NB + 1];
  __shared__ double s_a[FA_NB][FA_NB + 1];
  const int tid = threadIdx.x;
  const int r0 = k * FA_NB;
  const int nk = min(FA_NB, n - r0);          // size of the diagonal block (the last one may be smaller)
  const int i = k + (int)blockIdx.x;
  const int i0 = i * FA_NB;
  const int ni = min(FA_NB, n - i0);
  for (int e = tid; e < FA_NB * FA_NB; e += 256) {
    const int r = e / FA_NB, c = e % FA_NB;
    s_d[r][c] = (r < nk && c < nk) ? L[(int64_t)(r0 + r) * n + r0 + c] : (r == c ? 1.0 : 0.0);
    if (blockIdx.x > 0) s_a[r][c] = (r < ni && c < nk) ? L[(int64_t)(i0 + r) * n + r0 + c] : 0.0;
  }
  __syncthreads();
  // unblocked factorisation of the diagonal block (right-looking, column by column)
  for (int c = 0; c < nk; ++c) {
    if (tid == 0) {
      const double p = s_d[c][c];
      if (!(p > 0.0)) { *fail = 1; s_d[c][c] = NAN; }
      else s_d[c][c] = sqrt(p);
    }
    __syncthreads();
    const double piv = s_d[c][c];
    if (tid > c && tid < nk) s_d[tid][c] /= piv;
    __syncthreads();
    // trailing update of the lower triangle: a[r][c2] -= a[r][c] a[c2][c], c < c2 <= r  (only the m x m block still active)
    const int m = nk - c - 1;
    for (int e = tid; e < m * m; e += 256) {
      const int r = c + 1 + e / m, c2 = c + 1 + e % m;
      if (c2 <= r) s_d[r][c2] = fma(-s_d[r][c], s_d[c2][c], s_d[r][c2]);
    }
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (int e = tid; e < FA_NB * FA_NB; e += 256) {
      const int r = e / FA_NB, c = e % FA_NB;
      if (r < nk && c <= r) L[(int64_t)(r0 + r) * n + r0 + c] = s_d[r][c];
    }
    return;
  }
  // X L_kk^T = A_ik  =>  x[r][c] = (a[r][c] - sum_{t<c} x[r][t] l[c][t]) / l[c][c].  Right-looking over the columns: column c is
  // final after the division, then every later column of every row takes its term -- 64 rows x (nk - c - 1) columns at a time
  // over all 256 threads instead of one thread walking a whole row.
  for (int c = 0; c < nk; ++c) {
    if (tid < ni) s_a[tid][c] /= s_d[c][c];
    __syncthreads();
    const int m = nk - c - 1;
    for (int e = tid; e < FA_NB * m; e += 256) {
      const int r = e / m, c2 = c + 1 + e % m;
      s_a[r][c2] = fma(-s_a[r][c], s_d[c2][c], s_a[r][c2]);
    }
    __syncthreads();
  }
  for (int e = tid; e < FA_NB * FA_NB; e += 256) {
    const int r = e / FA_NB, c = e % FA_NB;
    if (r < ni && c < nk) L[(int64_t)(i0 + r) * n + r0 + c] = s_a[r][c];
  }
}

typedef double fa_v4d __attribute__((ext_vector_type(4)));

// Trailing update of step k: A_ij -= L_ik L_jk^T for k < j <= i.  One workgroup (4 waves) per 64 x 64 tile (i, j); wave w owns the
// 16 rows 16 w .. 16 w + 15 of the tile and all four 16-column blocks.  v_mfma_f64_16x16x4_f64: A operand lane l = A[l & 15][l >> 4],
// B operand lane l = B[l >> 4][l & 15], result register t of lane l = D[(l >> 4) + 4 t][l & 15] (MI355X guide).
// `use_mfma` = 0: the same tile with plain fma (A/B measurement, NUTS_FA_MFMA=0).
__global__ __launch_bounds__(256) void k_chol_update(int n, double* __restrict__ L, int k, int nb, int use_mfma) {
  // tile index -> (i, j), k < j <= i < nb: enumerate rows of the strictly-trailing lower triangle
  const int m = nb - k - 1;                      // trailing block rows
  int t = (int)blockIdx.x, bi = 0;
  while (t >= bi + 1) { t -= bi + 1; ++bi; }     // (m <= 64: a short loop)
  const int i = k + 1 + bi, j = k + 1 + t;
  if (bi >= m) return;
  __shared__ double s_li[FA_NB][FA_NB + 1];      // L_ik
  __shared__ double s_lj[FA_NB][FA_NB + 1];      // L_jk
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i0 = i * FA_NB, j0 = j * FA_NB, c0 = k * FA_NB;
  const int ni = min(FA_NB, n - i0), nj = min(FA_NB, n - j0);
  for (int e = tid; e < FA_NB * FA_NB; e += 256) {
    const int r = e / FA_NB, c = e % FA_NB;
    s_li[r][c] = r < ni ? L[(int64_t)(i0 + r) * n + c0 + c] : 0.0;
    s_lj[r][c] = r < nj ? L[(int64_t)(j0 + r) * n + c0 + c] : 0.0;
  }
  __syncthreads();
  if (use_mfma) {
    fa_v4d acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[cb] = (fa_v4d){0.0, 0.0, 0.0, 0.0};
    const int ar = 16 * w + (lane & 15), kq = lane >> 4;
#pragma unroll 4
    for (int kk = 0; kk < FA_NB; kk += 4) {
      const double a = s_li[ar][kk + kq];                      // A[row][k]
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const double b = s_lj[16 * cb + (lane & 15)][kk + kq];   // B[k][col] = L_jk[col][k]
        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[cb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int r = 16 * w + (lane >> 4) + 4 * tt, c = 16 * cb + (lane & 15);
        if (r < ni && c < nj && (i > j || c <= r)) {
          const int64_t o = (int64_t)(i0 + r) * n + j0 + c;
          L[o] -= acc[cb][tt];
        }
      }
    }
  } else {
    for (int e = tid; e < FA_NB * FA_NB; e += 256) {
      const int r = e / FA_NB, c = e % FA_NB;
      if (r < ni && c < nj && (i > j || c <= r)) {
        double s = 0.0;
        for (int kk = 0; kk < FA_NB; ++kk) s = fma(s_li[r][kk], s_lj[c][kk], s);
        L[(int64_t)(i0 + r) * n + j0 + c] -= s;
      }
    }
  }
}

// ---- random(): L^T p = z, blocked back substitution ------------------------------------------------------------------------
// Launch for block b = nb-1 .. 0 with grid = max(1, ceil(b 64 / 256)) workgroups.  Every workgroup solves the 64 x 64 upper
// triangle L_bb^T x_b = rhs_b itself (in LDS); workgroup 0 stores x_b into p; all update their 256 columns of the remaining
// right-hand side: rhs[j] -= sum_r L[b0 + r][j] x_b[r], j < b0 (rows of L are contiguous along j: coalesced).
// `rhs` is a work copy of z (block b of it is final when launch b starts and is only read then); the solution goes to `out`.
__global__ __launch_bounds__(256) void k_trsv_block(int n, const double* __restrict__ L, double* __restrict__ rhs, double* __restrict__ out,
                                                    int b) {
  __shared__ double s_t[FA_NB][FA_NB + 1];
  __shared__ double s_x[FA_NB];
  const int tid = threadIdx.x;
  const int b0 = b * FA_NB;
  const int nbk = min(FA_NB, n - b0);
  for (int e = tid; e < FA_NB * FA_NB; e += 256) {
    const int r = e / FA_NB, c = e % FA_NB;
    s_t[r][c] = (r < nbk && c <= r) ? L[(int64_t)(b0 + r) * n + b0 + c] : 0.0;
  }
  if (tid < FA_NB) s_x[tid] = tid < nbk ? rhs[b0 + tid] : 0.0;
  __syncthreads();
  // (L^T x)[c] = sum_{r >= c} L[r][c] x[r]: solve from the last row up; thread c owns x[c]
  for (int r = nbk - 1; r >= 0; --r) {
    if (tid == r) s_x[r] = s_x[r] / s_t[r][r];
    __syncthreads();
    if (tid < r) s_x[tid] = fma(-s_t[r][tid], s_x[r], s_x[tid]);
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid < nbk) out[b0 + tid] = s_x[tid];
  const int j = blockIdx.x * 256 + tid;
  if (j < b0) {
    double v = rhs[j];
    for (int r = 0; r < nbk; ++r) v = fma(-L[(int64_t)(b0 + r) * n + j], s_x[r], v);
    rhs[j] = v;
  }
}

// the factorisation went through: it becomes the factor in use (a failed one leaves the previous factor in place, as the
// exception does in the reference, quadpotential.py:806-812)
__global__ __launch_bounds__(256) void k_fa_commit(int64_t nn, const double* __restrict__ L_new, double* __restrict__ L_cur, const int* fail,
                                                   int* fail_sticky) {
  if (*fail) { if (blockIdx.x == 0 && threadIdx.x == 0) *fail_sticky = 1; return; }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (int64_t)gridDim.x * 256) L_cur[e] = L_new[e];
}

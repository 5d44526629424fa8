// Experiment bench for the row-streaming access pattern (not product code).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rows_lab.hip -o gpurun_out/rows_lab && gpurun_out/rows_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int D = 8;

__device__ __forceinline__ double wsum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ void logit_row(double eta, double yk, double& lp, double& r) {
  const double e = exp(-fabs(eta));
  const double l1p = log1p(e);
  const double inv = 1.0 / (1.0 + e);
  const double sgm = eta >= 0 ? inv : e * inv;
  const double spl = (eta > 0 ? eta : 0.0) + l1p;
  lp = yk * eta - spl;
  r = yk - sgm;
}

// LAYOUT 0: column-major [D][Npad]; 1: tiled [span][D][SPAN] (SPAN = 64*RPL rows)
// MATH 0: sum only; 1: full logit math with uniform beta
template <int RPL, int LAYOUT, int MATH, int NT>
__global__ __launch_bounds__(256) void k(const double* __restrict__ X, const int8_t* __restrict__ y, int64_t Npad, int64_t n_spans,
                                         int n_waves, double* out, int rev, int64_t nt_from = -1) {
  constexpr int SPAN = 64 * RPL;
  const int lane = threadIdx.x & 63;
  int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_waves) return;
  if (rev) wave = n_waves - 1 - wave;
  const int64_t s0 = (int64_t)wave * n_spans / n_waves, s1 = (int64_t)(wave + 1) * n_spans / n_waves;
  double beta[D];
  for (int d = 0; d < D; ++d) beta[d] = 0.01 * (d + 1);
  double acc[D];
  for (int d = 0; d < D; ++d) acc[d] = 0;
  double lp = 0;
  for (int64_t sp = s0; sp < s1; ++sp) {
    const int64_t r0 = sp * SPAN + lane * RPL;
    double x[D][RPL];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double* col = LAYOUT == 0 ? X + (int64_t)d * Npad + r0 : X + (sp * D + d) * SPAN + lane * RPL;
#pragma unroll
      for (int k2 = 0; k2 < RPL; k2 += 2) {
        double2 a;
        if (NT == 1 || (NT == 2 && sp >= nt_from)) { a.x = __builtin_nontemporal_load(col + k2); a.y = __builtin_nontemporal_load(col + k2 + 1); }
        else a = *reinterpret_cast<const double2*>(col + k2);
        x[d][k2] = a.x; x[d][k2 + 1] = a.y;
      }
    }
    uint32_t yb = RPL == 2 ? *reinterpret_cast<const uint16_t*>(y + r0) : *reinterpret_cast<const uint32_t*>(y + r0);
#pragma unroll
    for (int k2 = 0; k2 < RPL; ++k2) {
      if (MATH) {
        double eta = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) eta = fma(x[d][k2], beta[d], eta);
        double l, r;
        logit_row(eta, (double)((yb >> (8 * k2)) & 0xff), l, r);
        lp += l;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fma(r, x[d][k2], acc[d]);
      } else {
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += x[d][k2];
        lp += (double)((yb >> (8 * k2)) & 0xff);
      }
    }
  }
  double tot = lp;
  for (int d = 0; d < D; ++d) tot += wsum(acc[d]);
  tot = wsum(tot);
  if (lane == 0) out[wave] = tot;
}

template <int RPL, int LAYOUT, int MATH, int NT>
void run(const char* name, const double* X, const int8_t* y, int64_t N, int wpc, int alt, double* out) {
  constexpr int SPAN = 64 * RPL;
  const int64_t Npad = N, n_spans = N / SPAN;
  int n_waves = (int)std::min<int64_t>(256 * wpc, n_spans);
  n_waves = (n_waves + 3) / 4 * 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<RPL, LAYOUT, MATH, NT>), dim3(n_waves / 4), dim3(256), 0, 0, X, y, Npad, n_spans, n_waves, out, alt ? (i & 1) : 0);
  hipDeviceSynchronize();
  const int reps = 30;
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<RPL, LAYOUT, MATH, NT>), dim3(n_waves / 4), dim3(256), 0, 0, X, y, Npad, n_spans, n_waves, out, alt ? ((i + 1) & 1) : 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps;
  printf("%-28s rpl=%d wpc=%2d alt=%d: %7.1f us  %7.1f GB/s (65 B/row actual)\n", name, RPL, wpc, alt, us, 65.0 * N / us / 1e3);
  fflush(stdout);
}

template <int RPL>
void run_split(const double* X, const int8_t* y, int64_t N, int wpc, double frac, double* out) {
  constexpr int SPAN = 64 * RPL;
  const int64_t n_spans = N / SPAN;
  int n_waves = (int)std::min<int64_t>(256 * wpc, n_spans);
  n_waves = (n_waves + 3) / 4 * 4;
  const int64_t nt_from = (int64_t)(frac * n_spans);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<RPL, 1, 1, 2>), dim3(n_waves / 4), dim3(256), 0, 0, X, y, N, n_spans, n_waves, out, 0, nt_from);
  hipDeviceSynchronize();
  const int reps = 30;
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<RPL, 1, 1, 2>), dim3(n_waves / 4), dim3(256), 0, 0, X, y, N, n_spans, n_waves, out, 0, nt_from);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("tiled math, cached fraction %.2f (rest nontemporal), wpc=%d: %7.1f us\n", frac, wpc, ms * 1e3 / reps);
  fflush(stdout);
}

int main() {
  const int64_t N = 4992000;
  std::vector<double> hx((size_t)N * D);
  std::vector<int8_t> hy(N);
  srand(1);
  for (auto& v : hx) v = (rand() / (double)RAND_MAX - 0.5) * 2;
  for (auto& v : hy) v = rand() & 1;
  double* X; int8_t* y; double* out;
  CK(hipMalloc(&X, hx.size() * 8)); CK(hipMalloc(&y, N)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(y, hy.data(), N, hipMemcpyHostToDevice));
  for (double f : {0.0, 0.3, 0.5, 0.6, 0.7, 0.8, 1.0}) run_split<2>(X, y, N, 16, f, out);
  if (getenv("SPLIT_ONLY")) return 0;
  for (int alt = 0; alt < 2; ++alt)
    for (int wpc : {8, 16, 32}) {
      run<2, 0, 0, 0>("colmajor sum", X, y, N, wpc, alt, out);
      run<2, 1, 0, 0>("tiled sum", X, y, N, wpc, alt, out);
      run<4, 0, 0, 0>("colmajor sum", X, y, N, wpc, alt, out);
      run<4, 1, 0, 0>("tiled sum", X, y, N, wpc, alt, out);
      run<2, 1, 0, 1>("tiled sum nt", X, y, N, wpc, alt, out);
      run<2, 0, 1, 0>("colmajor math", X, y, N, wpc, alt, out);
      run<2, 1, 1, 0>("tiled math", X, y, N, wpc, alt, out);
      run<4, 1, 1, 0>("tiled math", X, y, N, wpc, alt, out);
    }
  return 0;
}

// Measurement, not product code: what would the MvNormal mat-vec of C3 (P delta, P = 2048 x 2048 fp64, 33.5 MB) gain from the
// matrix cores if several chains advanced in LOCKSTEP on one GPU, so that P is read once for all of them?
//
//   Y[K][NC] = P[K][K] * Q[K][NC]      NC = chains sharing the pass
//
//   fma  : the product kernel's scheme (a workgroup owns R rows, four waves stream them, each lane keeps R x NC accumulators)
//   mfma : v_mfma_f64_16x16x4_f64, one wave per 16-row block and K-slice; NC <= 16 chains occupy NC of the tile's 16 columns
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_matvec_bench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
// Prints one line per variant: microseconds per pass (HIP events over 200 back-to-back launches), GB/s on 8 K^2 bytes, and
// the largest deviation from a host reference.  Results of this round: profiles/r02k_mfma_matvec.txt, discussed in DESIGN.md 4.2.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- plain fma: R rows per workgroup, NC chains ----
template <int R, int NC>
__global__ __launch_bounds__(256) void k_fma(const double* __restrict__ P, const double* __restrict__ Q, double* __restrict__ Y, int K) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row0 = blockIdx.x * R;
  __shared__ double s_w[R][NC][4];
  double s[R][NC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) s[r][c] = 0.0;
#pragma unroll 2
  for (int col = 2 * tid; col < K; col += 512) {
    double q0[NC], q1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { q0[c] = Q[(size_t)col * NC + c]; q1[c] = Q[(size_t)(col + 1) * NC + c]; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double2 p = *reinterpret_cast<const double2*>(P + (size_t)(row0 + r) * K + col);
#pragma unroll
      for (int c = 0; c < NC; ++c) { s[r][c] = fma(p.x, q0[c], s[r][c]); s[r][c] = fma(p.y, q1[c], s[r][c]); }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double t = wave_sum(s[r][c]);
      if (lane == 0) s_w[r][c][w] = t;
    }
  __syncthreads();
  if (tid < R * NC) {
    const int r = tid / NC, c = tid % NC;
    Y[(size_t)(row0 + r) * NC + c] = s_w[r][c][0] + s_w[r][c][1] + s_w[r][c][2] + s_w[r][c][3];
  }
}

// ---- matrix cores: workgroup (rb, s) = rows [16 rb, 16 rb + 16), K-slice s of S; its four waves split the slice ----
// operand layout of v_mfma_f64_16x16x4_f64: A lane l = A[l & 15][l >> 4], B lane l = B[l >> 4][l & 15], D reg t = D[(l >> 4) + 4 t][l & 15].
// A lane reads 32 contiguous bytes of its row (4 consecutive k) and uses them in 4 MFMAs; the k index a lane contributes in MFMA i
// is kk + 4 (l >> 4) + i on BOTH operands, which is all the instruction needs (the sum over k has no order).
template <int S>
__global__ __launch_bounds__(256) void k_mfma(const double* __restrict__ P, const double* __restrict__ Q16, double* __restrict__ part, int K) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rb = blockIdx.x / S, s = blockIdx.x % S;
  const int kslice = K / S, kw = kslice / 4;
  const int k0 = s * kslice + w * kw;
  const int row = rb * 16 + (lane & 15), ks = lane >> 4, col = lane & 15;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
  const double* pr = P + (size_t)row * K;
#pragma unroll 4
  for (int kk = k0; kk < k0 + kw; kk += 16) {
    const double2 a01 = *reinterpret_cast<const double2*>(pr + kk + 4 * ks);
    const double2 a23 = *reinterpret_cast<const double2*>(pr + kk + 4 * ks + 2);
    const double* qb = Q16 + (size_t)(kk + 4 * ks) * 16 + col;
    const double b0 = qb[0], b1 = qb[16], b2 = qb[32], b3 = qb[48];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.x, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a01.y, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.x, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a23.y, b3, acc, 0, 0, 0);
  }
  __shared__ double s_d[4][16][16];
#pragma unroll
  for (int t = 0; t < 4; ++t) s_d[w][ks + 4 * t][col] = acc[t];
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
  part[((size_t)s * K + rb * 16 + r) * 16 + c] = s_d[0][r][c] + s_d[1][r][c] + s_d[2][r][c] + s_d[3][r][c];
}
template <int S>
__global__ __launch_bounds__(256) void k_mfma_sum(const double* __restrict__ part, double* __restrict__ Y, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * 16) return;
  double t = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) t += part[(size_t)s * K * 16 + i];
  Y[i] = t;
}

template <typename F>
static double time_us(F launch, int iters) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 20; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) launch();
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return 1e3 * ms / iters;
}

int main() {
  const int K = 2048, ITERS = 200;
  std::vector<double> P((size_t)K * K), Q16((size_t)K * 16);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (auto& v : P) v = rnd();
  for (auto& v : Q16) v = rnd();
  double *dP, *dQ16, *dQn, *dY, *dPart;
  CHECK(hipMalloc(&dP, P.size() * 8)); CHECK(hipMalloc(&dQ16, Q16.size() * 8)); CHECK(hipMalloc(&dQn, Q16.size() * 8));
  CHECK(hipMalloc(&dY, (size_t)K * 16 * 8)); CHECK(hipMalloc(&dPart, (size_t)8 * K * 16 * 8));
  CHECK(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dQ16, Q16.data(), Q16.size() * 8, hipMemcpyHostToDevice));
  // host reference for all 16 columns
  std::vector<double> ref((size_t)K * 16, 0.0);
  for (int r = 0; r < K; ++r)
    for (int k = 0; k < K; ++k) {
      const double p = P[(size_t)r * K + k];
      for (int c = 0; c < 16; ++c) ref[(size_t)r * 16 + c] += p * Q16[(size_t)k * 16 + c];
    }
  auto pack = [&](int nc) {   // Q[K][nc] = the first nc columns
    std::vector<double> q((size_t)K * nc);
    for (int k = 0; k < K; ++k) for (int c = 0; c < nc; ++c) q[(size_t)k * nc + c] = Q16[(size_t)k * 16 + c];
    CHECK(hipMemcpy(dQn, q.data(), q.size() * 8, hipMemcpyHostToDevice));
  };
  auto check = [&](int nc) {
    std::vector<double> y((size_t)K * nc);
    CHECK(hipMemcpy(y.data(), dY, y.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int r = 0; r < K; ++r) for (int c = 0; c < nc; ++c) worst = std::fmax(worst, std::fabs(y[(size_t)r * nc + c] - ref[(size_t)r * 16 + c]));
    return worst;
  };
  const double bytes = 8.0 * K * K;
  auto report = [&](const char* what, int nc, double us, double err) {
    std::printf("%-44s chains %2d  %7.2f us per pass  %7.2f us per chain  %6.0f GB/s on P  max |err| %.1e\n", what, nc, us, us / nc, bytes / us * 1e-3, err);
  };
  std::printf("# Y[K][NC] = P[K][K] Q[K][NC], K = %d (P = %.1f MB, cache-resident between passes), %d launches per timing\n", K, bytes / 1e6, ITERS);
#define RUN_FMA(RR, NN)                                                                                                  \
  { pack(NN); double us = time_us([&]() { hipLaunchKernelGGL((k_fma<RR, NN>), dim3(K / RR), dim3(256), 0, 0, dP, dQn, dY, K); }, ITERS); \
    report("fma, " #RR " rows per workgroup", NN, us, check(NN)); }
  RUN_FMA(4, 1)
  RUN_FMA(4, 2)
  RUN_FMA(4, 4)
  RUN_FMA(2, 8)
  RUN_FMA(2, 16)
#define RUN_MFMA(SS)                                                                                                      \
  { double us = time_us([&]() { hipLaunchKernelGGL((k_mfma<SS>), dim3(K / 16 * SS), dim3(256), 0, 0, dP, dQ16, dPart, K);                \
                                hipLaunchKernelGGL((k_mfma_sum<SS>), dim3(K * 16 / 256), dim3(256), 0, 0, dPart, dY, K); }, ITERS);     \
    report("mfma 16x16x4 f64, K split " #SS " x 4 waves (+ sum)", 16, us, check(16)); }
  RUN_MFMA(2)
  RUN_MFMA(4)
  RUN_MFMA(8)
  return 0;
}

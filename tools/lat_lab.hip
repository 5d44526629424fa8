// Latency experiment (not product code): cost of one round of loads from a tiny kernel that runs right after
// a big streaming kernel, for (a) many separately allocated small buffers, (b) one slab.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void stream(const double* x, size_t n, double* out) {
  double s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  if (s == 123.456) out[0] = s;
}
struct Ptrs { const double* p[16]; };
__global__ void writer(Ptrs P, int nb) {  // emulate the producer kernel: many blocks write the buffers
  for (int k = 0; k < nb; ++k) ((double*)P.p[k])[blockIdx.x * 64 + threadIdx.x] = k + threadIdx.x;
}
__global__ void probe(Ptrs P, int nb, long long* ticks, double* out) {
  long long t0 = __builtin_readcyclecounter();
  double s = 0;
  for (int k = 0; k < nb; ++k) s += P.p[k][threadIdx.x];          // nb independent loads (one round)
  long long t1 = __builtin_readcyclecounter();
  double s2 = 0;
  for (int k = 0; k < nb; ++k) s2 += P.p[k][(int)(s * 0) + threadIdx.x + 64];  // dependent second round
  long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; }
  out[threadIdx.x] = s + s2;
}
int main() {
  const size_t NS = 50u << 20;  // 400 MB
  double* big; CK(hipMalloc(&big, NS * 8)); CK(hipMemset(big, 0, NS * 8));
  double* out; CK(hipMalloc(&out, 1 << 16));
  long long* ticks; CK(hipMalloc(&ticks, 64));
  for (int mode = 0; mode < 3; ++mode) {
    for (int nb : {1, 4, 12}) {
      Ptrs P{};
      double* slab = nullptr;
      if (mode == 0) { for (int k = 0; k < nb; ++k) { double* q; CK(hipMalloc(&q, 40 * 64 * 8)); P.p[k] = q; } }
      else if (mode == 1) { CK(hipMalloc(&slab, 16 * 40 * 64 * 8)); for (int k = 0; k < nb; ++k) P.p[k] = slab + k * 40 * 64; }
      else { CK(hipMalloc(&slab, (size_t)16 * (96 << 20))); for (int k = 0; k < nb; ++k) P.p[k] = slab + (size_t)k * (12 << 20); }  // 96 MB apart
      long long acc[2] = {0, 0};
      const int reps = 20;
      for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(stream, dim3(2048), dim3(256), 0, 0, big, NS, out);
        hipLaunchKernelGGL(writer, dim3(40), dim3(64), 0, 0, P, nb);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, P, nb, ticks, out);
        long long h[2]; CK(hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost));
        if (r >= 2) { acc[0] += h[0]; acc[1] += h[1]; }
      }
      printf("mode %d (%s) buffers=%2d: round1 %6lld cycles, round2 %6lld cycles\n", mode,
             mode == 0 ? "separate hipMalloc" : mode == 1 ? "one small slab" : "96 MB apart in one 1.5 GB alloc", nb, acc[0] / (reps - 2), acc[1] / (reps - 2));
    }
  }
  return 0;
}

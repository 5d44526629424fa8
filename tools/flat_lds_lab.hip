// Measurement, not product code: latency of a DEPENDENT load on one wave of an MI355X -- LDS through ds_read, LDS through a flat
// (generic-pointer) load, global memory that sits in L2 -- in shader-clock cycles per load (a pointer chase of 4096 steps).
// Why: the single-workgroup NUTS kernel (csrc/small_kernel.h) interprets a model's factors through chains of such loads.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/flat_lds_lab.hip -o /tmp/flat_lds_lab && /tmp/flat_lds_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return (long long)t;
}

__global__ void k_chase(const int* __restrict__ next_g, long long* out, int steps, const int* opaque_zero) {
  __shared__ int s_next[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_next[i] = next_g[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  int p = 0;
  long long t0 = now();
  for (int i = 0; i < steps; ++i) p = s_next[p];                      // ds_read
  long long t1 = now();
  const int* gen = opaque_zero[0] ? next_g : (const int*)s_next;      // a pointer the compiler cannot classify: flat loads
  int q = p & 0;
  for (int i = 0; i < steps; ++i) q = gen[q];
  long long t2 = now();
  int r = q & 0;
  for (int i = 0; i < steps; ++i) r = next_g[r];                      // global (L2 / L1 resident after the copy above)
  long long t3 = now();
  out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = p + q + r;
}

int main() {
  std::vector<int> nxt(4096);
  for (int i = 0; i < 4096; ++i) nxt[i] = (i * 1237 + 811) & 4095;
  int *d_next, *d_zero; long long* d_out;
  CHECK(hipMalloc(&d_next, 4096 * 4)); CHECK(hipMalloc(&d_zero, 4)); CHECK(hipMalloc(&d_out, 4 * 8));
  CHECK(hipMemcpy(d_next, nxt.data(), 4096 * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(d_zero, 0, 4));
  long long h[4];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d_next, d_out, 4096, d_zero);
    CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("cycles per dependent load: ds_read %.1f   flat -> LDS %.1f   global (cache-resident) %.1f\n", h[0] / 4096.0, h[1] / 4096.0, h[2] / 4096.0);
  }
  return 0;
}

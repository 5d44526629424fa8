// What does the vector pipe deliver in fp64 on this box?  (round 6: the rows chain group is bound by it, csrc/rows_gal_kernel.h)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fp64_rate_lab.hip -o gpurun_out/fp64_rate_lab && gpurun_out/fp64_rate_lab
// Kernels of ILP independent dependent-chains per lane, WPS waves per SIMD (grid = 256 CUs x 4 SIMDs x WPS waves), ITER steps:
//   fma   : x = fma(x, a, b)                         (v_fma_f64, two VGPR + ... operands)
//   fmas  : the same with a, b in scalar registers    (v_fma_f64 with SGPR operands, as the row pass reads beta / literals)
//   mix   : 3 fma + 1 add + 1 mul per step            (the row pass's mix)
//   rcp   : v_rcp_f64 + 2 fma (one Newton step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int ILP, int MODE>
__global__ __launch_bounds__(64) void k(double* out, int iters, double a_in, double b_in) {
  double x[ILP];
  const double a = MODE == 1 ? __longlong_as_double(((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(a_in) >> 32)) << 32) |
                                                    (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(a_in))) : a_in + 1e-18 * threadIdx.x;
  const double b = MODE == 1 ? b_in : b_in + 1e-18 * threadIdx.x;
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (MODE <= 1) x[i] = fma(x[i], a, b);
      else if (MODE == 2) { double t = fma(x[i], a, b); t = fma(t, a, x[i]); t = fma(t, b, a); x[i] = (t + b) * a; }
      else { double y = __builtin_amdgcn_rcp(x[i]); double e = fma(-x[i], y, 1.0); x[i] = fma(y, e, y) + 1.0; }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  if (s == 12345.678) out[threadIdx.x] = s;
}

template <int ILP, int MODE>
void run(const char* name, int wps, int ops_per_step, double* out) {
  const int iters = 20000, grid = 256 * 4 * wps;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<ILP, MODE>), dim3(grid), dim3(64), 0, 0, out, 200, 0.999999, 1e-6);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<ILP, MODE>), dim3(grid), dim3(64), 0, 0, out, iters, 0.999999, 1e-6);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double winstr = (double)grid * iters * ILP * ops_per_step;           // wave instructions
  const double per_simd_cycles_at_2400 = ms * 1e-3 * 2.4e9 / (winstr / 1024.0);
  printf("%-5s ILP %d  %d waves/SIMD: %.3f ms, %.2f G wave-instr/s, %.2f cycles of a 2.4 GHz SIMD per wave instruction, %.1f TFLOP/s if every one were an fma\n", name, ILP, wps, ms,
         winstr / (ms * 1e-3) / 1e9, per_simd_cycles_at_2400, winstr * 64 * 2 / (ms * 1e-3) / 1e12);
}

int main() {
  double* out; CK(hipMalloc(&out, 4096));
  for (int wps : {1, 2, 4, 5, 8}) {
    run<8, 0>("fma", wps, 1, out);
    run<8, 1>("fmas", wps, 1, out);
    run<4, 2>("mix", wps, 5, out);
    run<4, 3>("rcp", wps, 4, out);
  }
  run<1, 0>("fma", 1, 1, out);   // latency of one dependent chain
  run<2, 0>("fma", 1, 1, out);
  run<4, 0>("fma", 1, 1, out);
  return 0;
}

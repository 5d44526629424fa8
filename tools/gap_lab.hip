// Idle-gap experiment (not product code): what does the GPU do between two dependent launches of the C2-L row
// pass?  rocprofv3's kernel trace of the bench shows ~2-3 us between the end of one k_rows_ga and the start of the next (0.6-0.85
// us for the cache-resident passes).  Candidates, one knob each, every variant = 400 back-to-back launches on one stream, the gap
// read from `rocprofv3 --kernel-trace` (tools/rocpd_summary.py prints mean and median gap per kernel name):
//   KARG   bytes of by-value kernel arguments (GaArgs is ~1.6 KB; does the CP's kernarg fetch show?)
//   LDS    static LDS per workgroup (22 KB in k_rows_ga)
//   MODE   0 empty | 1 streams 256 KB per workgroup with dwordx4 loads (320 MB per launch: every L2 full of clean lines, HBM busy
//          until the last wave retires) | 2 = 1 + a write-through record and an agent-scope ticket per workgroup at the end
//          | 3 = 1 but the data is only 8 MB (cache-resident stream: same instruction mix, no HBM tail)
//   GRID   1249 x 192 threads (C2-L: every workgroup resident at once) or 157 x 512 (C2-S)
// usage: hipcc --offload-arch=gfx950 -O3 tools/gap_lab.hip -o /tmp/gap_lab && rocprofv3 --kernel-trace -d out -o gap -- /tmp/gap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int BYTES> struct Karg { const double* x; double* out; unsigned* ticket; size_t per_wg; size_t wrap; char pad[BYTES]; };

template <int KARG, int LDS, int MODE>
__global__ __launch_bounds__(512) void gap_kernel(Karg<KARG> a) {
  __shared__ char s_lds[LDS > 0 ? LDS : 16];
  if (LDS > 0 && threadIdx.x == 0) s_lds[a.per_wg & 15] = 1;
  if (MODE == 0) { if (a.per_wg == 12345 && a.pad[KARG - 1]) a.out[0] = s_lds[0]; return; }
  const size_t n = a.per_wg;                                        // doubles per workgroup
  const double2* src = reinterpret_cast<const double2*>(a.x + ((size_t)blockIdx.x * n) % a.wrap);
  double s = 0.0;
  for (size_t i = threadIdx.x; i < n / 2; i += blockDim.x) { const double2 v = src[i]; s += v.x + v.y; }
  if (MODE == 2) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.out + 16 + blockIdx.x * 104 + (threadIdx.x & 63)), (unsigned long long)__double_as_longlong(s),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(&a.ticket[blockIdx.x / 39], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old % 39 == 38) a.out[8] = s;
    }
  } else if (s == 123.456) a.out[0] = s + s_lds[1] + a.pad[KARG - 1];
}

template <int KARG, int LDS, int MODE>
static void run(const char* what, int grid, int block, const double* x, double* out, unsigned* ticket, size_t per_wg, size_t wrap, int reps = 400) {
  Karg<KARG> a{};
  a.x = x; a.out = out; a.ticket = ticket; a.per_wg = per_wg; a.wrap = wrap;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((gap_kernel<KARG, LDS, MODE>), dim3(grid), dim3(block), 0, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gap_kernel<KARG, LDS, MODE>), dim3(grid), dim3(block), 0, 0, a);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("gap_kernel<%4d, %5d, %d> grid %4d x %3d  %-58s %8.2f us per launch (launch + gap, HIP events)\n", KARG, LDS, MODE, grid, block, what, 1e3 * ms / reps);
}

int main() {
  const size_t PER = 32768;                       // doubles per workgroup: 256 KB
  const size_t TOT = (size_t)1249 * PER;          // 327 MB
  double* x; CK(hipMalloc(&x, TOT * 8)); CK(hipMemset(x, 0, TOT * 8));
  double* out; CK(hipMalloc(&out, (16 + 1249 * 104 + 64) * 8)); CK(hipMemset(out, 0, (16 + 1249 * 104 + 64) * 8));
  unsigned* ticket; CK(hipMalloc(&ticket, 64 * 4)); CK(hipMemset(ticket, 0, 64 * 4));
  run<16, 0, 0>("empty, 56 B of arguments", 1249, 192, x, out, ticket, PER, TOT);
  run<1600, 0, 0>("empty, 1.6 KB of arguments", 1249, 192, x, out, ticket, PER, TOT);
  run<1600, 22080, 0>("empty, 1.6 KB of arguments, 22 KB LDS", 1249, 192, x, out, ticket, PER, TOT);
  run<16, 0, 1>("streams 327 MB", 1249, 192, x, out, ticket, PER, TOT);
  run<1600, 22080, 1>("streams 327 MB, 1.6 KB of arguments, 22 KB LDS", 1249, 192, x, out, ticket, PER, TOT);
  run<1600, 22080, 2>("the same + write-through record + ticket per workgroup", 1249, 192, x, out, ticket, PER, TOT);
  run<1600, 22080, 3>("streams the same 8 MB over and over (cache-resident)", 1249, 192, x, out, ticket, PER, (size_t)1 << 20);
  run<16, 0, 0>("empty, C2-S grid", 157, 512, x, out, ticket, PER, TOT);
  run<1600, 28824, 3>("cache-resident stream, C2-S grid", 157, 512, x, out, ticket, PER / 4, (size_t)1 << 20);
  return 0;
}

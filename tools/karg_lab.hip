// Does the runtime take a kernel-argument segment of more than 4 KiB?  (round 6: the merged launch of the group-block pass would like to carry eight chains there)
//   hipcc --offload-arch=gfx950 -O2 tools/karg_lab.hip -o /tmp/karg_lab && /tmp/karg_lab
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double v[700]; };   // 5600 bytes
__global__ void k(Big b, double* out) { out[threadIdx.x] = b.v[threadIdx.x] + b.v[699 - threadIdx.x]; }
int main() {
  Big b; for (int i = 0; i < 700; ++i) b.v[i] = i;
  double* d; hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipDeviceSynchronize();
  double h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("err %s  h[3] = %g (expect 699)\n", hipGetErrorString(e), h[3]);
  return 0;
}

// Chain groups on the group-block row pass (rows_gb_kernel.h): the merged launch.
#pragma once
#include "rows_gb_kernel.h"
#include "rows_gal_kernel.h"

// ---- chain groups on the group-block pass (round 6; VERDICT r05 "next" 3: C2-S chains ran as independent engines) -----------------
// The pass is latency-bound and cache-resident (12.8 us per launch for 8.3 MB at C2-S, 156 workgroups on 256 CUs), so a merged launch
// does not share the stream of X -- it shares the LAUNCH: grid = (chains standing at a leaf) x (1 control slot + ga_nblk row
// workgroups), every workgroup runs the single-chain body on ITS chain's arguments, assembled in LDS from the base member's model
// (kernel argument), the chain's constant part (device memory, rows_gal_kernel.h GalConst) and its part of this launch.  The body
// is the single-chain kernel's own code: a chain in a group is bitwise the chain alone by construction.
struct GbmArgs { GalLeaf c[GAL_MAXC]; int nc, rev; };

template <int D, int DX = D>
__global__ __launch_bounds__(64 * GB_W) void k_rows_gb_multi(ModelDev md, const GalConst* __restrict__ konst, GbmArgs ma) {
  __shared__ __attribute__((aligned(16))) GaArgs s_a;
  const int per = md.lg.ga_nblk + 1;
  const int c = (int)blockIdx.x / per, b = (int)blockIdx.x - c * per;
  {
    // the base member's model -> LDS (1.3 KB), then the fields that are the chain's own
    static_assert(offsetof(GaArgs, md) == 0 && sizeof(ModelDev) % 8 == 0, "GaArgs starts with the model");
    const uint2* src = (const uint2*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int t = threadIdx.x; t < (int)(sizeof(ModelDev) / 8); t += (int)blockDim.x) reinterpret_cast<uint2*>(&s_a)[t] = src[t];
    __syncthreads();
    if (threadIdx.x == 0) {
      const GalLeaf& l = ma.c[c];
      const GalConst& k = konst[l.slot];
      s_a.md.lg.ga_part = k.ga_part; s_a.md.lg.ga_bpart = k.ga_bpart; s_a.md.lg.ga_ticket = k.ga_ticket; s_a.md.def_loc = k.def_loc;
      s_a.A = k.A; s_a.A.uniforms = l.uniforms; s_a.A.log_uniforms = l.log_uniforms;
      s_a.io = l.io; s_a.cio = l.cio;
      s_a.j = l.j; s_a.rev = ma.rev; s_a.fold = l.fold; s_a.par = l.par; s_a.d = l.d; s_a.max_depth = k.max_depth;
      s_a.Emax = k.Emax; s_a.st = k.st; s_a.cj = l.cj; s_a.cd = l.cd; s_a.cseq = l.cseq; s_a.cpad = 0;
    }
    __syncthreads();
  }
  gb_body<D, DX>(s_a, b, nullptr);
}

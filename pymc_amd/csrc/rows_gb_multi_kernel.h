// Chain groups on the group-block row pass (rows_gb_kernel.h): the merged launch.
#pragma once
#include "rows_gb_kernel.h"
#include "rows_gal_kernel.h"

// ---- chain groups on the group-block pass (round 6; VERDICT r05 "next" 3: C2-S chains ran as independent engines) -----------------
// The pass is latency-bound and cache-resident (12.8 us per launch for 8.3 MB at C2-S, 156 workgroups on 256 CUs), so a merged launch
// does not share the stream of X -- it shares the LAUNCH: grid = (chains standing at a leaf) x (1 control slot + ga_nblk row
// workgroups), every workgroup runs the single-chain body (rows_gb_kernel.h gb_body) on ITS chain's arguments and the base member's
// model.  The body is the single-chain kernel's own code: a chain in a group is bitwise the chain alone by construction.
// Everything a chain brings travels in the KERNEL ARGUMENTS -- its constant part too (the LDS-shared launch of the group-aligned
// pass reads that from device memory): the body was tuned with its arguments in scalar registers, and a first version that assembled
// a chain's argument block in LDS ran 28 us for four chains against 11.8 us for one (every field a `ds_read`, every pointer a vector
// register; profiles/r06i_*).  Here a workgroup selects its chain's blocks by a wave-uniform index into the kernarg segment: scalar
// loads, as in a launch of the chain's own.
struct GbmArgs { GalLeaf c[GAL_MAXC]; GalConst k[GAL_MAXC]; int nc, rev; };

// OCC: waves per SIMD the register budget is sized for.  The single-chain kernel takes 175 registers (two waves per SIMD: ONE
// workgroup of eight waves per CU -- enough for its 157 workgroups); a merged launch of c chains has 157 c workgroups and runs in
// ceil(157 c / (256 x workgroups per CU)) rounds, so it is built for more workgroups per CU at the price of a few spilled registers.
template <int D, int DX, int OCC>
__global__ __launch_bounds__(64 * GB_W, OCC) void k_rows_gb_multi(ModelDev md, GbmArgs ma) {
  const int per = md.lg.ga_nblk + 1;
  const int c = __builtin_amdgcn_readfirstlane((int)blockIdx.x / per), b = (int)blockIdx.x - c * per;
  const GalLeaf& l = ma.c[c];
  const GalConst& k = ma.k[c];
  ArenaDev A = k.A;
  A.uniforms = l.uniforms; A.log_uniforms = l.log_uniforms;
  const GbChain ch{A, l.io, l.cio, l.j, l.fold, l.par, l.d, k.max_depth, l.cj, l.cd, l.cseq, k.Emax, k.st, k.ga_bpart, k.def_loc};
  gb_body<D, DX>(md, ch, b, nullptr);
}

// Lockstep chains of ONE hierarchical-logit model on the group-aligned row pass, tiles shared through LDS (round 6).
//
// `k_rows_ga_multi` (rows_ga_multi_kernel.h, round 5) gave every wave of workgroup g a third of group g's tiles and ALL chains: NC
// coefficient vectors in scalar registers, NC sets of gradient accumulators in vector registers.  That made the launch fp64-issue
// bound at 168 registers per lane (three waves per SIMD: only 1024 of the benchmark's 1248 workgroups resident, one tile buffer with
// four chains, every tile's memory latency exposed to the wave that needs it): 106 us for four chains at C2-L against 53 us for one
// -- and no room for the benchmark's eight.  Here the roles are turned round:
//
//   * workgroup g has ONE WAVE PER CHAIN.  Every wave evaluates every tile of group g under its own chain's coefficient vector:
//     the registers of the single-chain kernel minus its second tile buffer, whatever the number of chains -- five workgroups per CU
//     at four chains, i.e. every group of the benchmark resident at once.
//   * the tiles cross HBM -> LDS once per workgroup by LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per wave instruction, no staging
//     registers; MI355X guide, cdna_hip_programming.md "Async global -> LDS copy"): the waves share the requests of a tile (a column of
//     X = one request), a ring of GAL_RING tile slots, tiles requested GAL_PF ahead.  One barrier per tile; the requests are issued
//     and counted BY HAND (`s_waitcnt vmcnt(N)` with N = the requests of the younger tiles) because hipcc drains the queue at every
//     barrier and before every LDS read otherwise (same guide, "Pipelining across barriers").  Nothing else in the loop touches
//     memory: no scratch, no compiler-counted loads.
//   * a chain's numbers are the single-chain kernel's: wave c walks the W chunks the model's layout gave to W waves, each chunk in
//     its two halves in tile order, the same lane for the same rows (`ga_tile`), one wave sum per half in `wave_sum`'s association
//     order, and the tail combines the 2 W half sums in the same order -- BITWISE the chain alone (tests/test_gpu_chain_group.py),
//     whatever the company.
//
// Prologue, tail, records, tickets and block partials are those of rows_ga_multi_kernel.h on the chain's own buffers; the chain's
// wave keeps its five per-lane prologue values in registers (the old kernel parked them in LDS for another wave to pick up).
// Arguments: a chain's constant part (arena, record and ticket pointers) lives in device memory, uploaded when the chain is first
// seen; the per-launch part travels in the kernarg segment -- eight chains' complete arguments would not fit its 4 KiB.
#pragma once
#include "rows_ga_multi_kernel.h"

#define GAL_MAXC GAM_MAXC   // chains per launch (BASELINE configs[1]: eight chains); the single-chain launches of a group go through k_rows_ga_multi<1>, same grid
#define GAL_PF 2            // tiles requested ahead of the one being evaluated
#define GAL_RING (GAL_PF + 1)

struct GalConst {           // what a member chain keeps for its whole life
  ArenaDev A;               // (uniforms / log_uniforms travel per launch)
  double Emax;
  HostStatus* st;
  double* ga_part; double* ga_bpart; unsigned* ga_ticket; double* def_loc;
  int max_depth, slot;
};
struct GalLeaf {            // one chain's part of ONE launch
  EvalIO io, cio;
  const double* uniforms; const double* log_uniforms;
  int j, fold, par, d, cj, cd, cseq, slot;
};
template <int NC>
struct GalArgs {
  GalLeaf c[NC];
  int rev, pad;
};

__device__ __forceinline__ void gal_assemble(GaLeafArgs& L, const GalConst& k, const GalLeaf& l) {
  L.A = k.A; L.A.uniforms = l.uniforms; L.A.log_uniforms = l.log_uniforms;
  L.io = l.io; L.cio = l.cio; L.Emax = k.Emax; L.st = k.st;
  L.ga_part = k.ga_part; L.ga_bpart = k.ga_bpart; L.ga_ticket = k.ga_ticket; L.def_loc = k.def_loc;
  L.j = l.j; L.fold = l.fold; L.par = l.par; L.d = l.d; L.max_depth = k.max_depth; L.cj = l.cj; L.cd = l.cd; L.cseq = l.cseq;
  L.slot = l.slot; L.pad = 0;
}

// One 16-byte-per-lane request HBM -> LDS: `gsrc` each lane's source, `lds_dst` the wave-uniform LDS byte address of lane 0's 16 bytes
// (lane l lands at lds_dst + 16 l).  M0 is written in the statement that reads it (the compiler does not preserve it around asm).
__device__ __forceinline__ void gal_dma16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void gal_dma4(const void* gsrc, uint32_t lds_dst) {    // 4 bytes per lane (lane l lands at lds_dst + 4 l)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void gal_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
__device__ __forceinline__ void gal_barrier() { asm volatile("s_barrier" ::: "memory"); }

// The tail wave of chain `L` in workgroup g (gam_tail_wave with the prologue values in registers).
template <int D>
__device__ __forceinline__ void gal_tail_wave(const ModelDev& md, const GaLeafArgs& L, int g, int W, const double (&s_acc)[GA_MAXW][2][D + 1],
                                              double* s_red, int* s_info, double hval, double hph, double zq, double zph, double s_lane,
                                              const MergePrefetch& mpf) {
  const RowsDev& R = md.lg;
  const ArenaDev& A = L.A;
  const int lane = threadIdx.x & (WAVE - 1);
  const int j = L.j, par = L.par, d = L.d;
  Leaf lf; QView qv;
  resolve_leaf(L.io, A, j, lf, qv);
  const int dl = lane % D;
  const int iz = R.off_z + g * D + dl;
  double db = 0.0, lpg = 0.0;
  for (int ww = 0; ww < W; ++ww) { db += s_acc[ww][0][dl] + s_acc[ww][1][dl]; lpg += s_acc[ww][0][D] + s_acc[ww][1][D]; }
  const bool zact = lane < D;
  int idx[1] = {iz};
  bool act[1] = {zact};
  double grad[1] = {0.0}, ph[1] = {zph};
  {
    const double r = zq - R.z_np_mu;                       // z ~ Normal(mu0, s0) in closed form (continuous.py:526-532)
    const double gx = -r * R.z_np_inv_var;
    const double lpz = -0.5 * r * r * R.z_np_inv_var - R.z_np_lognorm;
    grad[0] = gx + s_lane * db;                            // d/dz = prior + sigma_d * d/dbeta_d
    lpg += wave_sum(zact ? lpz : 0.0);
    if (zact) { A.G[lf.d_o + iz] = grad[0]; A.Q[lf.d_o + iz] = zq; }
  }
  if (g == 0) {   // the hyper-parameter elements' local parts + their q' (one workgroup does it for the launch)
    const int e = lane;
    const bool hact = e < 2 * D, is_mu = e < D;
    double gx, dxdq, dj, lpd;
    ga_def_local(R, is_mu, hval, gx, dxdq, dj, lpd);
    lpg += wave_sum(hact ? lpd : 0.0);
    if (hact) {
      const int dd = is_mu ? e : e - D;
      const int slot = (is_mu ? R.def_mu : R.def_sigma) + dd;
      double2* loc = reinterpret_cast<double2*>(L.def_loc + (int64_t)par * 4 * MAX_DEFERRED) + 2 * slot;
      loc[0] = make_double2(gx, dxdq);
      loc[1] = make_double2(dj, hph);
      A.Q[lf.d_o + (is_mu ? R.off_mu : R.off_sigma) + dd] = hval;
    }
  }
  int m = 0; bool last = false;
  leaf_post<1>(A, lf, j, d, true, idx, act, grad, ph, s_red, 1, m, last, &mpf, 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- this group's record, write-through ----
  double* rec = L.ga_part + (int64_t)g * PART_STRIDE;
  if (lane == 0) st_agent(rec + PART_LP, lpg);
  if (zact) { st_agent(rec + PART_DMU + lane, db); st_agent(rec + PART_DSG + lane, db * zq); }
  for (int k = lane; k < NDOT; k += WAVE)
    if (dot_needed(k, m, last)) st_agent(rec + PART_DOT + k, s_red[k]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record has left this CU before the ticket is taken

  // ---- ticket: the block's last arriver sums the block's records in group order ----
  const int blk = g / R.ga_bsz;
  const int cnt = min(R.G, (blk + 1) * R.ga_bsz) - blk * R.ga_bsz;
  unsigned old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(&L.ga_ticket[blk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
  const int is_last = (int)old + 1 == cnt;
  if (is_last && lane == 0) __hip_atomic_store(&L.ga_ticket[blk], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) { s_info[0] = is_last; s_info[1] = m; s_info[2] = last ? 1 : 0; }
}

// D = 8 covariates, two rows per lane; DX stored columns (7: the intercept column is not stored).  Grid: GAL_MAXC control workgroups +
// G group workgroups; block: NC waves, wave c = chain c of this launch.  OCC: waves per SIMD the register budget is sized for.
template <int NC, int DX, int OCC, int PF = GAL_PF>
__global__ __launch_bounds__(64 * NC, OCC) void k_rows_gal(ModelDev md, const GalConst* __restrict__ konst, GalArgs<NC> la) {
  constexpr int D = 8, SPAN = WAVE * 2, RING = PF + 1;
  constexpr int ITEMS = DX + 1;                              // requests of a tile: DX columns of 1 KiB + the 128 y bytes (as 64 x 4 B)
  constexpr int LPT = (ITEMS + NC - 1) / NC;                 // ... per wave (the same count in every wave: the waits are immediates)
  constexpr int SLOT = DX * 1024 + 256;                      // bytes of a ring slot
  constexpr int64_t TS = (int64_t)DX * SPAN;                 // doubles per tile
  const RowsDev& R = md.lg;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = __builtin_amdgcn_readfirstlane(tid >> 6);
  __shared__ GaLeafArgs s_L[NC];
  // the tile ring; after the stream the same bytes hold the tails' dot products and the block reduce's chunk buffer; in a control
  // workgroup they are the control work's LDS
  constexpr int TAILB = NC * NDOT * 8 + GA_MAXCHUNK * PART_STRIDE * 8;
  constexpr int RINGB = RING * SLOT > TAILB ? RING * SLOT : TAILB;
  static_assert(sizeof(CtlLds) <= RINGB, "the control work's LDS is lent from the tile ring");
  __shared__ __attribute__((aligned(16))) char s_ring[RINGB];
  if ((int)blockIdx.x < GAL_MAXC) {   // control workgroups: chain `slot`'s folded control work in workgroup `slot`
    int ci = -1;
#pragma unroll
    for (int c = 0; c < NC; ++c) ci = (int)blockIdx.x == la.c[c].slot ? c : ci;
    if (ci < 0) return;
    if (tid == 0) gal_assemble(s_L[0], konst[la.c[ci].slot], la.c[ci]);
    __syncthreads();
    const GaLeafArgs& L = s_L[0];
    if (L.fold & GA_FOLD_CTL) {
      UniPrefetch upf;
      uni_prefetch_none(upf);
      control_lean_in<false, 8, false>(md, L.A, L.cio, L.cj, L.cd, L.Emax, L.max_depth, L.st, L.cseq, gam_src(R, L, L.par ^ 1),
                                       NC * WAVE > VEC_THREADS ? VEC_THREADS : 0, false, upf, *reinterpret_cast<CtlLds*>(s_ring));
    }
    return;
  }
  const int g = (int)blockIdx.x - GAL_MAXC;
  const int W = R.ga_w;                                      // the chunks of the model's layout: one per wave of the single-chain kernel
  __shared__ double s_acc[NC][GA_MAXW][2][D + 1];            // [chain][chunk][first / second half of its tiles][d/dbeta, log-lik]
  __shared__ int s_info[NC][4];
  __shared__ int s_dead[NC];
  __shared__ double s_keep[NC][5][2 * D];                    // the prologue values the chain's tail needs again (lanes < 2 D), out of the stream's registers
  const uint32_t ring0 = (uint32_t)(uintptr_t)(&s_ring[0]);   // (LDS addresses are 32-bit offsets)

  // ---- the chains' arguments -> LDS ----
  {
    GaLeafArgs* Lw = &s_L[w];
    if (lane == 0) gal_assemble(*Lw, konst[la.c[w].slot], la.c[w]);
  }
  __syncthreads();
  const GaLeafArgs& L = s_L[w];
  const int dead = load_aborted(L.io, L.A);                  // this chain's tree has ended: its wave only helps with the tiles
  if (lane == 0) {
    s_dead[w] = dead;
#pragma unroll
    for (int ww = 0; ww < GA_MAXW; ++ww)
#pragma unroll
      for (int dd = 0; dd <= D; ++dd) { s_acc[w][ww][0][dd] = 0.0; s_acc[w][ww][1][dd] = 0.0; }
  }

  // ---- geometry: the group's tiles as the sequence (chunk 0: first-streamed half, second half; chunk 1: ...) ----
  int T; int64_t ng;
  if (R.ga_T_uni > 0) { T = R.ga_T_uni; ng = R.ga_ng_uni; }
  else {
    T = __builtin_amdgcn_readfirstlane(R.ga_tile0[g + 1] - R.ga_tile0[g]);
    ng = __builtin_amdgcn_readfirstlane((int)(R.gptr[g + 1] - R.gptr[g]));
  }
  const int rev = la.rev;
  const int n_last = (int)(ng - (int64_t)(T - 1) * SPAN);
  auto chunk_base = [&](int ww) -> int64_t {
    if (R.ga_T_uni > 0) return (int64_t)(g * W + ww) * R.ga_cstride_uni;
    const int64_t cb = R.ga_coff[g * W + ww];
    return ((int64_t)__builtin_amdgcn_readfirstlane((int)(cb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(cb & 0xffffffffll));
  };
  // iterator over the sequence: (ww, pos) -> element offset of the tile in Xt; advanced by one tile per call
  struct It { int ww, pos, n, nA, nsw, has_last; int64_t cbase; };
  auto it_chunk = [&](It& it, int ww) {
    it.ww = ww; it.pos = 0;
    // (k_rows_ga's `(int64_t)w * T / W`; W <= 4 and T < 2^24 tiles: the products fit 32 bits, and a 64-bit division is ~140 instructions)
    const int c0 = (int)((unsigned)ww * (unsigned)T / (unsigned)W), c2 = (int)((unsigned)(ww + 1) * (unsigned)T / (unsigned)W);
    it.n = c2 - c0; it.nA = (it.n + 1) / 2; it.nsw = rev ? it.n - it.nA : it.nA;
    it.has_last = c2 == T;                       // the group's last (zero-padded) tile is this chunk's last
    it.cbase = ww < W ? chunk_base(ww) : 0;
  };
  auto it_skip_empty = [&](It& it) { while (it.ww < W && it.pos >= it.n) it_chunk(it, it.ww + 1); };
  auto it_local = [&](const It& it) { return rev ? (it.pos < it.nsw ? it.nA + it.pos : it.pos - it.nsw) : it.pos; };

  // requests of one tile, this wave's share: items k = w, w + NC, ...; item DX = the y bytes; past the end: item 0 once more (the
  // same bytes to the same place: every wave issues exactly LPT requests per tile)
  const uint32_t l16 = (uint32_t)lane * 16u;
  auto request = [&](const It& it, int slot) {
    const int64_t off = it.cbase + (int64_t)it_local(it) * TS;
    const double* xt = R.Xt + off;
    const int8_t* yt = R.y + off / DX;
    const uint32_t dst = ring0 + (uint32_t)slot * SLOT;
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      int k = w + i * NC;
      k = k < ITEMS ? k : 0;
      if (k < DX) gal_dma16(reinterpret_cast<const char*>(xt + (int64_t)k * SPAN) + l16, dst + (uint32_t)k * 1024u);
      else gal_dma4(yt + 4 * (lane & 31), dst + (uint32_t)DX * 1024u);     // (lanes 32 .. 63 re-read the tile's own bytes: nothing past its end is touched)
    }
  };

  It ld; it_chunk(ld, 0); it_skip_empty(ld);     // the tile to request next
  It ev = ld;                                    // the tile to evaluate next
  It lastreq = ld;                               // (past the end the last tile is requested again: the counts stay what the waits assume)
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    if (ld.ww < W) { lastreq = ld; request(ld, p % RING); ld.pos++; it_skip_empty(ld); }
    else request(lastreq, p % RING);
  }

  // ---- prologue of this wave's chain: mu', sigma' of its leaf, z' of this group, beta_g (the first tiles are in flight) ----
  double beta[D];
  {
    double hval0 = 0.0, hph0 = 0.0, zq = 0.0, zph = 0.0, s_lane = 0.0;
    Leaf lf; QView qv;
    resolve_leaf(L.io, L.A, L.j, lf, qv);
    double bl = 0.0;
    if (!dead) {
      gam_hyper<D>(R, L, qv, lane, hval0, hph0);
      const int dl = lane % D;
      const int iz = R.off_z + g * D + dl;
      if (qv.composed) { zph = fma(qv.half, qv.g[iz], qv.p[iz]); zq = fma(qv.eps, qv.var[iz] * zph, qv.q[iz]); }
      else { zph = 0.0; zq = qv.q[iz]; }
      const double m_lane = __shfl(hval0, dl);
      const double sraw = __shfl(hval0, D + dl);
      s_lane = R.sigma_tr == NUTS_TR_LOG ? exp(sraw) : sraw;
      bl = fma(s_lane, zq, m_lane);
    }
#pragma unroll
    for (int dd = 0; dd < D; ++dd) beta[dd] = readlane_d(bl, dd);
    if (lane < 2 * D) { s_keep[w][0][lane] = hval0; s_keep[w][1][lane] = hph0; s_keep[w][2][lane] = zq; s_keep[w][3][lane] = zph; s_keep[w][4][lane] = s_lane; }
  }
  __syncthreads();   // (s_dead, s_acc; hipcc waits for the prologue's own loads here -- and for the first tiles with them, once per launch)
  int alldead = 1;
#pragma unroll
  for (int c = 0; c < NC; ++c) alldead &= s_dead[c];
  if (alldead) { gal_wait<0>(); return; }   // every chain's tree has ended: the launch drains (no tickets)

  // ---- the stream: every tile once through LDS, every wave its own chain ----
  {
    double acc[D], lp = 0.0;
#pragma unroll
    for (int dd = 0; dd < D; ++dd) acc[dd] = 0.0;
    auto flush = [&](int ww, int half) {
      int idx;
      const double t = wave_sum_many<D>(acc, lane, idx);
      if (lane < D && !dead) s_acc[w][ww][half][idx] = t;
      int idx2;
      const double lpa[1] = {lp};
      const double t2 = wave_sum_many<1>(lpa, lane, idx2);
      if (lane == 0 && !dead) s_acc[w][ww][half][D] = t2;
#pragma unroll
      for (int dd = 0; dd < D; ++dd) acc[dd] = 0.0;
      lp = 0.0;
    };
    for (int s = 0; s < T; ++s) {
      // tile s has landed when at most the requests of the PF - 1 younger tiles are outstanding -- in every wave
      gal_wait<LPT * (PF - 1)>();
      gal_barrier();
      // (behind the barrier every wave has finished tile s - 1: its slot takes tile s + PF)
      if (ld.ww < W) { lastreq = ld; request(ld, (s + PF) % RING); ld.pos++; it_skip_empty(ld); }
      else request(lastreq, (s + PF) % RING);
      const char* slot = s_ring + (s % RING) * SLOT;
      const uint32_t yy = *reinterpret_cast<const uint16_t*>(slot + DX * 1024 + lane * 2);
      const int local = it_local(ev);
      // the flush between the two halves of a chunk comes BEFORE the first tile of the second-streamed half (k_rows_ga: `if (I == nsw) flush()`)
      if (ev.pos == ev.nsw) flush(ev.ww, rev ? 1 : 0);
      const int nv = (ev.has_last && local == ev.n - 1) ? n_last : SPAN;
      if (!dead) {
        // ga_tile<8, 2>'s arithmetic, row by row the same operations in the same order -- with the tile read from LDS TWICE (forward:
        // eta; backward: d/dbeta) instead of held in 28 registers across the exp / log1p / reciprocal sequences, whose two rows run
        // side by side (logit_row2, rows_kernel.h): 96 registers, five waves per SIMD, two dependent chains in flight per wave
        auto col = [&](int k) { return *reinterpret_cast<const ga_v2d*>(slot + k * 1024 + lane * 16); };
        double eta[2] = {0.0, 0.0}, yk[2], l[2], rr[2];
        if constexpr (DX == 7) { eta[0] = fma(1.0, beta[0], eta[0]); eta[1] = fma(1.0, beta[0], eta[1]); }
#pragma unroll
        for (int k = 0; k < DX; ++k) {
          const ga_v2d v = col(k);
          eta[0] = fma(v.x, beta[k + (8 - DX)], eta[0]); eta[1] = fma(v.y, beta[k + (8 - DX)], eta[1]);
        }
        yk[0] = (double)(yy & 0xffu); yk[1] = (double)((yy >> 8) & 0xffu);
        asm volatile("" ::: "memory");   // (the columns are read AGAIN below: without this hipcc keeps the first reads alive in registers)
        logit_row2(eta, yk, l, rr);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const bool in = lane * 2 + k < nv;
          lp += in ? l[k] : 0.0;
          rr[k] = in ? rr[k] : 0.0;
        }
        if constexpr (DX == 7) { acc[0] = fma(rr[0], 1.0, acc[0]); acc[0] = fma(rr[1], 1.0, acc[0]); }
#pragma unroll
        for (int k = 0; k < DX; ++k) {
          const ga_v2d v = col(k);
          acc[k + (8 - DX)] = fma(rr[0], v.x, acc[k + (8 - DX)]); acc[k + (8 - DX)] = fma(rr[1], v.y, acc[k + (8 - DX)]);
        }
      }
      ev.pos++;
      if (ev.pos >= ev.n) {        // the chunk is done: its last-streamed half
        flush(ev.ww, (ev.pos > ev.nsw ? 1 : 0) ^ (rev ? 1 : 0));
        it_skip_empty(ev);
      }
    }
    gal_wait<0>();     // (the dummy requests past the end)
  }
  __syncthreads();     // every wave has left the ring: its bytes now serve the tails

  double* s_red = reinterpret_cast<double*>(s_ring) + w * NDOT;
  double (&s_cp)[GA_MAXCHUNK][PART_STRIDE] = *reinterpret_cast<double (*)[GA_MAXCHUNK][PART_STRIDE]>(s_ring + NC * NDOT * 8);
  if (!dead) {
    Leaf lf; QView qv;
    resolve_leaf(L.io, L.A, L.j, lf, qv);
    MergePrefetch mpf;
    merge_prefetch(L.A, lf, L.j, R.off_z + g * D + lane % D, mpf);
    // (lanes >= 2 D: values nobody looks at -- hval / hph feed the hyper-parameter elements' lanes, zq / zph / s_lane the D z lanes)
    const int kl = lane < 2 * D ? lane : 0;
    gal_tail_wave<D>(md, L, g, W, s_acc[w], s_red, s_info[w], s_keep[w][0][kl], s_keep[w][1][kl], s_keep[w][2][kl], s_keep[w][3][kl], s_keep[w][4][kl], mpf);
  } else if (lane == 0) s_info[w][0] = 0;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (!s_info[c][0]) continue;            // (workgroup-uniform)
    gam_block_partial<D>(R, s_L[c], g, s_info[c], s_cp);
    __syncthreads();
  }
}

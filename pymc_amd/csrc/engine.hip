// libnuts_mi355.so -- host side of the C ABI declared in include/nuts_mi355.h.
//
// One library stream per model; every kernel of a draw is enqueued on it (three launches per
// leapfrog, see kernels.h) and the host synchronises once per tree doubling (it only needs to
// know "stop or keep doubling").  All trajectory state stays in HBM for the whole chain; per
// draw the host uploads (q0, normals, uniforms) in one pinned copy and downloads
// (q, grad, stats).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <utility>
#include <ctime>
#include <string>
#include <vector>

#include "kernels.h"
#include "small_kernel.h"
#include "mixture_kernel.h"
#include "nuts_mi355.h"
#include "pcg64_stream.h"
#include "advi.h"

static thread_local std::string g_err;

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      g_err = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
      return NUTS_E_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define HIPCHK_NULL(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      g_err = std::string(#expr) + ": " + hipGetErrorString(_e);                            \
      return nullptr;                                                                       \
    }                                                                                       \
  } while (0)

// Schedule options (include/nuts_mi355.h, nuts_set_option): which of several equivalent kernel schedules a model / chain created
// from now on uses.  The library never reads the environment: whoever wants a non-default schedule (parity tests of every
// schedule, A/B measurements in tools/) says so through the ABI.  Process-wide, read when a model or chain is created.
static std::mutex g_opt_mu;
static std::vector<std::pair<std::string, int>> g_opts;
static int env_int(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  for (const auto& kv : g_opts) if (kv.first == name) return kv.second;
  return dflt;
}
extern "C" int nuts_set_option(const char* name, int32_t value) {
  if (!name || std::strncmp(name, "NUTS_", 5) != 0) { g_err = "option names start with NUTS_"; return NUTS_E_ARG; }
  std::lock_guard<std::mutex> lk(g_opt_mu);
  for (auto& kv : g_opts) if (kv.first == name) { kv.second = value; return NUTS_OK; }
  g_opts.emplace_back(name, value);
  return NUTS_OK;
}
extern "C" void nuts_clear_options(void) {
  std::lock_guard<std::mutex> lk(g_opt_mu);
  g_opts.clear();
}

template <typename T>
static T* dev_alloc(size_t count) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  // NUTS_POISON_ALLOC=1 (tests): every fresh allocation is filled with 0xFF bytes -- NaN as a double, 0xffffffff as a counter -- so
  // that a read of memory nobody wrote shows up in a process of any history (hipMalloc hands out zero pages in a fresh process and
  // whatever a freed block held in an old one)
  if (env_int("NUTS_POISON_ALLOC", 0)) { hipMemset(p, 0xFF, bytes); hipDeviceSynchronize(); }
  return static_cast<T*>(p);
}
template <typename T>
static T* dev_upload(const T* src, size_t count) {
  T* p = dev_alloc<T>(count);
  if (p && count) hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

#define PROF_RUN_MIN 16    // profiling: doublings of this many leaves or more are timed as ONE run of back-to-back launches
#define SMALL_MAX_ELEMS 16384   // ... and the most factor elements it walks per gradient
#define SMALL_MAX_N 1024  // largest model of the single-launch path (small_kernel.h: one thread per parameter, one workgroup)

// ===========================================================================
// model
// ===========================================================================
struct nuts_model {
  ModelDev md{};
  hipStream_t stream = nullptr;
  std::vector<void*> owned;  // device allocations
  double* q_dev = nullptr;   // [n] staging for nuts_model_logp_grad
  double* g_dev = nullptr;
  double* lp_dev = nullptr;
  double* host_pin = nullptr;  // pinned [2n+2]
  std::vector<nuts_data_ref> data_refs;   // (offset, size) of every data vector in the device pool
  int64_t data_epoch = 0;                 // bumped by nuts_model_set_data: start-state caches of older epochs are stale
  double* set_pin = nullptr; int64_t set_pin_len = 0; hipEvent_t set_ev = nullptr;   // nuts_model_set_data_many's staging buffer
  int rows_grid = 0, mvn_grid = 0, ept = 1;
  bool has_prog = false;   // some factor carries an expression program or a gathered operand: kernels with the interpreter compiled in
  int rows_rpl = 2, rows_alternate = 1, rows_flip = 0, rows_occ = 4;
  int vector_one_xcd = 0;
  int ga_variant = 42;         // 10 x (waves per SIMD of the register budget) + tiles in flight per wave
  int ga_struct_ok = 0;        // the spec is exactly what the group-aligned row pass evaluates in closed form (compile_spec)
  int ga_par = 0;              // parity of the last group-aligned launch (its block partials / local parts are double-buffered)
  unsigned* ga_sync = nullptr; // progress words of the persistent tree kernel (rows_ga_tree.h)
  int ga_tree_ok = 0;          // the whole grid of k_tree_ga is resident at once on this device (checked at model creation)
  long long* tree_dbg = nullptr; int tree_dbg_leaf = 0;   // NUTS_GA_TREE_DBG=<leaf + 1>: per-workgroup timeline of one leaf
  int64_t dom_units = 0;       // leapfrog passes covered by the timed launches (a tree launch covers a whole tree)
  int explicit_pre = 0;        // the position must be materialised before the dense pass (MvNormal node)
  int64_t alg_bytes = 0;
  // profiling of the dominant kernel
  bool profile = false;
  std::vector<hipEvent_t> ev;  // pairs
  std::vector<int32_t> ev_units;   // passes over the model data each pair covers (a single launch: 1; a bracketed run of launches: its length)
  int run_open = 0;            // a run of back-to-back launches (the leaves of one doubling) is being bracketed
  int64_t runs_seen = 0;
  size_t ev_used = 0;
  int64_t dom_launches = 0;
  int sample_every = 1;
  // chain group (lockstep chains of one MvNormal model, mvn_multi_kernel.h): while a member, `stream` is the group's stream
  hipStream_t own_stream = nullptr;
  struct nuts_group* group = nullptr;
  bool g_active = false;       // the model's chain is inside a tree (its leaf launches are deposited with the group)
  int gslot = 0;               // the model's place in the group
  int n_chains = 0;            // chains created on this model
  std::vector<int32_t> derived;   // NUTS_D_DERIVED factors (compile_spec)
  int64_t pool_extra = 0;         // doubles behind the spec's data pool: (values, seed) of every derived vector
  int64_t orphan_elems = 0, factor_elems = 0;   // elements of the factors without an owning variable / of all factors (compile_spec)
  int64_t rows_xt_len = 0, rows_y_len = 0;   // group-aligned row pass: elements of the tiled X / y copies (chain groups compare them)
  // linear predictors (dense node 5, lin_kernel.h): per (predictor, column) the factors that read it = (offset of their adjoints in
  // ModelDev.adj, factor size), collected by compile_spec; the device tables are built by build_lins once the data pool exists
  struct LinUse { int64_t adj_off; int32_t size; };
  std::vector<std::vector<std::vector<LinUse>>> lin_uses;
  std::vector<LinDev> lins_host;
  // what the resolved-operand sweep's tables are built from (build_sweep_fast): compile_spec's sweep lists, the data table with the
  // derived vectors behind the spec's, the broadcast terms
  std::vector<GSweepFactor> gsf_host;
  std::vector<GSlot> gslots_host;
  std::vector<nuts_data_ref> drefs_host;
  std::vector<FactorBT> fbt_host;
  std::vector<int32_t> bterm_var_host;

  template <typename T>
  T* keep(T* p) {
    owned.push_back((void*)p);
    return p;
  }
};

extern "C" int nuts_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" int nuts_set_device(int device) {
  HIPCHK(hipSetDevice(device));
  return NUTS_OK;
}
extern "C" const char* nuts_last_error(void) { return g_err.c_str(); }

// ---- chain group: chains of the same MvNormal model advance through ONE launch per leapfrog (mvn_multi_kernel.h) ----
// Every member chain is driven by its own host thread exactly as a chain alone (same calls, same order); all members submit to one
// in-order stream.  The only thing that changes is the leaf launch of the row-aligned pass inside a tree: the thread DEPOSITS its
// arguments and returns once the launch that carries them has been submitted -- by whichever thread completes the set of chains
// that are inside a tree.  A chain between two trees (finishing a draw, adapting, starting the next one) or past the end of its
// run is not waited for: the others go on without it, and it joins the next launch it is ready for.
#define GRP_MAXC MFM_MAXC        // members of a group: 4 on the plain-fma kernels, up to 16 on the matrix cores (mvn_mfma_kernel.h)
struct nuts_group {
  std::mutex mu;
  hipStream_t stream = nullptr;
  nuts_model* member[GRP_MAXC] = {};
  int n = 0;
  int nactive = 0, npend = 0;
  std::atomic<unsigned> gen{0};
  MvaLeafArgs pend[GRP_MAXC];
  int64_t launches[GRP_MAXC + 1] = {};   // submitted launches by the number of chains they carried
  // wide group (MvNormal models laid out 16 rows per workgroup): every merged launch goes through k_mvn_mfma_multi, whose chains'
  // arguments are read from a ring of blocks in pinned, device-visible host memory
  int cap = MVM_MAXC;
  ModelDev* md_dev = nullptr;                  // the base member's model in device memory (the control code's view of it)
  const nuts_model* md_of = nullptr;           // ... whose it is
  double* dpack = nullptr;                     // [k][16] q - mu of the chains of the launch being submitted (k_mfm_pack)
  MfmChainConst* konst_dev = nullptr;          // [GRP_MAXC] what a member chain keeps for its whole life (mvn_mfma_kernel.h)
  MfmChainConst konst_host[GRP_MAXC] = {};     // ... as last uploaded
  bool konst_set[GRP_MAXC] = {};
  // kind of the members' models: 1 = one MvNormal node on the row-aligned pass (mvn_multi_kernel.h), 2 = the hierarchical-logit rows
  // on the group-aligned pass (rows_ga_multi_kernel.h); fixed by the first member
  int kind = 0;                                // (3: the hierarchical-logit rows on the group-BLOCK pass, rows_gb_multi_kernel.h -- round 6)
  GaLeafArgs gpend[GAL_MAXC];
  int rows_flip = 0;
  // kind 2, launches that carry two chains or more (rows_gal_kernel.h): a member chain's constant arguments in device memory
  GalConst* gal_konst_dev = nullptr;           // [GAL_MAXC]
  GalConst gal_konst_host[GAL_MAXC] = {};      // ... as last uploaded
  bool gal_konst_set[GAL_MAXC] = {};
  int gbm_occ = 2;                             // option NUTS_GBM_OCC (A/B): register budget of the group-block pass's merged launch
  int gal_pf3 = 0;                             // option NUTS_GAL_PF3 (A/B): tiles requested three ahead instead of two (four and eight chains)
  int gal_occ5 = 1;                            // option NUTS_GAL_OCC5 (A/B, default 1): four to six chains at the 96-register budget (five waves per SIMD: every group of the benchmark resident) instead of 128
  int rows_lds = 1;                            // option NUTS_ROWS_GROUP_LDS (read when the first member joins); 0: the round-5 kernel (<= 4 chains)
};
static_assert(GAM_MAXNC == MVM_MAXC && GAL_MAXC == GAM_MAXC && GAL_MAXC <= GRP_MAXC, "group sizes");

static nuts_model* group_base(nuts_group* g) {   // whose copy of (P, mu) every launch reads: one copy stays cache-resident
  for (int i = 0; i < GRP_MAXC; ++i) if (g->member[i]) return g->member[i];
  return nullptr;
}

static void group_flush_rows_locked(nuts_group* g);
static void group_flush_locked(nuts_group* g) {
  const int nc = g->npend;
  if (!nc) return;
  if (g->kind >= 2) { group_flush_rows_locked(g); return; }
  const ModelDev& md = group_base(g)->md;
  int order[GRP_MAXC];   // (by place in the group, not by arrival: the launch does not depend on who came first)
  for (int a = 0; a < GRP_MAXC; ++a) order[a] = a;
  for (int a = 1; a < nc; ++a)
    for (int b = a; b > 0 && g->pend[order[b]].slot < g->pend[order[b - 1]].slot; --b) std::swap(order[b], order[b - 1]);
  if (g->cap > MVM_MAXC) {   // the wide group: matrix cores, whatever the number of chains in this launch
    MfmArgs ma;
    ma.nc = nc; ma.pad = 0;
    for (int c = 0; c < nc; ++c) {
      const MvaLeafArgs& L = g->pend[order[c]];
      MfmLeaf& l = ma.c[c];
      l.io = L.io; l.cio = L.cio; l.uniforms = L.A.uniforms; l.log_uniforms = L.A.log_uniforms; l.st = L.st; l.ctl = L.A.ctl;
      l.j = L.j; l.fold = L.fold; l.d = L.d; l.max_depth = L.max_depth; l.par = L.par; l.cj = L.cj; l.cd = L.cd; l.cseq = L.cseq;
      l.slot = L.slot; l.pad = 0;
      // the chain's constant part: uploaded when it is first seen (and should it ever change); stream-ordered before the launch
      MfmChainConst k{};
      k.A = L.A; k.A.uniforms = nullptr; k.A.log_uniforms = nullptr; k.Emax = L.Emax; k.al_part = L.al_part;
      if (!g->konst_set[L.slot] || std::memcmp(&k, &g->konst_host[L.slot], sizeof(k)) != 0) {
        g->konst_host[L.slot] = k; g->konst_set[L.slot] = true;
        hipMemcpyAsync(g->konst_dev + L.slot, &g->konst_host[L.slot], sizeof(k), hipMemcpyHostToDevice, g->stream);
      }
    }
    for (int c = nc; c < MFM_MAXC; ++c) ma.c[c] = ma.c[0];
    if (g->md_of != group_base(g)) {
      g->md_of = group_base(g);
      hipMemcpyAsync(g->md_dev, &g->md_of->md, sizeof(ModelDev), hipMemcpyHostToDevice, g->stream);
    }
    hipLaunchKernelGGL(k_mfm_pack, dim3((md.mv.k * MFM_MAXC + 255) / 256), dim3(256), 0, g->stream, md.mv, (const MfmChainConst*)g->konst_dev, ma, g->dpack);
    hipLaunchKernelGGL(k_mvn_mfma_multi, dim3(MFM_MAXC + md.mv.al_nwg), dim3(MFM_WAVES * WAVE), 0, g->stream, md.mv, (const ModelDev*)g->md_dev,
                       (const MfmChainConst*)g->konst_dev, ma, (const double*)g->dpack);
    g->launches[nc]++;
    g->npend = 0;
    g->gen.fetch_add(1, std::memory_order_release);
    return;
  }
  const dim3 grid(MVM_MAXC + md.mv.al_nwg);
#define MVM_LAUNCH(RR, NC)                                                                                                   \
  {                                                                                                                          \
    MvaMultiArgs<NC> ma;                                                                                                     \
    for (int c = 0; c < NC; ++c) ma.c[c] = g->pend[order[c]];                                                                \
    hipLaunchKernelGGL((k_mvn_aligned_multi<RR, NC>), grid, dim3(MVM_THREADS(NC)), 0, g->stream, md, ma);                    \
  }
#define MVM_BY_NC(RR)                                                                                                        \
  switch (nc) {                                                                                                              \
    case 1: MVM_LAUNCH(RR, 1) break;                                                                                         \
    case 2: MVM_LAUNCH(RR, 2) break;                                                                                         \
    case 3: MVM_LAUNCH(RR, 3) break;                                                                                         \
    default: MVM_LAUNCH(RR, 4) break;                                                                                        \
  }
  if (md.mv.aligned == 8) MVM_BY_NC(8) else MVM_BY_NC(4)
#undef MVM_BY_NC
#undef MVM_LAUNCH
  g->launches[nc]++;
  g->npend = 0;
  g->gen.fetch_add(1, std::memory_order_release);
}

// The merged launch of the group-aligned row pass: the base member's model (X, y, the closed-form priors), every pending chain's
// own arena / records / tickets.  Register budget by the number of chains: two sets of accumulators still fit the single-chain
// kernel's 128 registers (five workgroups of W waves per CU: all groups resident), three and four need the 168-register budget.
static void group_flush_rows_locked(nuts_group* g) {
  const int nc = g->npend;
  const nuts_model* base = group_base(g);
  const ModelDev& md = base->md;
  int order[GAL_MAXC];   // (by place in the group, not by arrival)
  for (int a = 0; a < GAL_MAXC; ++a) order[a] = a;
  for (int a = 1; a < nc; ++a)
    for (int b = a; b > 0 && g->gpend[order[b]].slot < g->gpend[order[b - 1]].slot; --b) std::swap(order[b], order[b - 1]);
  const int rev = base->rows_alternate ? (g->rows_flip ^= 1) : 0;
  auto gal_leaf = [&](GalLeaf& l, const GaLeafArgs& L) {
    l.io = L.io; l.cio = L.cio; l.uniforms = L.A.uniforms; l.log_uniforms = L.A.log_uniforms;
    l.j = L.j; l.fold = L.fold; l.par = L.par; l.d = L.d; l.cj = L.cj; l.cd = L.cd; l.cseq = L.cseq; l.slot = L.slot;
  };
  auto gal_upload_consts = [&]() {   // the chains' constant parts: uploaded when a chain is first seen (and should it ever change); stream-ordered before the launch
    for (int c = 0; c < nc; ++c) {
      const GaLeafArgs& L = g->gpend[order[c]];
      GalConst k{};
      k.A = L.A; k.A.uniforms = nullptr; k.A.log_uniforms = nullptr; k.Emax = L.Emax; k.st = L.st;
      k.ga_part = L.ga_part; k.ga_bpart = L.ga_bpart; k.ga_ticket = L.ga_ticket; k.def_loc = L.def_loc; k.max_depth = L.max_depth; k.slot = L.slot;
      if (!g->gal_konst_set[L.slot] || std::memcmp(&k, &g->gal_konst_host[L.slot], sizeof(k)) != 0) {
        g->gal_konst_host[L.slot] = k; g->gal_konst_set[L.slot] = true;
        hipMemcpyAsync(g->gal_konst_dev + L.slot, &g->gal_konst_host[L.slot], sizeof(k), hipMemcpyHostToDevice, g->stream);
      }
    }
  };
  if (g->kind == 3) {
    // the group-block pass (C2-S): the chains share the LAUNCH -- one control slot + ga_nblk row workgroups per chain, each running
    // the single-chain body on its chain's arguments (rows_gb_multi_kernel.h)
    GbmArgs ma;
    for (int c = 0; c < nc; ++c) {
      const GaLeafArgs& L = g->gpend[order[c]];
      gal_leaf(ma.c[c], L);
      GalConst& k = ma.k[c];
      k.A = L.A; k.A.uniforms = nullptr; k.A.log_uniforms = nullptr; k.Emax = L.Emax; k.st = L.st;
      k.ga_part = L.ga_part; k.ga_bpart = L.ga_bpart; k.ga_ticket = L.ga_ticket; k.def_loc = L.def_loc; k.max_depth = L.max_depth; k.slot = L.slot;
    }
    for (int c = nc; c < GAL_MAXC; ++c) { ma.c[c] = ma.c[0]; ma.k[c] = ma.k[0]; }
    ma.nc = nc; ma.rev = rev;
    const dim3 grid(nc * (md.lg.ga_nblk + 1)), block(WAVE * GB_W);
    // (register budget by the rounds the grid needs: NUTS_GBM_OCC = 2 / 4 / 6 waves per SIMD, i.e. 1 / 2 / 3 workgroups per CU)
#define GBM_LAUNCH(DXX)                                                                                              \
    switch (g->gbm_occ) {                                                                                            \
      default: hipLaunchKernelGGL((k_rows_gb_multi<8, DXX, 2>), grid, block, 0, g->stream, md, ma); break;           \
      case 6: hipLaunchKernelGGL((k_rows_gb_multi<8, DXX, 6>), grid, block, 0, g->stream, md, ma); break;            \
      case 4: hipLaunchKernelGGL((k_rows_gb_multi<8, DXX, 4>), grid, block, 0, g->stream, md, ma); break;            \
    }
    if (md.lg.ga_dx == 7) GBM_LAUNCH(7) else GBM_LAUNCH(8)
#undef GBM_LAUNCH
    g->launches[nc]++;
    g->npend = 0;
    g->gen.fetch_add(1, std::memory_order_release);
    return;
  }
  // Which merged launch: up to four chains the round-5 kernel (every wave all chains: 70 / 81 / 106 us at C2-L), five to eight the
  // LDS-shared one (one wave per chain: 213 - 234 us; its two- to four-chain instantiations measure 80 / 100 / 119 us).  Both are
  // bitwise the chain alone, so a chain may pass through either from one leaf to the next.  NUTS_ROWS_GROUP_LDS = 2: the LDS-shared
  // kernel for every launch of two chains or more (tests, A/B).
  if (g->rows_lds && (nc > GAM_MAXNC || (nc >= 2 && g->rows_lds == 2))) {
    // one wave per chain, the tiles shared through LDS (rows_gal_kernel.h).  OCC: waves per SIMD the register budget is sized for:
    // four (128 registers: the two rows of a lane side by side without a spill in the stream).
    const dim3 grid(GAL_MAXC + md.lg.G);
#define GAL_LAUNCH(NC, DXX, OCC)                                                                                     \
  {                                                                                                                  \
    GalArgs<NC> la;                                                                                                  \
    for (int c = 0; c < NC; ++c) gal_leaf(la.c[c], g->gpend[order[c]]);                                              \
    la.rev = rev; la.pad = 0;                                                                                        \
    hipLaunchKernelGGL((k_rows_gal<NC, DXX, OCC>), grid, dim3(WAVE * NC), 0, g->stream, md, (const GalConst*)g->gal_konst_dev, la); \
  }
#define GAL_LAUNCH3(NC, DXX, OCC)   /* three tiles requested ahead (A/B: NUTS_GAL_PF3) */                             \
  {                                                                                                                  \
    GalArgs<NC> la;                                                                                                  \
    for (int c = 0; c < NC; ++c) gal_leaf(la.c[c], g->gpend[order[c]]);                                              \
    la.rev = rev; la.pad = 0;                                                                                        \
    hipLaunchKernelGGL((k_rows_gal<NC, DXX, OCC, 3>), grid, dim3(WAVE * NC), 0, g->stream, md, (const GalConst*)g->gal_konst_dev, la); \
  }
#define GAL_BY_NC(DXX)                                                                                               \
  switch (nc) {                                                                                                      \
    case 2: GAL_LAUNCH(2, DXX, 4) break;                                                                             \
    case 3: GAL_LAUNCH(3, DXX, 4) break;                                                                             \
    case 4: if (g->gal_occ5) GAL_LAUNCH(4, DXX, 5) else if (g->gal_pf3) GAL_LAUNCH3(4, DXX, 4) else GAL_LAUNCH(4, DXX, 4) break; \
    case 5: if (g->gal_occ5) GAL_LAUNCH(5, DXX, 5) else GAL_LAUNCH(5, DXX, 4) break;                                 \
    case 6: if (g->gal_occ5) GAL_LAUNCH(6, DXX, 5) else GAL_LAUNCH(6, DXX, 4) break;                                 \
    case 7: GAL_LAUNCH(7, DXX, 4) break;                                                                             \
    default: if (g->gal_pf3) GAL_LAUNCH3(8, DXX, 4) else GAL_LAUNCH(8, DXX, 4) break;                                \
  }
    gal_upload_consts();
    if (md.lg.ga_dx == 7) GAL_BY_NC(7) else GAL_BY_NC(8)
#undef GAL_BY_NC
#undef GAL_LAUNCH
#undef GAL_LAUNCH3
    g->launches[nc]++;
    g->npend = 0;
    g->gen.fetch_add(1, std::memory_order_release);
    return;
  }
  const dim3 grid(GAM_MAXC + md.lg.G), block(WAVE * md.lg.ga_w);
#define GAM_LAUNCH(NC, OCC, DXX)                                                                                     \
  {                                                                                                                  \
    GaMultiArgs<NC> ma;                                                                                              \
    for (int c = 0; c < NC; ++c) ma.c[c] = g->gpend[order[c]];                                                       \
    ma.rev = rev; ma.pad = 0;                                                                                        \
    hipLaunchKernelGGL((k_rows_ga_multi<NC, OCC, DXX>), grid, block, 0, g->stream, md, ma);                          \
  }
#define GAM_BY_NC(DXX)                                                                                               \
  switch (nc) {                                                                                                      \
    case 1: GAM_LAUNCH(1, 4, DXX) break;                                                                             \
    case 2: GAM_LAUNCH(2, 3, DXX) break;                                                                             \
    case 3: GAM_LAUNCH(3, 3, DXX) break;                                                                             \
    default: GAM_LAUNCH(4, 3, DXX) break;                                                                            \
  }
  if (md.lg.ga_dx == 7) GAM_BY_NC(7) else GAM_BY_NC(8)
#undef GAM_BY_NC
#undef GAM_LAUNCH
  g->launches[nc]++;
  g->npend = 0;
  g->gen.fetch_add(1, std::memory_order_release);
}

// A deposit waits until the launch that carries it has been submitted -- by the partner that completes the set of chains standing
// inside a tree.  The wait is bounded: every partner is inside engine code with no callback into Python, so a launch follows within
// microseconds; should one not come for GROUP_WAIT_S seconds (a partner's thread died, a debugger holds it), the waiter submits
// what is pending itself and goes on -- lockstep is an optimisation, never a condition of progress.
#define GROUP_WAIT_S 2.0
static void group_wait(nuts_group* g, unsigned mine) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; g->gen.load(std::memory_order_acquire) == mine; ++spins) {
    if ((spins & 0x3ff) != 0x3ff) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#endif
      continue;
    }
    std::this_thread::yield();
    if ((spins & 0xfffff) == 0xfffff && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > GROUP_WAIT_S) {
      std::lock_guard<std::mutex> lk(g->mu);
      if (g->gen.load(std::memory_order_relaxed) == mine) group_flush_locked(g);
      return;
    }
  }
}

static void group_deposit(nuts_model* m, const MvaLeafArgs& L) {
  nuts_group* g = m->group;
  unsigned mine;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->pend[g->npend++] = L;
    mine = g->gen.load(std::memory_order_relaxed);
    if (g->npend >= g->nactive) { group_flush_locked(g); return; }
  }
  group_wait(g, mine);
}

static void group_deposit(nuts_model* m, const GaLeafArgs& L) {
  nuts_group* g = m->group;
  unsigned mine;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->gpend[g->npend++] = L;
    mine = g->gen.load(std::memory_order_relaxed);
    if (g->npend >= g->nactive) { group_flush_locked(g); return; }
  }
  group_wait(g, mine);
}

static void group_enter(nuts_model* m) {
  if (!m || !m->group || m->g_active) return;
  std::lock_guard<std::mutex> lk(m->group->mu);
  m->group->nactive++;
  m->g_active = true;
}

static void group_leave(nuts_model* m) {
  if (!m || !m->group || !m->g_active) return;
  nuts_group* g = m->group;
  std::lock_guard<std::mutex> lk(g->mu);
  g->nactive--;
  m->g_active = false;
  if (g->npend && g->npend >= g->nactive) group_flush_locked(g);   // the others were only waiting for this chain
}

struct GroupTreeScope {   // a chain is a lockstep partner from the first leaf of a tree to the tree's last status
  nuts_model* m;
  explicit GroupTreeScope(nuts_model* mm) : m(mm) { group_enter(m); }
  ~GroupTreeScope() { group_leave(m); }
};

// the control work of a leaf on the lean path as a launch of its own (kernels.h: control_lean; the row-aligned MvNormal pass
// leaves per-workgroup records that `mva_control` sums first)
static void launch_control_lean(nuts_model* m, const ArenaDev& A, const EvalIO& io, int j, int d, double Emax, int max_depth,
                                HostStatus* st, int seq) {
  if (m->md.has_mvn && m->md.mv.aligned)
    hipLaunchKernelGGL(k_mva_control, dim3(1), dim3(VEC_THREADS), 0, m->stream, m->md, A, io, j, d, Emax, max_depth, st, seq, m->ga_par);
  else
    hipLaunchKernelGGL(k_control_lean, dim3(1), dim3(VEC_THREADS), 0, m->stream, m->md, A, io, j, d, Emax, max_depth, st, seq, m->ga_par);
}

static void launch_vector(nuts_model* m, const ArenaDev& A, const EvalIO& io, int j, int d) {
  const ModelDev& md = m->md;
  // linear predictors (lin_kernel.h): eta = X coef at this leaf's position, before the factors that read them are swept
  if (md.n_lins > 0) {
    if (md.n_derived > 0) hipLaunchKernelGGL(k_derive, dim3(2), dim3(256), 0, m->stream, md, A, io, j);   // (coefficients that are a derived vector)
    for (int l = 0; l < md.n_lins; ++l) {
      const LinDev& L = m->lins_host[l];
      if (L.N == 1) { hipLaunchKernelGGL(k_lin_fwd1, dim3(L.K), dim3(1024), 0, m->stream, md, A, io, j, l); continue; }
      const dim3 grid((unsigned)std::min<int64_t>(4096, (L.N + 255) / 256));
      const size_t lds = (size_t)L.K * L.P * sizeof(double);
#define LIN_FWD(KT) hipLaunchKernelGGL(k_lin_fwd<KT>, grid, dim3(256), lds, m->stream, md, A, io, j, l)
      if (L.K == 1) LIN_FWD(1); else if (L.K == 2) LIN_FWD(2); else if (L.K <= 4) LIN_FWD(4); else if (L.K <= 8) LIN_FWD(8); else LIN_FWD(16);
#undef LIN_FWD
    }
  }
  // gathered adjoints (model_dev.h GSlot): every element of the factors that read variables through index vectors is swept once,
  // before the kernels whose gathers add the results up (B and C below; the dense node's seed of a derived vector is already there)
  if (md.n_gsf > 0) {
    if (md.n_swf > 0)
      hipLaunchKernelGGL(k_gsweep_fast, dim3(md.n_gs_blocks), dim3(GSL_THREADS), (size_t)md.sw_rows * GSL_THREADS * sizeof(double), m->stream, md, A, io, j);
    else if (md.gs_lds_rows > 0)
      hipLaunchKernelGGL(k_gsweep_lds, dim3(md.n_gs_blocks), dim3(GSL_THREADS), (size_t)md.gs_lds_bytes, m->stream, md, A, io, j);
    else
      hipLaunchKernelGGL(k_gsweep, dim3(md.n_gs_blocks), dim3(256), 0, m->stream, md, A, io, j);
  }
  if (md.n_glong > 0) hipLaunchKernelGGL(k_gadj_reduce, dim3(md.n_glong), dim3(256), 0, m->stream, md, A, io);
  // ... and the predictors' adjoints go back to the coefficients: X^T adj in row chunks, then every coefficient adds its partials up
  if (md.n_lins > 0) {
    for (int l = 0; l < md.n_lins; ++l) {
      const LinDev& L = m->lins_host[l];
      if (L.N == 1) { hipLaunchKernelGGL(k_lin_bwd1, dim3(L.K), dim3(1024), 0, m->stream, md, A, io, l); continue; }
      const dim3 grid((unsigned)L.nchunk, (unsigned)L.P);
#define LIN_BWD(KT) hipLaunchKernelGGL(k_lin_bwd<KT>, grid, dim3(256), 0, m->stream, md, A, io, l)
      if (L.K == 1) LIN_BWD(1); else if (L.K == 2) LIN_BWD(2); else if (L.K <= 4) LIN_BWD(4); else if (L.K <= 8) LIN_BWD(8); else LIN_BWD(16);
#undef LIN_BWD
    }
    if (md.n_lin_targets > 0) hipLaunchKernelGGL(k_lin_fin, dim3((md.n_lin_targets + 255) / 256), dim3(256), 0, m->stream, md, A, io);
  }
  if (md.lg.ga) return;   // group-aligned row pass: the O(n) work rides in the row pass itself (rows_ga_kernel.h)
  if (md.has_mvn && md.mv.aligned && io.lean) return;   // (the row-aligned MvNormal pass has finished the leapfrog itself)
  // small models: 8x oversubscribed launch, one XCD does the work (see k_vector); large ones use the whole chip
  const dim3 grid(m->vector_one_xcd ? md.nblk * 8 : md.nblk);
  // (has_prog: the instantiation that carries the expression-program interpreter, model_dev.h gather_element)
#define VEC_LAUNCH(E, P) hipLaunchKernelGGL((k_vector<E, P>), grid, dim3(VEC_THREADS), 0, m->stream, md, A, io, j, d)
  switch (m->ept) {
    case 1: if (m->has_prog) VEC_LAUNCH(1, true); else VEC_LAUNCH(1, false); break;
    case 4: if (m->has_prog) VEC_LAUNCH(4, true); else VEC_LAUNCH(4, false); break;
    default: if (m->has_prog) VEC_LAUNCH(16, true); else VEC_LAUNCH(16, false); break;
  }
#undef VEC_LAUNCH
}

// kernel A of the pipeline: the pass over the model data (timed when profiling is on)
// `fold` (lean path only, kernels.h): workgroup 0 of the row pass does the control work of leaf j-1
struct CtlJob {   // control work riding in workgroup 0 of a group-aligned row pass (rows_ga_kernel.h, GaArgs)
  EvalIO io; int j, d, seq; bool src_prev;
};
static void launch_dense(nuts_model* m, const ArenaDev& A, const EvalIO& io, int j, int fold = 0, int d = 0, double Emax = 0.0,
                         int max_depth = 0, HostStatus* st = nullptr, const CtlJob* job = nullptr) {
  ModelDev& md = m->md;
  if (md.has_mix) {   // mixture node (mixture_kernel.h): per-workgroup sums over the rows, then their total and the node's outputs
    const dim3 grid(md.mix.nwg), block(MIX_BLOCK);
    if (md.mix.K <= 4) hipLaunchKernelGGL(k_mix_rows<4>, grid, block, 0, m->stream, md, A, io, j);
    else if (md.mix.K <= 8) hipLaunchKernelGGL(k_mix_rows<8>, grid, block, 0, m->stream, md, A, io, j);
    else hipLaunchKernelGGL(k_mix_rows<16>, grid, block, 0, m->stream, md, A, io, j);
    hipLaunchKernelGGL(k_mix_reduce, dim3(1), dim3(WAVE), 0, m->stream, md, A, io, j);
  }
  if (!md.has_logit && !md.has_mvn && !md.has_glm) return;
  // derived vectors (a dense node's parameter that is an expression of the model's variables): evaluated before the pass reads them
  if (md.n_derived > 0) hipLaunchKernelGGL(k_derive, dim3(2), dim3(256), 0, m->stream, md, A, io, j);
  // (a member of a chain group shares its launches and its stream with other chains: event pairs around them would time the company)
  bool prof = m->profile && !m->group && (m->dom_launches % m->sample_every == 0) && m->ev_used + 2 <= m->ev.size();
  // Inside a tree the leaves of a doubling are back-to-back launches of the timed kernel (control folded into them): doublings of
  // PROF_RUN_MIN leaves or more are bracketed WHOLE -- two marker packets per run instead of two per launch (which stretch a
  // bracketed launch by ~2 us: the per-launch figure then did not add up to the step time) -- and single launches
  // are not sampled there.  A run some of whose launches drained behind a finished tree is dropped when the figures are read
  // (its average is far below the others': profile_sum_ms).
  bool run_begin = false, run_end = false;
  if (io.mode == MODE_TREE && (1 << d) >= PROF_RUN_MIN && !md.has_glm) {
    prof = false;
    if (m->profile && !m->group) {
      if (j == 0 && !m->run_open && m->ev_used + 2 <= m->ev.size() && (m->runs_seen++ % 2 == 0)) run_begin = true;
      if (j == (1 << d) - 1 && (m->run_open || run_begin)) run_end = true;
    }
  }
  if (run_begin) { hipEventRecord(m->ev[m->ev_used], m->stream); m->run_open = 1; }
  struct RunEnd {   // (after the launch, whichever branch below submits it)
    nuts_model* m; bool on; int units;
    ~RunEnd() { if (on) { hipEventRecord(m->ev[m->ev_used + 1], m->stream); m->ev_units[m->ev_used / 2] = units; m->ev_used += 2; m->dom_units += units; m->run_open = 0; } }
  } run_end_guard{m, run_end, 1 << d};
  if (prof) hipEventRecord(m->ev[m->ev_used], m->stream);
  if (md.has_glm) {   // GLM node (glm_kernel.h): the fused pass over X (the timed kernel), then the totals of its records
    const dim3 grid(md.glm.nwg), block(GLM_BLOCK);
#define GLM_LAUNCH(LL, CC) hipLaunchKernelGGL((k_glm_rows<LL, CC>), grid, block, 0, m->stream, md, A, io, j)
    switch (md.glm.lpr * 8 + md.glm.ch) {
      case 1 * 8 + 1: GLM_LAUNCH(1, 1); break;
      case 1 * 8 + 2: GLM_LAUNCH(1, 2); break;
      case 1 * 8 + 4: GLM_LAUNCH(1, 4); break;
      case 2 * 8 + 4: GLM_LAUNCH(2, 4); break;
      case 4 * 8 + 4: GLM_LAUNCH(4, 4); break;
      case 8 * 8 + 4: GLM_LAUNCH(8, 4); break;
      case 16 * 8 + 4: GLM_LAUNCH(16, 4); break;
      case 32 * 8 + 3: GLM_LAUNCH(32, 3); break;
      case 32 * 8 + 4: GLM_LAUNCH(32, 4); break;
      case 64 * 8 + 3: GLM_LAUNCH(64, 3); break;
      default: GLM_LAUNCH(64, 4); break;
    }
#undef GLM_LAUNCH
    if (prof) { hipEventRecord(m->ev[m->ev_used + 1], m->stream); m->ev_used += 2; m->dom_units += 1; }
    m->dom_launches++;
    hipLaunchKernelGGL(k_glm_reduce, dim3((md.glm.Ppad + 3 + GLM_RED_COLS - 1) / GLM_RED_COLS), dim3(GLM_RED_CHUNKS * GLM_RED_COLS), 0, m->stream, md, A, io, j);
    if (!md.has_mvn) return;
    // an MvNormal node next to the GLM node (e.g. a multivariate-normal prior on the coefficients): its mat-vec follows; the timed
    // kernel of such a model is the pass over X above
    prof = false; m->dom_launches--;
  }
  if (md.has_logit && md.lg.ga) {
    const int rev = m->rows_alternate ? (m->rows_flip ^= 1) : 0;
    const int par = (m->ga_par ^= 1);
    const dim3 grid(m->rows_grid + (fold ? 1 : 0) + md.lg.ga_naux), block(WAVE * md.lg.ga_w);   // [control] + groups / blocks + auxiliary
    GaArgs ga{md, A, io, j, rev, fold ? (GA_FOLD_CTL | GA_FOLD_SRC) : 0, par, d, max_depth, Emax, st, io, j - 1, d, 0, 0};
    if (fold && job) {   // the control work of another doubling's last leaf
      ga.fold = GA_FOLD_CTL | (job->src_prev ? GA_FOLD_SRC : 0);
      ga.cio = job->io; ga.cj = job->j; ga.cd = job->d; ga.cseq = job->seq;
    }
    if (m->group && m->g_active && io.mode == MODE_TREE && m->group->kind >= 2) {
      // a member of a chain group inside a tree: the launch is deposited; the partner that completes the set submits ONE launch
      // that streams X once for all of them (rows_ga_multi_kernel.h)
      GaLeafArgs L;
      L.A = A; L.io = io; L.cio = ga.cio; L.Emax = Emax; L.st = st;
      L.ga_part = md.lg.ga_part; L.ga_bpart = md.lg.ga_bpart; L.ga_ticket = md.lg.ga_ticket; L.def_loc = md.def_loc;
      L.j = j; L.fold = ga.fold; L.par = par; L.d = d; L.max_depth = max_depth; L.cj = ga.cj; L.cd = ga.cd; L.cseq = ga.cseq;
      L.slot = m->gslot; L.pad = 0;
      group_deposit(m, L);
    } else
    if (md.lg.ga_gpw) {   // group-block pass (small groups): same arguments, same protocol
      switch (md.lg.D) {
        case 8: if (md.lg.ga_dx == 7) hipLaunchKernelGGL((k_rows_gb<8, 7>), grid, block, 0, m->stream, ga);
                else hipLaunchKernelGGL((k_rows_gb<8>), grid, block, 0, m->stream, ga); break;
        case 4: hipLaunchKernelGGL((k_rows_gb<4>), grid, block, 0, m->stream, ga); break;
        case 2: hipLaunchKernelGGL((k_rows_gb<2>), grid, block, 0, m->stream, ga); break;
        case 1: hipLaunchKernelGGL((k_rows_gb<1>), grid, block, 0, m->stream, ga); break;
        case 3: hipLaunchKernelGGL((k_rows_gb<3>), grid, block, 0, m->stream, ga); break;
        case 5: hipLaunchKernelGGL((k_rows_gb<5>), grid, block, 0, m->stream, ga); break;
        case 6: hipLaunchKernelGGL((k_rows_gb<6>), grid, block, 0, m->stream, ga); break;
        default: hipLaunchKernelGGL((k_rows_gb<7>), grid, block, 0, m->stream, ga); break;
      }
    } else {
#define GA_LAUNCH(DD, OO, PP) hipLaunchKernelGGL((k_rows_ga<DD, 2, OO, PP>), grid, block, 0, m->stream, ga)
#define GA_BY_D(OO, PP)                      \
    switch (md.lg.D) {                       \
      case 8: if (md.lg.ga_dx == 7) hipLaunchKernelGGL((k_rows_ga<8, 2, OO, PP, 7>), grid, block, 0, m->stream, ga); \
              else GA_LAUNCH(8, OO, PP); break;   \
      case 4: GA_LAUNCH(4, OO, PP); break;   \
      default: GA_LAUNCH(2, OO, PP); break;  \
    }
    if (md.lg.D != 8 && md.lg.D != 4 && md.lg.D != 2) {   // the other covariate counts: the default budget only
      switch (md.lg.D) {
        case 1: GA_LAUNCH(1, 4, 2); break;
        case 3: GA_LAUNCH(3, 4, 2); break;
        case 5: GA_LAUNCH(5, 4, 2); break;
        case 6: GA_LAUNCH(6, 4, 2); break;
        default: GA_LAUNCH(7, 4, 2); break;
      }
    } else
    switch (m->ga_variant) {   // (register budget, tiles in flight): see rows_ga_kernel.h
      case 32: GA_BY_D(3, 2) break;
      case 33: GA_BY_D(3, 3) break;
      default: GA_BY_D(4, 2) break;
    }
#undef GA_BY_D
#undef GA_LAUNCH
    }
  } else if (md.has_logit) {
    const int rev = m->rows_alternate ? (m->rows_flip ^= 1) : 0;
    const dim3 grid(m->rows_grid + (fold ? 1 : 0)), block(ROWS_BLOCK);
#define ROWS_LAUNCH(DD, RR, OO) \
    hipLaunchKernelGGL((k_rows<DD, RR, OO>), grid, block, 0, m->stream, md, A, io, j, rev, fold, d, Emax, max_depth, st)
#define ROWS_BY_D(RR, OO)                                   \
    switch (md.lg.D) {                                      \
      case 8: ROWS_LAUNCH(8, RR, OO); break;                \
      case 4: ROWS_LAUNCH(4, RR, OO); break;                \
      default: ROWS_LAUNCH(2, RR, OO); break;               \
    }
    // (one launch geometry: two rows per lane, launch bounds for four waves per SIMD -- the variants with 3 / 5 / 6 waves and four
    // rows per lane lost their A/Bs in round 1 and were 21 more instantiations of the kernel to compile)
    switch (md.lg.D) {   // (widths that are not a power of two take the same geometry)
      case 1: ROWS_LAUNCH(1, 2, 4); break;
      case 3: ROWS_LAUNCH(3, 2, 4); break;
      case 5: ROWS_LAUNCH(5, 2, 4); break;
      case 6: ROWS_LAUNCH(6, 2, 4); break;
      case 7: ROWS_LAUNCH(7, 2, 4); break;
      default: ROWS_BY_D(2, 4)
    }
#undef ROWS_BY_D
#undef ROWS_LAUNCH
  }
  if (md.has_mvn && md.mv.winv) {
    // "cholesky" solver: delta, y = W delta, P delta = W^T y, outputs (chains on such a model do not fold the control work)
    const MvnDev& mv = md.mv;
    const int* abort_flag = io.mode == MODE_TREE ? &A.ctl->aborted : nullptr;
    const dim3 gk((mv.k + 255) / 256), gmv((mv.k + (256 / WAVE) - 1) / (256 / WAVE));
    hipLaunchKernelGGL(k_mvn_delta, gk, dim3(256), 0, m->stream, mv, A, io, j);
    hipLaunchKernelGGL(k_dense_mv, gmv, dim3(256), 0, m->stream, mv.winv, mv.wy, mv.wy + mv.k, mv.k, (const double*)nullptr, (double*)nullptr, 0.0, abort_flag);
    hipLaunchKernelGGL(k_dense_mv, gmv, dim3(256), 0, m->stream, mv.winv_t, mv.wy + mv.k, mv.wy + 2 * mv.k, mv.k, (const double*)nullptr, (double*)nullptr, 0.0, abort_flag);
    hipLaunchKernelGGL(k_mvn_finish, gk, dim3(256), 0, m->stream, mv, A, io);
  } else if (md.has_mvn && md.mv.aligned && io.lean) {
    const int par = (m->ga_par ^= 1);
    // control work riding in workgroup 0: leaf j - 1 of this doubling, or (look-ahead) the last leaf of the previous doubling
    const EvalIO& cio = job ? job->io : io;
    const int cj = job ? job->j : j - 1, cd = job ? job->d : d, cseq = job ? job->seq : 0;
#define MVA_LAUNCH(RR) hipLaunchKernelGGL(k_mvn_aligned<RR>, dim3(md.mv.al_nwg + 1), dim3(MVA_THREADS), 0, m->stream, md, A, io, j, \
                                           fold, d, Emax, max_depth, st, par, cio, cj, cd, cseq)
    if (m->group && m->g_active && io.mode == MODE_TREE) {
      MvaLeafArgs L;
      L.A = A; L.io = io; L.cio = cio; L.Emax = Emax; L.st = st; L.al_part = md.mv.al_part;
      L.j = j; L.fold = fold; L.d = d; L.max_depth = max_depth; L.par = par; L.cj = cj; L.cd = cd; L.cseq = cseq;
      L.slot = m->gslot; L.pad = 0;
      group_deposit(m, L);
    } else
    switch (md.mv.aligned) {
      case 2: MVA_LAUNCH(2); break;
      case 8: MVA_LAUNCH(8); break;
      case 16: MVA_LAUNCH(16); break;
      default: MVA_LAUNCH(4); break;
    }
#undef MVA_LAUNCH
  } else if (md.has_mvn) {
    const int mfold = md.has_logit ? 0 : fold;   // (the control work rides in exactly one launch)
    hipLaunchKernelGGL(k_mvn_matvec, dim3(m->mvn_grid + (mfold ? 1 : 0)), dim3(MVN_BLOCK), 0, m->stream, md, A, io, j, mfold, d, Emax,
                       max_depth, st);
  }
  if (prof) { hipEventRecord(m->ev[m->ev_used + 1], m->stream); m->ev_used += 2; m->dom_units += 1; }
  m->dom_launches++;
}

// logp + gradient at a plain position vector (ValueGradFunction.__call__): A, B, C in MODE_PLAIN
static void model_enqueue_plain(nuts_model* m, const double* q_dev, double* g_dev, double* lp_dev) {
  ArenaDev A{};
  A.n = m->md.n; A.nblk = m->md.nblk; A.ept = m->ept; A.S = 1;
  EvalIO io{};
  io.mode = MODE_PLAIN; io.q = q_dev; io.grad = g_dev; io.logp = lp_dev; io.lean = m->md.lean_ok;
  launch_dense(m, A, io, 0);
  launch_vector(m, A, io, 0, 0);
  if (io.lean) launch_control_lean(m, A, io, 0, 0, 0.0, 0, nullptr, 0);
  else if (m->has_prog) hipLaunchKernelGGL(k_control<true>, dim3(1), dim3(VEC_THREADS), 0, m->stream, m->md, A, io, 0, 0, 0.0, 0, (HostStatus*)nullptr, 0);
  else hipLaunchKernelGGL(k_control<false>, dim3(1), dim3(VEC_THREADS), 0, m->stream, m->md, A, io, 0, 0, 0.0, 0, (HostStatus*)nullptr, 0);
}

// "Compile" the spec: contributions per variable, broadcast terms, deferred elements, orphan factors.
static bool compile_spec(nuts_model* m, const nuts_model_spec* s, std::vector<VarDev>& vars) {
  ModelDev& md = m->md;
  const int nv = s->n_vars, nf = s->n_factors;
  vars.resize(nv);
  for (int k = 0; k < nv; ++k) {
    const nuts_var& v = s->vars[k];
    vars[k] = VarDev{};
    vars[k].offset = v.offset; vars[k].size = v.size; vars[k].transform = v.transform; vars[k].deferred = v.size == 1 ? 1 : 0;
    vars[k].lower = v.lower; vars[k].upper = v.upper;
    if (v.size <= 0) { g_err = "empty value variable"; return false; }
  }
  if (s->rows_N > 0) { vars[s->rows_mu].deferred = 1; vars[s->rows_sigma].deferred = 1; }
  if (s->n_lins < 0 || s->n_lins > NUTS_MAX_LINS || (s->n_lins > 0 && !s->lins)) { g_err = "bad number of linear predictors (NUTS_MAX_LINS)"; return false; }
  if (s->n_lins > 0 && (s->rows_N > 0 || s->mix_N > 0 || s->glm_N > 0)) { g_err = "linear predictors are not combined with the logit-rows, mixture or GLM node"; return false; }
  for (int l = 0; l < s->n_lins; ++l) {
    const nuts_lin& L = s->lins[l];
    if (L.N < 1 || L.P < 1 || L.K < 1 || L.K > NUTS_LIN_MAXK || !L.X) { g_err = "linear predictor with bad dimensions"; return false; }
    if (L.N > 1 && (L.P > NUTS_LIN_MAXP || (int64_t)L.K * L.P > 4096)) { g_err = "linear predictor: P <= 512 and K P <= 4096 for predictors with more than one row"; return false; }
    if (L.N * (int64_t)L.P > ((int64_t)1 << 31) || L.N > (int64_t)INT32_MAX) { g_err = "linear predictor: X too large"; return false; }
  }
  std::vector<std::vector<Contrib>> per_var(nv);
  std::vector<FactorBT> fbt(std::max(nf, 1));
  std::vector<int32_t> bterm_var, orphans;
  std::vector<nuts_factor> fac(s->factors, s->factors + nf);   // (pad is rewritten: 1 = the factor has gathered operands)
  std::vector<int32_t> csr;
  std::vector<std::pair<int, int>> gathered;   // (variable, index data id) pairs of the factor being compiled
  std::vector<int32_t>& derived = m->derived;  // NUTS_D_DERIVED factors, in factor order
  derived.clear();
  // gathered adjoints (model_dev.h GSlot): the factors whose elements are swept once, ahead of the gathers that read the result
  std::vector<GSweepFactor> gsf;
  std::vector<GSlot> gslots;
  std::vector<GLong> glong;
  int64_t adj_len = 0;
  int32_t gs_elems = 0;
  const bool gsweep_on = env_int("NUTS_GSWEEP", 1) != 0;
  std::vector<std::pair<int, int>> lin_used;   // (predictor, column) pairs of the factor being compiled (NUTS_OP_LIN operands)
  m->lin_uses.assign(std::max(s->n_lins, 0), {});
  for (int l = 0; l < s->n_lins; ++l) m->lin_uses[l].assign(std::max(std::min(s->lins[l].K, NUTS_LIN_MAXK), 0), {});
  // (round 6, last session) a LARGE factor with an expression program and no owning variable -- a likelihood over data whose
  // parameters are scalars: curve fits, robust regressions written out -- is swept as well, with no slot at all: the sweep accounts
  // its log-density and its scalars' adjoints (gs_part), and the scalar-driven sweep kernel walks it at a fraction of kernel B's cost
  // per element (NUTS_GSWEEP_ORPHANS = 0: kernel B walks it, as before).  Models the single-workgroup kernel takes are not touched.
  const int sweep_orphans_opt = env_int("NUTS_GSWEEP_ORPHANS", 1);     // (2: also large orphan factors WITHOUT a program -- A/B)
  const bool sweep_orphans = gsweep_on && sweep_orphans_opt != 0;
  auto finish_gather = [&](int fi, bool is_orphan = false) -> bool {
    // (a factor that reads a linear predictor MUST be swept: the sweep is where d logp / d eta comes from)
    const bool need = !lin_used.empty();
    const bool big_orphan = sweep_orphans && is_orphan && (s->factors[fi].n_instr > 0 || sweep_orphans_opt == 2) && s->factors[fi].size > SMALL_MAX_ELEMS &&
                            s->factors[fi].dist != NUTS_D_DERIVED;
    if (gathered.empty() && !need && !big_orphan) return true;
    // (a DERIVED vector is never swept: its "density" is the seed d logp / d element that the node reading it leaves behind -- a
    // linear predictor's backward pass runs after the sweep (launch_vector), and the sweep does not read the GLM node's seed either
    // -- so its gathers stay with kernels B / C, which run behind both.  Found on the device by `pm.ICAR` over 56 areas (the sum over
    // the edge list: a one-row predictor) and by tests/test_gpu_glm_fuzz.py (`beta = a[idx] + s z` under the GLM node): the
    // gradient of the gathered variable was lost)
    if (s->factors[fi].dist == NUTS_D_DERIVED) return true;
    if (big_orphan) m->has_prog = true;
    const int64_t fsize = s->factors[fi].size;
    const size_t nsl = gathered.size() + lin_used.size();
    const bool fits = nsl <= MAX_GSLOTS && adj_len + (int64_t)nsl * fsize <= ((int64_t)1 << 28) && (int64_t)gs_elems + fsize <= ((int64_t)1 << 30);   // (2 GiB of adjoints)
    if (need && !fits) { g_err = "a factor that reads linear predictors has too many gathered operands / elements for the adjoint sweep"; return false; }
    if ((!gsweep_on && !need) || !fits) return true;   // (keep the old path)
    GSweepFactor sf{fi, (int32_t)gslots.size(), (int32_t)nsl, gs_elems, 0, 0};
    for (const auto& gv : gathered) {
      gslots.push_back(GSlot{gv.first, gv.second, adj_len});
      for (Contrib& cb : per_var[gv.first])
        if (cb.f == fi && cb.arg == -2 && (int)cb.p[0] == gv.second) {
          cb.p[1] = (double)adj_len; cb.p[2] = 1.0; cb.p[3] = -1.0;
          // elements of the variable that many factor elements index: their lists are totalled by a workgroup each (GLong)
          const int vs = vars[gv.first].size;
          const int32_t* ptr = csr.data() + cb.dist;
          bool any = false;
          for (int e = 0; e < vs; ++e) any = any || (ptr[e + 1] - ptr[e] >= GADJ_LONG);
          if (any && glong.size() + (size_t)vs <= 16384) {
            std::vector<int32_t> map(vs, -1);
            for (int e = 0; e < vs; ++e)
              if (ptr[e + 1] - ptr[e] >= GADJ_LONG) {
                const int len = ptr[e + 1] - ptr[e], nc = (len + GADJ_CHUNK - 1) / GADJ_CHUNK;
                map[e] = (int32_t)glong.size();
                for (int c = 0; c < nc; ++c)   // (chunk c: entries [c GADJ_CHUNK, ...) of the list, a workgroup's work)
                  glong.push_back(GLong{adj_len, cb.pad + ptr[e] + c * GADJ_CHUNK, std::min(GADJ_CHUNK, len - c * GADJ_CHUNK), c == 0 ? nc : 0, 0});
              }
            cb.p[3] = (double)csr.size();
            csr.insert(csr.end(), map.begin(), map.end());
          }
        }
      adj_len += fsize;
    }
    for (const auto& lv : lin_used) {   // a predictor column is a slot too: var = -1 - predictor, did = column (model_dev.h push)
      gslots.push_back(GSlot{-1 - lv.first, lv.second, adj_len});
      m->lin_uses[lv.first][lv.second].push_back(nuts_model::LinUse{adj_len, (int32_t)fsize});
      adj_len += fsize;
    }
    gs_elems += (int32_t)fsize;
    gsf.push_back(sf);
    return true;
  };
  // NUTS_OP_LIN operand of factor fi: element-aligned with the factor, or a one-row predictor that broadcasts
  auto add_lin = [&](int fi, const nuts_operand& o) -> bool {
    const nuts_factor& f = s->factors[fi];
    const int k = (int)o.c;
    if (o.ref < 0 || o.ref >= s->n_lins || !s->lins) { g_err = "operand refers to a missing linear predictor"; return false; }
    const nuts_lin& L = s->lins[o.ref];
    if (k < 0 || k >= L.K || (double)k != o.c) { g_err = "operand refers to a missing column of a linear predictor"; return false; }
    if (L.N != f.size && L.N != 1) { g_err = "linear predictor: one row per element of the factor (or a single row that broadcasts)"; return false; }
    if (f.dist == NUTS_D_DERIVED) { g_err = "a derived vector cannot read a linear predictor"; return false; }
    for (const auto& lv : lin_used) if (lv.first == o.ref && lv.second == k) return true;
    lin_used.emplace_back(o.ref, k);
    fac[fi].pad = 1;
    m->has_prog = true;
    return true;
  };
  // NUTS_OP_GATHER operand of factor fi: the inverse index (which factor elements read element e of the variable, in order)
  auto add_gather = [&](int fi, const nuts_operand& o) -> bool {
    const nuts_factor& f = s->factors[fi];
    const int did = (int)o.c;
    if (o.ref < 0 || o.ref >= nv) { g_err = "gather refers to a missing variable"; return false; }
    if (did < 0 || did >= s->n_data || (double)did != o.c) { g_err = "gather refers to a missing index vector"; return false; }
    if (s->data[did].size != f.size) { g_err = "gather: one index per element of the factor"; return false; }
    // (a variable may be gathered into a factor through several index vectors -- the columns of `beta[idx]` summed over a small axis,
    // a broadcast of a vector against a matrix -- and be a direct operand of it as well: every (variable, index vector) pair is a
    // contribution of its own, and the reverse sweep credits a gather operand only to the pair it belongs to)
    for (const auto& gv : gathered)
      if (gv.first == o.ref && gv.second == did) return true;   // second occurrence of the same gather: already registered
    gathered.emplace_back(o.ref, did);
    const int vs = vars[o.ref].size;
    const double* idx = s->data_pool + s->data[did].offset;
    std::vector<int32_t> ptr(vs + 1, 0), lst(f.size);
    for (int i = 0; i < f.size; ++i) {
      const int e = (int)idx[i];
      if ((double)e != idx[i] || e < 0 || e >= vs) { g_err = "gather index out of range for its variable"; return false; }
      ptr[e + 1]++;
    }
    for (int e = 0; e < vs; ++e) ptr[e + 1] += ptr[e];
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int i = 0; i < f.size; ++i) lst[fill[(int)idx[i]]++] = i;   // increasing i within an element: fixed summation order
    Contrib cb{};
    cb.f = fi; cb.arg = -2; cb.slot = -1; cb.owner = 0; cb.fast = 0;
    cb.dist = (int32_t)csr.size(); csr.insert(csr.end(), ptr.begin(), ptr.end());
    cb.pad = (int32_t)csr.size(); csr.insert(csr.end(), lst.begin(), lst.end());
    cb.p[0] = (double)did;   // which gather operands of the factor this contribution stands for
    per_var[o.ref].push_back(cb);
    fac[fi].pad = 1;
    m->has_prog = true;
    return true;
  };
  for (int fi = 0; fi < nf; ++fi) {
    const nuts_factor& f = s->factors[fi];
    fac[fi].pad = 0;
    gathered.clear(); lin_used.clear();
    fbt[fi].n = 0; fbt[fi].pad = 0;
    if (f.nargs < 1 || f.nargs > 4 || f.size < 1) { g_err = "factor with a bad argument count or size"; return false; }
    if (f.dist < 0 || f.dist > NUTS_D_DERIVED) { g_err = "factor with an unknown distribution code"; return false; }
    if (f.dist == NUTS_D_DERIVED) {
      // a derived vector (include/nuts_mi355.h): two internal data vectors behind the spec's own -- the values (written by k_derive
      // before the dense pass) and the seed d logp / d element (written by the dense node that reads the vector) -- become the
      // factor's second and third argument, so that the interpreter sees an ordinary factor with logp 0 and d logp / d term = seed
      if (f.nargs != 1) { g_err = "a derived vector has one argument (its term)"; return false; }
      if ((int)derived.size() >= MAX_DERIVED) { g_err = "too many derived vectors (MAX_DERIVED)"; return false; }
      const int did = s->n_data + 2 * (int)derived.size();
      derived.push_back(fi);
      fac[fi].nargs = 3;
      for (int a = 1; a <= 2; ++a) {
        fac[fi].arg[a] = nuts_term{};
        fac[fi].arg[a].a.kind = NUTS_OP_DATA; fac[fi].arg[a].a.ref = did + (a - 1);
      }
    }
    if (f.dist == NUTS_D_TRUNCNORMAL) {   // the bounds carry no gradient here: lower must be a constant (upper is `konst`)
      const nuts_term& lo = f.arg[3];
      if (f.nargs != 4 || lo.a.kind >= NUTS_OP_VAR || lo.b.kind >= NUTS_OP_VAR || lo.c.kind >= NUTS_OP_VAR) {
        g_err = "TruncatedNormal: the bounds must be constants";
        return false;
      }
    }
    bool owned_already = false;
    if (f.n_instr > 0) m->has_prog = true;
    if (f.n_instr > 0) {
      // ---- a factor with an expression program (include/nuts_mi355.h): one contribution per VARIABLE that occurs anywhere in
      // its arguments or instructions (the device differentiates through the program, model_dev.h factor_eval_prog) ----
      if (f.n_instr > NUTS_MAX_FACTOR_INSTR || f.instr_off < 0 || !s->instrs || (int64_t)f.instr_off + f.n_instr > s->n_instrs) {
        g_err = "factor with a bad expression program (offset / length)"; return false;
      }
      std::vector<const nuts_operand*> ops;
      for (int i = 0; i < f.n_instr; ++i) {
        const nuts_instr& I = s->instrs[f.instr_off + i];
        if (I.op < 0 || I.op > NUTS_E_LAST) { g_err = "expression program with an unknown opcode"; return false; }
        const bool ternary = I.op == NUTS_E_SWITCH || I.op == NUTS_E_CLIP;
        const bool binary = ternary || I.op <= NUTS_E_DIV || (I.op >= NUTS_E_GT && I.op <= NUTS_E_OR) || I.op == NUTS_E_MAXIMUM || I.op == NUTS_E_MINIMUM ||
                            I.op == NUTS_E_POW || I.op == NUTS_E_LOGADDEXP || I.op == NUTS_E_CHECK;
        if ((I.x.kind == NUTS_OP_TMP && (I.x.ref < 0 || I.x.ref >= i)) || (binary && I.y.kind == NUTS_OP_TMP && (I.y.ref < 0 || I.y.ref >= i)) ||
            (ternary && I.z.kind == NUTS_OP_TMP && (I.z.ref < 0 || I.z.ref >= i))) {
          g_err = "expression program: an instruction may only use the results of earlier instructions"; return false;
        }
        ops.push_back(&I.x);
        if (binary) ops.push_back(&I.y);
        if (ternary) ops.push_back(&I.z);
      }
      for (int a = 0; a < f.nargs; ++a)
        for (const nuts_operand* o : {&f.arg[a].a, &f.arg[a].b, &f.arg[a].c}) {
          if (o->kind == NUTS_OP_TMP && (o->ref < 0 || o->ref >= f.n_instr)) { g_err = "factor argument refers to a missing instruction"; return false; }
          ops.push_back(o);
        }
      std::vector<int> seen;
      for (const nuts_operand* o : ops) {
        if (o->kind < NUTS_OP_CONST || o->kind > NUTS_OP_LIN) { g_err = "operand of an unknown kind"; return false; }
        if (o->kind == NUTS_OP_GATHER) { if (!add_gather(fi, *o)) return false; continue; }
        if (o->kind == NUTS_OP_LIN) { if (!add_lin(fi, *o)) return false; continue; }
        if (o->kind == NUTS_OP_DATA) {
          if (o->ref < 0 || o->ref >= s->n_data) { g_err = "factor refers to a missing data vector"; return false; }
          const int64_t ds = s->data[o->ref].size;
          if (ds != 1 && ds != f.size) { g_err = "data vector does not broadcast against its factor"; return false; }
        }
        if (o->kind != NUTS_OP_VAR) continue;
        if (o->ref < 0 || o->ref >= nv) { g_err = "factor refers to a missing variable"; return false; }
        if (std::find(seen.begin(), seen.end(), o->ref) != seen.end()) continue;
        seen.push_back(o->ref);
        const VarDev& v = vars[o->ref];
        if (v.size == f.size) {
          Contrib cb{};
          cb.f = fi; cb.arg = -1; cb.slot = -1; cb.owner = owned_already ? 0 : 1; cb.fast = 0; cb.dist = f.dist; cb.konst = f.konst;
          per_var[o->ref].push_back(cb);
          owned_already = true;
        } else if (v.size == 1) {
          int b = -1;
          for (size_t t = 0; t < bterm_var.size(); ++t) if (bterm_var[t] == o->ref) b = (int)t;
          if (b < 0) {
            if ((int)bterm_var.size() >= MAX_BTERMS) { g_err = "too many scalar variables broadcast against vector factors (MAX_BTERMS)"; return false; }
            b = (int)bterm_var.size();
            bterm_var.push_back(o->ref);
          }
          if (fbt[fi].n >= MAX_FACTOR_BT) { g_err = "too many scalar operands in one factor (MAX_FACTOR_BT)"; return false; }
          fbt[fi].e[fbt[fi].n].arg = -1; fbt[fi].e[fbt[fi].n].slot = -1; fbt[fi].e[fbt[fi].n].bterm = b;
          fbt[fi].n++;
        } else { g_err = "variable does not broadcast against its factor"; return false; }
      }
      if (!owned_already) orphans.push_back(fi);
      if (!finish_gather(fi, !owned_already)) return false;
      continue;
    }
    for (int a = 0; a < f.nargs; ++a) {
      const nuts_operand* ops[3] = {&f.arg[a].a, &f.arg[a].b, &f.arg[a].c};
      for (int sl = 0; sl < 3; ++sl) {
        const nuts_operand& o = *ops[sl];
        if (o.kind == NUTS_OP_TMP) { g_err = "factor argument refers to an instruction but the factor has no program"; return false; }
        if (o.kind == NUTS_OP_GATHER) {
          if (!add_gather(fi, o)) return false;
          continue;
        }
        if (o.kind == NUTS_OP_LIN) {
          if (!add_lin(fi, o)) return false;
          continue;
        }
        if (o.kind < NUTS_OP_CONST || o.kind > NUTS_OP_LIN) { g_err = "operand of an unknown kind"; return false; }
        if (o.kind == NUTS_OP_DATA) {
          if (o.ref < 0 || o.ref >= s->n_data) { g_err = "factor refers to a missing data vector"; return false; }
          const int64_t ds = s->data[o.ref].size;
          if (ds != 1 && ds != f.size) { g_err = "data vector does not broadcast against its factor"; return false; }
        }
        if (o.kind != NUTS_OP_VAR) continue;
        if (o.ref < 0 || o.ref >= nv) { g_err = "factor refers to a missing variable"; return false; }
        const VarDev& v = vars[o.ref];
        if (v.size == f.size) {
          Contrib cb{};
          cb.f = fi; cb.arg = (int16_t)a; cb.slot = (int16_t)sl; cb.owner = owned_already ? 0 : 1;
          // fast form: this operand is the whole argument (a + 0), every other argument folds to a constant
          bool fast = sl == 0 && f.dist != NUTS_D_DERIVED;   // (a derived vector's seed is a data operand the engine adds)
          for (int a2 = 0; a2 < f.nargs && fast; ++a2) {
            const nuts_term& t = f.arg[a2];
            const bool bc_zero = (t.b.kind == NUTS_OP_CONST && t.b.c == 0.0) || (t.c.kind == NUTS_OP_CONST && t.c.c == 0.0);
            if (a2 == a) fast = bc_zero && t.b.kind == NUTS_OP_CONST && t.c.kind == NUTS_OP_CONST;
            else {
              fast = t.a.kind == NUTS_OP_CONST && t.b.kind == NUTS_OP_CONST && t.c.kind == NUTS_OP_CONST;
              cb.p[a2] = t.a.c + t.b.c * t.c.c;
            }
          }
          cb.fast = fast ? 1 : 0; cb.dist = f.dist; cb.konst = f.konst;
          if (fast && a == 0 && f.dist == NUTS_D_NORMAL && cb.p[2] > 0) { cb.fast = 2; const double sg = cb.p[2]; cb.p[2] = 1.0 / sg; cb.p[3] = std::log(sg); }
          if (fast && a == 0 && f.dist == NUTS_D_HALFNORMAL && cb.p[1] > 0) { cb.fast = 2; const double sg = cb.p[1]; cb.p[2] = 1.0 / sg; cb.p[3] = std::log(sg); }
          per_var[o.ref].push_back(cb);
          owned_already = true;
        } else if (v.size == 1) {
          int b = -1;
          for (size_t t = 0; t < bterm_var.size(); ++t) if (bterm_var[t] == o.ref) b = (int)t;
          if (b < 0) {
            if ((int)bterm_var.size() >= MAX_BTERMS) { g_err = "too many scalar variables broadcast against vector factors (MAX_BTERMS)"; return false; }
            b = (int)bterm_var.size();
            bterm_var.push_back(o.ref);
          }
          if (fbt[fi].n >= MAX_FACTOR_BT) { g_err = "too many scalar operands in one factor (MAX_FACTOR_BT)"; return false; }
          fbt[fi].e[fbt[fi].n].arg = (int16_t)a; fbt[fi].e[fbt[fi].n].slot = (int16_t)sl; fbt[fi].e[fbt[fi].n].bterm = b;
          fbt[fi].n++;
        } else { g_err = "variable does not broadcast against its factor"; return false; }
      }
    }
    if (!owned_already) orphans.push_back(fi);
    if (!finish_gather(fi, !owned_already)) return false;
  }
  // peephole: untransformed vector variable whose only contribution is its own constant-parameter Normal prior
  for (int k = 0; k < nv; ++k) {
    if (vars[k].transform != NUTS_TR_NONE || vars[k].deferred || per_var[k].size() != 1) continue;
    const Contrib& cb = per_var[k][0];
    if (cb.fast != 2 || !cb.owner || cb.dist != NUTS_D_NORMAL || cb.arg != 0) continue;
    vars[k].normal_prior = 1;
    vars[k].np_mu = cb.p[1];
    vars[k].np_inv_var = cb.p[2] * cb.p[2];                       // p[2] = 1/sigma, p[3] = log sigma
    vars[k].np_lognorm = 0.91893853320467274178 + cb.p[3];
  }
  std::vector<int32_t> cptr(nv + 1, 0);
  std::vector<Contrib> contrib;
  for (int k = 0; k < nv; ++k) {
    cptr[k] = (int32_t)contrib.size();
    contrib.insert(contrib.end(), per_var[k].begin(), per_var[k].end());
  }
  cptr[nv] = (int32_t)contrib.size();
  std::vector<int32_t> deferred;   // (element, variable) pairs
  for (int k = 0; k < nv; ++k)
    if (vars[k].deferred) {
      vars[k].def_base = (int32_t)deferred.size() / 2;
      for (int i = 0; i < vars[k].size; ++i) { deferred.push_back(vars[k].offset + i); deferred.push_back(k); }
    }
  if ((int)deferred.size() / 2 > MAX_DEFERRED) { g_err = "too many scalar / hyper-parameter elements (MAX_DEFERRED)"; return false; }
  md.n_bterms = (int)bterm_var.size(); md.n_orphans = (int)orphans.size(); md.n_deferred = (int)deferred.size() / 2;
  // the orphans the adjoint sweep covers go behind the others: k_gsweep accounts their log-density and their scalars' adjoints, kernel
  // B walks the first n_orphans_b only (NUTS_GSWEEP_LP = 0: kernel B walks all of them, as before -- A/B)
  md.n_orphans_b = md.n_orphans;
  if (env_int("NUTS_GSWEEP_LP", 1) != 0) {
    auto swept = [&](int fi) { for (const GSweepFactor& g : gsf) if (g.f == fi) return true; return false; };
    std::stable_partition(orphans.begin(), orphans.end(), [&](int fi) { return !swept(fi); });
    md.n_orphans_b = 0;
    for (int fi : orphans) if (!swept(fi)) md.n_orphans_b++;
    for (GSweepFactor& g : gsf) g.orphan = std::find(orphans.begin() + md.n_orphans_b, orphans.end(), g.f) != orphans.end() ? 1 : 0;
  }
  m->orphan_elems = 0; m->factor_elems = 0;
  for (int o = 0; o < md.n_orphans_b; ++o) m->orphan_elems += s->factors[orphans[o]].size;
  for (int fi = 0; fi < nf; ++fi) m->factor_elems += s->factors[fi].size;
  md.orphans = m->keep(dev_upload(orphans.data(), orphans.size()));
  md.deferred_g = m->keep(dev_upload(deferred.data(), deferred.size()));
  md.def_loc = m->keep(dev_alloc<double>(2 * 4 * (size_t)MAX_DEFERRED));
  hipMemset(md.def_loc, 0, 2 * 4 * (size_t)MAX_DEFERRED * sizeof(double));
  // lean control path (kernels.h): no broadcast terms, and either
  // (a) a logit node (and no MvNormal node): the deferred elements are its mu / sigma, whose gradient is the local part + a
  //     cross-workgroup sum of the row pass, and possibly scalars with factors of their own, whose gradient IS the local part
  //     (a scalar that broadcasts into a vector factor would be a broadcast term) -- one control thread each, or
  md.lean_ok = 0;
  // (b) no deferred element at all and an MvNormal node: the control work is sums and tree logic only
  if (md.n_bterms == 0 && env_int("NUTS_LEAN", 1)) {
    if (s->rows_N > 0 && s->mvn_k <= 0) {
      md.lean_ok = md.n_deferred <= GA_AUX_MAXDEF ? 1 : 0;
      // NUTS_LEAN_STRICT=1: only the round-3 shape of the lean path (mu / sigma the only deferred elements) -- A/B and tests
      if (env_int("NUTS_LEAN_STRICT", 0))
        for (int k = 0; k < nv; ++k) if (vars[k].deferred && k != s->rows_mu && k != s->rows_sigma) md.lean_ok = 0;
      md.lg.def_mu = vars[s->rows_mu].def_base; md.lg.def_sigma = vars[s->rows_sigma].def_base;
    } else if (s->rows_N <= 0 && s->mvn_k > 0 && md.n_deferred == 0 && s->glm_N <= 0 && s->n_lins <= 0) {
      md.lean_ok = 1;
    }
  }
  // one-launch row passes (rows_ga_kernel.h, rows_gb_kernel.h): the row stream + the per-group finish of the z elements, which
  // needs z ~ Normal with constant parameters as its only factor (closed form in the group's own workgroup).  Everything else:
  //   ga_struct_ok = 1  exactly mu ~ Normal, sigma ~ HalfNormal (log-transformed or not) with constant parameters and nothing
  //                     else: closed forms in workgroup 0's tail (ga_def_local), no auxiliary workgroup -- the benchmark's model;
  //   ga_struct_ok = 2  any other hyper-prior, further variables and factors (no expression programs / gathers): auxiliary
  //                     workgroups run the interpreter over every element that is not a z element (rows_aux.h).
  // NUTS_GA_AUX=1 sends the closed-form model through the auxiliary workgroups too (tests: both routes, same model);
  // NUTS_GA_AUX=0 keeps models that would need them off the one-launch passes (A/B against the general path).
  m->ga_struct_ok = 0;
  if (md.lean_ok && s->rows_N > 0 && !m->has_prog) {
    const int km = s->rows_mu, ks = s->rows_sigma, kz = s->rows_z;
    auto only_prior = [&](int k, int dist) {
      return per_var[k].size() == 1 && per_var[k][0].fast == 2 && per_var[k][0].owner && per_var[k][0].arg == 0 && per_var[k][0].dist == dist;
    };
    if (vars[kz].normal_prior && vars[km].transform == NUTS_TR_NONE && (vars[ks].transform == NUTS_TR_LOG || vars[ks].transform == NUTS_TR_NONE)) {
      md.lg.z_np_mu = vars[kz].np_mu; md.lg.z_np_inv_var = vars[kz].np_inv_var; md.lg.z_np_lognorm = vars[kz].np_lognorm;
      const bool closed = nv == 3 && orphans.empty() && only_prior(km, NUTS_D_NORMAL) && only_prior(ks, NUTS_D_HALFNORMAL);
      const int aux_opt = env_int("NUTS_GA_AUX", -1);
      if (closed && aux_opt != 1) {
        m->ga_struct_ok = 1;
        const Contrib& cm = per_var[km][0];
        md.lg.mu_c[0] = cm.p[1]; md.lg.mu_c[1] = cm.p[2]; md.lg.mu_c[2] = cm.p[3];
        const Contrib& cs = per_var[ks][0];
        md.lg.sg_c[0] = cs.p[2]; md.lg.sg_c[1] = cs.p[3];
      } else if (aux_opt != 0) m->ga_struct_ok = 2;
    }
  }
  // pack the interpreter's tables into one blob (copied into LDS by kernels B and C)
  std::vector<char> blob;
  auto put = [&](const void* src, size_t bytes) {
    const size_t off = (blob.size() + 15) & ~(size_t)15;
    blob.resize(off + std::max<size_t>(bytes, 16), 0);
    if (bytes) std::memcpy(blob.data() + off, src, bytes);
    return (int32_t)off;
  };
  md.po_vars = put(vars.data(), vars.size() * sizeof(VarDev));
  md.po_cptr = put(cptr.data(), cptr.size() * sizeof(int32_t));
  md.po_contrib = put(contrib.data(), contrib.size() * sizeof(Contrib));
  md.po_factors = put(fac.data(), (size_t)nf * sizeof(nuts_factor));
  md.csr = m->keep(dev_upload(csr.data(), csr.size()));
  md.po_fbt = put(fbt.data(), (size_t)nf * sizeof(FactorBT));
  md.po_btvar = put(bterm_var.data(), bterm_var.size() * sizeof(int32_t));
  // data table: the spec's vectors, then (values, seed) of every derived vector behind the spec's pool
  std::vector<nuts_data_ref> drefs(s->data, s->data + s->n_data);
  m->pool_extra = 0;
  md.n_derived = (int32_t)derived.size();
  for (size_t t = 0; t < derived.size(); ++t) {
    const int64_t sz = s->factors[derived[t]].size;
    md.derived_f[t] = derived[t];
    md.derived_off[t] = s->data_pool_len + m->pool_extra;
    drefs.push_back(nuts_data_ref{s->data_pool_len + m->pool_extra, sz});
    drefs.push_back(nuts_data_ref{s->data_pool_len + m->pool_extra + sz, sz});
    m->pool_extra += 2 * sz;
  }
  md.po_data = put(drefs.data(), drefs.size() * sizeof(nuts_data_ref));
  md.po_deferred = put(deferred.data(), deferred.size() * sizeof(int32_t));
  md.po_instrs = put(s->instrs, s->instrs ? (size_t)std::max(s->n_instrs, 0) * sizeof(nuts_instr) : 0);
  md.n_gsf = (int32_t)gsf.size(); md.n_gs_elems = gs_elems;
  md.po_gsf = put(gsf.data(), gsf.size() * sizeof(GSweepFactor));
  md.po_gslot = put(gslots.data(), gslots.size() * sizeof(GSlot));
  md.adj = nullptr; md.adj_red = nullptr; md.glong = nullptr;
  md.n_glong = (int32_t)glong.size(); md.glong_pad = 0;
  if (!glong.empty()) {
    md.glong = m->keep(dev_upload(glong.data(), glong.size()));
    md.adj_red = m->keep(dev_alloc<double>(glong.size()));
    if (!md.glong || !md.adj_red) { g_err = "device allocation failed (gathered adjoints)"; return false; }
    hipMemset(md.adj_red, 0, glong.size() * sizeof(double));
  }
  if (adj_len > 0) {
    md.adj = m->keep(dev_alloc<double>((size_t)adj_len));
    if (!md.adj) { g_err = "device allocation failed (gathered adjoints)"; return false; }
    hipMemset(md.adj, 0, (size_t)adj_len * sizeof(double));
  }
  blob.resize((blob.size() + 15) & ~(size_t)15, 0);
  md.prog_bytes = (int32_t)blob.size();
  // the sweep's arrays in LDS (k_gsweep_lds, one wave per workgroup; the interpreter's tables and the broadcast accumulators share the
  // launch's dynamic LDS) -- when every element's wave is resident at once: with LDS as the limit a second round of workgroups costs
  // more than the scratch version's trips to the L2 (measured: a 79-row block at 100 000 elements 363 us against 190)
  md.gs_lds_rows = 0; md.gs_lds_bytes = 0;
  if (!gsf.empty() && env_int("NUTS_GSWEEP_LDS", 1) != 0) {
    int max_instr = 0, max_slots = 0;
    for (const GSweepFactor& g : gsf) { max_instr = std::max(max_instr, s->factors[g.f].n_instr); max_slots = std::max(max_slots, g.n_slots); }
    const int rows = std::max(2 * max_instr + max_slots, 1) + md.n_bterms;
    const int64_t bytes = (int64_t)rows * 64 * 8 + (int64_t)md.prog_bytes + 64;
    const int64_t waves = ((int64_t)gs_elems + 63) / 64;
    const int64_t per_cu = std::min<int64_t>(32, (160 * 1024) / std::max<int64_t>(bytes, 1));
    if (bytes <= 60 * 1024 && (waves <= 256 * per_cu || env_int("NUTS_GSWEEP_LDS", 1) == 2)) { md.gs_lds_rows = rows; md.gs_lds_bytes = (int32_t)bytes; }
  }
  md.n_gs_blocks = gsf.empty() ? 0 : (md.gs_lds_rows > 0 ? std::max(1, std::min(8192, (gs_elems + 63) / 64)) : std::max(1, std::min(2048, (gs_elems + 255) / 256)));
  md.gs_part = nullptr;
  if (md.n_gs_blocks > 0) {
    md.gs_part = m->keep(dev_alloc<double>((size_t)md.n_gs_blocks * (1 + MAX_BTERMS)));
    if (!md.gs_part) { g_err = "device allocation failed (gathered adjoints)"; return false; }
    hipMemset(md.gs_part, 0, (size_t)md.n_gs_blocks * (1 + MAX_BTERMS) * sizeof(double));
  }
  md.prog = m->keep(dev_upload(blob.data(), blob.size()));
  m->gsf_host = gsf; m->gslots_host = gslots; m->drefs_host = drefs; m->fbt_host = fbt; m->bterm_var_host = bterm_var;
  return true;
}

// The adjoint sweep with resolved operands (model_dev.h SwFactor, kernels.h k_gsweep_fast): every distinct leaf operand of a swept
// factor becomes an entry of the factor's leaf list -- what to load from where, as a pointer --, the programs and arguments are
// rewritten to refer to those entries, and the slots' adjoint offsets sit in a table of their own.  Runs once the data pool and the
// predictors' buffers exist.  The kernel reads the tables through the scalar unit (constant address space), one wave per 64-element
// block of ONE factor (SwFactor.blk0); taken whenever a thread's LDS column fits (NUTS_GSWEEP_FAST = 0: the generic sweeps).
static bool build_sweep_fast(nuts_model* m, const nuts_model_spec* s, const std::vector<VarDev>& vars) {
  ModelDev& md = m->md;
  md.n_swf = 0; md.sw_bytes = 0; md.sw_rows = 0; md.sw_blob = nullptr;
  const int opt = env_int("NUTS_GSWEEP_FAST", 1);
  if (m->gsf_host.empty() || opt == 0 || !md.pool) return true;
  std::vector<SwFactor> swf;
  std::vector<SwLeaf> leaves;
  std::vector<nuts_instr> instrs;
  std::vector<int64_t> slot_off;
  int max_leaves = 1, max_slots = 1, max_instr = 1;
  for (const GSweepFactor& g : m->gsf_host) {
    const nuts_factor& f = s->factors[g.f];
    SwFactor F{};
    F.f = g.f; F.elem0 = g.elem0; F.size = f.size; F.orphan = g.orphan;
    F.n_instr = f.n_instr; F.nargs = f.nargs; F.dist = f.dist; F.konst = f.konst;
    F.slot0 = (int32_t)slot_off.size(); F.n_slots = g.n_slots;
    for (int sl = 0; sl < g.n_slots; ++sl) slot_off.push_back(m->gslots_host[g.slot0 + sl].adj_off);
    struct Key { int kind, ref, did; };
    std::vector<Key> keys;          // leaves in order of first use
    std::vector<SwLeaf> lf;
    std::vector<int> push_to;       // per leaf: >= 0 slot, <= -2 broadcast accumulator, -1 nothing
    auto leaf_of = [&](const nuts_operand& o) -> int {
      const int did = (o.kind == NUTS_OP_GATHER || o.kind == NUTS_OP_LIN) ? (int)o.c : 0;
      for (size_t i = 0; i < keys.size(); ++i) if (keys[i].kind == o.kind && keys[i].ref == o.ref && keys[i].did == did) return (int)i;
      SwLeaf L{};
      int pt = -1;
      if (o.kind == NUTS_OP_DATA) {
        const nuts_data_ref r = m->drefs_host[o.ref];
        L.kind = SWL_DATA; L.bcast = r.size > 1 ? 0 : 1; L.ptr = md.pool + r.offset;
      } else if (o.kind == NUTS_OP_LIN) {
        const LinDev& D = m->lins_host[o.ref];
        L.kind = SWL_DATA; L.bcast = D.N > 1 ? 0 : 1; L.ptr = D.eta + (int64_t)did * D.N;
        for (int sl = 0; sl < g.n_slots; ++sl) { const GSlot& gs = m->gslots_host[g.slot0 + sl]; if (gs.var == -1 - o.ref && gs.did == did) { pt = sl; break; } }
      } else if (o.kind == NUTS_OP_GATHER) {
        const nuts_data_ref r = m->drefs_host[did];
        const VarDev& v = vars[o.ref];
        L.kind = SWL_GATHER; L.bcast = 0; L.ptr = md.pool + r.offset; L.voff = v.offset; L.transform = v.transform; L.lower = v.lower; L.upper = v.upper;
        for (int sl = 0; sl < g.n_slots; ++sl) { const GSlot& gs = m->gslots_host[g.slot0 + sl]; if (gs.var == o.ref && gs.did == did) { pt = sl; break; } }
      } else {   // NUTS_OP_VAR
        const VarDev& v = vars[o.ref];
        L.kind = SWL_VAR; L.bcast = v.size > 1 ? 0 : 1; L.ptr = md.pool; L.voff = v.offset; L.transform = v.transform; L.lower = v.lower; L.upper = v.upper;
        const FactorBT& bt = m->fbt_host[g.f];
        for (int b = 0; b < bt.n; ++b) if (m->bterm_var_host[bt.e[b].bterm] == o.ref) { pt = -2 - bt.e[b].bterm; break; }
      }
      keys.push_back(Key{o.kind, o.ref, did}); lf.push_back(L); push_to.push_back(pt);
      return (int)keys.size() - 1;
    };
    auto rewrite = [&](nuts_operand& o) {
      if (o.kind == NUTS_OP_CONST || o.kind == NUTS_OP_TMP) return;
      if (o.kind != NUTS_OP_DATA && o.kind != NUTS_OP_VAR && o.kind != NUTS_OP_GATHER && o.kind != NUTS_OP_LIN) { o.kind = NUTS_OP_CONST; o.c = 0.0; return; }
      const int l = leaf_of(o);
      o.kind = SW_LEAF; o.ref = l; o.c = (double)push_to[l];
    };
    F.instr0 = (int32_t)instrs.size();
    for (int i = 0; i < f.n_instr; ++i) {
      nuts_instr I = s->instrs[f.instr_off + i];
      rewrite(I.x); rewrite(I.y); rewrite(I.z);
      instrs.push_back(I);
    }
    for (int k = 0; k < 4; ++k) {
      nuts_operand a{}, b{}, c{};
      if (k < f.nargs) { a = f.arg[k].a; b = f.arg[k].b; c = f.arg[k].c; rewrite(a); rewrite(b); rewrite(c); }
      F.arg[k][0] = a; F.arg[k][1] = b; F.arg[k][2] = c;
    }
    // the leaves that need the position first (kernel: one pass of direct loads over all leaves, one of position loads over these)
    std::vector<int> order, where(lf.size());
    for (size_t i = 0; i < lf.size(); ++i) if (lf[i].kind != SWL_DATA) order.push_back((int)i);
    F.n2 = (int32_t)order.size();
    for (size_t i = 0; i < lf.size(); ++i) if (lf[i].kind == SWL_DATA) order.push_back((int)i);
    for (size_t i = 0; i < order.size(); ++i) where[order[i]] = (int)i;
    auto remap = [&](nuts_operand& o) { if (o.kind == SW_LEAF) o.ref = where[o.ref]; };
    for (int i = 0; i < f.n_instr; ++i) { nuts_instr& I = instrs[F.instr0 + i]; remap(I.x); remap(I.y); remap(I.z); }
    for (int k = 0; k < 4; ++k) for (int u = 0; u < 3; ++u) remap(F.arg[k][u]);
    F.leaf0 = (int32_t)leaves.size(); F.n_leaves = (int32_t)lf.size();
    for (int i : order) leaves.push_back(lf[i]);
    max_leaves = std::max(max_leaves, F.n_leaves); max_slots = std::max(max_slots, F.n_slots); max_instr = std::max(max_instr, F.n_instr);
    swf.push_back(F);
  }
  int32_t n_blocks = 0;
  for (SwFactor& F : swf) { F.blk0 = n_blocks; n_blocks += (F.size + 63) / 64; }
  std::vector<char> blob;
  auto put = [&](const void* src, size_t bytes) {
    const size_t off = (blob.size() + 63) & ~(size_t)63;
    blob.resize(off + std::max<size_t>(bytes, 64), 0);
    if (bytes) std::memcpy(blob.data() + off, src, bytes);
    return (int32_t)off;
  };
  put(swf.data(), swf.size() * sizeof(SwFactor));
  const int32_t po_leaf = put(leaves.data(), leaves.size() * sizeof(SwLeaf));
  const int32_t po_instr = put(instrs.data(), instrs.size() * sizeof(nuts_instr));
  const int32_t po_slot = put(slot_off.data(), slot_off.size() * sizeof(int64_t));
  blob.resize((blob.size() + 63) & ~(size_t)63, 0);
  // the launch's LDS: a column per thread of broadcast accumulators, leaf values, slot adjoints, instruction values and adjoints
  const int rows = md.n_bterms + max_leaves + max_slots + 2 * max_instr;
  const int64_t bytes = (int64_t)rows * 64 * 8;
  if (bytes > 64 * 1024) return true;   // (the generic sweeps stay: a program of more than ~ 60 instructions)
  md.sw_blob = m->keep(dev_upload(blob.data(), blob.size()));
  if (!md.sw_blob) { g_err = "device allocation failed (resolved-operand sweep)"; return false; }
  md.n_swf = (int32_t)swf.size(); md.sw_bytes = (int32_t)blob.size(); md.sw_rows = rows;
  md.sw_po_leaf = po_leaf; md.sw_po_instr = po_instr; md.sw_po_slot = po_slot;
  md.sw_max_leaves = max_leaves; md.sw_max_slots = max_slots; md.sw_max_instr = max_instr; md.sw_blocks = n_blocks;
  // (one wave per workgroup and per block; grid-stride beyond 8192 -- the records of gs_part were sized for the generic sweeps' grids)
  const int nb = std::max(1, std::min(8192, n_blocks));
  if (nb > md.n_gs_blocks) {
    md.gs_part = m->keep(dev_alloc<double>((size_t)nb * (1 + MAX_BTERMS)));
    if (!md.gs_part) { g_err = "device allocation failed (gathered adjoints)"; return false; }
    hipMemset(md.gs_part, 0, (size_t)nb * (1 + MAX_BTERMS) * sizeof(double));
  }
  md.n_gs_blocks = nb;
  return true;
}

// Linear predictors (dense node 5, lin_kernel.h): X transposed, the predictors' buffers, where every coefficient lives and which
// partial sums make up its gradient.  Runs once the data pool exists (a coefficient may be an element of a derived vector).
static bool build_lins(nuts_model* m, const nuts_model_spec* s, const std::vector<VarDev>& vars) {
  ModelDev& md = m->md;
  md.n_lins = 0; md.n_lin_targets = 0; md.lins = nullptr; md.lin_targets = nullptr; md.lin_srcs = nullptr; md.lin_gdense = nullptr;
  if (s->n_lins <= 0) return true;
  const int n = md.n;
  md.lin_gdense = m->keep(dev_alloc<double>((size_t)n));
  if (!md.lin_gdense) { g_err = "device allocation failed (linear predictors)"; return false; }
  hipMemset(md.lin_gdense, 0, (size_t)n * sizeof(double));
  struct Src { int64_t key; LinSrc src; double* dst; };
  std::vector<Src> srcs;
  m->lins_host.assign(s->n_lins, LinDev{});
  for (int l = 0; l < s->n_lins; ++l) {
    const nuts_lin& L = s->lins[l];
    LinDev& D = m->lins_host[l];
    D.N = L.N; D.P = L.P; D.K = L.K;
    D.nchunk = L.N > 1 ? (int32_t)((L.N + LIN_CHUNK - 1) / LIN_CHUNK) : 1;
    std::vector<double> xt((size_t)L.N * L.P);
    for (int64_t i = 0; i < L.N; ++i)
      for (int p = 0; p < L.P; ++p) xt[(size_t)p * L.N + i] = L.X[(size_t)i * L.P + p];
    D.Xt = m->keep(dev_upload(xt.data(), xt.size()));
    D.eta = m->keep(dev_alloc<double>((size_t)L.K * L.N));
    const size_t npart = L.N > 1 ? (size_t)L.K * L.P * D.nchunk : (size_t)L.K;
    D.part = m->keep(dev_alloc<double>(npart));
    if (!D.Xt || !D.eta || !D.part) { g_err = "device allocation failed (linear predictors)"; return false; }
    hipMemset(D.eta, 0, (size_t)L.K * L.N * sizeof(double));
    hipMemset(D.part, 0, npart * sizeof(double));
    std::vector<int32_t> coef((size_t)L.K * L.P);
    for (int k = 0; k < L.K; ++k) {
      LinCol& C = D.col[k];
      const auto& uses = m->lin_uses[l][k];
      if (uses.size() > LIN_MAXUSE) { g_err = "a linear predictor column is read by too many factors (LIN_MAXUSE)"; return false; }
      C.n_use = (int32_t)uses.size();
      for (size_t u = 0; u < uses.size(); ++u) { C.adj_off[u] = uses[u].adj_off; C.use_size[u] = uses[u].size; }
      const int64_t last = (int64_t)L.off[k] + (int64_t)(L.P - 1) * L.stride[k];
      if (L.off[k] < 0 || L.stride[k] < 0 || (L.stride[k] == 0 && L.P > 1)) { g_err = "linear predictor: bad coefficient offset / stride"; return false; }
      double* dst_base = nullptr; int64_t cbase = 0;
      if (L.var[k] >= 0) {
        if (L.var[k] >= md.n_vars) { g_err = "linear predictor refers to a missing variable"; return false; }
        const VarDev& v = vars[L.var[k]];
        if (last >= v.size) { g_err = "linear predictor: coefficients beyond the end of their variable"; return false; }
        C.transform = v.transform; C.lower = v.lower; C.upper = v.upper;
        cbase = v.offset; dst_base = md.lin_gdense + v.offset;
      } else {
        const int fi = -(L.var[k] + 1);
        int t = -1;
        for (int u = 0; u < md.n_derived; ++u) if (md.derived_f[u] == fi) t = u;
        if (t < 0) { g_err = "linear predictor: coefficients must be a variable or a NUTS_D_DERIVED factor"; return false; }
        const int64_t sz = s->factors[fi].size;
        if (last >= sz) { g_err = "linear predictor: coefficients beyond the end of their derived vector"; return false; }
        C.transform = -1; C.lower = 0.0; C.upper = 0.0;
        cbase = md.derived_off[t]; dst_base = const_cast<double*>(md.pool) + md.derived_off[t] + sz;   // (the vector's seed follows its values)
      }
      for (int p = 0; p < L.P; ++p) {
        const int64_t e = (int64_t)L.off[k] + (int64_t)p * L.stride[k];
        coef[(size_t)k * L.P + p] = (int32_t)(cbase + e);
        if (C.n_use > 0) srcs.push_back(Src{(int64_t)reinterpret_cast<intptr_t>(dst_base + e), LinSrc{l, k, p, 0}, dst_base + e});
      }
    }
    D.coef = m->keep(dev_upload(coef.data(), coef.size()));
    m->alg_bytes += 16 * L.N * (int64_t)L.P;   // X once forwards, once backwards
  }
  // a coefficient that several columns share (the same vector against two matrices) adds their partial sums up, in table order
  std::stable_sort(srcs.begin(), srcs.end(), [](const Src& a, const Src& b) { return a.key < b.key; });
  std::vector<LinTarget> targets;
  std::vector<LinSrc> flat;
  for (size_t i = 0; i < srcs.size(); ++i) {
    if (i == 0 || srcs[i].key != srcs[i - 1].key) targets.push_back(LinTarget{srcs[i].dst, (int32_t)flat.size(), 0});
    targets.back().n_src++;
    flat.push_back(srcs[i].src);
  }
  md.n_lins = s->n_lins; md.n_lin_targets = (int32_t)targets.size();
  md.lins = m->keep(dev_upload(m->lins_host.data(), m->lins_host.size()));
  md.lin_targets = m->keep(dev_upload(targets.data(), targets.size()));
  md.lin_srcs = m->keep(dev_upload(flat.data(), flat.size()));
  return true;
}

extern "C" nuts_model* nuts_model_create(const nuts_model_spec* s) {
  if (!s || s->n_vars <= 0) { g_err = "empty model spec"; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device visible: libnuts_mi355 requires an MI355X (gfx950); there is no CPU fallback";
    return nullptr;
  }
  auto* m = new nuts_model();
  ModelDev& md = m->md;
  HIPCHK_NULL(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  m->own_stream = m->stream;
  int n = 0;
  for (int i = 0; i < s->n_vars; ++i) n = std::max(n, s->vars[i].offset + s->vars[i].size);
  md.n = n; md.n_vars = s->n_vars; md.n_factors = s->n_factors; md.n_data = s->n_data;
  md.fdead_mode = 0;
  md.fdead = m->keep(dev_alloc<int32_t>((size_t)std::max(s->n_factors, 1)));
  if (md.fdead) hipMemset(md.fdead, 0, (size_t)std::max(s->n_factors, 1) * sizeof(int32_t));
  std::vector<VarDev> vars;
  if (!compile_spec(m, s, vars)) { nuts_model_destroy(m); return nullptr; }
  {   // the spec's data pool, then the engine's own vectors (derived values / seeds), zeroed
    double* pool = m->keep(dev_alloc<double>((size_t)std::max<int64_t>(s->data_pool_len + m->pool_extra, 1)));
    if (pool) {
      hipMemset(pool, 0, (size_t)std::max<int64_t>(s->data_pool_len + m->pool_extra, 1) * sizeof(double));
      if (s->data_pool_len > 0) hipMemcpy(pool, s->data_pool, (size_t)s->data_pool_len * sizeof(double), hipMemcpyHostToDevice);
    }
    md.pool = pool;
    md.pool_len = (int32_t)std::min<int64_t>(s->data_pool_len + m->pool_extra, INT32_MAX);
  }
  m->data_refs.assign(s->data, s->data + s->n_data);
  m->ept = n <= 65536 ? 1 : (n <= 262144 ? 4 : 16);
  md.nblk = (n + VEC_THREADS * m->ept - 1) / (VEC_THREADS * m->ept);
  // (round 6) ... and enough workgroups for the factors without an owning variable, whose elements kernel B walks grid-stride: a
  // 100 000-row likelihood over fifteen coefficients was ONE workgroup's work (36 ms per gradient; four elements per thread now)
  if (m->orphan_elems > 4 * VEC_THREADS * (int64_t)md.nblk)
    md.nblk = (int)std::min<int64_t>(256, (m->orphan_elems + 4 * VEC_THREADS - 1) / (4 * VEC_THREADS));
  md.tick_j = env_int("NUTS_TICK_J", -1);
  md.ticks = m->keep(dev_alloc<long long>(64));
  hipMemset(md.ticks, 0, 64 * sizeof(long long));
  md.part_stride = PART_STRIDE;
  m->vector_one_xcd = env_int("NUTS_VECTOR_ONE_XCD", 0);
  md.part = m->keep(dev_alloc<double>((size_t)md.nblk * PART_STRIDE));
  hipMemset(md.part, 0, (size_t)md.nblk * PART_STRIDE * sizeof(double));
  m->q_dev = m->keep(dev_alloc<double>(n));
  m->g_dev = m->keep(dev_alloc<double>(n));
  m->lp_dev = m->keep(dev_alloc<double>(2));
  HIPCHK_NULL(hipHostMalloc((void**)&m->host_pin, (2 * (size_t)n + 2) * sizeof(double), hipHostMallocDefault));
  m->alg_bytes = 0;

  hipDeviceProp_t prop;
  int dev = 0;
  hipGetDevice(&dev);
  hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

  if (s->rows_N > 0) {
    const int D = s->rows_D;
    if (D < 1 || D > LOGIT_MAXD) { g_err = "logit rows: 1 <= D <= 8 covariates"; nuts_model_destroy(m); return nullptr; }
    // (D = 2, 4, 8 have every schedule; the other widths run the span-partitioned pass and the group-block pass, one element per
    // thread in kernel B)
    const bool d_pow2 = D == 8 || D == 4 || D == 2;
    if (!d_pow2 && m->ept != 1) { g_err = "logit rows: D must be 2, 4 or 8 for models beyond 65 536 parameters"; nuts_model_destroy(m); return nullptr; }
    RowsDev& lg = md.lg;
    md.has_logit = 1;
    // launch geometry (tunable for experiments; defaults chosen from measurements)
    m->rows_rpl = 2;
    m->rows_alternate = env_int("NUTS_ROWS_ALTERNATE", 1) ? 1 : 0;
    m->rows_occ = 4;
    // 16 waves per CU are resident at a time (4 per SIMD at 113 VGPRs); two such sets of shorter waves balance the
    // tail better than one (measured with the folded control: 61.3 us per pass vs 63.0 us at 16, 62.8 at 48, 65.8 at 64)
    // -- for passes long enough to give every wave a few spans (C2-L: 4.8 per wave); shorter, cache-resident passes are
    // better off with one set (R = 800 rows per group: 23.5 us at 16 vs 26.2 us at 32)
    int wpc = env_int("NUTS_ROWS_WAVES_PER_CU", 0);
    if (wpc <= 0) {
      const int64_t spans = (s->rows_N + (int64_t)WAVE * m->rows_rpl - 1) / ((int64_t)WAVE * m->rows_rpl);
      wpc = spans >= (int64_t)cus * 32 * 4 ? 32 : 16;
    }
    const int SPAN = WAVE * m->rows_rpl;
    lg.N = s->rows_N; lg.D = D; lg.G = s->rows_G;
    lg.Npad = (lg.N + SPAN - 1) / SPAN * SPAN;
    lg.n_spans = lg.Npad / SPAN;
    const nuts_var &vmu = s->vars[s->rows_mu], &vsg = s->vars[s->rows_sigma], &vz = s->vars[s->rows_z];
    if (vmu.size != D || vsg.size != D || vz.size != (int64_t)lg.G * D || vmu.transform != NUTS_TR_NONE ||
        vz.transform != NUTS_TR_NONE || !(vsg.transform == NUTS_TR_NONE || vsg.transform == NUTS_TR_LOG)) {
      g_err = "logit rows: mu/sigma/z shapes or transforms unsupported"; nuts_model_destroy(m); return nullptr;
    }
    lg.off_mu = vmu.offset; lg.off_sigma = vsg.offset; lg.off_z = vz.offset; lg.sigma_tr = vsg.transform;
    lg.var_mu = s->rows_mu; lg.var_sigma = s->rows_sigma; lg.var_z = s->rows_z;
    std::vector<int8_t> yy;
    std::vector<int64_t> gptr(lg.G + 1, 0);
    const int32_t* gid = s->rows_gid;
    for (int64_t i = 0; i < lg.N; ++i) {
      if (i > 0 && gid[i] < gid[i - 1]) { g_err = "logit rows: group ids must be sorted"; nuts_model_destroy(m); return nullptr; }
      if (gid[i] < 0 || gid[i] >= lg.G) { g_err = "logit rows: group id out of range"; nuts_model_destroy(m); return nullptr; }
      gptr[gid[i] + 1] = i + 1;
    }
    for (int g = 0; g < lg.G; ++g) gptr[g + 1] = std::max(gptr[g + 1], gptr[g]);  // empty groups
    lg.gptr = m->keep(dev_upload(gptr.data(), gptr.size()));
    // ---- group-aligned row pass (rows_ga_kernel.h)?  NUTS_ROWS_GA: 0 never, 1 when the shape suits it (default), 2 whenever
    // the model structure allows it (tests: ragged / empty / tiny groups through the same kernel) ----
    {
      const int want = (s->rows_opts & NUTS_ROWS_NO_GROUP_ALIGNED) ? 0 : env_int("NUTS_ROWS_GA", 1);
      std::vector<int32_t> tile0(lg.G + 1, 0);
      int64_t maxT = 0;
      for (int g = 0; g < lg.G; ++g) {
        const int64_t T = (gptr[g + 1] - gptr[g] + SPAN - 1) / SPAN;
        maxT = std::max(maxT, T);
        tile0[g + 1] = tile0[g] + (int32_t)T;
      }
      const int64_t n_tiles = tile0[lg.G];
      m->ga_variant = env_int("NUTS_GA_VARIANT", 42);
      if (m->ga_variant != 32 && m->ga_variant != 33) m->ga_variant = 42;
      const int occ = m->ga_variant / 10;
      int W = std::min(GA_MAXW, (4 * occ * cus) / std::max(lg.G, 1));   // all G workgroups resident at once (4 occ waves per CU)
      if (env_int("NUTS_ROWS_GA_W", 0) > 0) W = std::max(1, std::min(GA_MAXW, env_int("NUTS_ROWS_GA_W", 0)));
      const double meanT = (double)n_tiles / std::max(lg.G, 1);
      bool use = false;
      // (every covariate count 1 .. 8 has an instantiation of the group-aligned pass; the register-budget variants 32 / 33 exist
      // for D = 2, 4, 8 only)
      if (!d_pow2) m->ga_variant = 42;
      if (want >= 2) { use = m->ga_struct_ok && m->ept == 1; W = std::max(1, W); }
      else if (want == 1) use = m->ga_struct_ok && m->ept == 1 && W >= 1 && lg.G >= 2 * cus && meanT >= 4.0 * W && (double)maxT <= 1.5 * meanT + 1.0;
      // group-BLOCK pass (rows_gb_kernel.h) for small groups: the same closed-form model, a workgroup owns GPW whole groups and
      // nothing crosses workgroups inside the launch.  NUTS_ROWS_GB=0 keeps such models on the general path (A/B, tests).
      int gpw = 0;
      if (!use && want == 1 && !(s->rows_opts & NUTS_ROWS_NO_GROUP_BLOCK) && env_int("NUTS_ROWS_GB", 1) && m->ga_struct_ok && m->ept == 1 &&
          lg.G >= 64 && maxT <= 16) {
        gpw = GB_W * (int)((lg.G + GB_W * 512 - 1) / (GB_W * 512));   // a group per wave; more (in sequence) only to stay <= 512 workgroups
        gpw = std::min(gpw, GB_MAXGPW);
        if (env_int("NUTS_ROWS_GPW", 0) > 0) gpw = std::max(1, std::min(GB_MAXGPW, env_int("NUTS_ROWS_GPW", 0)));
        // (slot_sum: at most SLOT_SUM_MAXR records per lane -- block partials + the records of the auxiliary workgroups)
        const int aux_threads = WAVE * GB_W, auxel = n - lg.G * D;
        const int naux = m->ga_struct_ok == 2 ? (auxel + aux_threads - 1) / aux_threads : 0;
        if ((lg.G + gpw - 1) / gpw + naux > WAVE * SLOT_SUM_MAXR) gpw = 0;
      }
      if (gpw) { use = true; W = 1; }   // (layout: one chunk per group)
      if (use && n_tiles * (int64_t)SPAN < ((int64_t)1 << 31) && (gpw || lg.G <= 32 * 8 * GA_MAXCHUNK)) {
        lg.ga = 1; lg.ga_w = gpw ? GB_W : W;
        lg.ga_gpw = gpw;
        lg.ga_bsz = gpw ? gpw : (lg.G + 31) / 32;
        lg.ga_flags = env_int("NUTS_GA_FLAGS", 0);
        lg.ga_T_uni = 0; lg.ga_ng_uni = 0;
        {
          bool uni = lg.G > 0;
          for (int g = 1; g < lg.G; ++g) uni = uni && (gptr[g + 1] - gptr[g] == gptr[1] - gptr[0]);
          if (uni && maxT > 0) { lg.ga_T_uni = (int32_t)maxT; lg.ga_ng_uni = gptr[1] - gptr[0]; }
        }
        lg.ga_nblk = (lg.G + lg.ga_bsz - 1) / lg.ga_bsz;
        // auxiliary workgroups (rows_aux.h): one thread per element that is not a z element
        lg.ga_auxel = n - lg.G * D;
        lg.ga_naux = m->ga_struct_ok == 2 ? (lg.ga_auxel + WAVE * lg.ga_w - 1) / (WAVE * lg.ga_w) : 0;
        lg.ga_nrec = lg.ga_nblk + lg.ga_naux;
        // chunk (g, w) = the tiles wave w of workgroup g streams, [w T_g / W, (w + 1) T_g / W) -- the split the kernel makes.  Chunks
        // are placed one after the other with `GA_SKEW` doubles (8448 B = 33 x 256 B) between them, so consecutive chunk starts
        // differ by an ODD multiple of 256 B modulo any power-of-two channel interleave.
        // An intercept column (x_{i,0} = 1 for every row, SURVEY 8d's C2) carries no information: it is not stored, the tiles
        // are [D - 1][SPAN] and the kernel multiplies by the literal 1.0 (rows_ga_kernel.h, GaTileRegs7) -- 57 B per row
        // instead of 65 on a pass that is bound by the bytes it moves.  NUTS_GA_ONES0=0 keeps the column (A/B, tests).
        bool ones0 = D == 8 && env_int("NUTS_GA_ONES0", 1) != 0 && lg.N > 0;
        for (int64_t i = 0; ones0 && i < lg.N; ++i) ones0 = s->rows_X[i * D] == 1.0;
        const int DX = ones0 ? D - 1 : D;
        lg.ga_dx = DX;
        // (the skew between chunks is an ODD multiple of 256 B and a multiple of the tile's column count, so that the y bytes
        // of a tile sit at its element offset / DX)
        const int64_t TS = (int64_t)DX * SPAN, GA_SKEW = env_int("NUTS_GA_SKEW", DX == 7 || DX == 5 ? 1120 : 1056);   // 32 x 35, 32 x 33
        if (GA_SKEW % DX != 0) { g_err = "NUTS_GA_SKEW must be a multiple of the stored column count"; nuts_model_destroy(m); return nullptr; }
        std::vector<int64_t> coff((size_t)lg.G * W, 0);
        int64_t pos = 0, max_ct = 0;
        for (int g = 0; g < lg.G; ++g) {
          const int64_t T = tile0[g + 1] - tile0[g];
          for (int w = 0; w < W; ++w) {
            const int64_t ct = (int64_t)(w + 1) * T / W - (int64_t)w * T / W;
            coff[(size_t)g * W + w] = pos;
            pos += ct * TS + GA_SKEW;
            max_ct = std::max(max_ct, ct);
          }
        }
        lg.ga_cstride_uni = 0;
        if (lg.ga_T_uni > 0 && lg.ga_T_uni % W == 0) {   // equal chunks: offsets follow from the chunk index
          lg.ga_cstride_uni = (lg.ga_T_uni / W) * TS + GA_SKEW;
        } else lg.ga_T_uni = 0;
        lg.Npad = n_tiles * SPAN; lg.n_spans = n_tiles;
        // (one tile of slack at the end: a wave without tiles still issues its unconditional first request)
        std::vector<double> xt((size_t)(pos + TS), 0.0);
        yy.assign((size_t)(pos / DX + 2 * SPAN), 0);
        for (int g = 0; g < lg.G; ++g) {
          const int64_t T = tile0[g + 1] - tile0[g];
          for (int64_t i = gptr[g]; i < gptr[g + 1]; ++i) {
            const int64_t r = i - gptr[g], t = r / SPAN, rr = r % SPAN;
            int w = (int)(((t + 1) * W - 1) / std::max<int64_t>(T, 1));          // the wave whose range [w T / W, (w + 1) T / W) holds tile t
            while (w > 0 && (int64_t)w * T / W > t) --w;
            while (w + 1 < W && (int64_t)(w + 1) * T / W <= t) ++w;
            const int64_t base = coff[(size_t)g * W + w] + (t - (int64_t)w * T / W) * TS;
            for (int d = D - DX; d < D; ++d) xt[(size_t)(base + (int64_t)(d - (D - DX)) * SPAN + rr)] = s->rows_X[i * D + d];
            yy[(size_t)(base / DX + rr)] = s->rows_y[i];
          }
        }
        lg.Xt = m->keep(dev_upload(xt.data(), xt.size()));
        lg.y = m->keep(dev_upload(yy.data(), yy.size()));
        m->rows_xt_len = (int64_t)xt.size(); m->rows_y_len = (int64_t)yy.size();
        lg.ga_coff = m->keep(dev_upload(coff.data(), coff.size()));
        lg.ga_tile0 = m->keep(dev_upload(tile0.data(), tile0.size()));
        lg.ga_part = m->keep(dev_alloc<double>((size_t)lg.G * PART_STRIDE));
        // (group-block pass: slot-major, every slot padded to a multiple of 64 records -- the padding stays zero)
        const size_t bpart_len = gpw ? 2 * (size_t)PART_STRIDE * ((lg.ga_nrec + WAVE - 1) / WAVE * WAVE) : 2 * (size_t)lg.ga_nrec * PART_STRIDE;
        lg.ga_bpart = m->keep(dev_alloc<double>(bpart_len));
        lg.ga_ticket = m->keep(dev_alloc<unsigned>(lg.ga_nblk));
        if (lg.ga_part) hipMemset(lg.ga_part, 0, (size_t)lg.G * PART_STRIDE * sizeof(double));
        if (lg.ga_bpart) hipMemset(lg.ga_bpart, 0, bpart_len * sizeof(double));
        if (lg.ga_ticket) hipMemset(lg.ga_ticket, 0, lg.ga_nblk * sizeof(unsigned));
        if (env_int("NUTS_GA_TREE_DBG", 0) > 0) {
          m->tree_dbg_leaf = env_int("NUTS_GA_TREE_DBG", 0) - 1;
          m->tree_dbg = m->keep(dev_alloc<long long>((size_t)lg.G * 8));
          if (m->tree_dbg) hipMemset(m->tree_dbg, 0, (size_t)lg.G * 8 * sizeof(long long));
        }
        m->ga_sync = m->keep(dev_alloc<unsigned>(GA_SYNC_WORDS));
        if (m->ga_sync) hipMemset(m->ga_sync, 0, GA_SYNC_WORDS * sizeof(unsigned));
        m->rows_grid = gpw ? lg.ga_nblk : lg.G;
        // persistent tree kernel (rows_ga_tree.h): needs every one of its G + 1 workgroups resident at once.  The occupancy
        // query is asked for the real block size and capped by the wave slots of the register budget; the query can be
        // optimistic (MI355X guide, "Residency and cooperative launch"), which is why every wait in the kernel is bounded.
        m->ga_tree_ok = 0;
        // (only at the 168-register budget, variant 32: at 128 registers the allocator spills inside the streaming loop)
        if (D == 8 && m->ga_variant == 32 && !gpw && lg.ga_naux == 0 && env_int("NUTS_GA_TREE", 1) != 0) {
          int per_cu = 0;
          const hipError_t e = lg.ga_dx == 7 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tree_ga<3, 7>, WAVE * W, 0)
                                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tree_ga<3, 8>, WAVE * W, 0);
          if (e == hipSuccess) {
            const int hw = std::min(per_cu, (4 * occ) / std::max(W, 1));   // `occ` waves per SIMD at this register budget
            m->ga_tree_ok = (int64_t)hw * cus >= (int64_t)lg.G + 1;
          }
        }
      }
    }
    if (!lg.ga) {
    // HBM layout: X in span tiles [n_spans][D][SPAN] (one contiguous block per wave-iteration, each column a
    // coalesced 16 B/lane load), y int8, group structure as G+1 row pointers (rows are sorted by group)
    {
      std::vector<double> xt((size_t)D * lg.Npad, 0.0);
      for (int64_t i = 0; i < lg.N; ++i) {
        const int64_t sp = i / SPAN, r = i % SPAN;
        for (int d = 0; d < D; ++d) xt[((size_t)sp * D + d) * SPAN + r] = s->rows_X[i * D + d];
      }
      lg.Xt = m->keep(dev_upload(xt.data(), xt.size()));
    }
    yy.assign(lg.Npad, 0);
    for (int64_t i = 0; i < lg.N; ++i) yy[i] = s->rows_y[i];
    lg.y = m->keep(dev_upload(yy.data(), yy.size()));
    const int waves_per_block = ROWS_BLOCK / WAVE;
    int64_t want_waves = std::min<int64_t>((int64_t)cus * wpc, lg.n_spans);
    const int nb_main = (int)((want_waves + waves_per_block - 1) / waves_per_block);
    lg.n_waves = nb_main * waves_per_block;
    // static tables.  A span that lies entirely inside one group is "uniform" (streamed by the main waves);
    // the others (group boundary or padding rows inside) are "mixed" and get one wave each.
    std::vector<int32_t> span_gid(lg.n_spans), mixed_g0, mixed_seg_base, mixed_seg_gid;
    std::vector<int64_t> mixed_span;
    for (int64_t sp = 0; sp < lg.n_spans; ++sp) {
      const int64_t r0 = sp * SPAN, r1 = r0 + SPAN;
      if (r1 <= lg.N && gid[r0] == gid[r1 - 1]) { span_gid[sp] = gid[r0]; continue; }
      span_gid[sp] = -1;
      mixed_span.push_back(sp);
      mixed_g0.push_back(gid[r0]);
      mixed_seg_base.push_back((int32_t)mixed_seg_gid.size());
      int prev = -1;
      for (int64_t r = r0; r < std::min<int64_t>(r1, lg.N); ++r)
        if (gid[r] != prev) { prev = gid[r]; mixed_seg_gid.push_back(prev); }
    }
    lg.n_mixed = (int32_t)mixed_span.size();
    lg.n_mixed_seg = (int32_t)mixed_seg_gid.size();
    std::vector<int32_t> run_ptr(lg.n_waves + 1, 0), seg_gid;
    std::vector<int4> runs;
    for (int w = 0; w < lg.n_waves; ++w) {
      const int64_t s0 = (int64_t)w * lg.n_spans / lg.n_waves, s1 = (int64_t)(w + 1) * lg.n_spans / lg.n_waves;
      run_ptr[w] = (int32_t)runs.size();
      for (int64_t sp = s0; sp < s1; ++sp) {
        if (span_gid[sp] < 0) continue;
        // runs of the same group inside one wave share a segment even when a mixed span sits between them
        if (!runs.empty() && (int)runs.size() > run_ptr[w] && runs.back().z == span_gid[sp] && runs.back().x + runs.back().y == sp) {
          runs.back().y++;
          continue;
        }
        int4 r; r.x = (int)sp; r.y = 1; r.z = span_gid[sp]; r.w = (int)seg_gid.size();
        runs.push_back(r);
        seg_gid.push_back(span_gid[sp]);
      }
    }
    run_ptr[lg.n_waves] = (int32_t)runs.size();
    lg.n_seg = (int32_t)seg_gid.size();
    auto group_ptr = [&](const std::vector<int32_t>& sg) {
      std::vector<int32_t> p(lg.G + 1, 0);
      for (int32_t g : sg) p[g + 1]++;
      for (int g = 0; g < lg.G; ++g) p[g + 1] += p[g];
      return p;
    };
    // segments are emitted in row order and rows are sorted by group => the segments of a group are contiguous
    const std::vector<int32_t> gsp = group_ptr(seg_gid), gmp = group_ptr(mixed_seg_gid);
    // fixed-slot segment layout (model_dev.h) when every group has few main segments and kernel B runs one element per
    // thread: slot = group * segK + ordinal of the segment inside its group
    int kmain = 1;
    for (int g = 0; g < lg.G; ++g) kmain = std::max(kmain, gsp[g + 1] - gsp[g]);
    lg.segK = 0;
    if (kmain <= SEG_MAIN_MAX && m->ept == 1 && env_int("NUTS_SEG_FIXED", 1)) {
      lg.segK = kmain + 2;
      for (auto& r : runs) r.w = r.z * lg.segK + (r.w - gsp[r.z]);
    }
    const size_t seg_doubles = lg.segK ? (size_t)lg.G * lg.segK * D : (size_t)lg.n_seg * D;
    lg.run_ptr = m->keep(dev_upload(run_ptr.data(), run_ptr.size()));
    lg.runs = m->keep(dev_upload(runs.data(), runs.size()));
    lg.gseg_ptr = m->keep(dev_upload(gsp.data(), gsp.size()));
    lg.seg_part = m->keep(dev_alloc<double>(std::max<size_t>(seg_doubles, 1)));
    if (lg.seg_part) hipMemset(lg.seg_part, 0, std::max<size_t>(seg_doubles, 1) * sizeof(double));
    lg.mixed_span = m->keep(dev_upload(mixed_span.data(), mixed_span.size()));
    lg.mixed_g0 = m->keep(dev_upload(mixed_g0.data(), mixed_g0.size()));
    lg.mixed_seg_base = m->keep(dev_upload(mixed_seg_base.data(), mixed_seg_base.size()));
    lg.gmix_ptr = m->keep(dev_upload(gmp.data(), gmp.size()));
    lg.mixed_part = m->keep(dev_alloc<double>((size_t)lg.n_mixed_seg * D));
    lg.wave_lp = m->keep(dev_alloc<double>((size_t)lg.n_waves + lg.n_mixed));
    m->rows_grid = nb_main + (lg.n_mixed + waves_per_block - 1) / waves_per_block;
    }   // span-partitioned pass
    m->alg_bytes += lg.N * (8 * (int64_t)D + 1 + 4);  // X row + y + group id per row
  }
  if (s->mvn_k > 0) {
    md.has_mvn = 1;
    m->explicit_pre = 1;
    MvnDev& mv = md.mv;
    mv.k = s->mvn_k; mv.off = s->vars[s->mvn_var].offset;
    mv.mu = m->keep(dev_upload(s->mvn_mu, mv.k));
    mv.prec = m->keep(dev_upload(s->mvn_prec, (size_t)mv.k * mv.k));
    mv.rowq = m->keep(dev_alloc<double>(mv.k));
    mv.gdense = m->keep(dev_alloc<double>(n));
    hipMemset(mv.gdense, 0, n * sizeof(double));
    mv.konst = -0.5 * mv.k * std::log(2.0 * M_PI) - s->mvn_logdet;
    mv.winv = nullptr; mv.winv_t = nullptr; mv.wy = nullptr;
    if (s->mvn_winv) {   // "cholesky" solver (include/nuts_mi355.h): W and W^T row-major, so that both mat-vecs read rows
      const size_t kk = (size_t)mv.k * mv.k;
      std::vector<double> wt(kk);
      for (int r = 0; r < mv.k; ++r)
        for (int cc = 0; cc < mv.k; ++cc) wt[(size_t)cc * mv.k + r] = s->mvn_winv[(size_t)r * mv.k + cc];
      mv.winv = m->keep(dev_upload(s->mvn_winv, kk));
      mv.winv_t = m->keep(dev_upload(wt.data(), kk));
      mv.wy = m->keep(dev_alloc<double>(3 * (size_t)mv.k));
    }
    m->mvn_grid = mv.k;   // one workgroup per row
    m->alg_bytes += 8 * (int64_t)mv.k * mv.k;
    // the model IS this node (one untransformed vector variable, no other factor): the row-aligned pass finishes the leapfrog
    // in the mat-vec's own workgroups (kernels.h, k_mvn_aligned)
    mv.aligned = 0; mv.al_nwg = 0; mv.al_part = nullptr;
    if (md.lean_ok && !md.has_logit && !mv.winv && mv.off == 0 && mv.k == n && s->n_vars == 1 && s->n_factors == 0 &&
        s->vars[0].transform == NUTS_TR_NONE && md.n_deferred == 0 && md.n_orphans == 0) {
      // rows per workgroup; 0: the two-kernel leapfrog.  Fewer, larger workgroups = fewer records for the control work to total and
      // a cheaper launch: at k = 2048 (C3) 8 rows measured 120 k leapfrog/s against 108 k with 4 and 88 k with 2
      const int R = env_int("NUTS_MVN_ALIGNED", mv.k >= 1024 ? 8 : 4);
      if (R == 2 || R == 4 || R == 8 || R == 16) {
        mv.aligned = R;
        mv.al_nwg = (mv.k + R - 1) / R;
        mv.al_part = m->keep(dev_alloc<double>(2 * (size_t)MVA_RS * mv.al_nwg));
        if (mv.al_part) hipMemset(mv.al_part, 0, 2 * (size_t)MVA_RS * mv.al_nwg * sizeof(double));
      }
    }
  }
  md.has_mix = 0;
  if (s->mix_N > 0) {
    MixDev& mx = md.mix;
    auto bad = [&](const char* msg) { g_err = msg; nuts_model_destroy(m); return (nuts_model*)nullptr; };
    if (s->rows_N > 0 || s->mvn_k > 0) return bad("mixture node: not together with another dense node");
    if (s->mix_K < 1 || s->mix_K > MIX_MAXK) return bad("mixture node: 1 <= K <= 16 components");
    if (!s->mix_y) return bad("mixture node: no observations");
    auto var_ok = [&](int v) { return v >= 0 && v < s->n_vars && s->vars[v].size == s->mix_K && !vars[v].deferred; };
    if (!var_ok(s->mix_mu)) return bad("mixture node: mu must be a variable with K elements (not a scalar that broadcasts into another factor)");
    if (s->mix_sigma >= 0 && (!var_ok(s->mix_sigma) || (s->vars[s->mix_sigma].transform != NUTS_TR_NONE && s->vars[s->mix_sigma].transform != NUTS_TR_LOG)))
      return bad("mixture node: sigma must be a variable with K elements, untransformed or log-transformed");
    if (s->mix_sigma < 0 && !s->mix_sigma_const) return bad("mixture node: sigma is neither a variable nor a constant");
    if (s->mix_w_simplex) {
      const int v = s->mix_w_logits;
      if (s->mix_K < 3 || v < 0 || v >= s->n_vars || s->vars[v].size != s->mix_K - 1 || vars[v].deferred || s->vars[v].transform != NUTS_TR_NONE || !s->mix_w_alpha)
        return bad("mixture node: Dirichlet weights are a variable of K - 1 elements (the simplex-transformed value, K >= 3) with K concentrations");
      for (int k = 0; k < s->mix_K; ++k)
        if (!(s->mix_w_alpha[k] > 0)) return bad("mixture node: Dirichlet concentrations a > 0");   // multivariate.py Dirichlet.logp check_parameters
    } else
    if (s->mix_w_logits >= 0 && (!var_ok(s->mix_w_logits) || s->vars[s->mix_w_logits].transform != NUTS_TR_NONE))
      return bad("mixture node: the weight logits must be an untransformed variable with K elements");
    if (s->mix_w_logits < 0 && !s->mix_w_const) return bad("mixture node: the weights are neither softmax(logits) nor constants");
    if (s->vars[s->mix_mu].transform != NUTS_TR_NONE) return bad("mixture node: mu must be untransformed");
    if (s->mix_assign >= s->n_data || (s->mix_assign >= 0 && s->data[s->mix_assign].size != s->mix_N))
      return bad("mixture node: one assignment per observed row");
    mx.N = s->mix_N; mx.K = s->mix_K;
    mx.off_mu = s->vars[s->mix_mu].offset;
    mx.off_sigma = s->mix_sigma >= 0 ? s->vars[s->mix_sigma].offset : -1;
    mx.tr_sigma = s->mix_sigma >= 0 ? s->vars[s->mix_sigma].transform : NUTS_TR_NONE;
    mx.off_w = s->mix_w_logits >= 0 ? s->vars[s->mix_w_logits].offset : -1;
    for (int k = 0; k < MIX_MAXK; ++k) { mx.sigma_c[k] = 1.0; mx.logw_c[k] = 0.0; mx.alpha[k] = 1.0; }
    mx.w_simplex = s->mix_w_simplex ? 1 : 0; mx.pad_ = 0; mx.w_konst = 0.0;
    if (mx.w_simplex) {
      double sa = 0.0, sl = 0.0;
      for (int k = 0; k < mx.K; ++k) { mx.alpha[k] = s->mix_w_alpha[k]; sa += mx.alpha[k]; sl += std::lgamma(mx.alpha[k]); }
      mx.w_konst = std::lgamma(sa) - sl + std::log((double)mx.K);
    }
    if (s->mix_sigma < 0)
      for (int k = 0; k < mx.K; ++k) {
        if (!(s->mix_sigma_const[k] > 0)) return bad("mixture node: sigma > 0");   // continuous.py:532 check_parameters
        mx.sigma_c[k] = s->mix_sigma_const[k];
      }
    if (s->mix_w_logits < 0) {
      double sum = 0.0;
      for (int k = 0; k < mx.K; ++k) { if (!(s->mix_w_const[k] >= 0 && s->mix_w_const[k] <= 1)) return bad("mixture node: 0 <= weights <= 1, sum(weights) == 1"); sum += s->mix_w_const[k]; }
      if (std::fabs(sum - 1.0) > 1e-8) return bad("mixture node: 0 <= weights <= 1, sum(weights) == 1");   // mixture.py:487-493
      for (int k = 0; k < mx.K; ++k) mx.logw_c[k] = std::log(s->mix_w_const[k]);
    }
    mx.y = m->keep(dev_upload(s->mix_y, (size_t)s->mix_N));
    mx.assign = s->mix_assign >= 0 ? md.pool + s->data[s->mix_assign].offset : nullptr;
    mx.nwg = (int)std::max<int64_t>(1, std::min<int64_t>(512, (s->mix_N + 4 * MIX_BLOCK - 1) / (4 * MIX_BLOCK)));
    mx.part = m->keep(dev_alloc<double>((size_t)mx.nwg * (3 * MIX_MAXK + 1)));
    mx.gdense = m->keep(dev_alloc<double>((size_t)n + 1));
    mx.lp = mx.gdense + n;
    if (mx.part) hipMemset(mx.part, 0, (size_t)mx.nwg * (3 * MIX_MAXK + 1) * sizeof(double));
    if (mx.gdense) hipMemset(mx.gdense, 0, ((size_t)n + 1) * sizeof(double));
    md.has_mix = 1;
    m->alg_bytes += 8 * s->mix_N + (s->mix_assign >= 0 ? 8 * s->mix_N : 0);
  }
  md.has_glm = 0;
  if (s->glm_N > 0) {
    GlmDev& gm = md.glm;
    auto bad = [&](const char* msg) { g_err = msg; nuts_model_destroy(m); return (nuts_model*)nullptr; };
    if (s->rows_N > 0 || s->mix_N > 0) return bad("GLM node: together with an MvNormal node only, not with the logit rows or the mixture node");
    if (s->mvn_k > 0 && s->mvn_winv) return bad("GLM node next to an MvNormal node: the MvNormal node's precision solver only");
    if (s->glm_P < 1 || s->glm_P > NUTS_GLM_MAXP) return bad("GLM node: 1 <= P <= 512 covariates");
    if (!s->glm_X || !s->glm_y) return bad("GLM node: no design matrix / observations");
    if (s->glm_family < NUTS_GLM_NORMAL || s->glm_family > NUTS_GLM_POISSON) return bad("GLM node: unknown family");
    const int vb = s->glm_beta;
    int dslot = -1;   // beta a derived vector: its place in the model's list of derived vectors
    if (vb < 0) {
      for (int t = 0; t < md.n_derived; ++t) if (md.derived_f[t] == s->glm_beta_derived) dslot = t;
      if (dslot < 0 || s->factors[s->glm_beta_derived].size != s->glm_P) return bad("GLM node: glm_beta_derived must name a NUTS_D_DERIVED factor of P elements");
    } else if (vb >= s->n_vars || s->vars[vb].size != s->glm_P || s->vars[vb].transform != NUTS_TR_NONE)
      return bad("GLM node: beta must be an untransformed variable with P elements (or a derived vector)");
    auto scalar_ok = [&](int v, bool log_ok) {
      return v >= 0 && v < s->n_vars && s->vars[v].size == 1 && (s->vars[v].transform == NUTS_TR_NONE || (log_ok && s->vars[v].transform == NUTS_TR_LOG));
    };
    if (s->glm_intercept >= 0 && !scalar_ok(s->glm_intercept, false)) return bad("GLM node: the intercept must be an untransformed scalar variable");
    if (s->glm_family == NUTS_GLM_NORMAL) {
      if (s->glm_sigma >= 0 && !scalar_ok(s->glm_sigma, true)) return bad("GLM node: sigma must be a scalar variable, untransformed or log-transformed");
      if (s->glm_sigma < 0 && !(s->glm_sigma_const > 0)) return bad("GLM node: sigma > 0");   // continuous.py:532 check_parameters
    } else if (s->glm_sigma >= 0) return bad("GLM node: only the Normal family has a sigma");
    gm.N = s->glm_N; gm.P = s->glm_P; gm.family = s->glm_family;
    // register layout: a row in the lanes of a group of `lpr`, `ch` 16-byte chunks per lane (glm_kernel.h)
    int lpr = 1;
    while (8 * lpr < gm.P) lpr *= 2;
    int ch = (gm.P + 2 * lpr - 1) / (2 * lpr);
    if (lpr == 1) ch = ch == 3 ? 4 : ch;       // instantiated: 1, 2, 4
    else if (lpr < 32) ch = 4;                 // 4 only
    else ch = std::max(ch, 3);                 // 3 or 4
    gm.lpr = lpr; gm.ch = ch; gm.Ppad = 2 * lpr * ch;
    gm.off_beta = vb >= 0 ? s->vars[vb].offset : -1;
    gm.beta_buf = dslot >= 0 ? const_cast<double*>(md.pool) + md.derived_off[dslot] : nullptr;
    gm.beta_seed = dslot >= 0 ? const_cast<double*>(md.pool) + md.derived_off[dslot] + gm.P : nullptr;
    gm.off_icpt = s->glm_intercept >= 0 ? s->vars[s->glm_intercept].offset : -1;
    gm.off_sigma = s->glm_sigma >= 0 ? s->vars[s->glm_sigma].offset : -1;
    gm.tr_sigma = s->glm_sigma >= 0 ? s->vars[s->glm_sigma].transform : NUTS_TR_NONE;
    gm.sigma_c = s->glm_sigma >= 0 ? 1.0 : (s->glm_family == NUTS_GLM_NORMAL ? s->glm_sigma_const : 1.0);
    gm.konst = 0.0;
    for (int64_t i = 0; i < gm.N; ++i) {
      const double yi = s->glm_y[i];
      if (s->glm_family == NUTS_GLM_BERNOULLI && !(yi == 0.0 || yi == 1.0)) return bad("GLM node: Bernoulli observations must be 0 or 1");
      if (s->glm_family == NUTS_GLM_POISSON) {
        if (!(yi >= 0.0) || yi != std::floor(yi)) return bad("GLM node: Poisson observations must be non-negative integers");
        gm.konst -= std::lgamma(yi + 1.0);     // factln(y): parameter-free (discrete.py:581-597)
      }
    }
    gm.xstride = (gm.P + 1) & ~1; gm.xpad_ = 0;   // 16-byte rows; the layout's last chunks read on into the next row (glm_kernel.h)
    {   // rows at their own stride + one layout width of zeros behind the last row (what its last chunks read), uploaded in slabs
      const size_t total = (size_t)gm.N * gm.xstride + (size_t)gm.Ppad;
      double* xd = m->keep(dev_alloc<double>(total));
      gm.X = xd;
      if (xd) {
        hipMemset(xd + (size_t)gm.N * gm.xstride, 0, (size_t)gm.Ppad * sizeof(double));
        if (gm.xstride == gm.P) hipMemcpy(xd, s->glm_X, (size_t)gm.N * gm.P * sizeof(double), hipMemcpyHostToDevice);
        else {
          const int64_t slab = std::max<int64_t>(1, (int64_t)(1 << 22) / gm.xstride);
          std::vector<double> buf((size_t)slab * gm.xstride, 0.0);
          for (int64_t r0 = 0; r0 < gm.N; r0 += slab) {
            const int64_t nr = std::min<int64_t>(slab, gm.N - r0);
            for (int64_t r = 0; r < nr; ++r) std::memcpy(&buf[(size_t)r * gm.xstride], s->glm_X + (size_t)(r0 + r) * gm.P, (size_t)gm.P * sizeof(double));
            hipMemcpy(xd + (size_t)r0 * gm.xstride, buf.data(), (size_t)nr * gm.xstride * sizeof(double), hipMemcpyHostToDevice);
          }
        }
      }
    }
    gm.y = m->keep(dev_upload(s->glm_y, (size_t)gm.N));
    gm.Xt = nullptr;
    if (gm.P <= GLM_SMALL_P && gm.N <= GLM_SMALL_N) {   // the single-workgroup kernel's copy (small_kernel.h): columns contiguous
      std::vector<double> xt((size_t)gm.N * gm.P);
      for (int64_t i = 0; i < gm.N; ++i)
        for (int p = 0; p < gm.P; ++p) xt[(size_t)p * gm.N + i] = s->glm_X[(size_t)i * gm.P + p];
      gm.Xt = m->keep(dev_upload(xt.data(), xt.size()));
    }
    // grid: every CU gets NUTS_GLM_WG_PER_CU workgroups of four waves (default 4: 16 waves per CU, each with one row-iteration in
    // flight and one being evaluated).  Measured at configs[3]'s shape on two boxes (profiles/r04h_glm_sweep_workgroups_per_cu.txt,
    // r04i): 1421 / 1381 / 1380 / 1328 and 1426 / 1387 / 1365 / 1345 leapfrog/s at 4 / 8 / 12 / 16 -- more workgroups shave the
    // tail of the pass itself but leave more records for k_glm_reduce and a longer launch ramp; the whole leapfrog is fastest at 4
    // (rocprofv3, same command: 4351 ms of GPU time for 7610 launches at 4 against 5591 ms for 8826 at 12).
    const int64_t iters = (gm.N + (WAVE / lpr) - 1) / (WAVE / lpr);
    int64_t nwg = (int64_t)cus * std::max(1, env_int("NUTS_GLM_WG_PER_CU", 4));
    nwg = std::max<int64_t>(1, std::min<int64_t>(nwg, (iters + 4 * (GLM_BLOCK / WAVE) - 1) / (4 * (GLM_BLOCK / WAVE))));
    gm.nwg = (int)nwg;
    gm.part = m->keep(dev_alloc<double>((size_t)gm.nwg * (gm.Ppad + 4)));
    gm.gdense = m->keep(dev_alloc<double>((size_t)n + 1));
    gm.lp = gm.gdense ? gm.gdense + n : nullptr;
    if (gm.part) hipMemset(gm.part, 0, (size_t)gm.nwg * (gm.Ppad + 4) * sizeof(double));
    if (gm.gdense) hipMemset(gm.gdense, 0, ((size_t)n + 1) * sizeof(double));
    md.has_glm = 1;
    m->alg_bytes += 8 * gm.N * (int64_t)gm.P;   // one read of X (SURVEY 8d convention: the node's data once per evaluation)
  }
  if (!build_lins(m, s, vars)) { nuts_model_destroy(m); return nullptr; }
  if (!build_sweep_fast(m, s, vars)) { nuts_model_destroy(m); return nullptr; }
  for (void* p : m->owned)
    if (!p) { g_err = "device allocation failed"; nuts_model_destroy(m); return nullptr; }
  HIPCHK_NULL(hipDeviceSynchronize());
  return m;
}

extern "C" int nuts_model_set_data(nuts_model* m, int32_t data_id, const double* values, int64_t n) {
  if (!m || !values) { g_err = "null argument"; return NUTS_E_ARG; }
  if (data_id < 0 || data_id >= (int)m->data_refs.size()) { g_err = "nuts_model_set_data: no such data vector"; return NUTS_E_ARG; }
  const nuts_data_ref& r = m->data_refs[data_id];
  if (n != r.size) { g_err = "nuts_model_set_data: the length of a data vector cannot change"; return NUTS_E_ARG; }
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipMemcpy(const_cast<double*>(m->md.pool) + r.offset, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  m->data_epoch++;
  return NUTS_OK;
}

// Several data vectors at once (the extra inputs another step method has just changed: `ValueGradFunction.set_extra_values`,
// model/core.py:275-278): `values` holds them one after the other.  The bytes go through a pinned staging buffer and ONE
// stream-ordered copy per run of vectors that are neighbours in the data pool -- no host synchronisation: launches submitted
// afterwards see the new values, launches in flight the old ones.
extern "C" int nuts_model_set_data_many(nuts_model* m, int32_t count, const int32_t* data_ids, const double* values, const int64_t* lens) {
  if (!m || count < 0 || (count > 0 && (!data_ids || !values || !lens))) { g_err = "null argument"; return NUTS_E_ARG; }
  int64_t total = 0;
  for (int i = 0; i < count; ++i) {
    if (data_ids[i] < 0 || data_ids[i] >= (int)m->data_refs.size()) { g_err = "nuts_model_set_data_many: no such data vector"; return NUTS_E_ARG; }
    if (lens[i] != m->data_refs[data_ids[i]].size) { g_err = "nuts_model_set_data_many: the length of a data vector cannot change"; return NUTS_E_ARG; }
    total += lens[i];
  }
  if (total == 0) return NUTS_OK;
  if (m->set_pin_len < total) {
    if (m->set_ev) HIPCHK(hipEventSynchronize(m->set_ev));
    if (m->set_pin) hipHostFree(m->set_pin);
    m->set_pin = nullptr; m->set_pin_len = 0;
    HIPCHK(hipHostMalloc((void**)&m->set_pin, (size_t)total * sizeof(double), hipHostMallocDefault));
    m->set_pin_len = total;
  }
  if (!m->set_ev) HIPCHK(hipEventCreateWithFlags(&m->set_ev, hipEventDisableTiming));
  else HIPCHK(hipEventSynchronize(m->set_ev));          // the previous call's copies have left the staging buffer
  std::memcpy(m->set_pin, values, (size_t)total * sizeof(double));
  int64_t at = 0;
  for (int i = 0; i < count;) {
    const int64_t off0 = m->data_refs[data_ids[i]].offset;
    int64_t run = lens[i];
    int j = i + 1;
    while (j < count && m->data_refs[data_ids[j]].offset == off0 + run) { run += lens[j]; ++j; }
    if (run > 0)
      HIPCHK(hipMemcpyAsync(const_cast<double*>(m->md.pool) + off0, m->set_pin + at, (size_t)run * sizeof(double), hipMemcpyHostToDevice, m->stream));
    at += run;
    i = j;
  }
  HIPCHK(hipEventRecord(m->set_ev, m->stream));
  m->data_epoch++;
  return NUTS_OK;
}

static void group_remove_model(struct nuts_group* g, nuts_model* m);
extern "C" void nuts_model_destroy(nuts_model* m) {
  if (!m) return;
  if (m->stream) hipStreamSynchronize(m->stream);
  if (m->set_ev) hipEventDestroy(m->set_ev);
  if (m->set_pin) hipHostFree(m->set_pin);
  if (m->group) group_remove_model(m->group, m);
  for (void* p : m->owned) if (p) hipFree(p);
  for (auto e : m->ev) hipEventDestroy(e);
  if (m->host_pin) hipHostFree(m->host_pin);
  if (m->own_stream) hipStreamDestroy(m->own_stream);
  delete m;
}

// Diagnostics: shader-clock timestamps of the phases of the last B / C launch (zeros unless the library was
// built with -DNUTS_KTIMING).  64 entries.
extern "C" int nuts_model_debug_ticks(nuts_model* m, int64_t* out) {
  if (!m || !out) return NUTS_E_ARG;
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipMemcpy(out, m->md.ticks, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  return NUTS_OK;
}

extern "C" int nuts_model_debug_tree(nuts_model* m, int64_t* out, int64_t cap) {
  if (!m || !out) return NUTS_E_ARG;
  if (!m->tree_dbg) { g_err = "no per-workgroup timeline (create the model with NUTS_GA_TREE_DBG=<leaf + 1>)"; return NUTS_E_ARG; }
  HIPCHK(hipStreamSynchronize(m->stream));
  const int64_t nwords = std::min<int64_t>(cap, (int64_t)m->md.lg.G * 8);
  HIPCHK(hipMemcpy(out, m->tree_dbg, (size_t)nwords * sizeof(long long), hipMemcpyDeviceToHost));
  return NUTS_OK;
}
extern "C" int32_t nuts_model_ndim(const nuts_model* m) { return m ? m->md.n : -1; }
extern "C" int nuts_model_get_scalar(const nuts_model* m, const char* name, double* out) {
  if (!m || !name || !out) return NUTS_E_ARG;
  const std::string k(name);
  if (k == "rows_group_aligned") *out = m->md.lg.ga;
  else if (k == "rows_group_block") *out = m->md.lg.ga_gpw;
  else if (k == "rows_aux_workgroups") *out = m->md.lg.ga ? m->md.lg.ga_naux : 0;
  else if (k == "mixture_workgroups") *out = m->md.has_mix ? m->md.mix.nwg : 0;
  else if (k == "glm_workgroups") *out = m->md.has_glm ? m->md.glm.nwg : 0;
  else if (k == "glm_row_stride") *out = m->md.has_glm ? m->md.glm.xstride : 0;
  else if (k == "glm_layout_width") *out = m->md.has_glm ? m->md.glm.Ppad : 0;
  else if (k == "mvn_row_aligned") *out = m->md.has_mvn ? m->md.mv.aligned : 0;
  else if (k == "rows_waves") *out = m->md.lg.ga ? m->md.lg.ga_w : m->md.lg.n_waves;
  else if (k == "lean") *out = m->md.lean_ok;
  else if (k == "tree_kernel_ok") *out = m->ga_tree_ok;
  else if (k == "rows_stored_columns") *out = m->md.lg.ga ? m->md.lg.ga_dx : m->md.lg.D;
  else if (k == "chain_group_kind") {   // what a chain group of this model's chains would merge: 0 nothing, 1 the MvNormal row-aligned pass, 2 the group-aligned row pass
    const RowsDev& lg = m->md.lg;
    const bool is_mvn = m->md.has_mvn && (m->md.mv.aligned == 4 || m->md.mv.aligned == 8 || m->md.mv.aligned == 16);
    const bool is_rows = m->md.has_logit && !m->md.has_mvn && lg.ga && lg.ga_naux == 0 && lg.D == 8 && m->ga_struct_ok == 1 && (lg.ga_gpw > 0 || m->ga_variant == 42);
    *out = is_mvn ? 1.0 : (is_rows ? (lg.ga_gpw > 0 ? 3.0 : 2.0) : 0.0);   // (3: the group-block row pass, round 6)
  }
  // 1: chains of this model can form a WIDE group (up to 16 chains per launch through the matrix cores, mvn_mfma_kernel.h) once the
  // model is laid out 8 rows per workgroup (NUTS_MVN_ALIGNED = 8: the default from k = 1024) and the chain was created under NUTS_GROUP_WIDE = 1
  else if (k == "chain_group_wide_ok") *out = (m->md.has_mvn && m->md.mv.aligned > 0 && m->md.mv.k % 16 == 0) ? 1.0 : 0.0;
  else if (k == "chain_group_wide_rows") *out = 8.0;   // rows per workgroup (NUTS_MVN_ALIGNED) the members' models must be laid out with
  else if (k == "single_workgroup_ok") *out = (m->md.n <= SMALL_MAX_N && m->ept == 1 && !m->md.has_logit && !m->md.has_mvn && !m->md.has_mix && !m->md.has_glm && m->md.n_lins == 0 && m->factor_elems <= SMALL_MAX_ELEMS) ? 1.0 : 0.0;
  else { g_err = "unknown model scalar " + k; return NUTS_E_ARG; }
  return NUTS_OK;
}
extern "C" int64_t nuts_model_algorithmic_bytes(const nuts_model* m) { return m ? m->alg_bytes : 0; }

extern "C" int nuts_model_logp_grad(nuts_model* m, const double* q, double* logp, double* grad) {
  if (!m || !q || !logp) { g_err = "null argument"; return NUTS_E_ARG; }
  const int n = m->md.n;
  std::memcpy(m->host_pin, q, n * sizeof(double));
  HIPCHK(hipMemcpyAsync(m->q_dev, m->host_pin, n * sizeof(double), hipMemcpyHostToDevice, m->stream));
  m->md.fdead_mode = 1;   // (record factors whose parameter check fails; costs nothing unless one does)
  model_enqueue_plain(m, m->q_dev, m->g_dev, m->lp_dev);
  m->md.fdead_mode = 0;
  HIPCHK(hipMemcpyAsync(m->host_pin + n, m->g_dev, n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipMemcpyAsync(m->host_pin + 2 * n, m->lp_dev, sizeof(double), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipGetLastError());
  *logp = m->host_pin[2 * n];
  if (m->md.n_factors > 0 && !std::isfinite(*logp)) {
    // logp = -inf: did a PARAMETER check fail?  Then the reference's gradient is 0 for every element of that factor, not only
    // for the offending one (ModelDev.fdead): the pass above recorded such factors, a second pass honours the record.
    std::vector<int32_t> fd(m->md.n_factors);
    HIPCHK(hipMemcpy(fd.data(), m->md.fdead, fd.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    bool any = false;
    for (int32_t v : fd) any = any || v != 0;
    if (any) {
      if (grad) {
        m->md.fdead_mode = 2;
        model_enqueue_plain(m, m->q_dev, m->g_dev, m->lp_dev);
        m->md.fdead_mode = 0;
        HIPCHK(hipMemcpyAsync(m->host_pin + n, m->g_dev, n * sizeof(double), hipMemcpyDeviceToHost, m->stream));
      }
      HIPCHK(hipMemsetAsync(m->md.fdead, 0, fd.size() * sizeof(int32_t), m->stream));   // (the record is empty between calls)
      HIPCHK(hipStreamSynchronize(m->stream));
      HIPCHK(hipGetLastError());
    }
  }
  if (grad) std::memcpy(grad, m->host_pin + n, n * sizeof(double));
  return NUTS_OK;
}

static void profile_enable(nuts_model* m, bool on, int sample_every, size_t max_pairs) {
  m->profile = on;
  m->sample_every = std::max(1, sample_every);
  m->ev_used = 0;
  m->dom_launches = 0;
  m->dom_units = 0;
  m->run_open = 0; m->runs_seen = 0;
  while (on && m->ev.size() < 2 * max_pairs) {
    hipEvent_t e;
    hipEventCreate(&e);
    m->ev.push_back(e);
  }
  m->ev_units.assign(m->ev.size() / 2 + 1, 1);
}
// Sum of the bracketed times; `m->dom_units` is set to the passes they cover.  Runs (pairs that cover more than one launch) whose
// average per launch is below 0.8 x the median run's contained launches that drained behind a finished tree: dropped.
static double profile_sum_ms(nuts_model* m, int64_t* pairs) {
  std::vector<float> ms(m->ev_used / 2, 0.f);
  std::vector<double> run_avg;
  for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
    if (hipEventElapsedTime(&ms[i / 2], m->ev[i], m->ev[i + 1]) != hipSuccess) ms[i / 2] = -1.f;
    else if (m->ev_units[i / 2] > 1) run_avg.push_back(ms[i / 2] / m->ev_units[i / 2]);
  }
  double med = 0.0;
  if (!run_avg.empty()) { std::sort(run_avg.begin(), run_avg.end()); med = run_avg[run_avg.size() / 2]; }
  double tot = 0.0;
  int64_t units = 0, kept = 0;
  for (size_t p = 0; p < ms.size(); ++p) {
    if (ms[p] < 0.f) continue;
    const int u = m->ev_units[p];
    if (u > 1 && ms[p] / u < 0.8 * med) continue;
    tot += ms[p]; units += u; kept++;
  }
  m->dom_units = units;
  if (pairs) *pairs = kept;
  return tot;
}

extern "C" int nuts_model_time_logp_grad(nuts_model* m, const double* q, int reps, double* ms_total, double* ms_dominant) {
  if (!m || !q || reps <= 0) { g_err = "bad argument"; return NUTS_E_ARG; }
  const int n = m->md.n;
  HIPCHK(hipMemcpy(m->q_dev, q, n * sizeof(double), hipMemcpyHostToDevice));
  for (int i = 0; i < 3; ++i) model_enqueue_plain(m, m->q_dev, m->g_dev, m->lp_dev);
  HIPCHK(hipStreamSynchronize(m->stream));
  profile_enable(m, true, 1, (size_t)reps);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, m->stream);
  for (int i = 0; i < reps; ++i) model_enqueue_plain(m, m->q_dev, m->g_dev, m->lp_dev);
  hipEventRecord(b, m->stream);
  HIPCHK(hipStreamSynchronize(m->stream));
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  int64_t pairs = 0;
  const double dom = profile_sum_ms(m, &pairs);
  if (ms_total) *ms_total = ms / reps;
  if (ms_dominant) *ms_dominant = pairs ? dom / pairs : 0.0;
  profile_enable(m, false, 1, 0);
  return NUTS_OK;
}

// ===========================================================================
// chain
// ===========================================================================
struct DualAvg {  // pymc/step_methods/step_sizes.py:41-84
  double initial_step, target, gamma, k, t0;
  double log_step, log_bar, hbar, mu;
  int64_t count;
  void reset() {
    log_step = std::log(initial_step); log_bar = log_step; hbar = 0.0; count = 1; mu = std::log(10 * initial_step);
  }
  double current(bool tune) const { return tune ? std::exp(log_step) : std::exp(log_bar); }
  void update(double accept, bool tune) {
    if (!tune) return;
    const double w = 1.0 / (count + t0);
    hbar = (1 - w) * hbar + w * (target - accept);
    log_step = mu - hbar * std::sqrt((double)count) / gamma;
    const double mk = std::pow((double)count, -k);
    log_bar = mk * log_step + (1 - mk) * log_bar;
    count += 1;
  }
};

struct StateHeader {  // host scalars of the sampling state
  int64_t magic, n;
  DualAvg da;
  int64_t iter_count, divergences, n_samples, adaptation_window;
  int32_t tune, fg_is_a, pad0, pad1;
  double fg_count, bg_count;
  int64_t fa_previous_update;   // NUTS_POT_FULL_ADAPT
  double fa_fg_n, fa_bg_n;
};

#define DOM_RING 4
struct nuts_chain {
  nuts_model* m = nullptr;
  nuts_chain_config cfg{};
  int n = 0;
  ArenaDev A{};
  std::vector<void*> owned;
  // potential (device)
  double *var = nullptr, *stds = nullptr, *inv_stds = nullptr;
  double *dense_C = nullptr, *dense_W = nullptr;   // NUTS_POT_FULL: velocity = C p, random = W z
  int dense = 0, mv_grid = 0;
  // NUTS_POT_HOST: the potential's methods are callbacks into the caller (nuts_chain_set_host_potential)
  int host_pot = 0, cb_err = 0;
  nuts_velocity_fn hp_velocity = nullptr;
  nuts_energy_fn hp_energy = nullptr;
  nuts_velocity_energy_fn hp_velocity_energy = nullptr;
  void* hp_user = nullptr;
  std::vector<double> hp_p, hp_v;
  double* kin_user_dev = nullptr;
  // NUTS_POT_FULL_ADAPT (dense_adapt.h): estimators, factor in use (fa_L), factorisation workspace, counters
  int full_adapt = 0, fa_mfma = 1;
  double *fa_L = nullptr, *fa_Lw = nullptr, *fa_fg_mean = nullptr, *fa_fg_raw = nullptr, *fa_bg_mean = nullptr, *fa_bg_raw = nullptr;
  double *fa_scratch = nullptr, *fa_rhs = nullptr;
  int *fa_fail = nullptr;          // [2]: {last factorisation failed, any factorisation failed since the last reset}
  double fa_fg_n = 0, fa_bg_n = 0;
  int64_t fa_previous_update = 0;
  std::vector<double> fa_initial_cov;
  double *wa_mean = nullptr, *wa_m2 = nullptr, *wb_mean = nullptr, *wb_m2 = nullptr;  // two Welford estimators
  bool fg_is_a = true;
  double fg_count = 0, bg_count = 0;
  std::vector<double> initial_mean, initial_diag;
  int64_t n_samples = 0, adaptation_window = 101;
  bool window_switched = false;   // the last potential update swapped foreground <- background (quadpotential.py:350-353)
  // step
  DualAvg da{};
  bool tune = true;
  int64_t iter_count = 0, divergences = 0;
  double step_size = 0;
  // staging
  double* stage_dev = nullptr;   // [2n + NUNI]
  double* stage_host = nullptr;  // pinned
  double* out_dev = nullptr;     // [2n]
  double* out_dev2 = nullptr;    // [2n] second output buffer (single-launch path: output and start-state cache alternate)
  bool small = false;            // latency regime: whole draw in one launch (small_kernel.h)
  int small_lds_slots = 0;       // SmallDrawArgs.lds_slots
  int small_one_wave = 1;        // n <= 64: the single-workgroup kernel with ONE wave (NUTS_SMALL_ONE_WAVE=0: four, as before round 5)
  int group_wide = 0;            // option NUTS_GROUP_WIDE when the chain was created: a group this chain founds takes up to 16 chains (matrix cores)
  double* out_host = nullptr;    // pinned [2n]
  HostStatus* st_dev = nullptr;
  HostStatus* st_host = nullptr;   // pinned + device-mapped; st_dev is its device alias
  int seq = 0;
  // divergence points of the last diverging draw (DivergenceInfo.state / state_div, nuts.py:433-440)
  std::vector<double> div_source, div_dest;
  // start-state cache: (q, grad) of the last returned proposal stay in out_dev, its logp here
  bool cache_ok = false;
  std::vector<double> last_q;
  double last_logp = 0.0;
  DrawOut* do_dev = nullptr;
  DrawOut* do_host = nullptr;
  DrawOutMapped* dom_host = nullptr;   // pinned + device-mapped copies of the last DOM_RING draws' records (multi-draw calls), slot = seq % DOM_RING
  DrawOutMapped* dom_dev = nullptr;
  unsigned dom_seq = 0;
  double* kin_part = nullptr;    // [nblk] kinetic-energy partials of the initial state
  int64_t cache_epoch = -1;      // model data epoch the start-state cache belongs to
  int fold_ctl = 1;              // lean path: overlap the control work of leaf j with the row pass of leaf j+1
  int tree_mode = 0;             // group-aligned row pass: one persistent launch per NUTS tree (rows_ga_tree.h)
  int64_t tree_launches = 0;
  CtlJob pend{}; bool pend_valid = false;   // control work of a doubling's last leaf waiting for the next doubling's first row pass
  bool defer_last_ctl = false; int xfold = 1;   // NUTS_XFOLD: fold control work across doublings (group-aligned row pass)
  // row-aligned MvNormal pass: the doubling being queued will be followed (look-ahead) by one in direction `next_dir`; its last
  // leaf then also materialises the first half of that doubling's first leaf (EvalIO.pre_next 1 / 3), and `pre_done` tells the
  // first leaf of the next doubling that its k_leaf_pre launch is not needed
  int next_dir = 0; bool pre_done = false; int xpre = 1;
  int tree_opts = 0;             // GA_TREE_* switches (NUTS_GA_TREE_OPTS, NUTS_GA_TREE_TICKS)
  int tree_prof_pair = 0;        // ... which event pair that is
  int tree_prof_pending = 0;     // the last tree launch is being timed: its leaf count is added when the draw's record arrives
  int spec_max = 10, last_depth = 0;   // look-ahead over the doublings, as deep as the previous tree went (run_tree)
  int pipe_draws = 1;                  // NUTS_PIPE_DRAWS, latched at creation: post-tuning draws of a batch are queued behind each other
  int logs_done = 0, logs_total = 0;  // logarithms of the pre-drawn uniforms taken / needed at most this draw
  // staging of nuts_chain_draw_many (grown on demand)
  double* many_in_host = nullptr; double* many_in_dev = nullptr; char* many_out_host = nullptr; char* many_out_dev = nullptr;
  size_t many_in_cap = 0, many_out_cap = 0;
  double t_begin = 0, t_loop = 0, t_wait = 0, t_finish = 0, t_post = 0;   // host seconds per phase of nuts_chain_draw, summed
  // host seconds per phase of nuts_chain_draw_many, summed (tools/draw_host_phases.py): batch set-up before the first launch,
  // launching a draw's start kernels, the doubling loop (`t_wait` of it spinning on status words), draw-finish launch until its
  // record is seen, host arithmetic per draw, batch tear-down (trace copy-back)
  double tm_pre = 0, tm_start = 0, tm_tree = 0, tm_record = 0, tm_host = 0, tm_post = 0;
  int64_t tm_draws = 0, tm_batches = 0;
  int64_t leapfrogs = 0;
  int n_uni_cap = 0;
  template <typename T>
  T* keep(T* p) { owned.push_back((void*)p); return p; }
};

extern "C" void nuts_chain_config_default(nuts_chain_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->step_scale = 0.25; c->Emax = 1000; c->target_accept = 0.8; c->gamma = 0.05; c->k = 0.75; c->t0 = 10;
  c->adapt_step_size = 1; c->max_treedepth = 10; c->early_max_treedepth = 8;
  c->potential = NUTS_POT_DIAG_ADAPT; c->initial_weight = 0; c->adaptation_window = 101; c->discard_window = 50;
  c->exp_alpha = 0.02; c->exp_stop_adaptation = INFINITY; c->exp_use_grads = 0;
  c->adaptation_window_multiplier = 1; c->early_update = 0;
}

// ---- NUTS_POT_FULL_ADAPT (dense_adapt.h) ----
// cov = raw / denom into dense_C, blocked Cholesky of it in the workspace, committed to the factor in use when it went through
static void fa_factor(nuts_chain* c, const double* raw, double denom) {
  const int n = c->n;
  hipStream_t s = c->m->stream;
  const dim3 g2((n + 255) / 256, n);
  hipLaunchKernelGGL(k_fa_cov, g2, dim3(256), 0, s, n, raw, denom, c->dense_C, c->fa_Lw, c->fa_fail);
  const int nb = (n + FA_NB - 1) / FA_NB;
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(k_chol_panel, dim3(nb - k), dim3(256), 0, s, n, c->fa_Lw, k, c->fa_fail);
    const int m = nb - k - 1;
    if (m > 0) hipLaunchKernelGGL(k_chol_update, dim3(m * (m + 1) / 2), dim3(256), 0, s, n, c->fa_Lw, k, nb, c->fa_mfma);
  }
  hipLaunchKernelGGL(k_fa_commit, dim3(1024), dim3(256), 0, s, (int64_t)n * n, c->fa_Lw, c->fa_L, c->fa_fail, c->fa_fail + 1);
}
// random(): p = solve(chol^T, z) (quadpotential.py:709-711)
static void fa_random(nuts_chain* c, const double* z_dev, double* p_out) {
  const int n = c->n;
  hipStream_t s = c->m->stream;
  hipMemcpyAsync(c->fa_rhs, z_dev, n * sizeof(double), hipMemcpyDeviceToDevice, s);
  const int nb = (n + FA_NB - 1) / FA_NB;
  for (int b = nb - 1; b >= 0; --b)
    hipLaunchKernelGGL(k_trsv_block, dim3(std::max(1, (b * FA_NB + 255) / 256)), dim3(256), 0, s, n, c->fa_L, c->fa_rhs, p_out, b);
}
static int fa_reset(nuts_chain* c) {   // QuadPotentialFullAdapt.reset (quadpotential.py:794-804)
  const int n = c->n;
  const size_t nn = (size_t)n * n;
  const double w = c->cfg.initial_weight;
  std::vector<double> raw(nn);
  for (size_t e = 0; e < nn; ++e) raw[e] = c->fa_initial_cov[e] * w;   // `raw_cov[:] *= n_samples`
  HIPCHK(hipMemcpy(c->fa_fg_raw, raw.data(), nn * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->fa_fg_mean, c->initial_mean.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(c->fa_bg_raw, 0, nn * sizeof(double)));    // `_WeightedCovariance(n)`: eye * 0
  HIPCHK(hipMemset(c->fa_bg_mean, 0, n * sizeof(double)));
  HIPCHK(hipMemset(c->fa_fail, 0, 2 * sizeof(int)));
  c->fa_fg_n = w; c->fa_bg_n = 0; c->fa_previous_update = 0;
  c->adaptation_window = c->cfg.adaptation_window;
  // the covariance in use is the INITIAL covariance and its factor (not the estimator's): stage it through the bg buffer
  HIPCHK(hipMemcpy(c->fa_Lw, c->fa_initial_cov.data(), nn * sizeof(double), hipMemcpyHostToDevice));
  {
    hipStream_t s = c->m->stream;
    HIPCHK(hipMemcpyAsync(c->fa_bg_raw, c->fa_Lw, nn * sizeof(double), hipMemcpyDeviceToDevice, s));
    fa_factor(c, c->fa_bg_raw, 1.0);
    HIPCHK(hipMemsetAsync(c->fa_bg_raw, 0, nn * sizeof(double), s));
    HIPCHK(hipStreamSynchronize(s));
  }
  int fail[2] = {0, 0};
  HIPCHK(hipMemcpy(fail, c->fa_fail, sizeof(fail), hipMemcpyDeviceToHost));
  if (fail[0]) { g_err = "the initial covariance of QuadPotentialFullAdapt is not positive definite"; return NUTS_E_LINALG; }
  return NUTS_OK;
}
// QuadPotentialFullAdapt.update (quadpotential.py:819-843)
static int fa_update(nuts_chain* c, const double* x_dev) {
  if (!c->tune) return NUTS_OK;
  const int n = c->n;
  hipStream_t s = c->m->stream;
  const int64_t delta = c->n_samples - c->fa_previous_update;
  c->fa_fg_n += 1; c->fa_bg_n += 1;
  hipLaunchKernelGGL(k_fa_diffs, dim3((n + 255) / 256), dim3(256), 0, s, n, x_dev, c->fa_fg_mean, c->fa_bg_mean, c->fa_fg_n, c->fa_bg_n, c->fa_scratch);
  hipLaunchKernelGGL(k_fa_rank1, dim3((n + 255) / 256, n), dim3(256), 0, s, n, c->fa_scratch, c->fa_fg_raw, c->fa_bg_raw);
  if ((delta + 1) % c->cfg.fa_update_window == 0) fa_factor(c, c->fa_fg_raw, c->fa_fg_n - 1.0);
  if (delta >= c->adaptation_window) {   // foreground <- background, fresh background, window grows
    std::swap(c->fa_fg_raw, c->fa_bg_raw); std::swap(c->fa_fg_mean, c->fa_bg_mean);
    c->fa_fg_n = c->fa_bg_n; c->fa_bg_n = 0;
    HIPCHK(hipMemsetAsync(c->fa_bg_raw, 0, (size_t)n * n * sizeof(double), s));
    HIPCHK(hipMemsetAsync(c->fa_bg_mean, 0, n * sizeof(double), s));
    c->fa_previous_update = c->n_samples;
    c->adaptation_window = (int64_t)(c->adaptation_window * c->cfg.adaptation_window_multiplier);
  }
  c->n_samples += 1;
  return NUTS_OK;
}

static int potential_reset(nuts_chain* c) {  // quadpotential.py:297-306
  const int n = c->n;
  // the blocking copies below run on the null stream; the model stream is non-blocking, so a Welford update queued by the
  // last tuning draw has to be waited for explicitly
  HIPCHK(hipStreamSynchronize(c->m->stream));
  std::vector<double> st(n), inv(n), m2(n);
  const double w = c->cfg.initial_weight;
  for (int i = 0; i < n; ++i) { st[i] = std::sqrt(c->initial_diag[i]); inv[i] = 1.0 / st[i]; m2[i] = c->initial_diag[i] * w; }
  HIPCHK(hipMemcpy(c->var, c->initial_diag.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->stds, st.data(), n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->inv_stds, inv.data(), n * sizeof(double), hipMemcpyHostToDevice));
  if (c->cfg.potential == NUTS_POT_DIAG_ADAPT) {
    c->fg_is_a = true;
    HIPCHK(hipMemcpy(c->wa_mean, c->initial_mean.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->wa_m2, m2.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(c->wb_mean, 0, n * sizeof(double)));
    HIPCHK(hipMemset(c->wb_m2, 0, n * sizeof(double)));
    c->fg_count = w; c->bg_count = 0;
  }
  if (c->full_adapt) { const int rc = fa_reset(c); if (rc) return rc; }
  if (c->cfg.potential == NUTS_POT_DIAG_ADAPT_EXP) {   // the estimators start at sample `discard_window` (kernel flag 1)
    HIPCHK(hipMemset(c->wa_mean, 0, n * sizeof(double))); HIPCHK(hipMemset(c->wa_m2, 0, n * sizeof(double)));
    HIPCHK(hipMemset(c->wb_mean, 0, n * sizeof(double))); HIPCHK(hipMemset(c->wb_m2, 0, n * sizeof(double)));
  }
  c->n_samples = 0;
  return NUTS_OK;
}

extern "C" nuts_chain* nuts_chain_create(nuts_model* m, const nuts_chain_config* cfg) {
  if (!m || !cfg) { g_err = "null argument"; return nullptr; }
  if (cfg->potential < NUTS_POT_DIAG_ADAPT || cfg->potential > NUTS_POT_HOST) { g_err = "unknown potential kind"; return nullptr;
  }
  if (cfg->potential == NUTS_POT_FULL && (!cfg->dense_cov || !cfg->dense_rand)) { g_err = "dense potential needs dense_cov and dense_rand"; return nullptr; }
  if (cfg->potential == NUTS_POT_FULL_ADAPT && !cfg->dense_cov) { g_err = "NUTS_POT_FULL_ADAPT needs dense_cov (the initial covariance)"; return nullptr; }
  if (cfg->max_treedepth < 1 || cfg->max_treedepth > MAX_LEVELS - 1) { g_err = "max_treedepth out of range (1..11)"; return nullptr; }
  if ((cfg->potential == NUTS_POT_FULL || cfg->potential == NUTS_POT_FULL_ADAPT || cfg->potential == NUTS_POT_HOST) && m->md.lg.ga) {
    g_err = "a dense mass matrix needs the span-partitioned row pass: create the model with NUTS_ROWS_NO_GROUP_ALIGNED";
    return nullptr;
  }
  auto* c = new nuts_chain();
  c->m = m; c->cfg = *cfg; c->n = m->md.n;
  m->n_chains++;
  const int n = c->n;
  c->initial_mean.assign(n, 0.0);
  c->initial_diag.assign(n, 1.0);
  if (cfg->initial_mean) c->initial_mean.assign(cfg->initial_mean, cfg->initial_mean + n);
  if (cfg->initial_diag) c->initial_diag.assign(cfg->initial_diag, cfg->initial_diag + n);
  else if (cfg->potential == NUTS_POT_DIAG_ADAPT) c->cfg.initial_weight = 1;  // quadpotential.py:280-282
  c->cfg.initial_mean = nullptr; c->cfg.initial_diag = nullptr;
  c->host_pot = cfg->potential == NUTS_POT_HOST;
  // (a host potential runs the dense schedule: the velocity is produced BETWEEN kernels, there by a mat-vec, here by the callback)
  c->dense = cfg->potential == NUTS_POT_FULL || cfg->potential == NUTS_POT_FULL_ADAPT || c->host_pot;
  c->full_adapt = cfg->potential == NUTS_POT_FULL_ADAPT;
  c->fa_mfma = env_int("NUTS_FA_MFMA", 1);
  if (c->cfg.fa_update_window < 1) c->cfg.fa_update_window = 1;
  if (c->dense) c->initial_diag.assign(m->md.n, 1.0);  // the diagonal vectors stay allocated (unused)
  c->cfg.dense_cov = nullptr; c->cfg.dense_rand = nullptr;
  c->adaptation_window = cfg->adaptation_window;
  ArenaDev& A = c->A;
  A.n = n;
  const int maxd = std::max(cfg->max_treedepth, cfg->early_max_treedepth);
  A.S = 1 << maxd;
  A.ept = m->ept;
  A.nblk = m->md.nblk;
  const size_t arena = (size_t)A.S * n;
  A.Q = c->keep(dev_alloc<double>(arena)); A.P = c->keep(dev_alloc<double>(arena));
  A.V = c->keep(dev_alloc<double>(arena)); A.G = c->keep(dev_alloc<double>(arena));
  A.E = c->keep(dev_alloc<double>(A.S)); A.LOGP = c->keep(dev_alloc<double>(A.S));
  A.PS = c->keep(dev_alloc<double>((size_t)MAX_LEVELS * n));
  A.PSUM = c->keep(dev_alloc<double>(n));
  c->kin_part = c->keep(dev_alloc<double>((size_t)A.nblk));
  A.ctl = c->keep(dev_alloc<Ctl>(1));
  c->var = c->keep(dev_alloc<double>(n)); c->stds = c->keep(dev_alloc<double>(n)); c->inv_stds = c->keep(dev_alloc<double>(n));
  c->wa_mean = c->keep(dev_alloc<double>(n)); c->wa_m2 = c->keep(dev_alloc<double>(n));
  c->wb_mean = c->keep(dev_alloc<double>(n)); c->wb_m2 = c->keep(dev_alloc<double>(n));
  A.var = c->var; A.inv_stds = c->inv_stds;
  A.ga_ticket = m->md.lg.ga ? m->md.lg.ga_ticket : nullptr; A.ga_nticket = m->md.lg.ga ? m->md.lg.ga_nblk : 0;
  A.ga_sync = m->md.lg.ga ? m->ga_sync : nullptr;
  // NUTS_GA_TREE=0: one launch per leapfrog (also what several chains SHARING a GPU must use: the tree kernel needs the chip)
  c->tree_opts = env_int("NUTS_GA_TREE_OPTS", 0) | (env_int("NUTS_GA_TREE_TICKS", 0) << GA_TREE_TICK_SHIFT);
  c->tree_mode = m->md.lg.ga && m->ga_tree_ok && !c->dense && m->md.lean_ok && env_int("NUTS_GA_TREE", 1) != 0;
  if (c->full_adapt) {
    const size_t nn = (size_t)n * n;
    c->fa_initial_cov.assign(cfg->dense_cov, cfg->dense_cov + nn);
    c->dense_C = c->keep(dev_alloc<double>(nn));
    c->fa_L = c->keep(dev_alloc<double>(nn)); c->fa_Lw = c->keep(dev_alloc<double>(nn));
    c->fa_fg_raw = c->keep(dev_alloc<double>(nn)); c->fa_bg_raw = c->keep(dev_alloc<double>(nn));
    c->fa_fg_mean = c->keep(dev_alloc<double>(n)); c->fa_bg_mean = c->keep(dev_alloc<double>(n));
    c->fa_scratch = c->keep(dev_alloc<double>(4 * (size_t)n)); c->fa_rhs = c->keep(dev_alloc<double>(n));
    c->fa_fail = c->keep(dev_alloc<int>(2));
    c->mv_grid = (n + (256 / WAVE) - 1) / (256 / WAVE);
  } else if (c->host_pot) {
    c->hp_p.assign(n, 0.0); c->hp_v.assign(n, 0.0);
    c->kin_user_dev = c->keep(dev_alloc<double>(1));
    A.kin_user = c->kin_user_dev;
  } else if (c->dense) {
    c->dense_C = c->keep(dev_upload(cfg->dense_cov, (size_t)n * n));
    c->dense_W = c->keep(dev_upload(cfg->dense_rand, (size_t)n * n));
    c->mv_grid = (n + (256 / WAVE) - 1) / (256 / WAVE);
  }
  c->n_uni_cap = (1 << maxd) + 2 * maxd + 16;
  c->stage_dev = c->keep(dev_alloc<double>(2 * (size_t)n + 2 * (size_t)c->n_uni_cap));
  c->out_dev = c->keep(dev_alloc<double>(2 * (size_t)n));
  c->out_dev2 = c->keep(dev_alloc<double>(2 * (size_t)n + 2));
  c->fold_ctl = env_int("NUTS_FOLD_CTL", 1) != 0 && !(m->md.has_mvn && m->md.mv.winv);   // (the four-launch MvNormal pass has no workgroup 0 for it)
  c->spec_max = env_int("NUTS_SPEC_MAX", 10);
  c->xfold = env_int("NUTS_XFOLD", 1);
  c->pipe_draws = env_int("NUTS_PIPE_DRAWS", 1) != 0;
  c->xpre = env_int("NUTS_XPRE", 1);
  // NUTS_SMALL_LDS=0: the single-workgroup kernel with its tree in global memory, as before round 5 (A/B); NUTS_SMALL_LDS_SLOTS=k
  // (a power of two): at most k slots in LDS, so that ordinary trees cross over to the global arena (tests)
  c->small_lds_slots = env_int("NUTS_SMALL_LDS", 1) == 0 ? -1 : env_int("NUTS_SMALL_LDS_SLOTS", 0);
  c->small_one_wave = env_int("NUTS_SMALL_ONE_WAVE", 1);
  c->group_wide = env_int("NUTS_GROUP_WIDE", 0);
  // (... and whose factors are small too: the single workgroup walks every factor element itself -- SMALL_MAX_ELEMS, round 6)
  // (round 6: a GLM node with few covariates and rows is evaluated inside that launch -- small_kernel.h GLM_SMALL_P / GLM_SMALL_N;
  // NUTS_GLM_SMALL = 0: the general path's four launches per leapfrog, A/B and tests)
  const bool glm_small = m->md.has_glm && env_int("NUTS_GLM_SMALL", 1) != 0 && m->md.glm.Xt && !m->md.glm.beta_buf && m->md.glm.off_beta >= 0 && m->md.n_derived == 0 && n <= 256;   // (the 256-thread variant carries the node)
  c->small = env_int("NUTS_SMALL_KERNEL", 1) != 0 && n <= SMALL_MAX_N && m->ept == 1 && !m->md.has_logit && !m->md.has_mvn && !m->md.has_mix &&
             (!m->md.has_glm || glm_small) &&
             m->md.n_lins == 0 && m->factor_elems <= SMALL_MAX_ELEMS && !c->dense;   // (the single-workgroup kernel knows diagonal potentials only)
  c->do_dev = c->keep(dev_alloc<DrawOut>(1));
  A.uniforms = c->stage_dev + 2 * (size_t)n;
  A.log_uniforms = A.uniforms + c->n_uni_cap;
  for (void* p : c->owned)
    if (!p) { g_err = "device allocation failed (trajectory arena needs 4*2^max_treedepth*n*8 bytes)"; nuts_chain_destroy(c); return nullptr; }
  if (hipHostMalloc((void**)&c->stage_host, (2 * (size_t)n + 2 * (size_t)c->n_uni_cap) * sizeof(double), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->out_host, 2 * (size_t)n * sizeof(double), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->st_host, sizeof(HostStatus), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->st_dev, c->st_host, 0) != hipSuccess ||
      hipHostMalloc((void**)&c->do_host, sizeof(DrawOut), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->dom_host, DOM_RING * sizeof(DrawOutMapped), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->dom_dev, c->dom_host, 0) != hipSuccess) {
    g_err = "pinned host allocation failed"; nuts_chain_destroy(c); return nullptr;
  }
  hipMemset(A.ctl, 0, sizeof(Ctl));
  std::memset(c->st_host, 0, sizeof(HostStatus));
  std::memset(c->dom_host, 0, DOM_RING * sizeof(DrawOutMapped));
  c->step_size = cfg->step_scale / std::pow((double)n, 0.25);  // base_hmc.py:161
  c->da = DualAvg{c->step_size, cfg->target_accept, cfg->gamma, cfg->k, cfg->t0, 0, 0, 0, 0, 1};
  c->da.reset();
  if (potential_reset(c) != NUTS_OK) { nuts_chain_destroy(c); return nullptr; }
  return c;
}

extern "C" void nuts_chain_destroy(nuts_chain* c) {
  if (!c) return;
  if (c->m) hipStreamSynchronize(c->m->stream);
  if (c->m && c->m->group) group_remove_model(c->m->group, c->m);
  if (c->m) c->m->n_chains--;
  for (void* p : c->owned) if (p) hipFree(p);
  if (c->stage_host) hipHostFree(c->stage_host);
  if (c->out_host) hipHostFree(c->out_host);
  if (c->st_host) hipHostFree(c->st_host);
  if (c->do_host) hipHostFree(c->do_host);
  if (c->dom_host) hipHostFree(c->dom_host);
  if (c->many_in_host) hipHostFree(c->many_in_host);
  if (c->many_out_host) hipHostFree(c->many_out_host);
  if (c->many_in_dev) hipFree(c->many_in_dev);
  if (c->many_out_dev) hipFree(c->many_out_dev);
  delete c;
}

// ---- chain groups (include/nuts_mi355.h) ----
static void group_remove_model(nuts_group* g, nuts_model* m) {
  if (!g || !m || m->group != g) return;
  hipStreamSynchronize(g->stream);
  std::lock_guard<std::mutex> lk(g->mu);
  for (int i = 0; i < GRP_MAXC; ++i)
    if (g->member[i] == m) { g->member[i] = nullptr; g->n--; g->konst_set[i] = false; if (i < GAL_MAXC) g->gal_konst_set[i] = false; }
  if (m->g_active) { g->nactive--; m->g_active = false; }
  m->group = nullptr;
  m->stream = m->own_stream;
}

extern "C" nuts_group* nuts_group_create(void) {
  nuts_group* g = new nuts_group();
  if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { g_err = "nuts_group_create: no stream"; delete g; return nullptr; }
  return g;
}

extern "C" int nuts_group_add(nuts_group* g, nuts_chain* c) {
  if (!g || !c || !c->m) return NUTS_E_ARG;
  nuts_model* m = c->m;
  const MvnDev& mv = m->md.mv;
  if (m->group) { g_err = "nuts_group_add: the chain's model already belongs to a group"; return NUTS_E_ARG; }
  if (m->n_chains != 1) { g_err = "nuts_group_add: a member model carries exactly one chain (its launch parity and records are the chain's)"; return NUTS_E_ARG; }
  const RowsDev& lg = m->md.lg;
  const bool is_mvn = m->md.has_mvn && (mv.aligned == 4 || mv.aligned == 8);
  // the hierarchical-logit rows on the group-aligned pass, closed-form model (the benchmark's), D = 8: rows_ga_multi_kernel.h
  // ... or on the group-BLOCK pass (small groups, C2-S): rows_gb_multi_kernel.h
  const bool is_rows = m->md.has_logit && !m->md.has_mvn && lg.ga && lg.ga_naux == 0 && lg.D == 8 && m->ga_struct_ok == 1 && (lg.ga_gpw > 0 || m->ga_variant == 42);
  const int rows_kind = lg.ga_gpw > 0 ? 3 : 2;
  if (!(is_mvn || is_rows) || c->dense || c->host_pot) {
    g_err = "nuts_group_add: chain groups advance models that are one constant-covariance MvNormal node on the row-aligned pass or the "
            "hierarchical-logit rows on the group-aligned pass (diagonal mass matrix); this chain is neither";
    return NUTS_E_ARG;
  }
  // a group of MvNormal models laid out 8 rows per workgroup whose FIRST member asked for it (option NUTS_GROUP_WIDE = 1 when the
  // member's chain was created) is WIDE: up to 16 chains per launch through the matrix cores
  // (the rows group carries up to eight chains through the LDS-shared launch, rows_gal_kernel.h; NUTS_ROWS_GROUP_LDS = 0 when the first
  // member joins: the round-5 kernel and its four)
  const int rows_lds = g->n == 0 ? env_int("NUTS_ROWS_GROUP_LDS", 1) : g->rows_lds;
  const int cap = g->n == 0 ? (is_rows ? ((rows_lds || rows_kind == 3) ? GAL_MAXC : GAM_MAXNC) : ((is_mvn && mv.aligned == 8 && mv.k % 16 == 0 && c->group_wide) ? GRP_MAXC : MVM_MAXC)) : g->cap;
  if (g->n >= cap) {
    g_err = cap == GRP_MAXC ? "nuts_group_add: a wide group holds at most 16 chains" : (cap == GAL_MAXC ? "nuts_group_add: a rows group holds at most 8 chains" : "nuts_group_add: a group holds at most 4 chains");
    return NUTS_E_ARG;
  }
  if (g->n > 0 && g->kind != (is_rows ? rows_kind : 1)) { g_err = "nuts_group_add: not the same model as the group's"; return NUTS_E_ARG; }
  HIPCHK(hipStreamSynchronize(m->stream));
  if (is_rows) {
    if (nuts_model* base = group_base(g)) {
      // every launch streams the first member's copy of (X, y): the newcomer's must be the same numbers in the same layout
      const RowsDev& bl = base->md.lg;
      if (bl.N != lg.N || bl.G != lg.G || bl.D != lg.D || bl.ga_dx != lg.ga_dx || bl.ga_w != lg.ga_w || bl.ga_gpw != lg.ga_gpw || bl.ga_nblk != lg.ga_nblk || bl.Npad != lg.Npad || bl.ga_T_uni != lg.ga_T_uni ||
          bl.ga_cstride_uni != lg.ga_cstride_uni || bl.ga_bsz != lg.ga_bsz || bl.ga_nrec != lg.ga_nrec || bl.sigma_tr != lg.sigma_tr ||
          bl.off_mu != lg.off_mu || bl.off_sigma != lg.off_sigma || bl.off_z != lg.off_z || base->md.n != m->md.n ||
          std::memcmp(&bl.z_np_mu, &lg.z_np_mu, 8 * sizeof(double)) != 0) {
        g_err = "nuts_group_add: not the same model as the group's"; return NUTS_E_ARG;
      }
      const size_t nx = (size_t)base->rows_xt_len, ny = (size_t)base->rows_y_len;
      if (nx != (size_t)m->rows_xt_len || ny != (size_t)m->rows_y_len) { g_err = "nuts_group_add: not the same model as the group's"; return NUTS_E_ARG; }
      // (compared in slabs: X is 320 MB at the benchmark's shape)
      const size_t slab = (size_t)1 << 22;
      std::vector<double> a(slab), b(slab);
      for (size_t o = 0; o < nx; o += slab) {
        const size_t k = std::min(slab, nx - o);
        HIPCHK(hipMemcpy(a.data(), bl.Xt + o, k * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), lg.Xt + o, k * sizeof(double), hipMemcpyDeviceToHost));
        if (std::memcmp(a.data(), b.data(), k * sizeof(double)) != 0) { g_err = "nuts_group_add: not the same model as the group's (the design matrices differ)"; return NUTS_E_ARG; }
      }
      std::vector<int8_t> ya(ny), yb(ny);
      HIPCHK(hipMemcpy(ya.data(), bl.y, ny, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(yb.data(), lg.y, ny, hipMemcpyDeviceToHost));
      if (std::memcmp(ya.data(), yb.data(), ny) != 0) { g_err = "nuts_group_add: not the same model as the group's (the observations differ)"; return NUTS_E_ARG; }
    }
  } else
  if (nuts_model* base = group_base(g)) {
    // every launch reads the first member's (P, mu): the newcomer's must be the same numbers
    const MvnDev& bv = base->md.mv;
    if (bv.k != mv.k || bv.aligned != mv.aligned || bv.konst != mv.konst) { g_err = "nuts_group_add: not the same model as the group's"; return NUTS_E_ARG; }
    const size_t kk = (size_t)mv.k * mv.k;
    std::vector<double> a(kk + mv.k), b(kk + mv.k);
    HIPCHK(hipMemcpy(a.data(), bv.prec, kk * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data(), mv.prec, kk * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(a.data() + kk, bv.mu, mv.k * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(b.data() + kk, mv.mu, mv.k * sizeof(double), hipMemcpyDeviceToHost));
    if (std::memcmp(a.data(), b.data(), a.size() * sizeof(double)) != 0) { g_err = "nuts_group_add: not the same model as the group's (precision or mean differ)"; return NUTS_E_ARG; }
  }
  if (is_rows && (rows_lds || rows_kind == 3) && !g->gal_konst_dev) {
    HIPCHK(hipMalloc((void**)&g->gal_konst_dev, GAL_MAXC * sizeof(GalConst)));
    HIPCHK(hipMemset(g->gal_konst_dev, 0, GAL_MAXC * sizeof(GalConst)));
  }
  if (!is_rows && cap > MVM_MAXC && !g->konst_dev) {
    HIPCHK(hipMalloc((void**)&g->md_dev, sizeof(ModelDev)));
    HIPCHK(hipMalloc((void**)&g->dpack, (size_t)mv.k * MFM_MAXC * sizeof(double)));
    HIPCHK(hipMalloc((void**)&g->konst_dev, GRP_MAXC * sizeof(MfmChainConst)));
    HIPCHK(hipMemset(g->konst_dev, 0, GRP_MAXC * sizeof(MfmChainConst)));
  }
  std::lock_guard<std::mutex> lk(g->mu);
  g->cap = cap;
  g->rows_lds = rows_lds;
  // (NUTS_GBM_OCC, measured at C2-S with 4 / 8 chains: 107 k / 128 k at the unbounded 175 registers, 88 k / 114 k at 128, 58 k / 66 k
  // at 80 -- the spills cost more than the rounds they save)
  if (g->n == 0) { g->gal_occ5 = env_int("NUTS_GAL_OCC5", 1); g->gal_pf3 = env_int("NUTS_GAL_PF3", 0); g->gbm_occ = env_int("NUTS_GBM_OCC", 2); }
  for (int i = 0; i < cap; ++i)
    if (!g->member[i]) { g->member[i] = m; m->gslot = i; break; }
  g->n++;
  g->kind = is_rows ? rows_kind : 1;
  m->group = g;
  m->stream = g->stream;
  return NUTS_OK;
}

extern "C" int nuts_group_remove(nuts_group* g, nuts_chain* c) {
  if (!g || !c || !c->m || c->m->group != g) return NUTS_E_ARG;
  group_remove_model(g, c->m);
  return NUTS_OK;
}

extern "C" void nuts_group_destroy(nuts_group* g) {
  if (!g) return;
  for (int i = 0; i < GRP_MAXC; ++i)
    if (g->member[i]) group_remove_model(g, g->member[i]);
  hipStreamDestroy(g->stream);
  if (g->konst_dev) hipFree(g->konst_dev);
  if (g->gal_konst_dev) hipFree(g->gal_konst_dev);
  if (g->md_dev) hipFree(g->md_dev);
  if (g->dpack) hipFree(g->dpack);
  delete g;
}

extern "C" int nuts_group_launches(nuts_group* g, int64_t* by_chains) {
  if (!g || !by_chains) return NUTS_E_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  for (int i = 0; i <= MVM_MAXC; ++i) by_chains[i] = g->launches[i];
  return NUTS_OK;
}
// ... of a wide group: by_chains [17]; returns the group's capacity (4 or 16) in *cap
extern "C" int nuts_group_launches_wide(nuts_group* g, int64_t* by_chains, int32_t* cap) {
  if (!g || !by_chains) return NUTS_E_ARG;
  std::lock_guard<std::mutex> lk(g->mu);
  for (int i = 0; i <= GRP_MAXC; ++i) by_chains[i] = g->launches[i];
  if (cap) *cap = g->cap;
  return NUTS_OK;
}
extern "C" int nuts_unset_option(const char* name) {
  if (!name) return NUTS_E_ARG;
  std::lock_guard<std::mutex> lk(g_opt_mu);
  for (size_t i = 0; i < g_opts.size(); ++i) if (g_opts[i].first == name) { g_opts.erase(g_opts.begin() + i); break; }
  return NUTS_OK;
}

extern "C" int nuts_chain_reset_tuning(nuts_chain* c) {  // base_hmc.py:290-298
  if (!c) return NUTS_E_ARG;
  c->da.reset();
  c->iter_count = 0; c->divergences = 0; c->tune = true;
  return potential_reset(c);
}
extern "C" int nuts_chain_set_tune(nuts_chain* c, int tune) { if (!c) return NUTS_E_ARG; c->tune = tune != 0; return NUTS_OK; }
extern "C" int nuts_chain_set_iter_count(nuts_chain* c, int64_t it) { if (!c) return NUTS_E_ARG; c->iter_count = it; return NUTS_OK; }

static int check_mass_matrix(nuts_chain* c) {  // quadpotential.py:357-393 raise_ok
  if (c->full_adapt) {   // quadpotential.py:845-847: a factorisation that failed during adaptation is reported here
    int fail[2] = {0, 0};
    HIPCHK(hipMemcpy(fail, c->fa_fail, sizeof(fail), hipMemcpyDeviceToHost));
    if (fail[1]) { g_err = "the adapted covariance is not positive definite (Cholesky factorisation failed)"; return NUTS_E_LINALG; }
    return NUTS_OK;
  }
  if (c->dense) return NUTS_OK;
  std::vector<double> st(c->n);
  HIPCHK(hipMemcpy(st.data(), c->stds, c->n * sizeof(double), hipMemcpyDeviceToHost));
  for (double s : st) {
    if (s == 0) { g_err = "Mass matrix contains zeros on the diagonal. "; return NUTS_E_BAD_ENERGY; }
    if (!std::isfinite(s)) { g_err = "Mass matrix contains non-finite values on the diagonal. "; return NUTS_E_BAD_ENERGY; }
  }
  return NUTS_OK;
}

static int potential_update(nuts_chain* c, const double* x_dev, const double* g_dev) {  // quadpotential.py:335-355
  c->window_switched = false;
  if (c->full_adapt) return fa_update(c, x_dev);
  if (c->cfg.potential == NUTS_POT_DIAG_ADAPT_EXP) {   // quadpotential.py:534-569
    if (!(c->tune && (double)c->n_samples < c->cfg.exp_stop_adaptation)) return NUTS_OK;
    const int64_t k = c->n_samples, dw = c->cfg.discard_window;
    int flags = 0;
    if (k > dw) flags |= 2;
    else if (k == dw) flags |= 1;
    if (k > 2 * dw) flags |= 4;
    if (flags) {
      const int grid = std::max(1, std::min(1024, (c->n + VEC_THREADS - 1) / VEC_THREADS));
      hipLaunchKernelGGL(k_potential_update_exp, dim3(grid), dim3(VEC_THREADS), 0, c->m->stream, c->n, x_dev, g_dev, c->wa_mean,
                         c->wa_m2, c->wb_mean, c->wb_m2, c->cfg.exp_alpha, 1.0 - c->cfg.exp_alpha, c->cfg.exp_use_grads, c->var, c->stds,
                         c->inv_stds, flags);
    }
    c->n_samples += 1;
    return NUTS_OK;
  }
  if (c->cfg.potential != NUTS_POT_DIAG_ADAPT || !c->tune) return NUTS_OK;
  hipStream_t s = c->m->stream;
  int flags = 0;
  if (c->n_samples > c->cfg.discard_window) { flags |= 1; c->fg_count += 1; c->bg_count += 1; }
  if (c->cfg.early_update || c->n_samples > c->adaptation_window) flags |= 2;
  double *fm = c->fg_is_a ? c->wa_mean : c->wb_mean, *f2 = c->fg_is_a ? c->wa_m2 : c->wb_m2;
  double *bm = c->fg_is_a ? c->wb_mean : c->wa_mean, *b2 = c->fg_is_a ? c->wb_m2 : c->wa_m2;
  if (flags) {
    if ((flags & 2) && c->fg_count == 0) { g_err = "Can not compute variance without samples."; return NUTS_E_ARG; }
    const int grid = std::max(1, std::min(1024, (c->n + VEC_THREADS - 1) / VEC_THREADS));
    hipLaunchKernelGGL(k_potential_update, dim3(grid), dim3(VEC_THREADS), 0, s, c->n, x_dev, fm, f2, c->fg_count, bm, b2,
                       c->bg_count, c->var, c->stds, c->inv_stds, flags);
  }
  if (c->n_samples > 0 && c->n_samples % c->adaptation_window == 0) {
    c->fg_is_a = !c->fg_is_a;  // foreground <- background
    c->window_switched = true;
    c->fg_count = c->bg_count;
    HIPCHK(hipMemsetAsync(fm, 0, c->n * sizeof(double), s));  // old foreground becomes the fresh background
    HIPCHK(hipMemsetAsync(f2, 0, c->n * sizeof(double), s));
    c->bg_count = 0;
    c->adaptation_window = (int64_t)(c->adaptation_window * c->cfg.adaptation_window_multiplier);
  }
  c->n_samples += 1;
  return NUTS_OK;
}

// Upload (q0, normals|p, uniforms), evaluate the model at q0 (plain A/B/C) and initialise the trajectory.
//   p_exact: the second vector is the momentum itself (integrator tests) instead of standard normals
//   dir_forced: +1/-1 fixes the direction (HMC / integrator tests); 0 = draw it from uniforms[0] (nuts.py:215)
#define LOGS_FIRST 80   // doublings 0..5 consume uniform indices < 2^6 + 6 = 70

// make sure the logarithms of uniforms [0, upto) are on the device before the launches that may read them are queued
static int ensure_logs(nuts_chain* c, int upto) {
  upto = std::min(upto, c->logs_total);
  if (upto <= c->logs_done) return NUTS_OK;
  const int n = c->n;
  double* u = c->stage_host + 2 * n;
  double* lu = u + c->n_uni_cap;
  for (int i = c->logs_done; i < upto; ++i) lu[i] = std::log(u[i]);
  HIPCHK(hipMemcpyAsync(c->stage_dev + 2 * (size_t)n + c->n_uni_cap + c->logs_done, lu + c->logs_done,
                        (size_t)(upto - c->logs_done) * sizeof(double), hipMemcpyHostToDevice, c->m->stream));
  c->logs_done = upto;
  return NUTS_OK;
}

// velocity of a chain whose potential is not diagonal, y = velocity(x) for device vectors x, y (and q_out = q_in + eps y when
// given: the first half of a leapfrog, integration.py:121-127).  Dense matrices: one mat-vec launch.  Host potentials
// (NUTS_POT_HOST): the stream is drained, x goes to the host, the caller's method runs, y comes back -- `site` says which of the
// reference's three call sites this is, because `energy` / `velocity_energy` also return the kinetic energy the next control
// kernel must use (kernels.h, ArenaDev.kin_user).  Errors are parked in c->cb_err (the callers queue launches and cannot return).
enum { VEL_ONLY = 0, VEL_START = 1, VEL_LEAF = 2 };
static void dense_velocity(nuts_chain* c, const double* x, double* y, const double* q_in, double* q_out, double eps, const int* abort_flag,
                           int site) {
  const int n = c->n;
  hipStream_t s = c->m->stream;
  if (!c->host_pot) {
    hipLaunchKernelGGL(k_dense_mv, dim3(c->mv_grid), dim3(256), 0, s, c->dense_C, x, y, n, q_in, q_out, eps, abort_flag);
    return;
  }
  if (c->cb_err) return;
  auto hip_failed = [&](hipError_t e) {
    if (e == hipSuccess) return false;
    g_err = std::string("host potential: ") + hipGetErrorString(e); c->cb_err = NUTS_E_HIP;
    return true;
  };
  if (hip_failed(hipStreamSynchronize(s))) return;
  if (abort_flag) {   // the tree already ended inside this doubling: the remaining leaves only drain, the caller's code is not run for them
    int ab = 0;
    if (hip_failed(hipMemcpy(&ab, abort_flag, sizeof(int), hipMemcpyDeviceToHost))) return;
    if (ab) return;
  }
  if (hip_failed(hipMemcpy(c->hp_p.data(), x, n * sizeof(double), hipMemcpyDeviceToHost))) return;
  double kin = 0.0;
  int rc;
  if (site == VEL_LEAF) rc = c->hp_velocity_energy(c->hp_user, n, c->hp_p.data(), c->hp_v.data(), &kin);
  else {
    rc = c->hp_velocity(c->hp_user, n, c->hp_p.data(), c->hp_v.data());
    if (!rc && site == VEL_START) rc = c->hp_energy(c->hp_user, n, c->hp_p.data(), c->hp_v.data(), &kin);
  }
  if (rc) { g_err = "host potential: a callback reported an error"; c->cb_err = NUTS_E_CALLBACK; return; }
  if (hip_failed(hipMemcpy(y, c->hp_v.data(), n * sizeof(double), hipMemcpyHostToDevice))) return;
  if (site != VEL_ONLY && hip_failed(hipMemcpy(c->kin_user_dev, &kin, sizeof(double), hipMemcpyHostToDevice))) return;
  if (q_out) hipLaunchKernelGGL(k_host_pot_drift, dim3((n + 255) / 256), dim3(256), 0, s, q_in, (const double*)y, q_out, eps, n);
}

// a failed host-potential callback: whatever was queued behind it is drained, the error goes to the caller
static int host_pot_error(nuts_chain* c) {
  const int rc = c->cb_err;
  c->cb_err = 0;
  (void)hipStreamSynchronize(c->m->stream);
  c->cache_ok = false;
  return rc;
}

static int draw_begin(nuts_chain* c, const double* q0, const double* normals, const double* uniforms, int n_uniforms,
                      double step_size, int max_depth, bool p_exact, int dir_forced, bool allow_cache = false) {
  const int n = c->n;
  if (c->host_pot) {
    if (!c->hp_velocity) { g_err = "NUTS_POT_HOST chain: nuts_chain_set_host_potential has not been called"; return NUTS_E_ARG; }
    p_exact = true;   // `normals` IS potential.random()
  }
  // the usual case inside a chain: q0 is bit for bit the proposal this chain returned last time, whose gradient and
  // logp are still on the device -- the model pass at q0 would reproduce exactly those numbers
  const bool cached = allow_cache && c->cache_ok && c->cache_epoch == c->m->data_epoch && std::memcmp(q0, c->last_q.data(), n * sizeof(double)) == 0;
  c->cache_ok = false;
  hipStream_t s = c->m->stream;
  ArenaDev& A = c->A;
  const int nu = std::min(n_uniforms, c->n_uni_cap);
  std::memcpy(c->stage_host, q0, n * sizeof(double));
  std::memcpy(c->stage_host + n, normals, n * sizeof(double));
  if (nu > 0) {
    // the uniforms and, right behind them (at a fixed offset), their logarithms: `np.log(rng.random())` is what the
    // tree compares (nuts.py:371,466); taking the log on the host keeps a long scalar chain off the control kernel
    double* u = c->stage_host + 2 * n;
    double* lu = u + c->n_uni_cap;
    std::memcpy(u, uniforms, nu * sizeof(double));
    // most trees stop long before the worst case: only the logarithms the first doublings can consume are taken
    // here, the rest just before the doubling that could reach them is queued (ensure_logs)
    const int first = std::min(nu, LOGS_FIRST);
    for (int i = 0; i < first; ++i) lu[i] = std::log(u[i]);
    c->logs_done = first; c->logs_total = nu;
    HIPCHK(hipMemcpyAsync(c->stage_dev, c->stage_host, (2 * (size_t)n + c->n_uni_cap + first) * sizeof(double), hipMemcpyHostToDevice, s));
  } else {
    HIPCHK(hipMemcpyAsync(c->stage_dev, c->stage_host, 2 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  }
  if (!cached) {
    HIPCHK(hipMemcpyAsync(A.Q, c->stage_dev, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    model_enqueue_plain(c->m, A.Q, A.G, A.LOGP);
  }
  if (c->dense) {
    // p0 = W z (or the given momentum), v0 = C p0   (quadpotential.py:704-711)
    if (p_exact) HIPCHK(hipMemcpyAsync(A.P, c->stage_dev + n, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    else if (c->full_adapt) fa_random(c, c->stage_dev + n, A.P);
    else hipLaunchKernelGGL(k_dense_mv, dim3(c->mv_grid), dim3(256), 0, s, c->dense_W, c->stage_dev + n, A.P, n, (const double*)nullptr,
                            (double*)nullptr, 0.0, (const int*)nullptr);
    dense_velocity(c, A.P, A.V, nullptr, nullptr, 0.0, nullptr, VEL_START);
    if (c->cb_err) return host_pot_error(c);
  }
  hipLaunchKernelGGL(k_draw_start, dim3(A.nblk), dim3(VEC_THREADS), 0, s, A, c->stage_dev + n,
                     p_exact ? (const double*)(c->stage_dev + n) : (const double*)nullptr, c->kin_part,
                     cached ? (const double*)c->out_dev : (const double*)nullptr, cached ? (const double*)(c->out_dev + n) : (const double*)nullptr,
                     c->dense);
  hipLaunchKernelGGL(k_draw_ctl_start, dim3(1), dim3(64), 0, s, A, c->kin_part, step_size, dir_forced, max_depth, c->st_dev,
                     cached ? 1 : 0, c->last_logp, (const DrawOut*)nullptr);
  return NUTS_OK;
}

static int sync_status(nuts_chain* c) {   // the status record is host memory: a stream sync makes it current
  HIPCHK(hipStreamSynchronize(c->m->stream));
  return NUTS_OK;
}

// Wait until the control kernel of the last leaf of a doubling has published sequence number `seq`
// (spin on the mapped record; falls back to an error after 60 s so that a lost kernel cannot hang the process).
static int wait_status(nuts_chain* c, int seq, unsigned* flags, int* cursor = nullptr) {
  volatile unsigned long long* word = &c->st_host->word[seq & (ST_SLOTS - 1)];
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long w;
  // (member of a chain group: the chain stays a lockstep partner while it waits -- everything this status depends on has been
  // submitted, since a deposit returns only after its launch was; partners that are ahead wait at their next deposit until this
  // chain has seen its status and deposits too, which keeps the chains of a group leaf by leaf in the same launches.  Letting
  // the others go on instead was measured: the host runs far ahead of the device, every chain spends most of its time here, and
  // the chains took turns -- 1.17 chains per launch.)
  for (unsigned spins = 0; (unsigned)((w = *word) >> 32) != (unsigned)seq; ++spins) {
    if ((spins & 0xfffff) == 0xfffff) {
      if (hipStreamQuery(c->m->stream) == hipSuccess && (unsigned)(*word >> 32) != (unsigned)seq) {
        g_err = "control kernel finished without publishing its status";
        return NUTS_E_HIP;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { g_err = "timed out waiting for the device"; return NUTS_E_HIP; }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  *flags = (unsigned)w & 0xffu;
  if (cursor) *cursor = (int)(((unsigned)w >> 8) & 0xffffffu);
  return NUTS_OK;
}

// one leapfrog leaf = A (data pass) + B (O(n) work) + C (control); see kernels.h
struct Geometry {  // of the doubling being built: direction, edge state, current edge indices, signed step
  int dir, edge, left, right;
  double eps;
};

// `n_simple`: MODE_SIMPLE only -- the length of the fixed trajectory being queued (leaf j of n_simple), so that the folded
// schedules below know which leaf is the last one
static inline void enqueue_leaf(nuts_chain* c, const Geometry& gm, int j, int d, int mode, int max_depth, int seq = 0,
                                int n_simple = 0) {
  ArenaDev& A = c->A;
  nuts_model* m = c->m;
  hipStream_t s = m->stream;
  EvalIO io{};
  io.mode = mode; io.explicit_pre = m->explicit_pre || c->dense; io.dense = c->dense;
  io.dir = gm.dir; io.edge = gm.edge; io.left = gm.left; io.right = gm.right; io.eps = gm.eps;
  const int* abort_flag = mode == MODE_TREE ? &A.ctl->aborted : nullptr;
  const int t = gm.edge + gm.dir * (j + 1), src = gm.edge + gm.dir * j;
  const int64_t d_o = (int64_t)(t & (A.S - 1)) * A.n, so = (int64_t)(src & (A.S - 1)) * A.n;
  HostStatus* const st = mode == MODE_TREE ? c->st_dev : (HostStatus*)nullptr;
  io.lean = m->md.lean_ok && !c->dense && (!io.explicit_pre || m->md.n_deferred == 0);
  const bool foldable = c->fold_ctl && (mode == MODE_TREE || (mode == MODE_SIMPLE && n_simple > 0));
  const bool last = mode == MODE_TREE ? j + 1 == (1 << d) : j + 1 == n_simple;
  if (io.lean && io.explicit_pre && foldable) {
    // MvNormal model on the lean path: kernel B of leaf j also materialises the first half of leaf j+1, the control
    // work of leaf j-1 rides in workgroup 0 of this leaf's mat-vec; the last leaf gets a control launch of its own
    const bool al_xpre = m->md.mv.aligned && mode == MODE_TREE && c->xpre;
    if (j == 0 && !(al_xpre && c->pre_done)) hipLaunchKernelGGL(k_leaf_pre, dim3(A.nblk), dim3(VEC_THREADS), 0, s, A, io, j);
    if (j == 0) c->pre_done = false;
    io.pre_next = last ? 0 : 1;
    if (last && al_xpre && c->defer_last_ctl && c->next_dir != 0) {
      // the next doubling is queued right behind this leaf: same side -> its first leaf starts from this one (what pre_next = 1
      // writes); other side -> from the tree's other edge state (pre_next = 3: k_leaf_pre's arithmetic on this workgroup's rows)
      io.pre_next = c->next_dir == gm.dir ? 1 : 3;
      c->pre_done = true;
    }
    // row-aligned pass: as for the group-aligned row pass below, the control work of a doubling's LAST leaf rides in the first
    // launch of the next doubling when the look-ahead queues that doubling right behind it (run_tree sets `defer_last_ctl`)
    const bool al_tree_leaf = m->md.mv.aligned && mode == MODE_TREE;
    CtlJob* job = (al_tree_leaf && j == 0 && c->pend_valid) ? &c->pend : nullptr;
    launch_dense(m, A, io, j, (j > 0 || job) ? 1 : 0, d, c->cfg.Emax, max_depth, st, job);
    if (job) c->pend_valid = false;
    launch_vector(m, A, io, j, d);
    if (last) {
      if (al_tree_leaf && c->defer_last_ctl) { c->pend = CtlJob{io, j, d, seq, false}; c->pend_valid = true; }
      else launch_control_lean(m, A, io, j, d, c->cfg.Emax, max_depth, st, seq);
    }
    c->leapfrogs++;
    return;
  }
  if (io.explicit_pre) hipLaunchKernelGGL(k_leaf_pre, dim3(A.nblk), dim3(VEC_THREADS), 0, s, A, io, j);
  if (c->dense)   // v = C p_half ; q' = q + eps v   (integration.py:121-127 with a dense velocity)
    dense_velocity(c, A.P + d_o, A.V + d_o, A.Q + so, A.Q + d_o, gm.eps, abort_flag, VEL_ONLY);
  if (io.lean && foldable) {
    // folded control (kernels.h): the control work of leaf j-1 rides in workgroup 0 of this leaf's row pass; only the
    // last leaf of the doubling (of the fixed-length trajectory) -- whose result the host waits for -- gets a control
    // launch of its own
    // Group-aligned row pass: the control work of a doubling's LAST leaf can ride in the first row pass of the next doubling
    // when that doubling is queued by the look-ahead right behind it (run_tree sets `defer_last_ctl`): one launch less per
    // doubling between two row passes.
    const bool ga_tree_leaf = m->md.lg.ga && mode == MODE_TREE;
    CtlJob* job = nullptr;
    if (ga_tree_leaf && j == 0 && c->pend_valid) {
      c->pend.src_prev = c->pend.io.dir == io.dir;   // growing on the same side: this leaf starts from the leaf just finished
      job = &c->pend;
    }
    launch_dense(m, A, io, j, (j > 0 || job) ? 1 : 0, d, c->cfg.Emax, max_depth, st, job);
    if (job) c->pend_valid = false;
    launch_vector(m, A, io, j, d);
    if (last) {
      if (ga_tree_leaf && c->defer_last_ctl) { c->pend = CtlJob{io, j, d, seq, false}; c->pend_valid = true; }
      else launch_control_lean(m, A, io, j, d, c->cfg.Emax, max_depth, st, seq);
    }
    c->leapfrogs++;
    return;
  }
  // (the group-aligned row pass finishes the leaf's z elements itself -- merges included -- so it needs `d` also when no
  // control work rides in it)
  launch_dense(m, A, io, j, 0, d, c->cfg.Emax, max_depth, st);
  launch_vector(m, A, io, j, d);
  if (io.lean) launch_control_lean(m, A, io, j, d, c->cfg.Emax, max_depth, st, seq);
  else if (m->has_prog) hipLaunchKernelGGL(k_control<true>, dim3(1), dim3(VEC_THREADS), 0, s, m->md, A, io, j, d, c->cfg.Emax, max_depth, st, seq);
  else hipLaunchKernelGGL(k_control<false>, dim3(1), dim3(VEC_THREADS), 0, s, m->md, A, io, j, d, c->cfg.Emax, max_depth, st, seq);
  if (c->dense) {   // v' = C p', then the tree work on the stored (p', v')
    dense_velocity(c, A.P + d_o, A.V + d_o, nullptr, nullptr, 0.0, abort_flag, VEL_LEAF);
    const dim3 grid(m->md.nblk);
    switch (m->ept) {
      case 1: hipLaunchKernelGGL(k_tree_vec<1>, grid, dim3(VEC_THREADS), 0, s, m->md, A, io, j, d); break;
      case 4: hipLaunchKernelGGL(k_tree_vec<4>, grid, dim3(VEC_THREADS), 0, s, m->md, A, io, j, d); break;
      default: hipLaunchKernelGGL(k_tree_vec<16>, grid, dim3(VEC_THREADS), 0, s, m->md, A, io, j, d); break;
    }
    hipLaunchKernelGGL(k_tree_ctl, dim3(1), dim3(VEC_THREADS), 0, s, m->md, A, io, j, d, c->cfg.Emax, max_depth,
                       mode == MODE_TREE ? c->st_dev : (HostStatus*)nullptr, seq);
  }
  c->leapfrogs++;
}

// The same transition as ONE launch of the persistent tree kernel (rows_ga_tree.h): the leaf loop and the doubling loop run on
// the device, the host waits for the status word of the whole tree.  Every logarithm the tree can consume must be on the
// device before the launch (`ensure_logs` up to the worst case).
static int run_tree_ga(nuts_chain* c, const double* uniforms, double step_size, int max_depth, unsigned* flags_out, bool* exhausted_out,
                       int* cursor_out = nullptr) {
  nuts_model* m = c->m;
  ArenaDev& A = c->A;
  int rc = ensure_logs(c, (1 << max_depth) + max_depth + 1);
  if (rc) return rc;
  const int seq = ++c->seq;
  GaTreeArgs ga{};
  ga.md = m->md; ga.A = A;
  ga.max_depth = max_depth;
  ga.spec_depth = c->last_depth - 1;   // speculate across doubling boundaries as deep as the previous draw's tree went
  ga.rev0 = m->rows_flip; ga.par0 = m->ga_par; ga.alternate = m->rows_alternate;
  ga.first_dir = uniforms[0] < 0.5 ? 1 : -1;
  ga.seq = seq;
  ga.Emax = c->cfg.Emax; ga.eps_abs = step_size;
  ga.st = c->st_dev;
  ga.timeout = (long long)env_int("NUTS_GA_TREE_TIMEOUT_MS", 50) * 100000ll;   // 100 MHz ticks
  ga.opts = c->tree_opts;
  ga.dbg = m->tree_dbg; ga.dbg_leaf = m->tree_dbg_leaf;
  const bool prof = m->profile && m->ev_used + 2 <= m->ev.size();
  if (prof) hipEventRecord(m->ev[m->ev_used], m->stream);
  if (m->md.lg.ga_dx == 7) hipLaunchKernelGGL((k_tree_ga<3, 7>), dim3(m->rows_grid + 1), dim3(WAVE * m->md.lg.ga_w), 0, m->stream, ga);
  else hipLaunchKernelGGL((k_tree_ga<3, 8>), dim3(m->rows_grid + 1), dim3(WAVE * m->md.lg.ga_w), 0, m->stream, ga);
  if (prof) { hipEventRecord(m->ev[m->ev_used + 1], m->stream); c->tree_prof_pair = (int)(m->ev_used / 2); m->ev_used += 2; c->tree_prof_pending = 1; }
  m->dom_launches++;
  c->tree_launches++;
  unsigned flags = 0;
  rc = wait_status(c, seq, &flags, cursor_out);
  if (rc) return rc;
  if (flags & ST_TIMEOUT) {
    g_err = "persistent tree kernel: a wait between workgroups timed out (are all of its workgroups resident? another process on "
            "this GPU? run with NUTS_GA_TREE=0)";
    hipStreamSynchronize(m->stream);
    return NUTS_E_HIP;
  }
  *flags_out = flags;
  *exhausted_out = !(flags & (ST_DIVERGING | ST_TURNING));
  return NUTS_OK;
}

// The doubling loop of one transition (NUTS._hamiltonian_step, nuts.py:204-225): queue the leaves of each doubling, wait for the
// status word of its last leaf.  `uniforms`: the host copy of the pre-drawn `step.rng.random()` values of THIS draw.
static int run_tree(nuts_chain* c, const double* uniforms, double step_size, int max_depth, unsigned* flags_out, bool* exhausted_out,
                    int* cursor_out = nullptr) {
  using clk = std::chrono::steady_clock;
  int rc = NUTS_OK;
  bool exhausted = true;
  // (a chain whose leaves do not go through the row-aligned launch -- a dense mass matrix set after it joined -- is nobody's partner:
  // the others would wait for deposits that never come)
  GroupTreeScope lockstep(c->dense || c->host_pot ? nullptr : c->m);
  // geometry of the first doubling: `(rng.random() < 0.5) * 2 - 1` on uniforms[0] (nuts.py:215); later ones come
  // back in the status record (the device advances the uniform cursor, the host only mirrors the result)
  unsigned flags = 0;
  Geometry gm{uniforms[0] < 0.5 ? 1 : -1, 0, 0, 0, 0.0};
  gm.eps = gm.dir > 0 ? step_size : -step_size;
  // geometry of the doubling after `g` (which had 2^d leaves) once its direction is known (nuts.py:347-362)
  auto next_geometry = [&](const Geometry& g, int d, int dir) {
    Geometry r = g;
    if (g.dir > 0) r.right += 1 << d; else r.left -= 1 << d;   // the finished subtree's far end is the new edge state
    r.dir = dir;
    r.edge = dir > 0 ? r.right : r.left;
    r.eps = dir > 0 ? step_size : -step_size;
    return r;
  };
  // (a host potential is called back leaf by leaf: nothing is queued ahead of a status it has not seen)
  const int spec = c->host_pot ? 0 : std::min(c->spec_max, c->last_depth - 1);
  auto enqueue_doubling = [&](const Geometry& g, int d) {
    const int nleaf = 1 << d;
    ensure_logs(c, (2 << d) + d + 1);   // this doubling reads uniform indices < 2^(d+1) + d + 1
    const int seq = ++c->seq;
    // will doubling d + 1 be queued (look-ahead) before the status of this one is waited for?  Then its first row pass takes
    // this doubling's last control work with it.
    c->defer_last_ctl = c->xfold && c->fold_ctl && d + 1 < max_depth && d + 1 <= spec;
    c->next_dir = c->defer_last_ctl ? (uniforms[(2 << d) + d] < 0.5 ? 1 : -1) : 0;   // (the direction of doubling d + 1 whenever it is needed at all)
    for (int j = 0; j < nleaf; ++j) enqueue_leaf(c, g, j, d, MODE_TREE, max_depth, j + 1 == nleaf ? seq : 0);
    c->defer_last_ctl = false;
    return seq;
  };
  c->pre_done = false;
  c->pend_valid = false;   // (a look-ahead doubling behind the end of the previous tree may have left its control work unclaimed: it
                           // would only have drained)
  auto flush_pending = [&](int seq_waited) {   // (not pending by construction: never wait on a status nobody will publish)
    if (!c->pend_valid || c->pend.seq != seq_waited) return;
    const CtlJob& p = c->pend;
    launch_control_lean(c->m, c->A, p.io, p.j, p.d, c->cfg.Emax, max_depth, c->st_dev, p.seq);
    c->pend_valid = false;
  };
  // Look-ahead for the short doublings, where the host round trip (status word over PCIe, then the first launch of
  // the next doubling) is comparable to the doubling itself: a doubling that runs to completion consumes a fixed
  // number of uniforms (2^d - 1 merges, the `extend` acceptance, the next direction), so the direction of doubling
  // d+1 is uniforms[2^(d+1) + d] whenever it is needed at all, and its launches can be queued before the status of
  // doubling d arrives.  If the tree stops at d they drain as no-ops behind the `aborted` flag.  Only done as far as
  // the previous draw's tree went, so a wasted look-ahead is rare.  The prediction is checked against the device.
  Geometry ahead{};
  int ahead_seq = 0;
  int seq = enqueue_doubling(gm, 0);
  int depth_done = 0;
  for (int d = 0; d < max_depth; ++d) {
    if (d + 1 < max_depth && d + 1 <= spec) {
      ahead = next_geometry(gm, d, uniforms[(2 << d) + d] < 0.5 ? 1 : -1);
      ahead_seq = enqueue_doubling(ahead, d + 1);
    } else ahead_seq = 0;
    flush_pending(seq);
    if (c->cb_err) return host_pot_error(c);
    const auto tw0 = clk::now();
    rc = wait_status(c, seq, &flags, cursor_out);
    c->t_wait += std::chrono::duration<double>(clk::now() - tw0).count();
    if (rc) return rc;
    depth_done = d + 1;
    if (flags & ST_BAD_ENERGY) break;
    if (flags & (ST_DIVERGING | ST_TURNING)) { exhausted = false; break; }
    if (d + 1 >= max_depth) break;
    const int dir = (flags & ST_DIR_POS) ? 1 : -1;
    if (ahead_seq) {
      if (ahead.dir != dir) { g_err = "internal error: look-ahead mispredicted the direction of a doubling"; return NUTS_E_HIP; }
      gm = ahead; seq = ahead_seq;
    } else {
      gm = next_geometry(gm, d, dir);
      seq = enqueue_doubling(gm, d + 1);
    }
  }
  c->last_depth = depth_done;
  c->pend_valid = false;
  *flags_out = flags;
  *exhausted_out = exhausted;
  return NUTS_OK;
}

// the single-launch path (small_kernel.h): one workgroup, one thread per parameter
static void launch_small(nuts_chain* c, const ArenaDev& A, const SmallDrawArgs& a) {
#define SMALL_LAUNCH(NT, P) hipLaunchKernelGGL((k_small_draw<NT, P>), dim3(1), dim3(NT), 0, c->m->stream, c->m->md, A, a)
  const bool hp = c->m->has_prog;
  // (a GLM node's rows are the workgroup's work: four waves at least, the tree still in LDS for n <= 64)
  if (c->n <= WAVE && c->small_one_wave && !c->m->md.has_glm) { if (hp) SMALL_LAUNCH(64, true); else SMALL_LAUNCH(64, false); }
  else if (c->n <= 256) { if (hp) SMALL_LAUNCH(256, true); else SMALL_LAUNCH(256, false); }
  else if (c->n <= 512) { if (hp) SMALL_LAUNCH(512, true); else SMALL_LAUNCH(512, false); }
  else { if (hp) SMALL_LAUNCH(1024, true); else SMALL_LAUNCH(1024, false); }
#undef SMALL_LAUNCH
}

// What the host does after a transition (nuts.py:478-489, base_hmc.py:238-282): step-size adaptation, mass-matrix update
// (a kernel), divergence bookkeeping, the statistics record.  `result_dev`: (q, grad) of the proposal on the device.
static int finish_draw_host(nuts_chain* c, const DrawOut& o, bool adapt, bool exhausted, const double* result_dev, int64_t evals,
                            double perf_start, double perf_diff, double cpu_diff, nuts_draw_stats* stats) {
  const int n = c->n;
  ArenaDev& A = c->A;
  const double accept = std::exp(o.log_accept_sum) / o.n_proposals;
  if (c->tree_mode) {
    c->last_depth = o.depth;
    c->leapfrogs += o.n_proposals;
    if (c->tree_prof_pending) { c->m->dom_units += o.n_proposals; c->m->ev_units[c->tree_prof_pair] = std::max(1, (int)o.n_proposals); c->tree_prof_pending = 0; }
  }
  c->da.update(accept, adapt);
  int rc = potential_update(c, result_dev, result_dev + c->n);
  if (rc) return rc;
  const bool diverging = o.diverging != 0;
  if (diverging) {   // keep the leaf the integrator started from and the one it diverged to (base_hmc.py:249-258)
    const int dir = o.div_t > 0 ? 1 : -1;   // a leaf's index is its parent's + sign(eps) and the start state is 0
    c->div_source.resize(n); c->div_dest.resize(n);
    HIPCHK(hipMemcpy(c->div_dest.data(), A.Q + (int64_t)(o.div_t & (A.S - 1)) * n, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c->div_source.data(), A.Q + (int64_t)((o.div_t - dir) & (A.S - 1)) * n, n * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (!c->tune) c->divergences += diverging;
  c->iter_count += 1;
  std::memset(stats, 0, sizeof(*stats));
  stats->depth = o.depth;
  stats->step_size = std::exp(c->da.log_step);
  stats->step_size_bar = std::exp(c->da.log_bar);
  stats->mean_tree_accept = accept;
  stats->tree_size = o.n_proposals;
  stats->diverging = diverging;
  stats->reached_max_treedepth = (exhausted && !c->tune) ? 1 : 0;  // nuts.py:220-221
  stats->divergences = c->divergences;
  stats->energy_error = o.energy - o.E0;
  stats->energy = o.energy;
  stats->max_energy_error = o.max_energy_change;
  stats->model_logp = o.logp;
  stats->index_in_trajectory = o.proposal;
  stats->n_uniforms_consumed = o.cursor;
  stats->warning = diverging ? 1 : 0;
  stats->divergence_energy_change = o.div_dE;
  stats->n_model_evals = evals;
  stats->perf_counter_start = perf_start;
  stats->perf_counter_diff = perf_diff;
  stats->process_time_diff = cpu_diff;
  return NUTS_OK;
}

extern "C" int nuts_chain_draw(nuts_chain* c, const double* q0, const double* normals, const double* uniforms,
                               int32_t n_uniforms, double* q_out, double* grad_out, nuts_draw_stats* stats) {
  if (!c || !q0 || !normals || !uniforms || !q_out || !stats) { g_err = "null argument"; return NUTS_E_ARG; }
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  const double perf_start = std::chrono::duration<double>(t0.time_since_epoch()).count();
  const std::clock_t c0 = std::clock();
  const int n = c->n;
  hipStream_t s = c->m->stream;
  ArenaDev& A = c->A;
  // base_hmc.py:226-228 ; nuts.py:205-208
  const bool adapt = c->tune && c->cfg.adapt_step_size;
  const double step_size = c->da.current(adapt);
  c->step_size = step_size;
  const int max_depth = (c->tune && c->iter_count < 200) ? c->cfg.early_max_treedepth : c->cfg.max_treedepth;
  const int need_uni = (1 << max_depth) + max_depth + 1;
  if (n_uniforms < need_uni) { g_err = "not enough uniforms for the worst-case tree"; return NUTS_E_ARG; }

  int rc = NUTS_OK;
  bool exhausted = true;
  int64_t evals = 1;
  if (c->small) {
    // latency regime: the whole transition in one launch of one workgroup (small_kernel.h)
    const bool cached = c->cache_ok && c->cache_epoch == c->m->data_epoch && std::memcmp(q0, c->last_q.data(), n * sizeof(double)) == 0;
    c->cache_ok = false;
    std::memcpy(c->stage_host, q0, n * sizeof(double));
    std::memcpy(c->stage_host + n, normals, n * sizeof(double));
    // (this path takes log(u) on the device, small_kernel.h: only the uniforms travel)
    std::memcpy(c->stage_host + 2 * n, uniforms, need_uni * sizeof(double));
    HIPCHK(hipMemcpyAsync(c->stage_dev, c->stage_host, (2 * (size_t)n + need_uni) * sizeof(double), hipMemcpyHostToDevice, s));
    if (!cached) HIPCHK(hipMemcpyAsync(A.Q, c->stage_dev, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    SmallDrawArgs a{};
    a.normals = c->stage_dev + n;
    a.q_src = cached ? c->out_dev2 : nullptr; a.g_src = cached ? c->out_dev2 + n : nullptr; a.cached_logp = c->last_logp;
    a.step_size = step_size; a.Emax = c->cfg.Emax; a.max_depth = max_depth;
    a.n_draws = 1; a.n_uniforms = need_uni; a.worst_uniforms = need_uni;
    a.q_out = c->out_dev; a.g_out = c->out_dev + n; a.trace_q = nullptr; a.out = c->do_dev; a.n_done = nullptr; a.st = nullptr; a.seq = 0; a.lds_slots = c->small_lds_slots;
    launch_small(c, A, a);
    // the next draw's start-state cache must not alias this draw's output buffer
    std::swap(c->out_dev, c->out_dev2);
    HIPCHK(hipMemcpyAsync(c->out_host, c->out_dev2, 2 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->do_host, c->do_dev, sizeof(DrawOut), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    if (c->do_host->bad_energy) {
      rc = check_mass_matrix(c);
      if (rc == NUTS_OK) g_err = "Bad initial energy, check any log probabilities that are inf or -inf, nan or very small";
      return NUTS_E_BAD_ENERGY;
    }
    exhausted = !(c->do_host->diverging || c->do_host->turning);
  } else {
  rc = draw_begin(c, q0, normals, uniforms, need_uni, step_size, max_depth, false, 0, true);
  if (rc) return rc;
  const auto tb = clk::now();
  c->t_begin += std::chrono::duration<double>(tb - t0).count();
  unsigned flags = 0;
  rc = c->tree_mode ? run_tree_ga(c, uniforms, step_size, max_depth, &flags, &exhausted)
                    : run_tree(c, uniforms, step_size, max_depth, &flags, &exhausted);
  if (rc) return rc;
  const auto tl = clk::now();
  c->t_loop += std::chrono::duration<double>(tl - tb).count();
  if (flags & ST_BAD_ENERGY) {
    // base_hmc.py:205-224: SamplingError("Bad initial energy"), after potential.raise_ok
    rc = check_mass_matrix(c);
    if (rc == NUTS_OK) g_err = "Bad initial energy, check any log probabilities that are inf or -inf, nan or very small";
    return NUTS_E_BAD_ENERGY;
  }
  const int fgrid = std::max(1, std::min(256, (n + VEC_THREADS - 1) / VEC_THREADS));
  hipLaunchKernelGGL(k_draw_finish, dim3(fgrid), dim3(VEC_THREADS), 0, s, A, c->out_dev, c->out_dev + n, c->do_dev, (double*)nullptr,
                     (DrawOutMapped*)nullptr, 0u);
  HIPCHK(hipMemcpyAsync(c->out_host, c->out_dev, 2 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->do_host, c->do_dev, sizeof(DrawOut), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  c->t_finish += std::chrono::duration<double>(clk::now() - tl).count();
  }
  const double* const result_dev = c->small ? c->out_dev2 : c->out_dev;   // (q, grad) of the proposal on the device
  const DrawOut& o = *c->do_host;
  evals += o.n_proposals;
  const auto t1 = clk::now();
  const std::clock_t c1 = std::clock();

  rc = finish_draw_host(c, o, adapt, exhausted, result_dev, evals, perf_start, std::chrono::duration<double>(t1 - t0).count(),
                        (double)(c1 - c0) / CLOCKS_PER_SEC, stats);
  if (rc) return rc;
  std::memcpy(q_out, c->out_host, n * sizeof(double));
  if (grad_out) std::memcpy(grad_out, c->out_host + n, n * sizeof(double));
  c->last_q.assign(c->out_host, c->out_host + n); c->last_logp = o.logp; c->cache_ok = true; c->cache_epoch = c->m->data_epoch;
  return NUTS_OK;
}

// K consecutive transitions of a model on the GENERAL path (one or more launches per leapfrog) inside one call: the host
// stays in the doubling loop (it decides what to queue), but nothing else crosses PCIe per draw -- momentum normals and
// uniforms of the whole batch are uploaded once, a draw starts from the previous proposal where it lies on the device,
// positions go to a device trace buffer, and the per-draw record comes back through pinned mapped memory (no stream
// synchronisation).  Tuning draws are allowed (dual averaging is host arithmetic on that record, the mass-matrix update a
// kernel).  Same semantics as the single-launch variant below: stops early after a divergent draw or when the uniforms could
// not cover another worst-case tree; `stats[i].n_uniforms_consumed` is cumulative.
static int draw_many_general(nuts_chain* c, const double* q0, const double* normals, const double* uniforms, int32_t n_uniforms,
                             int32_t K, double* q_out, nuts_draw_stats* stats, int32_t* n_done) {
  using clk = std::chrono::steady_clock;
  const int n = c->n;
  hipStream_t s = c->m->stream;
  ArenaDev& A = c->A;
  const size_t U = (size_t)n_uniforms;
  const size_t in_doubles = (size_t)n + (size_t)K * n + 2 * U;
  const size_t out_bytes = (size_t)K * n * sizeof(double);
  if (in_doubles > c->many_in_cap || out_bytes > c->many_out_cap) {
    HIPCHK(hipStreamSynchronize(s));
    if (c->many_in_host) hipHostFree(c->many_in_host);
    if (c->many_out_host) hipHostFree(c->many_out_host);
    if (c->many_in_dev) hipFree(c->many_in_dev);
    if (c->many_out_dev) hipFree(c->many_out_dev);
    c->many_in_host = nullptr; c->many_out_host = nullptr; c->many_in_dev = nullptr; c->many_out_dev = nullptr;
    c->many_in_cap = c->many_out_cap = 0;
    HIPCHK(hipHostMalloc((void**)&c->many_in_host, in_doubles * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->many_out_host, out_bytes, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&c->many_in_dev, in_doubles * sizeof(double)));
    HIPCHK(hipMalloc((void**)&c->many_out_dev, out_bytes));
    c->many_in_cap = in_doubles; c->many_out_cap = out_bytes;
  }
  const auto tm0 = clk::now();
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  const bool cached0 = c->cache_ok && c->cache_epoch == c->m->data_epoch && std::memcmp(q0, c->last_q.data(), n * sizeof(double)) == 0;
  c->cache_ok = false;
  double* const h_q = c->many_in_host;
  double* const h_norm = h_q + n;
  double* const h_u = h_norm + (size_t)K * n;
  double* const h_lu = h_u + U;
  std::memcpy(h_q, q0, n * sizeof(double));
  std::memcpy(h_norm, normals, (size_t)K * n * sizeof(double));
  std::memcpy(h_u, uniforms, U * sizeof(double));
  for (size_t i = 0; i < U; ++i) h_lu[i] = std::log(h_u[i]);   // `np.log(rng.random())` is what the tree compares (nuts.py:371,466)
  HIPCHK(hipMemcpyAsync(c->many_in_dev, c->many_in_host, in_doubles * sizeof(double), hipMemcpyHostToDevice, s));
  const double* const d_norm = c->many_in_dev + n;
  const double* const d_u = d_norm + (size_t)K * n;
  const double* const d_lu = d_u + U;
  double* const trace_dev = reinterpret_cast<double*>(c->many_out_dev);
  if (!cached0) {
    HIPCHK(hipMemcpyAsync(A.Q, c->many_in_dev, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    model_enqueue_plain(c->m, A.Q, A.G, A.LOGP);
  }
  const double* const save_u = A.uniforms; const double* const save_lu = A.log_uniforms;
  const int save_done = c->logs_done, save_total = c->logs_total;
  c->logs_done = c->logs_total = 1 << 30;   // every logarithm of the batch is already on the device (ensure_logs has nothing to do)
  int rc = NUTS_OK, done = 0;
  size_t consumed = 0;
  const int fgrid = std::max(1, std::min(256, (n + VEC_THREADS - 1) / VEC_THREADS));
  c->tm_pre += secs(tm0, clk::now()); c->tm_batches++;
  // After tuning nothing the host computes between two draws feeds the next one (step size and mass matrix are fixed): the status word of
  // the tree's last doubling carries the uniform cursor, so the NEXT draw's start kernels are queued right behind this draw's finish
  // kernel, and this draw's record -- statistics only -- is read while the next tree is already running.  One host round trip per draw
  // (the status) instead of three (status, record, then an idle queue to restart): ~50 us of GPU idle per draw on C2-S / C3
  // (profiles/r03g_profile_c3.txt: 50.7 us idle before k_draw_start).  A divergent draw is finished synchronously, as before.
  const bool pipe = !c->tune && !c->tree_mode && !c->full_adapt && c->pipe_draws;   // (the option is latched when the chain is created)
  struct Pending {
    bool valid = false; int k = 0; unsigned seq = 0; bool exhausted = false, adapt = false; size_t consumed_after = 0; int cursor = 0;
    double perf_start = 0.0, wall = 0.0, cpu = 0.0;
  } pend;
  auto enqueue_start = [&](int k, double step_size, int max_depth, bool from_prev) {   // momentum, start state, control block of draw k
    if (c->dense) {
      if (c->full_adapt) fa_random(c, d_norm + (size_t)k * n, A.P);
      else hipLaunchKernelGGL(k_dense_mv, dim3(c->mv_grid), dim3(256), 0, s, c->dense_W, d_norm + (size_t)k * n, A.P, n, (const double*)nullptr,
                         (double*)nullptr, 0.0, (const int*)nullptr);
      hipLaunchKernelGGL(k_dense_mv, dim3(c->mv_grid), dim3(256), 0, s, c->dense_C, A.P, A.V, n, (const double*)nullptr, (double*)nullptr, 0.0,
                         (const int*)nullptr);
    }
    hipLaunchKernelGGL(k_draw_start, dim3(A.nblk), dim3(VEC_THREADS), 0, s, A, d_norm + (size_t)k * n, (const double*)nullptr, c->kin_part,
                       from_prev ? (const double*)c->out_dev : (const double*)nullptr,
                       from_prev ? (const double*)(c->out_dev + n) : (const double*)nullptr, c->dense);
    hipLaunchKernelGGL(k_draw_ctl_start, dim3(1), dim3(64), 0, s, A, c->kin_part, step_size, 0, max_depth, c->st_dev, from_prev ? 1 : 0,
                       c->last_logp, k > 0 ? (const DrawOut*)c->do_dev : (const DrawOut*)nullptr);
  };
  // host side of a draw whose finish kernel has been queued: wait for its record, adaptation + statistics
  auto complete = [&](const Pending& p, bool* diverged) -> int {
    const auto tb = clk::now();
    DrawOutMapped* slot = c->dom_host + (p.seq % DOM_RING);
    volatile unsigned long long* w = &slot->seq;
    for (unsigned spins = 0; (unsigned)*w != p.seq; ++spins) {
      if ((spins & 0xfffff) == 0xfffff) {
        if (hipStreamQuery(s) == hipSuccess && (unsigned)*w != p.seq) { g_err = "k_draw_finish ended without publishing its record"; return NUTS_E_HIP; }
        if (clk::now() - tb > std::chrono::seconds(60)) { g_err = "timed out waiting for the device"; return NUTS_E_HIP; }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const DrawOut o = slot->o;
    const auto t1 = clk::now();
    c->tm_record += secs(tb, t1);
    if (o.cursor != p.cursor) { g_err = "internal error: the status word and the draw record disagree on the uniforms consumed"; return NUTS_E_HIP; }
    int r = finish_draw_host(c, o, p.adapt, p.exhausted, c->out_dev, o.n_proposals + ((p.k == 0 && !cached0) ? 1 : 0), p.perf_start, p.wall, p.cpu,
                             stats + p.k);
    c->tm_host += secs(t1, clk::now()); c->tm_draws++;
    if (r) return r;
    stats[p.k].n_uniforms_consumed = (int32_t)p.consumed_after;
    c->last_logp = o.logp;
    done = p.k + 1;
    *diverged = o.diverging != 0;
    return NUTS_OK;
  };
  bool started = false;   // the start kernels of draw k are already in the queue (put there behind draw k - 1's finish kernel)
  for (int k = 0; k < K; ++k) {
    const auto t0 = clk::now();
    const double perf_start = std::chrono::duration<double>(t0.time_since_epoch()).count();
    const std::clock_t c0 = std::clock();
    const bool adapt = c->tune && c->cfg.adapt_step_size;
    const double step_size = c->da.current(adapt);
    c->step_size = step_size;
    const int max_depth = (c->tune && c->iter_count < 200) ? c->cfg.early_max_treedepth : c->cfg.max_treedepth;
    const size_t need_uni = ((size_t)1 << max_depth) + max_depth + 1;
    if (!started) {
      if (U - consumed < need_uni) {
        if (k == 0) { g_err = "not enough uniforms for the worst-case tree"; rc = NUTS_E_ARG; }
        break;
      }
      A.uniforms = d_u + consumed; A.log_uniforms = d_lu + consumed;
      enqueue_start(k, step_size, max_depth, k > 0 || cached0);   // (from the previous proposal: (q, grad) in out_dev, logp in do_dev / last_logp)
    }
    started = false;
    unsigned flags = 0;
    bool exhausted = true;
    int cursor = 0;
    const auto ta = clk::now();
    c->tm_start += secs(t0, ta);
    rc = c->tree_mode ? run_tree_ga(c, h_u + consumed, step_size, max_depth, &flags, &exhausted, &cursor)
                      : run_tree(c, h_u + consumed, step_size, max_depth, &flags, &exhausted, &cursor);
    const auto tb = clk::now();
    c->tm_tree += secs(ta, tb);
    if (rc) break;
    if (flags & ST_BAD_ENERGY) {
      if (pend.valid) { bool dv = false; rc = complete(pend, &dv); pend.valid = false; if (rc) break; }
      rc = check_mass_matrix(c);
      if (rc == NUTS_OK) g_err = "Bad initial energy, check any log probabilities that are inf or -inf, nan or very small";
      rc = NUTS_E_BAD_ENERGY;
      break;
    }
    const unsigned seq = ++c->dom_seq;
    hipLaunchKernelGGL(k_draw_finish, dim3(fgrid), dim3(VEC_THREADS), 0, s, A, c->out_dev, c->out_dev + n, c->do_dev, trace_dev + (size_t)k * n,
                       c->dom_dev + (seq % DOM_RING), seq);
    const size_t consumed_after = consumed + (size_t)cursor;
    if (pipe && !(flags & ST_DIVERGING) && k + 1 < K && U - consumed_after >= need_uni) {
      A.uniforms = d_u + consumed_after; A.log_uniforms = d_lu + consumed_after;
      enqueue_start(k + 1, step_size, max_depth, true);
      started = true;
    }
    Pending cur;
    cur.valid = true; cur.k = k; cur.seq = seq; cur.exhausted = exhausted; cur.adapt = adapt; cur.consumed_after = consumed_after; cur.cursor = cursor;
    cur.perf_start = perf_start;
    cur.wall = std::chrono::duration<double>(clk::now() - t0).count();
    cur.cpu = (double)(std::clock() - c0) / CLOCKS_PER_SEC;
    bool diverged = false;
    if (pend.valid) { rc = complete(pend, &diverged); pend.valid = false; if (rc) break; }   // (draw k - 1: never divergent, see above)
    if (started) pend = cur;                                  // its record is read while draw k + 1 runs
    else { rc = complete(cur, &diverged); if (rc) break; }
    consumed = consumed_after;
    if (diverged) break;   // its two phase-space points were read from the arena; the caller sees the warning before going on
  }
  if (pend.valid) {   // (left over by an error in the draw behind it)
    bool dv = false;
    const int r2 = complete(pend, &dv);
    pend.valid = false;
    if (!rc) rc = r2;
  }
  A.uniforms = save_u; A.log_uniforms = save_lu;
  c->logs_done = save_done; c->logs_total = save_total;
  *n_done = done;
  const auto tm1 = clk::now();
  struct PostTimer { nuts_chain* c; clk::time_point t; ~PostTimer() { c->tm_post += std::chrono::duration<double>(clk::now() - t).count(); } } post_timer{c, tm1};
  if (done > 0) {
    HIPCHK(hipMemcpyAsync(c->many_out_host, c->many_out_dev, (size_t)done * n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    const double* trace_host = reinterpret_cast<const double*>(c->many_out_host);
    std::memcpy(q_out, trace_host, (size_t)done * n * sizeof(double));
    c->last_q.assign(trace_host + (size_t)(done - 1) * n, trace_host + (size_t)done * n);
    c->cache_ok = true; c->cache_epoch = c->m->data_epoch;
  } else {
    hipStreamSynchronize(s);
  }
  if (rc && done > 0 && rc != NUTS_E_BAD_ENERGY) return rc;
  return done > 0 ? NUTS_OK : (rc ? rc : NUTS_E_ARG);
}

// K consecutive post-tuning NUTS transitions in ONE launch (single-workgroup models; small_kernel.h).
// `normals` is [K][n] (K calls of `potential.rng.normal(size=n)` draw exactly the values of one call of size K n),
// `uniforms` the next `n_uniforms` values of `step.rng.random()`.  On return `*n_done` draws were made (fewer than
// K after a divergent draw or when the uniforms could not cover another worst-case tree), `q_out` holds their
// positions [n_done][n], `stats[i].n_uniforms_consumed` the TOTAL consumed up to and including draw i.
extern "C" int nuts_chain_draw_many(nuts_chain* c, const double* q0, const double* normals, const double* uniforms,
                                    int32_t n_uniforms, int32_t K, double* q_out, nuts_draw_stats* stats, int32_t* n_done) {
  if (!c || !q0 || !normals || !uniforms || !q_out || !stats || !n_done || K <= 0) { g_err = "null argument"; return NUTS_E_ARG; }
  if (c->host_pot) { g_err = "nuts_chain_draw_many: a host potential is consulted between draws (random, update): one nuts_chain_draw per transition"; return NUTS_E_ARG; }
  if (!c->small) return draw_many_general(c, q0, normals, uniforms, n_uniforms, K, q_out, stats, n_done);
  if (c->tune) { g_err = "nuts_chain_draw_many: the chain is still tuning (the single-launch batch has no adaptation between its draws)"; return NUTS_E_ARG; }
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  const double perf_start = std::chrono::duration<double>(t0.time_since_epoch()).count();
  const std::clock_t c0 = std::clock();
  const int n = c->n;
  hipStream_t s = c->m->stream;
  ArenaDev& A = c->A;
  const double step_size = c->da.current(false);   // exp(log_step_size_bar) (step_sizes.py:60-64)
  c->step_size = step_size;
  const int max_depth = c->cfg.max_treedepth;
  const int need_uni = (1 << max_depth) + max_depth + 1;
  if (n_uniforms < need_uni) { g_err = "not enough uniforms for the worst-case tree"; return NUTS_E_ARG; }
  // staging: q0 | normals [K][n] | uniforms   ->  trace [K][n] | DrawOut [K] | n_done
  const size_t in_doubles = (size_t)n + (size_t)K * n + (size_t)n_uniforms;
  const size_t out_bytes = (size_t)K * n * sizeof(double) + (size_t)K * sizeof(DrawOut) + 16;
  if (in_doubles > c->many_in_cap || out_bytes > c->many_out_cap) {
    HIPCHK(hipStreamSynchronize(s));
    if (c->many_in_host) hipHostFree(c->many_in_host);
    if (c->many_out_host) hipHostFree(c->many_out_host);
    if (c->many_in_dev) hipFree(c->many_in_dev);
    if (c->many_out_dev) hipFree(c->many_out_dev);
    c->many_in_host = nullptr; c->many_out_host = nullptr; c->many_in_dev = nullptr; c->many_out_dev = nullptr;
    c->many_in_cap = c->many_out_cap = 0;
    HIPCHK(hipHostMalloc((void**)&c->many_in_host, in_doubles * sizeof(double), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->many_out_host, out_bytes, hipHostMallocDefault));
    HIPCHK(hipMalloc((void**)&c->many_in_dev, in_doubles * sizeof(double)));
    HIPCHK(hipMalloc((void**)&c->many_out_dev, out_bytes));
    c->many_in_cap = in_doubles; c->many_out_cap = out_bytes;
  }
  const bool cached = c->cache_ok && c->cache_epoch == c->m->data_epoch && std::memcmp(q0, c->last_q.data(), n * sizeof(double)) == 0;
  c->cache_ok = false;
  std::memcpy(c->many_in_host, q0, n * sizeof(double));
  std::memcpy(c->many_in_host + n, normals, (size_t)K * n * sizeof(double));
  std::memcpy(c->many_in_host + n + (size_t)K * n, uniforms, (size_t)n_uniforms * sizeof(double));
  HIPCHK(hipMemcpyAsync(c->many_in_dev, c->many_in_host, in_doubles * sizeof(double), hipMemcpyHostToDevice, s));
  if (!cached) HIPCHK(hipMemcpyAsync(A.Q, c->many_in_dev, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  double* trace_dev = reinterpret_cast<double*>(c->many_out_dev);
  DrawOut* outs_dev = reinterpret_cast<DrawOut*>(c->many_out_dev + (size_t)K * n * sizeof(double));
  int* ndone_dev = reinterpret_cast<int*>(c->many_out_dev + (size_t)K * n * sizeof(double) + (size_t)K * sizeof(DrawOut));
  ArenaDev Am = A;
  Am.uniforms = c->many_in_dev + n + (size_t)K * n;
  SmallDrawArgs a{};
  a.normals = c->many_in_dev + n;
  a.q_src = cached ? c->out_dev2 : nullptr; a.g_src = cached ? c->out_dev2 + n : nullptr; a.cached_logp = c->last_logp;
  a.step_size = step_size; a.Emax = c->cfg.Emax; a.max_depth = max_depth;
  a.n_draws = K; a.n_uniforms = n_uniforms; a.worst_uniforms = need_uni;
  a.q_out = c->out_dev; a.g_out = c->out_dev + n; a.trace_q = trace_dev; a.out = outs_dev; a.n_done = ndone_dev; a.st = nullptr; a.seq = 0; a.lds_slots = c->small_lds_slots;
  launch_small(c, Am, a);
  std::swap(c->out_dev, c->out_dev2);   // (q, grad) of the last proposal: the next call's start-state cache
  HIPCHK(hipMemcpyAsync(c->many_out_host, c->many_out_dev, out_bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  const double* trace_host = reinterpret_cast<const double*>(c->many_out_host);
  const DrawOut* outs = reinterpret_cast<const DrawOut*>(c->many_out_host + (size_t)K * n * sizeof(double));
  const int done = *reinterpret_cast<const int*>(c->many_out_host + (size_t)K * n * sizeof(double) + (size_t)K * sizeof(DrawOut));
  *n_done = done;
  if (done <= 0 || done > K) { g_err = "nuts_chain_draw_many: the device reported an impossible draw count"; return NUTS_E_HIP; }
  if (outs[done - 1].bad_energy) {
    int rc = check_mass_matrix(c);
    if (rc == NUTS_OK) g_err = "Bad initial energy, check any log probabilities that are inf or -inf, nan or very small";
    return NUTS_E_BAD_ENERGY;
  }
  const auto t1 = clk::now();
  const std::clock_t c1 = std::clock();
  const double wall = std::chrono::duration<double>(t1 - t0).count() / done;
  const double cpu = (double)(c1 - c0) / CLOCKS_PER_SEC / done;
  for (int i = 0; i < done; ++i) {
    const DrawOut& o = outs[i];
    const bool diverging = o.diverging != 0;
    c->divergences += diverging;          // (not tuning: base_hmc.py:270-273)
    c->iter_count += 1;
    nuts_draw_stats* st = stats + i;
    std::memset(st, 0, sizeof(*st));
    st->depth = o.depth;
    st->step_size = std::exp(c->da.log_step);
    st->step_size_bar = std::exp(c->da.log_bar);
    st->mean_tree_accept = std::exp(o.log_accept_sum) / o.n_proposals;
    st->tree_size = o.n_proposals;
    st->diverging = diverging;
    st->reached_max_treedepth = !(o.diverging || o.turning) ? 1 : 0;   // nuts.py:220-221 (tune is false here)
    st->divergences = c->divergences;
    st->energy_error = o.energy - o.E0;
    st->energy = o.energy;
    st->max_energy_error = o.max_energy_change;
    st->model_logp = o.logp;
    st->index_in_trajectory = o.proposal;
    st->n_uniforms_consumed = o.cursor;
    st->warning = diverging ? 1 : 0;
    st->divergence_energy_change = o.div_dE;
    st->n_model_evals = o.n_proposals + (i == 0 && !cached ? 1 : 0);
    st->perf_counter_start = perf_start + i * wall;
    st->perf_counter_diff = wall;
    st->process_time_diff = cpu;
  }
  const DrawOut& last = outs[done - 1];
  if (last.diverging) {   // only the last draw of a batch can be divergent: its two points are still in the arena
    const int dir = last.div_t > 0 ? 1 : -1;
    c->div_source.resize(n); c->div_dest.resize(n);
    HIPCHK(hipMemcpy(c->div_dest.data(), A.Q + (int64_t)(last.div_t & (A.S - 1)) * n, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(c->div_source.data(), A.Q + (int64_t)((last.div_t - dir) & (A.S - 1)) * n, n * sizeof(double), hipMemcpyDeviceToHost));
  }
  std::memcpy(q_out, trace_host, (size_t)done * n * sizeof(double));
  c->last_q.assign(trace_host + (size_t)(done - 1) * n, trace_host + (size_t)done * n);
  c->last_logp = last.logp; c->cache_ok = true; c->cache_epoch = c->m->data_epoch;
  return NUTS_OK;
}

// HamiltonianMC._hamiltonian_step (pymc/step_methods/hmc/hmc.py:130-184)
extern "C" int nuts_chain_draw_hmc(nuts_chain* c, const double* q0, const double* normals, const double* uniforms,
                                   double path_length, int32_t max_steps, double* q_out, double* grad_out,
                                   nuts_hmc_stats* stats) {
  if (!c || !q0 || !normals || !uniforms || !q_out || !stats) { g_err = "null argument"; return NUTS_E_ARG; }
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  const std::clock_t c0 = std::clock();
  const int n = c->n;
  hipStream_t s = c->m->stream;
  ArenaDev& A = c->A;
  const bool adapt = c->tune && c->cfg.adapt_step_size;
  double step_size = c->da.current(adapt);
  c->step_size = step_size;
  step_size = (0.85 + (1.15 - 0.85) * uniforms[0]) * step_size;  // `unif` step_rand, hmc.py:35-36
  int n_steps = std::max(1, (int)(path_length / step_size));
  n_steps = std::min<int>(max_steps, n_steps);
  // a fixed-length trajectory only ever needs its previous state: the arena is used as a ring (slot = index mod S), and what
  // the end of the transition needs from the START state is kept aside before the ring can wrap over slot 0
  int rc = draw_begin(c, q0, normals, uniforms, 0, step_size, 1, false, +1);
  if (rc) return rc;
  double* const start_keep = c->out_dev2;   // [0, n): gradient at the start state ; [n]: E, [n + 1]: logp of the start state
  HIPCHK(hipMemcpyAsync(start_keep, A.G, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(start_keep + n, A.E, sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(start_keep + n + 1, A.LOGP, sizeof(double), hipMemcpyDeviceToDevice, s));
  const Geometry gm{+1, 0, 0, 0, step_size};
  for (int j = 0; j < n_steps; ++j) enqueue_leaf(c, gm, j, 0, MODE_SIMPLE, 1, 0, n_steps);
  if (c->cb_err) return host_pot_error(c);
  std::vector<double> Eh(2), lph(2);
  const int last = n_steps & (A.S - 1);
  HIPCHK(hipMemcpyAsync(c->out_host, A.Q + (int64_t)last * n, n * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->out_host + n, A.G + (int64_t)last * n, n * sizeof(double), hipMemcpyDeviceToHost, s));
  rc = sync_status(c);
  if (rc) return rc;
  HIPCHK(hipMemcpy(&Eh[0], start_keep + n, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&Eh[1], A.E + last, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&lph[0], start_keep + n + 1, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&lph[1], A.LOGP + last, sizeof(double), hipMemcpyDeviceToHost));
  if (c->st_host->bad_energy) {
    rc = check_mass_matrix(c);
    if (rc == NUTS_OK) g_err = "Bad initial energy, check any log probabilities that are inf or -inf, nan or very small";
    return NUTS_E_BAD_ENERGY;
  }
  bool div = false;
  if (!std::isfinite(Eh[1])) div = true;           // hmc.py:147-148
  double dE = Eh[1] - Eh[0];
  if (std::isnan(dE)) dE = INFINITY;
  if (std::fabs(dE) > c->cfg.Emax) div = true;     // hmc.py:152-158
  const double accept = std::min(1.0, std::exp(-dE));
  // hmc.py:162: `div_info is not None or rng.random() >= accept` -- the accept draw is NOT consumed on a divergence
  const bool accepted = !(div || uniforms[1] >= accept);
  const auto t1 = clk::now();
  const std::clock_t c1 = std::clock();
  c->da.update(accept, adapt);
  // (a rejected transition stays at q0, which is still in the staging buffer of this draw)
  const double* xsel = accepted ? (A.Q + (int64_t)last * n) : c->stage_dev;
  // (gradient at q0 on a rejection: the copy kept aside above -- slot 0 of the ring is overwritten once n_steps >= S)
  rc = potential_update(c, xsel, accepted ? (A.G + (int64_t)last * n) : start_keep);
  if (rc) return rc;
  if (!c->tune) c->divergences += div;
  c->iter_count += 1;
  if (accepted) {
    std::memcpy(q_out, c->out_host, n * sizeof(double));
    if (grad_out) std::memcpy(grad_out, c->out_host + n, n * sizeof(double));
  } else {
    std::memcpy(q_out, q0, n * sizeof(double));
    if (grad_out) HIPCHK(hipMemcpy(grad_out, start_keep, n * sizeof(double), hipMemcpyDeviceToHost));
  }
  HIPCHK(hipStreamSynchronize(s));
  std::memset(stats, 0, sizeof(*stats));
  stats->step_size = std::exp(c->da.log_step); stats->step_size_bar = std::exp(c->da.log_bar);
  stats->accept = accept; stats->energy_error = dE; stats->energy = Eh[1]; stats->model_logp = lph[1];
  stats->path_length = path_length; stats->n_steps = n_steps; stats->divergences = c->divergences;
  stats->diverging = div; stats->accepted = accepted;
  stats->perf_counter_start = std::chrono::duration<double>(t0.time_since_epoch()).count();
  stats->perf_counter_diff = std::chrono::duration<double>(t1 - t0).count();
  stats->process_time_diff = (double)(c1 - c0) / CLOCKS_PER_SEC;
  return NUTS_OK;
}

// integrator property tests (tests/step_methods/hmc/test_hmc.py:49-74): n_steps leapfrogs from (q, p).
// Like the reference integrator this never looks at the energy of the start state.
extern "C" int nuts_chain_leapfrog_test(nuts_chain* c, const double* q, const double* p, double eps, int32_t n_steps,
                                        double* q_out, double* p_out, double* energy_out) {
  if (!c || !q || !p || n_steps < 0) { g_err = "bad argument"; return NUTS_E_ARG; }
  const int n = c->n;
  ArenaDev& A = c->A;
  hipStream_t s = c->m->stream;
  if (n_steps >= A.S) { g_err = "n_steps exceeds the trajectory arena"; return NUTS_E_ARG; }
  const int dir = eps >= 0 ? 1 : -1;
  int rc = draw_begin(c, q, p, nullptr, 0, std::fabs(eps), 1, true, dir);
  if (rc) return rc;
  const Geometry gm{dir, 0, 0, 0, eps};
  for (int j = 0; j < n_steps; ++j) enqueue_leaf(c, gm, j, 0, MODE_SIMPLE, 1, 0, n_steps);
  if (c->cb_err) return host_pot_error(c);
  const int last = (dir * n_steps) & (A.S - 1);
  HIPCHK(hipStreamSynchronize(s));
  if (q_out) HIPCHK(hipMemcpy(q_out, A.Q + (int64_t)last * n, n * sizeof(double), hipMemcpyDeviceToHost));
  if (p_out) HIPCHK(hipMemcpy(p_out, A.P + (int64_t)last * n, n * sizeof(double), hipMemcpyDeviceToHost));
  if (energy_out) HIPCHK(hipMemcpy(energy_out, A.E + last, sizeof(double), hipMemcpyDeviceToHost));
  return NUTS_OK;
}

extern "C" int nuts_chain_set_host_potential(nuts_chain* c, nuts_velocity_fn velocity, nuts_energy_fn energy,
                                             nuts_velocity_energy_fn velocity_energy, void* user) {
  if (!c || !velocity || !energy || !velocity_energy) { g_err = "nuts_chain_set_host_potential: null argument"; return NUTS_E_ARG; }
  if (!c->host_pot) { g_err = "nuts_chain_set_host_potential: the chain was not created with NUTS_POT_HOST"; return NUTS_E_ARG; }
  c->hp_velocity = velocity; c->hp_energy = energy; c->hp_velocity_energy = velocity_energy; c->hp_user = user;
  return NUTS_OK;
}

// ---- sampling_state ---------------------------------------------------------
static const int64_t STATE_MAGIC = 0x4e5554534d493335LL;
extern "C" int64_t nuts_chain_state_size(const nuts_chain* c) {
  if (!c) return 0;
  // (NUTS_POT_FULL_ADAPT: covariance in use, its factor, both estimators)
  const size_t extra = c->full_adapt ? (4 * (size_t)c->n * c->n + 2 * (size_t)c->n) : 0;
  return (int64_t)(sizeof(StateHeader) + (7 * (size_t)c->n + extra) * sizeof(double));
}
extern "C" int nuts_chain_get_state(nuts_chain* c, void* blob) {
  if (!c || !blob) return NUTS_E_ARG;
  HIPCHK(hipStreamSynchronize(c->m->stream));
  StateHeader h{};
  h.magic = STATE_MAGIC; h.n = c->n; h.da = c->da; h.iter_count = c->iter_count; h.divergences = c->divergences;
  h.n_samples = c->n_samples; h.adaptation_window = c->adaptation_window; h.tune = c->tune; h.fg_is_a = c->fg_is_a;
  h.fg_count = c->fg_count; h.bg_count = c->bg_count;
  h.fa_previous_update = c->fa_previous_update; h.fa_fg_n = c->fa_fg_n; h.fa_bg_n = c->fa_bg_n;
  std::memcpy(blob, &h, sizeof(h));
  double* v = reinterpret_cast<double*>(static_cast<char*>(blob) + sizeof(h));
  const double* src[7] = {c->var, c->stds, c->inv_stds, c->wa_mean, c->wa_m2, c->wb_mean, c->wb_m2};
  for (int k = 0; k < 7; ++k) HIPCHK(hipMemcpy(v + (size_t)k * c->n, src[k], c->n * sizeof(double), hipMemcpyDeviceToHost));
  if (c->full_adapt) {
    const size_t n = c->n, nn = n * n;
    double* w = v + 7 * n;
    const double* mats[4] = {c->dense_C, c->fa_L, c->fa_fg_raw, c->fa_bg_raw};
    for (int k = 0; k < 4; ++k) HIPCHK(hipMemcpy(w + (size_t)k * nn, mats[k], nn * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(w + 4 * nn, c->fa_fg_mean, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(w + 4 * nn + n, c->fa_bg_mean, n * sizeof(double), hipMemcpyDeviceToHost));
  }
  return NUTS_OK;
}
extern "C" int nuts_chain_set_state(nuts_chain* c, const void* blob) {
  if (!c || !blob) return NUTS_E_ARG;
  StateHeader h;
  std::memcpy(&h, blob, sizeof(h));
  if (h.magic != STATE_MAGIC || h.n != c->n) { g_err = "sampling state does not belong to this chain (frozen fields differ)"; return NUTS_E_ARG; }
  HIPCHK(hipStreamSynchronize(c->m->stream));
  c->cache_ok = false;
  c->da = h.da; c->iter_count = h.iter_count; c->divergences = h.divergences; c->n_samples = h.n_samples;
  c->adaptation_window = h.adaptation_window; c->tune = h.tune != 0; c->fg_is_a = h.fg_is_a != 0;
  c->fg_count = h.fg_count; c->bg_count = h.bg_count;
  c->fa_previous_update = h.fa_previous_update; c->fa_fg_n = h.fa_fg_n; c->fa_bg_n = h.fa_bg_n;
  const double* v = reinterpret_cast<const double*>(static_cast<const char*>(blob) + sizeof(h));
  double* dst[7] = {c->var, c->stds, c->inv_stds, c->wa_mean, c->wa_m2, c->wb_mean, c->wb_m2};
  for (int k = 0; k < 7; ++k) HIPCHK(hipMemcpy(dst[k], v + (size_t)k * c->n, c->n * sizeof(double), hipMemcpyHostToDevice));
  if (c->full_adapt) {
    const size_t n = c->n, nn = n * n;
    const double* w = v + 7 * n;
    double* mats[4] = {c->dense_C, c->fa_L, c->fa_fg_raw, c->fa_bg_raw};
    for (int k = 0; k < 4; ++k) HIPCHK(hipMemcpy(mats[k], w + (size_t)k * nn, nn * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->fa_fg_mean, w + 4 * nn, n * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->fa_bg_mean, w + 4 * nn + n, n * sizeof(double), hipMemcpyHostToDevice));
  }
  return NUTS_OK;
}

extern "C" int nuts_chain_get_scalar(nuts_chain* c, const char* name, double* out) {
  if (!c || !name || !out) return NUTS_E_ARG;
  const std::string k(name);
  if (k == "log_step") *out = c->da.log_step;
  else if (k == "log_bar") *out = c->da.log_bar;
  else if (k == "hbar") *out = c->da.hbar;
  else if (k == "count") *out = (double)c->da.count;
  else if (k == "mu") *out = c->da.mu;
  else if (k == "iter_count") *out = (double)c->iter_count;
  else if (k == "divergences") *out = (double)c->divergences;
  else if (k == "n_samples") *out = (double)c->n_samples;
  else if (k == "adaptation_window") *out = (double)c->adaptation_window;
  else if (k == "tune") *out = c->tune;
  else if (k == "fg_count") *out = c->fg_count;
  else if (k == "bg_count") *out = c->bg_count;
  else if (k == "step_size") *out = c->step_size;
  else if (k == "leapfrogs") *out = (double)c->leapfrogs;
  else if (k == "tree_kernel") *out = (double)c->tree_mode;
  else if (k == "fa_fg_n") *out = c->fa_fg_n;
  else if (k == "fa_bg_n") *out = c->fa_bg_n;
  else if (k == "fa_previous_update") *out = (double)c->fa_previous_update;
  else if (k == "tree_launches") *out = (double)c->tree_launches;
  else if (k == "single_launch") *out = c->small ? 1.0 : 0.0;
  else if (k == "window_switched") *out = c->window_switched ? 1.0 : 0.0;
  else if (k == "t_begin") *out = c->t_begin;
  else if (k == "t_loop") *out = c->t_loop;
  else if (k == "t_wait") *out = c->t_wait;
  else if (k == "tm_pre") *out = c->tm_pre;
  else if (k == "tm_start") *out = c->tm_start;
  else if (k == "tm_tree") *out = c->tm_tree;
  else if (k == "tm_record") *out = c->tm_record;
  else if (k == "tm_host") *out = c->tm_host;
  else if (k == "tm_post") *out = c->tm_post;
  else if (k == "tm_draws") *out = (double)c->tm_draws;
  else if (k == "tm_batches") *out = (double)c->tm_batches;
  else if (k == "t_finish") *out = c->t_finish;
  else { g_err = "unknown scalar " + k; return NUTS_E_ARG; }
  return NUTS_OK;
}
extern "C" int nuts_chain_get_vector(nuts_chain* c, const char* name, double* out) {
  if (!c || !name || !out) return NUTS_E_ARG;
  const std::string k(name);
  const double* src = nullptr;
  if (k == "var") src = c->var;
  else if (k == "stds") src = c->stds;
  else if (k == "inv_stds") src = c->inv_stds;
  else if (k == "fg_mean") src = c->fg_is_a ? c->wa_mean : c->wb_mean;
  else if (k == "fg_m2") src = c->fg_is_a ? c->wa_m2 : c->wb_m2;
  else if (k == "bg_mean") src = c->fg_is_a ? c->wb_mean : c->wa_mean;
  else if (k == "bg_m2") src = c->fg_is_a ? c->wb_m2 : c->wa_m2;
  else if (k == "divergence_source" || k == "divergence_dest") {
    const std::vector<double>& v = k == "divergence_source" ? c->div_source : c->div_dest;
    if ((int)v.size() != c->n) { g_err = "no divergence recorded"; return NUTS_E_ARG; }
    std::memcpy(out, v.data(), c->n * sizeof(double));
    return NUTS_OK;
  }
  else if (c->full_adapt && (k == "fa_cov" || k == "fa_chol" || k == "fa_fg_raw" || k == "fa_bg_raw")) {   // [n][n]
    src = k == "fa_cov" ? c->dense_C : k == "fa_chol" ? c->fa_L : k == "fa_fg_raw" ? c->fa_fg_raw : c->fa_bg_raw;
    HIPCHK(hipStreamSynchronize(c->m->stream));
    HIPCHK(hipMemcpy(out, src, (size_t)c->n * c->n * sizeof(double), hipMemcpyDeviceToHost));
    return NUTS_OK;
  }
  else if (c->full_adapt && (k == "fa_fg_mean" || k == "fa_bg_mean")) src = k == "fa_fg_mean" ? c->fa_fg_mean : c->fa_bg_mean;
  else if (k == "start_p" || k == "start_v") src = k == "start_p" ? c->A.P : c->A.V;   // momentum / velocity of the last draw's start state (arena slot 0)
  else { g_err = "unknown vector " + k; return NUTS_E_ARG; }
  HIPCHK(hipStreamSynchronize(c->m->stream));
  HIPCHK(hipMemcpy(out, src, c->n * sizeof(double), hipMemcpyDeviceToHost));
  return NUTS_OK;
}

// ---- host-adapted mass matrices: push the new matrix after a host-side update ------------------------------
extern "C" int nuts_chain_set_dense(nuts_chain* c, const double* cov, const double* rand) {
  if (!c || !cov || !rand) return NUTS_E_ARG;
  if (!c->dense || c->full_adapt) { g_err = "nuts_chain_set_dense: the chain was not created with NUTS_POT_FULL"; return NUTS_E_ARG; }
  const size_t nn = (size_t)c->n * c->n;
  HIPCHK(hipStreamSynchronize(c->m->stream));
  HIPCHK(hipMemcpy(c->dense_C, cov, nn * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->dense_W, rand, nn * sizeof(double), hipMemcpyHostToDevice));
  return NUTS_OK;
}
extern "C" int nuts_chain_set_diag(nuts_chain* c, const double* var, const double* stds, const double* inv_stds) {
  if (!c || !var || !stds || !inv_stds) return NUTS_E_ARG;
  if (c->dense) { g_err = "nuts_chain_set_diag: the chain has a dense potential"; return NUTS_E_ARG; }
  HIPCHK(hipStreamSynchronize(c->m->stream));
  HIPCHK(hipMemcpy(c->var, var, c->n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->stds, stds, c->n * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->inv_stds, inv_stds, c->n * sizeof(double), hipMemcpyHostToDevice));
  return NUTS_OK;
}

// ---- pooled adaptation hooks (opt-in; not reference behaviour) -----------------
extern "C" int nuts_chain_welford_export(nuts_chain* c, double* buf) {
  if (!c || !buf) return NUTS_E_ARG;
  const size_t n = c->n;
  hipStream_t s = c->m->stream;   // (every copy below is queued on the model stream, behind a pending k_potential_update)
  const double* fm = c->fg_is_a ? c->wa_mean : c->wb_mean; const double* f2 = c->fg_is_a ? c->wa_m2 : c->wb_m2;
  const double* bm = c->fg_is_a ? c->wb_mean : c->wa_mean; const double* b2 = c->fg_is_a ? c->wb_m2 : c->wa_m2;
  HIPCHK(hipMemcpyAsync(buf, &c->fg_count, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(buf + 1, fm, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(buf + 1 + n, f2, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(buf + 1 + 2 * n, &c->bg_count, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(buf + 2 + 2 * n, bm, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(buf + 2 + 3 * n, b2, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return NUTS_OK;
}
extern "C" int nuts_chain_welford_import(nuts_chain* c, const double* buf) {
  if (!c || !buf) return NUTS_E_ARG;
  const size_t n = c->n;
  hipStream_t s = c->m->stream;
  double* fm = c->fg_is_a ? c->wa_mean : c->wb_mean; double* f2 = c->fg_is_a ? c->wa_m2 : c->wb_m2;
  double* bm = c->fg_is_a ? c->wb_mean : c->wa_mean; double* b2 = c->fg_is_a ? c->wb_m2 : c->wa_m2;
  HIPCHK(hipMemcpyAsync(&c->fg_count, buf, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(fm, buf + 1, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(f2, buf + 1 + n, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(&c->bg_count, buf + 1 + 2 * n, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(bm, buf + 2 + 2 * n, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(b2, buf + 2 + 3 * n, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return NUTS_OK;
}
extern "C" int nuts_chain_set_log_step_bar(nuts_chain* c, double log_step, double log_bar) {
  if (!c) return NUTS_E_ARG;
  c->da.log_step = log_step; c->da.log_bar = log_bar;
  return NUTS_OK;
}

extern "C" int nuts_chain_profile(nuts_chain* c, int enable) {
  if (!c) return NUTS_E_ARG;
  HIPCHK(hipStreamSynchronize(c->m->stream));
  // every 61st launch of the dominant kernel is bracketed by a pair of events (a prime: no resonance with the 2^d leaves of a doubling,
  // every position of the tree gets sampled).  The pair is not free -- two marker packets the queue waits for, ~7 us: at one launch
  // in 8 the C2-S bench ran 14 % below the un-instrumented loop (54.8 k vs 63.8 k leapfrog/s, tools/draw_host_phases.py); at one
  // in 61 the cost is below 1 %
  profile_enable(c->m, enable != 0, 61, 4096);
  c->leapfrogs = 0;
  return NUTS_OK;
}
extern "C" int nuts_chain_profile_read(nuts_chain* c, double* ms_sum, int64_t* launches, int64_t* leapfrogs) {
  if (!c) return NUTS_E_ARG;
  HIPCHK(hipStreamSynchronize(c->m->stream));
  int64_t pairs = 0;
  const double tot = profile_sum_ms(c->m, &pairs);
  if (ms_sum) *ms_sum = tot;
  // (passes over the model data covered by the timed launches: one per launch, or the leaves of the tree for a tree launch)
  if (launches) *launches = c->m->dom_units;
  if (leapfrogs) *leapfrogs = c->leapfrogs;
  return NUTS_OK;
}


// ===========================================================================
// categorical Gibbs within Metropolis for mixture assignments (include/nuts_mi355.h)
// ===========================================================================
// `rng.shuffle(order)` (pcg64_stream.h `shuffle`: the candidates without a branch on acceptance, then the swaps); the scratch
// index vector is kept per host thread
static void shuffle_order(Pcg64Replay& r, int64_t n, int32_t* order) {
  static thread_local std::vector<int32_t> js;
  if ((int64_t)js.size() < n) js.resize((size_t)n);
  r.shuffle(n, order, js.data());
}

extern "C" int nuts_gibbs_plan(nuts_pcg64* rng, int64_t n, int32_t shuffle, int32_t* order, const int32_t* k_of_dim, int32_t* cand_raw,
                               double* uniform) {
  if (!rng || !order || !k_of_dim || !cand_raw || !uniform || n < 0) { g_err = "null argument"; return NUTS_E_ARG; }
  Pcg64Replay r;
  r.state = ((unsigned __int128)rng->state_hi << 64) | rng->state_lo;
  r.inc = ((unsigned __int128)rng->inc_hi << 64) | rng->inc_lo;
  r.has_uint32 = rng->has_uint32; r.uinteger = rng->uinteger;
  if (shuffle) shuffle_order(r, n, order);   // Generator.shuffle on a Python list: Fisher-Yates from the top (numpy/random/_generator.pyx, untyped path)
  r.begin_bulk();   // the raw outputs a block ahead, four interleaved lanes (pcg64_stream.h)
  bool same_k = n > 0;   // (every dimension with the same number of categories -- a mixture's assignments: no gather through the shuffled order)
  for (int64_t t = 1; t < n && same_k; ++t) same_k = k_of_dim[t] == k_of_dim[0];
  int64_t t0 = 0;
  if (same_k && k_of_dim[0] > 2) { r.end_bulk(); t0 = r.draws_same_k(n, (uint32_t)k_of_dim[0], cand_raw, uniform); r.begin_bulk(); }
  for (int64_t t = t0; t < n; ++t) {
    const int32_t k = same_k ? k_of_dim[0] : k_of_dim[order[t]];
    if (k < 2) { g_err = "a categorical dimension needs at least two categories"; return NUTS_E_ARG; }
    cand_raw[t] = (int32_t)r.integers((uint32_t)(k - 1));   // rng.choice(k - 1)
    uniform[t] = r.next_double();                           // rng.uniform()
  }
  r.end_bulk();
  rng->state_hi = (uint64_t)(r.state >> 64); rng->state_lo = (uint64_t)r.state;
  rng->has_uint32 = r.has_uint32; rng->uinteger = r.uinteger;
  return NUTS_OK;
}

// The two halves of `nuts_gibbs_plan` as calls of their own, so that a sweep's per-element draws and the NEXT sweep's shuffle can be
// replayed at the same time on two host threads (pymc_amd/gibbs.py `_PlanPipeline`): `..._shuffle` leaves the generator as it
// stands after `rng.shuffle`, `..._draws` replays the per-element loop from there, `..._skip` jumps over that loop without replaying
// it (every dimension with the same k; `*clean` of `..._draws` says whether the jump's assumption -- no Lemire rejection -- held).
static void pcg_load(Pcg64Replay& r, const nuts_pcg64* rng) {
  r.state = ((unsigned __int128)rng->state_hi << 64) | rng->state_lo;
  r.inc = ((unsigned __int128)rng->inc_hi << 64) | rng->inc_lo;
  r.has_uint32 = rng->has_uint32; r.uinteger = rng->uinteger;
}
static void pcg_store(const Pcg64Replay& r, nuts_pcg64* rng) {
  rng->state_hi = (uint64_t)(r.state >> 64); rng->state_lo = (uint64_t)r.state;
  rng->has_uint32 = r.has_uint32; rng->uinteger = r.uinteger;
}
extern "C" int nuts_gibbs_plan_shuffle(nuts_pcg64* rng, int64_t n, int32_t* order) {
  if (!rng || !order || n < 0) { g_err = "null argument"; return NUTS_E_ARG; }
  Pcg64Replay r;
  pcg_load(r, rng);
  shuffle_order(r, n, order);
  pcg_store(r, rng);
  return NUTS_OK;
}
extern "C" int nuts_gibbs_plan_draws(nuts_pcg64* rng, int64_t n, const int32_t* order, const int32_t* k_of_dim, int32_t* cand_raw, double* uniform,
                                     int32_t* clean) {
  if (!rng || !order || !k_of_dim || !cand_raw || !uniform || n < 0) { g_err = "null argument"; return NUTS_E_ARG; }
  Pcg64Replay r;
  pcg_load(r, rng);
  r.begin_bulk();
  bool same_k = n > 0;
  for (int64_t t = 1; t < n && same_k; ++t) same_k = k_of_dim[t] == k_of_dim[0];
  int64_t t0 = 0;
  if (same_k && k_of_dim[0] > 2) { r.end_bulk(); t0 = r.draws_same_k(n, (uint32_t)k_of_dim[0], cand_raw, uniform); r.begin_bulk(); }
  for (int64_t t = t0; t < n; ++t) {
    const int32_t k = same_k ? k_of_dim[0] : k_of_dim[order[t]];
    if (k < 2) { g_err = "a categorical dimension needs at least two categories"; return NUTS_E_ARG; }
    cand_raw[t] = (int32_t)r.integers((uint32_t)(k - 1));
    uniform[t] = r.next_double();
  }
  r.end_bulk();
  pcg_store(r, rng);
  if (clean) *clean = (same_k && r.rejections == 0) ? 1 : 0;
  return NUTS_OK;
}
extern "C" int nuts_gibbs_plan_skip(nuts_pcg64* rng, int64_t n, int32_t k) {
  if (!rng || n < 0 || k < 2) { g_err = "bad argument"; return NUTS_E_ARG; }
  Pcg64Replay r;
  pcg_load(r, rng);
  r.skip_draws((uint64_t)n, (uint32_t)k);
  pcg_store(r, rng);
  return NUTS_OK;
}

extern "C" int nuts_gibbs_plan_doubles(nuts_pcg64* rng, int64_t n, int32_t shuffle, int32_t* order, int64_t n_doubles, double* out) {
  if (!rng || !order || (n_doubles > 0 && !out) || n < 0 || n_doubles < 0) { g_err = "null argument"; return NUTS_E_ARG; }
  Pcg64Replay r;
  r.state = ((unsigned __int128)rng->state_hi << 64) | rng->state_lo;
  r.inc = ((unsigned __int128)rng->inc_hi << 64) | rng->inc_lo;
  r.has_uint32 = rng->has_uint32; r.uinteger = rng->uinteger;
  if (shuffle) shuffle_order(r, n, order);
  // the generator is handed back as it stands AFTER the shuffle: the doubles are looked at, not consumed (a double never touches
  // the buffered 32-bit half, so the caller's `advance(k)` lands exactly where k `random()` calls would)
  rng->state_hi = (uint64_t)(r.state >> 64); rng->state_lo = (uint64_t)r.state;
  rng->has_uint32 = r.has_uint32; rng->uinteger = r.uinteger;
  for (int64_t t = 0; t < n_doubles; ++t) out[t] = r.next_double();
  return NUTS_OK;
}

#define GIBBS_BLOCK 256
#define GIBBS_MAXK 32
// one thread per position of the plan; per-workgroup sufficient statistics in fixed order (wave sums, waves in order)
// PARV: the 3 K parameters travel in the kernel arguments (the staged sweep: nothing to upload but the plan, which is already there)
struct GibbsParV { double v[3 * GIBBS_MAXK]; };
template <bool PARV>
__global__ __launch_bounds__(GIBBS_BLOCK) void k_gibbs_sweep(int64_t n, int K, const double* __restrict__ y, int32_t* __restrict__ c,
                                                            const double* __restrict__ par /* [3][K]: log w, mu, sigma */,
                                                            const int32_t* __restrict__ order, const int32_t* __restrict__ cand_raw,
                                                            const double* __restrict__ log_u, double* __restrict__ part /* [nblk][3 K + 2] */,
                                                            GibbsParV parv) {
  __shared__ double s_par[3 * GIBBS_MAXK];
  __shared__ double s_w[GIBBS_BLOCK / WAVE][3 * GIBBS_MAXK + 2];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  for (int i = tid; i < 3 * K; i += GIBBS_BLOCK) s_par[i] = PARV ? parv.v[i] : par[i];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * GIBBS_BLOCK + tid;
  int knew = -1;
  double yi = 0.0, acc = 0.0, nonf = 0.0;
  if (t < n) {
    const int dim = order[t];
    const int cur = c[dim];
    int cand = cand_raw[t];
    if (cand >= cur) cand += 1;                                     // sample_except (metropolis.py:1225-1229)
    yi = y[dim];
    auto lp = [&](int k) {                                          // log w_k + log Normal(y | mu_k, sigma_k) up to the common constant
      const double sg = s_par[2 * K + k], z = (yi - s_par[K + k]) / sg;
      return s_par[k] - log(sg) - 0.5 * z * z;
    };
    const double mr = lp(cand) - lp(cur);                           // = logp(proposal) - logp(current) of the full model
    const bool fin = isfinite(mr);
    const bool ok = fin && log_u[t] < mr;                           // metrop_select (arraystep.py:208-235)
    knew = ok ? cand : cur;
    c[dim] = knew;
    acc = ok ? 1.0 : 0.0;
    nonf = fin ? 0.0 : 1.0;
  }
  for (int k = 0; k < K; ++k) {
    const bool m = knew == k;
    const double a = wave_sum(m ? 1.0 : 0.0), b = wave_sum(m ? yi : 0.0), d2 = wave_sum(m ? yi * yi : 0.0);
    if (lane == 0) { s_w[w][k] = a; s_w[w][K + k] = b; s_w[w][2 * K + k] = d2; }
  }
  {
    const double a = wave_sum(acc), b = wave_sum(nonf);
    if (lane == 0) { s_w[w][3 * K] = a; s_w[w][3 * K + 1] = b; }
  }
  __syncthreads();
  for (int i = tid; i < 3 * K + 2; i += GIBBS_BLOCK) {
    double sum = 0.0;
    for (int ww = 0; ww < GIBBS_BLOCK / WAVE; ++ww) sum += s_w[ww][i];
    part[(int64_t)blockIdx.x * (3 * K + 2) + i] = sum;
  }
}

struct nuts_gibbs {
  int64_t n = 0;
  int K = 0, nblk = 0;
  hipStream_t stream = nullptr;
  double *y = nullptr, *par = nullptr, *logu = nullptr, *part = nullptr;
  int32_t *c = nullptr, *order = nullptr, *cand = nullptr;
  int32_t* c2 = nullptr; double* u2 = nullptr; int8_t* flags = nullptr;   // proposal="proportional": output assignment, second uniform, finite flags
  std::vector<double> part_host;
  // plans staged on the device ahead of their sweep (nuts_gibbs_stage, called from the host thread that drew the plan) and pinned
  // landing buffers for what a sweep hands back: the sweep's own host thread then issues ONE launch and two small copies
  static constexpr int NSLOT = 8;
  int device = 0;
  hipStream_t up_stream = nullptr;
  // (round 6: ONE upload per staged plan and ONE download per sweep -- order | candidates | log-uniforms packed into one pinned block
  // and one device block per slot; assignments and per-workgroup sums side by side in one device block and one pinned landing
  // buffer; the 3 K parameters in the kernel arguments.  Round 5 issued six copies per sweep: 27 % of the GPU's time on configs[4])
  char* s_plan[NSLOT] = {};          // device: [order int32 n | cand int32 n | log u double n]
  char* s_plan_pinned[NSLOT] = {};
  int32_t *s_order[NSLOT] = {}, *s_cand[NSLOT] = {};
  double* s_logu[NSLOT] = {};
  char* cpart = nullptr;             // device: [c int32 n (padded to 8 bytes) | part double nblk (3 K + 2)]
  char* cpart_pinned = nullptr;
  size_t c_bytes = 0;
  int32_t* c_pinned = nullptr;
  double *part_pinned = nullptr;
};

extern "C" nuts_gibbs* nuts_gibbs_create(int64_t n, int32_t K, const double* y) {
  if (n <= 0 || K < 2 || K > GIBBS_MAXK || !y) { g_err = "nuts_gibbs_create: bad argument (2 <= K <= 32)"; return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device visible: libnuts_mi355 requires an MI355X (gfx950); there is no CPU fallback";
    return nullptr;
  }
  auto* g = new nuts_gibbs();
  g->n = n; g->K = K; g->nblk = (int)((n + GIBBS_BLOCK - 1) / GIBBS_BLOCK);
  HIPCHK_NULL(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
  g->y = dev_upload(y, (size_t)n);
  g->par = dev_alloc<double>(3 * (size_t)K);
  g->logu = dev_alloc<double>((size_t)n);
  g->c_bytes = (((size_t)n * sizeof(int32_t)) + 7) & ~(size_t)7;
  g->cpart = dev_alloc<char>(g->c_bytes + (size_t)g->nblk * (3 * K + 2) * sizeof(double));
  g->c = (int32_t*)g->cpart;
  g->part = g->cpart ? (double*)(g->cpart + g->c_bytes) : nullptr;
  g->order = dev_alloc<int32_t>((size_t)n); g->cand = dev_alloc<int32_t>((size_t)n);
  if (!g->y || !g->par || !g->logu || !g->part || !g->c || !g->order || !g->cand) { g_err = "device allocation failed"; nuts_gibbs_destroy(g); return nullptr; }
  g->part_host.resize((size_t)g->nblk * (3 * K + 2));
  HIPCHK_NULL(hipGetDevice(&g->device));
  return g;
}

extern "C" void nuts_gibbs_destroy(nuts_gibbs* g) {
  if (!g) return;
  if (g->stream) hipStreamSynchronize(g->stream);
  for (void* p : {(void*)g->y, (void*)g->par, (void*)g->logu, (void*)g->cpart, (void*)g->order, (void*)g->cand, (void*)g->c2, (void*)g->u2,
                  (void*)g->flags}) if (p) hipFree(p);
  if (g->up_stream) { hipStreamSynchronize(g->up_stream); hipStreamDestroy(g->up_stream); }
  for (int i = 0; i < nuts_gibbs::NSLOT; ++i) {
    if (g->s_plan[i]) hipFree(g->s_plan[i]);
    if (g->s_plan_pinned[i]) hipHostFree(g->s_plan_pinned[i]);
  }
  if (g->cpart_pinned) hipHostFree(g->cpart_pinned);
  if (g->stream) hipStreamDestroy(g->stream);
  delete g;
}

// Number of device slots a plan can be staged into (the look-ahead of pymc_amd/gibbs.py's plan pipeline must stay below it).
extern "C" int nuts_gibbs_stage_slots(void) { return nuts_gibbs::NSLOT; }

// Upload one sweep's plan (shuffled order, raw candidates, log-uniforms: what nuts_gibbs_plan produced) into device slot `slot`,
// from whatever host thread drew it; returns when the plan is on the device.  The sweep that uses it (nuts_gibbs_sweep_staged)
// then has nothing to upload but the 3 K parameters.
extern "C" int nuts_gibbs_stage(nuts_gibbs* g, int32_t slot, const int32_t* order, const int32_t* cand_raw, const double* log_u) {
  if (!g || !order || !cand_raw || !log_u || slot < 0 || slot >= nuts_gibbs::NSLOT) { g_err = "nuts_gibbs_stage: bad argument"; return NUTS_E_ARG; }
  HIPCHK(hipSetDevice(g->device));
  const size_t n = (size_t)g->n;
  if (!g->up_stream) HIPCHK(hipStreamCreateWithFlags(&g->up_stream, hipStreamNonBlocking));
  const size_t bytes = 2 * n * sizeof(int32_t) + n * sizeof(double);   // (2 n int32 = 8 n bytes: the doubles behind them are aligned)
  if (!g->s_plan[slot]) {
    g->s_plan[slot] = dev_alloc<char>(bytes);
    if (!g->s_plan[slot]) { g_err = "device allocation failed"; return NUTS_E_HIP; }
    HIPCHK(hipHostMalloc((void**)&g->s_plan_pinned[slot], bytes, hipHostMallocDefault));
    g->s_order[slot] = (int32_t*)g->s_plan[slot]; g->s_cand[slot] = g->s_order[slot] + n; g->s_logu[slot] = (double*)(g->s_plan[slot] + 2 * n * sizeof(int32_t));
  }
  char* pin = g->s_plan_pinned[slot];
  std::memcpy(pin, order, n * sizeof(int32_t));
  std::memcpy(pin + n * sizeof(int32_t), cand_raw, n * sizeof(int32_t));
  std::memcpy(pin + 2 * n * sizeof(int32_t), log_u, n * sizeof(double));
  HIPCHK(hipMemcpyAsync(g->s_plan[slot], pin, bytes, hipMemcpyHostToDevice, g->up_stream));   // ONE upload per plan
  HIPCHK(hipStreamSynchronize(g->up_stream));
  return NUTS_OK;
}

// One sweep on a staged plan.  `c_in` (int32, n entries) is uploaded when given; NULL = the assignments the previous sweep of this
// handle left on the device (the caller knows that nothing else has touched them).  The new assignments are written to `c_out`
// as int64 (`c_out_is64`) or int32 -- the dtype the point carries, so that the caller converts nothing.
extern "C" int nuts_gibbs_sweep_staged(nuts_gibbs* g, int32_t slot, const int32_t* c_in, void* c_out, int32_t c_out_is64, const double* log_w,
                                       const double* mu, const double* sigma, int64_t* n_accepted, int64_t* n_nonfinite, double* cnt, double* s1,
                                       double* s2) {
  if (!g || !c_out || !log_w || !mu || !sigma || !cnt || !s1 || !s2 || slot < 0 || slot >= nuts_gibbs::NSLOT || !g->s_order[slot]) {
    g_err = "nuts_gibbs_sweep_staged: bad argument (or a slot nothing was staged into)"; return NUTS_E_ARG;
  }
  const int K = g->K;
  const size_t n = (size_t)g->n;
  const size_t npart = (size_t)g->nblk * (3 * K + 2);
  hipStream_t s = g->stream;
  if (!g->cpart_pinned) {
    HIPCHK(hipHostMalloc((void**)&g->cpart_pinned, g->c_bytes + npart * sizeof(double), hipHostMallocDefault));
    g->c_pinned = (int32_t*)g->cpart_pinned; g->part_pinned = (double*)(g->cpart_pinned + g->c_bytes);
  }
  GibbsParV pv;
  for (int k = 0; k < K; ++k) { pv.v[k] = log_w[k]; pv.v[K + k] = mu[k]; pv.v[2 * K + k] = sigma[k]; }
  if (c_in) HIPCHK(hipMemcpyAsync(g->c, c_in, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_gibbs_sweep<true>, dim3(g->nblk), dim3(GIBBS_BLOCK), 0, s, g->n, K, g->y, g->c, (const double*)nullptr, g->s_order[slot],
                     g->s_cand[slot], g->s_logu[slot], g->part, pv);
  HIPCHK(hipMemcpyAsync(g->cpart_pinned, g->cpart, g->c_bytes + npart * sizeof(double), hipMemcpyDeviceToHost, s));   // ONE download per sweep
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  if (c_out_is64) { int64_t* o = (int64_t*)c_out; for (size_t i = 0; i < n; ++i) o[i] = g->c_pinned[i]; }
  else std::memcpy(c_out, g->c_pinned, n * sizeof(int32_t));
  const int stride = 3 * K + 2;
  double acc = 0.0, nonf = 0.0;
  for (int k = 0; k < K; ++k) { cnt[k] = s1[k] = s2[k] = 0.0; }
  for (int b = 0; b < g->nblk; ++b) {   // workgroups in order, as nuts_gibbs_sweep adds them
    const double* p = g->part_pinned + (size_t)b * stride;
    for (int k = 0; k < K; ++k) { cnt[k] += p[k]; s1[k] += p[K + k]; s2[k] += p[2 * K + k]; }
    acc += p[3 * K]; nonf += p[3 * K + 1];
  }
  if (n_accepted) *n_accepted = (int64_t)acc;
  if (n_nonfinite) *n_nonfinite = (int64_t)nonf;
  return NUTS_OK;
}

extern "C" int nuts_gibbs_sweep(nuts_gibbs* g, int32_t* c, const double* log_w, const double* mu, const double* sigma, const int32_t* order,
                                const int32_t* cand_raw, const double* log_u, int64_t* n_accepted, int64_t* n_nonfinite, double* cnt, double* s1,
                                double* s2) {
  if (!g || !c || !log_w || !mu || !sigma || !order || !cand_raw || !log_u || !cnt || !s1 || !s2) { g_err = "null argument"; return NUTS_E_ARG; }
  const int K = g->K;
  const size_t n = (size_t)g->n;
  hipStream_t s = g->stream;
  std::vector<double> par(3 * (size_t)K);
  for (int k = 0; k < K; ++k) { par[k] = log_w[k]; par[K + k] = mu[k]; par[2 * K + k] = sigma[k]; }
  HIPCHK(hipMemcpyAsync(g->par, par.data(), par.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->c, c, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->order, order, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->cand, cand_raw, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->logu, log_u, n * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_gibbs_sweep<false>, dim3(g->nblk), dim3(GIBBS_BLOCK), 0, s, g->n, K, g->y, g->c, g->par, g->order, g->cand, g->logu, g->part, GibbsParV{});
  HIPCHK(hipMemcpyAsync(c, g->c, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(g->part_host.data(), g->part, g->part_host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  const int stride = 3 * K + 2;
  double acc = 0.0, nonf = 0.0;
  for (int k = 0; k < K; ++k) { cnt[k] = s1[k] = s2[k] = 0.0; }
  for (int b = 0; b < g->nblk; ++b) {   // workgroups in order
    const double* p = g->part_host.data() + (size_t)b * stride;
    for (int k = 0; k < K; ++k) { cnt[k] += p[k]; s1[k] += p[K + k]; s2[k] += p[2 * K + k]; }
    acc += p[3 * K]; nonf += p[3 * K + 1];
  }
  if (n_accepted) *n_accepted = (int64_t)acc;
  if (n_nonfinite) *n_nonfinite = (int64_t)nonf;
  return NUTS_OK;
}


// NumPy's float64 `add.reduce` over a short contiguous vector (numpy/_core/src/umath/loops_utils.h.src, pairwise sum): fewer than 8
// elements are added in order; up to 128 go through eight running sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the
// remainder is added in order.  K <= 32 here, so the recursive split above 128 never happens.
__device__ __forceinline__ double np_sum_k(const double* a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

// `metropolis_proportional` (metropolis.py:803-826) for every element at once; one thread per position of the plan
__global__ __launch_bounds__(GIBBS_BLOCK) void k_gibbs_sweep_prop(int64_t n, int K, const double* __restrict__ y, const int32_t* __restrict__ c_in,
                                                                 int32_t* __restrict__ c_out, const double* __restrict__ par,
                                                                 const int32_t* __restrict__ order, const double* __restrict__ u_choice,
                                                                 const double* __restrict__ u_accept, int8_t* __restrict__ fin_flags,
                                                                 double* __restrict__ part /* [nblk][3 K + 2] */) {
  __shared__ double s_par[3 * GIBBS_MAXK];
  __shared__ double s_w[GIBBS_BLOCK / WAVE][3 * GIBBS_MAXK + 2];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  for (int i = tid; i < 3 * K; i += GIBBS_BLOCK) s_par[i] = par[i];
  __syncthreads();
  const int64_t t = (int64_t)blockIdx.x * GIBBS_BLOCK + tid;
  int knew = -1;
  double yi = 0.0, acc = 0.0;
  if (t < n) {
    const int dim = order[t];
    const int cur = c_in[dim];
    yi = y[dim];
    double p[GIBBS_MAXK];
    double mx = -INFINITY;
    for (int k = 0; k < K; ++k) {          // log w_k + log Normal(y | mu_k, sigma_k) up to the constant every category shares
      const double sg = s_par[2 * K + k], z = (yi - s_par[K + k]) / sg;
      p[k] = s_par[k] - log(sg) - 0.5 * z * z;
      mx = fmax(mx, p[k]);
    }
    for (int k = 0; k < K; ++k) p[k] = exp(p[k] - mx);       // scipy.special.softmax: exp(x - max) / sum(exp(x - max))
    const double tot = np_sum_k(p, K);
    for (int k = 0; k < K; ++k) p[k] = p[k] / tot;
    const double prob_curr = p[cur];
    p[cur] = 0.0;
    const double rem = 1.0 - prob_curr;
    for (int k = 0; k < K; ++k) p[k] = p[k] / rem;           // probs /= 1.0 - prob_curr
    // Generator.choice(K, p=probs): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, side="right")
    double cdf[GIBBS_MAXK];
    double run = 0.0;
    for (int k = 0; k < K; ++k) { run = (k == 0) ? p[0] : run + p[k]; cdf[k] = run; }
    const double last = cdf[K - 1];
    const double u = u_choice[t];
    int prop = 0;
    for (int k = 0; k < K; ++k) prop += (cdf[k] / last <= u) ? 1 : 0;
    if (prop > K - 1) prop = K - 1;
    const double ratio = rem / (1.0 - p[prop]);              // (1 - prob_curr) / (1 - probs[proposed])
    const bool fin = isfinite(ratio);
    const bool ok = fin && !(u_accept[t] >= ratio);          // `not isfinite(r) or uniform() >= r` -> stay
    knew = ok ? prop : cur;
    c_out[dim] = knew;
    fin_flags[t] = fin ? 1 : 0;
    acc = ok ? 1.0 : 0.0;
  }
  for (int k = 0; k < K; ++k) {
    const bool m = knew == k;
    const double a = wave_sum(m ? 1.0 : 0.0), b = wave_sum(m ? yi : 0.0), d2 = wave_sum(m ? yi * yi : 0.0);
    if (lane == 0) { s_w[w][k] = a; s_w[w][K + k] = b; s_w[w][2 * K + k] = d2; }
  }
  {
    const double a = wave_sum(acc);
    if (lane == 0) { s_w[w][3 * K] = a; s_w[w][3 * K + 1] = 0.0; }
  }
  __syncthreads();
  for (int i = tid; i < 3 * K + 2; i += GIBBS_BLOCK) {
    double sum = 0.0;
    for (int ww = 0; ww < GIBBS_BLOCK / WAVE; ++ww) sum += s_w[ww][i];
    part[(int64_t)blockIdx.x * (3 * K + 2) + i] = sum;
  }
}

extern "C" int nuts_gibbs_sweep_prop(nuts_gibbs* g, const int32_t* c_in, int32_t* c_out, const double* log_w, const double* mu, const double* sigma,
                                     const int32_t* order, const double* u_choice, const double* u_accept, int8_t* finite_flags,
                                     int64_t* n_accepted, double* cnt, double* s1, double* s2) {
  if (!g || !c_in || !c_out || !log_w || !mu || !sigma || !order || !u_choice || !u_accept || !finite_flags || !cnt || !s1 || !s2) {
    g_err = "null argument"; return NUTS_E_ARG;
  }
  const int K = g->K;
  const size_t n = (size_t)g->n;
  hipStream_t s = g->stream;
  if (!g->c2 || !g->u2 || !g->flags) {   // (all three or none: a partial failure must not leave a later call with a null buffer)
    if (g->c2) hipFree(g->c2);
    if (g->u2) hipFree(g->u2);
    if (g->flags) hipFree(g->flags);
    g->c2 = dev_alloc<int32_t>(n); g->u2 = dev_alloc<double>(n); g->flags = (int8_t*)dev_alloc<int32_t>((n + 3) / 4);
    if (!g->c2 || !g->u2 || !g->flags) {
      if (g->c2) hipFree(g->c2);
      if (g->u2) hipFree(g->u2);
      if (g->flags) hipFree(g->flags);
      g->c2 = nullptr; g->u2 = nullptr; g->flags = nullptr;
      g_err = "device allocation failed"; return NUTS_E_HIP;
    }
  }
  std::vector<double> par(3 * (size_t)K);
  for (int k = 0; k < K; ++k) { par[k] = log_w[k]; par[K + k] = mu[k]; par[2 * K + k] = sigma[k]; }
  HIPCHK(hipMemcpyAsync(g->par, par.data(), par.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->c, c_in, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->c2, c_in, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->order, order, n * sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->logu, u_choice, n * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(g->u2, u_accept, n * sizeof(double), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_gibbs_sweep_prop, dim3(g->nblk), dim3(GIBBS_BLOCK), 0, s, g->n, K, g->y, g->c, g->c2, g->par, g->order, g->logu, g->u2, g->flags, g->part);
  HIPCHK(hipMemcpyAsync(c_out, g->c2, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(finite_flags, g->flags, n * sizeof(int8_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(g->part_host.data(), g->part, g->part_host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  const int stride = 3 * K + 2;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) { cnt[k] = s1[k] = s2[k] = 0.0; }
  for (int b = 0; b < g->nblk; ++b) {   // workgroups in order
    const double* p = g->part_host.data() + (size_t)b * stride;
    for (int k = 0; k < K; ++k) { cnt[k] += p[k]; s1[k] += p[K + k]; s2[k] += p[2 * K + k]; }
    acc += p[3 * K];
  }
  if (n_accepted) *n_accepted = (int64_t)acc;
  return NUTS_OK;
}


// ===========================================================================
// full-rank minibatch ADVI on a GLM (include/nuts_mi355.h, csrc/advi.h)
// ===========================================================================
struct nuts_advi {
  AdviDev d{};
  hipStream_t stream = nullptr;
  std::vector<void*> owned;
  int64_t steps_done = 0;
  size_t in_cap = 0, hist_cap = 0;
  int64_t* idx_dev = nullptr; double* z0_dev = nullptr;
  template <typename T> T* keep(T* p) { owned.push_back((void*)p); return p; }
};

extern "C" nuts_advi* nuts_advi_create(const nuts_advi_config* c) {
  if (!c || !c->X || !c->y || c->N <= 0 || c->P <= 0 || c->P > WAVE * ADVI_MAXP_PER_LANE || c->batch <= 0 || c->n_win <= 0 || c->n_win > 64 ||
      (c->family != 0 && c->family != 1)) {
    g_err = "nuts_advi_create: bad configuration (1 <= P <= 1024, family 0 / 1)"; return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device visible: libnuts_mi355 requires an MI355X (gfx950); there is no CPU fallback";
    return nullptr;
  }
  auto* a = new nuts_advi();
  AdviDev& d = a->d;
  HIPCHK_NULL(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
  d.N = c->N; d.P = c->P; d.family = c->family; d.B = c->batch; d.n_win = c->n_win;
  d.sigma = c->sigma; d.prior_sd = c->prior_sd; d.lr = c->learning_rate; d.eps = c->epsilon;
  d.nc = c->scale_cost_to_minibatch ? (double)c->N / (double)c->batch : 1.0;   // opvi.py:1314-1332
  const int P = d.P;
  const size_t T = (size_t)P * (P + 1) / 2;
  d.X = a->keep(dev_upload(c->X, (size_t)c->N * P));
  d.y = a->keep(dev_upload(c->y, (size_t)c->N));
  std::vector<double> mu(P, 0.0), lt(T, 0.0);
  if (c->start) mu.assign(c->start, c->start + P);
  for (int i = 0; i < P; ++i) lt[(size_t)i * (i + 1) / 2 + i] = 1.0;   // eye(P)[tril_indices] (approximations.py:138-141)
  d.mu = a->keep(dev_upload(mu.data(), mu.size()));
  d.Lt = a->keep(dev_upload(lt.data(), lt.size()));
  d.acc_mu = a->keep(dev_alloc<double>((size_t)P * d.n_win));
  d.acc_L = a->keep(dev_alloc<double>(T * d.n_win));
  d.z = a->keep(dev_alloc<double>(2 * (size_t)P)); d.diag = a->keep(dev_alloc<double>(2 * (size_t)P)); d.rowq = a->keep(dev_alloc<double>(2 * (size_t)P));
  d.g = a->keep(dev_alloc<double>(P));
  d.par = 0; d.pad_ = 0;
  const int rows_per_wg = (256 / WAVE) * ADVI_ROWS_PER_WAVE;
  d.nwg = (d.B + rows_per_wg - 1) / rows_per_wg;
  d.gpart = a->keep(dev_alloc<double>((size_t)d.nwg * P));
  d.llpart = a->keep(dev_alloc<double>((size_t)d.nwg));
  for (void* p : a->owned) if (!p) { g_err = "device allocation failed"; nuts_advi_destroy(a); return nullptr; }
  hipMemset(d.acc_mu, 0, (size_t)P * d.n_win * sizeof(double));
  hipMemset(d.acc_L, 0, T * d.n_win * sizeof(double));
  HIPCHK_NULL(hipDeviceSynchronize());
  return a;
}

extern "C" void nuts_advi_destroy(nuts_advi* a) {
  if (!a) return;
  if (a->stream) hipStreamSynchronize(a->stream);
  for (void* p : a->owned) if (p) hipFree(p);
  if (a->idx_dev) hipFree(a->idx_dev);
  if (a->z0_dev) hipFree(a->z0_dev);
  if (a->d.hist) hipFree(a->d.hist);
  if (a->stream) hipStreamDestroy(a->stream);
  delete a;
}

extern "C" int nuts_advi_steps(nuts_advi* a, int32_t n_steps, const int64_t* idx, const double* z0, double* loss) {
  if (!a || !idx || !z0 || n_steps <= 0) { g_err = "null argument"; return NUTS_E_ARG; }
  AdviDev& d = a->d;
  hipStream_t s = a->stream;
  for (int64_t t = 0; t < (int64_t)n_steps * d.B; ++t)
    if (idx[t] < 0 || idx[t] >= d.N) { g_err = "minibatch row index out of range"; return NUTS_E_ARG; }
  if ((size_t)n_steps > a->in_cap) {
    HIPCHK(hipStreamSynchronize(s));
    if (a->idx_dev) hipFree(a->idx_dev);
    if (a->z0_dev) hipFree(a->z0_dev);
    if (d.hist) hipFree(d.hist);
    a->idx_dev = nullptr; a->z0_dev = nullptr; d.hist = nullptr;
    HIPCHK(hipMalloc((void**)&a->idx_dev, (size_t)n_steps * d.B * sizeof(int64_t)));
    HIPCHK(hipMalloc((void**)&a->z0_dev, (size_t)n_steps * d.P * sizeof(double)));
    HIPCHK(hipMalloc((void**)&d.hist, (size_t)n_steps * sizeof(double)));
    a->in_cap = (size_t)n_steps;
  }
  HIPCHK(hipMemcpyAsync(a->idx_dev, idx, (size_t)n_steps * d.B * sizeof(int64_t), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(a->z0_dev, z0, (size_t)n_steps * d.P * sizeof(double), hipMemcpyHostToDevice, s));
  const int P = d.P;
  const int64_t T = (int64_t)P * (P + 1) / 2;
  const int zgrid = (P + 3) / 4, ugrid = (P + 1) / 2;
  for (int st = 0; st < n_steps; ++st) {
    const double* z0s = a->z0_dev + (size_t)st * P;
    // (z of step t + 1 is formed by the update of step t from the entries it has just written; only the first step of a call
    // reads L for it)
    if (st == 0) hipLaunchKernelGGL(k_advi_z, dim3(zgrid), dim3(256), 0, s, d, z0s);
    hipLaunchKernelGGL(k_advi_rows, dim3(d.nwg), dim3(256), 0, s, d, a->idx_dev + (size_t)st * d.B);
    const double* z0n = st + 1 < n_steps ? z0s + P : (const double*)nullptr;
    const int slot = (int)(a->steps_done % d.n_win);
    if (P <= 256) hipLaunchKernelGGL(k_advi_row_update<1>, dim3(ugrid), dim3(256), 0, s, d, z0s, z0n, slot, st);
    else if (P <= 512) hipLaunchKernelGGL(k_advi_row_update<2>, dim3(ugrid), dim3(256), 0, s, d, z0s, z0n, slot, st);
    else hipLaunchKernelGGL(k_advi_row_update<4>, dim3(ugrid), dim3(256), 0, s, d, z0s, z0n, slot, st);
    d.par ^= 1;
    a->steps_done++;
  }
  if (loss) HIPCHK(hipMemcpyAsync(loss, d.hist, (size_t)n_steps * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipGetLastError());
  return NUTS_OK;
}

extern "C" int nuts_advi_get_params(nuts_advi* a, double* mu, double* L_tril) {
  if (!a) return NUTS_E_ARG;
  HIPCHK(hipStreamSynchronize(a->stream));
  const size_t P = a->d.P, T = P * (P + 1) / 2;
  if (mu) HIPCHK(hipMemcpy(mu, a->d.mu, P * sizeof(double), hipMemcpyDeviceToHost));
  if (L_tril) HIPCHK(hipMemcpy(L_tril, a->d.Lt, T * sizeof(double), hipMemcpyDeviceToHost));
  return NUTS_OK;
}
extern "C" int nuts_advi_set_params(nuts_advi* a, const double* mu, const double* L_tril) {
  if (!a) return NUTS_E_ARG;
  HIPCHK(hipStreamSynchronize(a->stream));
  const size_t P = a->d.P, T = P * (P + 1) / 2;
  if (mu) HIPCHK(hipMemcpy(a->d.mu, mu, P * sizeof(double), hipMemcpyHostToDevice));
  if (L_tril) HIPCHK(hipMemcpy(a->d.Lt, L_tril, T * sizeof(double), hipMemcpyHostToDevice));
  return NUTS_OK;
}

// The leapfrog pipeline: three launches per leapfrog step.
//
//   A  k_rows / k_mvn_matvec         the HBM-streaming pass over the model data (dominant kernel).
//                                    Reads the position through a QView: in chain mode the first half
//                                    of the leapfrog (p_half = p + eps/2 g ; q' = q + eps M^-1 p_half,
//                                    integration.py:118-127) is recomputed on the fly for the few
//                                    parameters a wave needs, so nothing has to be materialised first.
//   B  k_vector                      everything that is O(n): q' (stored), value transforms + Jacobians,
//                                    element-wise factors (reverse mode as a gather), combine of the
//                                    row-pass segments into d/dz, second half kick p' = p_half + eps/2 g',
//                                    v' = M^-1 p', kinetic energy and all U-turn dot products of the tree
//                                    merges this leaf completes (nuts.py:452-463, 380-390) as per-workgroup
//                                    partials.
//   C  k_control                     one workgroup: fixed-order sum of the partials, the "deferred"
//                                    elements (scalars and hyper-parameters whose gradient is a
//                                    cross-workgroup reduction), the energy, and the scalar tree logic of
//                                    nuts.py:334-476 consuming the pre-drawn uniforms in reference order.
//
// On the "lean" path (hierarchical-logit and MvNormal models, see control_lean below) kernel C's work of leaf j is
// folded into workgroup 0 of kernel A of leaf j+1 and overlaps the data pass: two launches per leapfrog.
//
// All reductions have a fixed order (no FP atomics): results are bit-reproducible run to run, which the
// reference promises for a fixed seed (tests/sampling/test_mcmc.py:80-109).
//
// Trajectory arena.  Every phase-space point of the current trajectory lives in HBM at slot
// (index_in_trajectory & (S-1)), S = 2^max_treedepth: Q,P,V,G are [S][n].  A trajectory occupies at most S
// consecutive indices containing 0, so the slot is unique, and tree nodes are plain integers: a proposal is
// the trajectory index of a leaf (no vector is copied when a proposal is selected), a subtree is
// (first leaf, last leaf, p_sum).
//
// Recursion -> binary counter.  `_build_subtree` (nuts.py:442-476) is a post-order traversal; leaf j of a
// 2^d-leaf subtree completes m = (number of trailing one bits of j) merges, level l merging the stored left
// sibling of 2^l leaves with the just-finished right sibling.  The vector work of those merges is static
// given j, so kernel B computes it speculatively and kernel C replays the scalar decisions.
#pragma once
#include "model_dev.h"
#include "rows_kernel.h"

#define MAX_LEVELS 12            // supports max_treedepth <= 11
#define NDOT (1 + 6 * (MAX_LEVELS + 1))
#define DOT_TOP (1 + 6 * MAX_LEVELS)
#define VEC_THREADS 256
#define PART_STRIDE (PART_DOT + NDOT)

#define ROWS_BLOCK 256

#define MODE_PLAIN 0
#define MODE_TREE 1
#define MODE_SIMPLE 2

struct Ctl {
  // trajectory-level (nuts.py:318-332)
  double E0;
  double log_size, log_accept_sum, max_energy_change;
  double div_dE;
  int n_proposals, depth, left, right, proposal, cursor;
  int aborted, turning, diverging, bad_energy;
  // current doubling
  int dir, edge;
  double eps;      // signed step (nuts.py:348,357)
  double eps_abs;
  // pending left siblings of the subtree under construction
  double st_ls[MAX_LEVELS];
  int st_prop[MAX_LEVELS];
  int n_leaves_total;
  int div_t;       // trajectory index of the state whose energy error was too large (nuts.py:433-435)
};

struct ArenaDev {
  int n, S, nblk, ept;     // dimension, slots, vector-kernel workgroups, elements per thread
  double *Q, *P, *V, *G;   // [S][n]
  double *E, *LOGP;        // [S]
  double *PS;              // [MAX_LEVELS][n] pending-sibling p_sum (level 0 unused: read from P)
  double *PSUM;            // [n] whole-tree p_sum (nuts.py:330)
  const double *var, *inv_stds;  // diagonal potential (quadpotential.py:308-326)
  Ctl* ctl;
  const double* uniforms;   // pre-drawn `step.rng.random()` values
  const double* log_uniforms;  // their logarithms, taken on the host (the tree compares log(u), nuts.py:371,466);
                               // nullptr (single-launch path): the control code takes log(u) itself
  unsigned* ga_ticket;         // arrival counters of the group-aligned row pass (rows_ga_kernel.h), reset at the end of a draw
  int ga_nticket;
  unsigned* ga_sync;           // [GA_SYNC_WORDS] progress words of the persistent tree kernel (rows_ga_tree.h), reset at the start of a draw
  const double* kin_user;      // NUTS_POT_HOST: the kinetic energy the host's `energy` / `velocity_energy` returned for the state the
                               // next control kernel looks at (nullptr: kinetic = p.v / 2, the dots the kernels took)
};
#define GA_SYNC_DONE 0         // number of block partials published since the start of the draw: leaf L is complete at (L + 1) ga_nblk
#define GA_SYNC_CTL 1          // leaves whose control work is finished (and written back)
#define GA_SYNC_ERR 2          // != 0: a wait inside the tree kernel timed out (1: row workgroup, 2: control workgroup)
#define GA_SYNC_WORDS 4

// Lives in pinned, device-mapped host memory.  `word[seq % ST_SLOTS]` = (sequence number << 32) | ST_* flags is written with ONE
// system-scope store by the control work of the LAST leaf of a doubling; the host spins on it instead of paying a stream
// synchronisation + copy per doubling, and everything it needs to go on (stop / keep doubling, next direction) is in
// that word -- no fence, no second PCIe write to wait for.  The other fields are written once per draw by
// k_draw_ctl_start and read after a stream synchronisation (fixed-length HMC).
#define ST_ABORTED 1u
#define ST_TURNING 2u
#define ST_DIVERGING 4u
#define ST_BAD_ENERGY 8u
#define ST_DIR_POS 16u
#define ST_TIMEOUT 32u   // persistent tree kernel: a wait on another workgroup did not complete (not all workgroups resident?)
#define ST_SLOTS 4   // doublings publish round-robin (the host may have queued the next doubling before reading this one)
struct HostStatus {
  unsigned long long word[ST_SLOTS];
  int aborted, turning, diverging, bad_energy, depth, cursor, n_proposals, proposal, dir, edge;
};

struct EvalIO {
  int mode;          // MODE_*
  int explicit_pre;  // leaf modes: q' and p_half were materialised by k_leaf_pre
  int dense;         // dense mass matrix: v = C p is a mat-vec between the kernels, so B/C stop after the kick and the
                     // tree work runs in its own pair of kernels (k_tree_vec / k_tree_ctl)
  int lean;          // lean control path: kernel B evaluates the local part of the deferred elements (control_lean)
  int pre_next;      // explicit_pre models without deferred elements: kernel B also materialises the first half of the
                     // NEXT leaf of the doubling (what k_leaf_pre would do), so only leaf 0 needs that launch
  const double* q;   // MODE_PLAIN: position in
  double* grad;      // MODE_PLAIN: gradient out
  double* logp;      // MODE_PLAIN: logp out
  // leaf modes: geometry of the current doubling.  The host learns (dir, edge) from the status record it
  // reads after every doubling, so the kernels get them as arguments and never wait on the control block for
  // an address; the only thing they read from it is the `aborted` flag.
  int dir, edge;
  int left, right;   // trajectory indices of the tree's current edge states (nuts.py:326)
  double eps;        // signed step (nuts.py:348,357)
};

struct Leaf {
  int dir, edge, src, t, left, right;
  int64_t so, d_o;
  double eps, half;
};

__device__ __forceinline__ int64_t slot_off(const ArenaDev& A, int t) { return (int64_t)(t & (A.S - 1)) * A.n; }

// Resolve what this launch works on (no memory access: everything comes from the kernel arguments).
__device__ __forceinline__ void resolve_leaf(const EvalIO& io, const ArenaDev& A, int j, Leaf& lf, QView& qv) {
  if (io.mode == MODE_PLAIN) {
    qv.q = io.q; qv.p = qv.g = qv.var = nullptr; qv.eps = qv.half = 0.0; qv.composed = 0;
    lf.dir = 1; lf.edge = lf.src = lf.t = lf.left = lf.right = 0; lf.so = lf.d_o = 0; lf.eps = lf.half = 0.0;
    return;
  }
  lf.dir = io.dir; lf.edge = io.edge; lf.left = io.left; lf.right = io.right;
  lf.src = lf.edge + lf.dir * j;
  lf.t = lf.src + lf.dir;
  lf.eps = io.eps; lf.half = 0.5 * io.eps;
  lf.so = slot_off(A, lf.src); lf.d_o = slot_off(A, lf.t);
  if (io.explicit_pre) {
    qv.q = A.Q + lf.d_o; qv.p = qv.g = qv.var = nullptr; qv.eps = qv.half = 0.0; qv.composed = 0;
  } else {
    qv.q = A.Q + lf.so; qv.p = A.P + lf.so; qv.g = A.G + lf.so; qv.var = A.var;
    qv.eps = lf.eps; qv.half = lf.half; qv.composed = 1;
  }
}

// The host enqueues a whole doubling ahead; once the trajectory has terminated the remaining launches drain
// as no-ops.  The flag is loaded first and tested late, so the load overlaps the first data loads.
__device__ __forceinline__ int load_aborted(const EvalIO& io, const ArenaDev& A) {
  return io.mode == MODE_TREE ? __hip_atomic_load(&A.ctl->aborted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
}

__device__ __forceinline__ void publish_status(const Ctl* c, HostStatus* st, int seq) {
  if (!seq) return;   // only the last leaf of a doubling reports
  const unsigned flags = (c->aborted ? ST_ABORTED : 0u) | (c->turning ? ST_TURNING : 0u) | (c->diverging ? ST_DIVERGING : 0u) |
                         (c->bad_energy ? ST_BAD_ENERGY : 0u) | (c->dir > 0 ? ST_DIR_POS : 0u);
  // (bits 8 .. 31: the uniforms this draw has consumed so far -- with it the host can queue the NEXT draw's start behind this tree
  // without waiting for the draw's record)
  __hip_atomic_store(&st->word[seq & (ST_SLOTS - 1)],
                     ((unsigned long long)(unsigned)seq << 32) | ((unsigned long long)((unsigned)c->cursor & 0xffffffu) << 8) | flags,
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void publish_fields(const Ctl* c, HostStatus* st) {
  st->aborted = c->aborted; st->turning = c->turning; st->diverging = c->diverging; st->bad_energy = c->bad_energy;
  st->depth = c->depth; st->cursor = c->cursor; st->n_proposals = c->n_proposals; st->proposal = c->proposal;
  st->dir = c->dir; st->edge = c->edge;
}

// ---------------------------------------------------------------------------
// explicit first half of a leapfrog (only when the position cannot be composed on the fly)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(VEC_THREADS) void k_leaf_pre(ArenaDev A, EvalIO io, int j) {
  if (load_aborted(io, A)) return;  // (MODE_SIMPLE never aborts)
  const int src = io.edge + io.dir * j, dst = src + io.dir;
  const double eps = io.eps, half = 0.5 * eps;
  const int64_t so = slot_off(A, src), d_o = slot_off(A, dst);
  const int base = blockIdx.x * VEC_THREADS * A.ept;
  for (int e = 0; e < A.ept; ++e) {
    const int i = base + e * VEC_THREADS + threadIdx.x;
    if (i < A.n) {
      const double ph = fma(half, A.G[so + i], A.P[so + i]);
      A.P[d_o + i] = ph;
      if (!io.dense) A.Q[d_o + i] = fma(eps, A.var[i] * ph, A.Q[so + i]);   // dense: k_dense_mv finishes q'
    }
  }
}

// ---------------------------------------------------------------------------
// dense mass matrix (QuadPotentialFull / FullInv, quadpotential.py:633-725): y = C x, one wave per row
//   mode 0: y = C x          mode 1: y = C x and  q_out = q_in + eps * y   (first half of the leapfrog, q')
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dense_mv(const double* __restrict__ C, const double* __restrict__ x, double* __restrict__ y,
                                                  int n, const double* __restrict__ q_in, double* __restrict__ q_out, double eps,
                                                  const int* __restrict__ abort_flag) {
  if (abort_flag && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const int lane = threadIdx.x & (WAVE - 1);
  const int row = blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* __restrict__ cr = C + (int64_t)row * n;
  double s = 0.0;
  const int n2 = n & ~1;
  for (int c = lane * 2; c < n2; c += 2 * WAVE) {
    const double2 a = *reinterpret_cast<const double2*>(cr + c);
    s = fma(a.x, x[c], s);
    s = fma(a.y, x[c + 1], s);
  }
  if (lane == 0 && (n & 1)) s = fma(cr[n - 1], x[n - 1], s);
  s = wave_sum(s);
  if (lane == 0) {
    y[row] = s;
    if (q_out) q_out[row] = fma(eps, s, q_in[row]);
  }
}

// NUTS_POT_HOST: the position update behind a velocity the host produced, q' = q + eps v  (integration.py:124-127)
__global__ __launch_bounds__(256) void k_host_pot_drift(const double* __restrict__ q_in, const double* __restrict__ v, double* __restrict__ q_out,
                                                        double eps, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) q_out[i] = fma(eps, v[i], q_in[i]);
}

// ---------------------------------------------------------------------------
// second half of the leapfrog + the vector work of the tree merges a leaf completes
// (shared by kernel B for ordinary elements and kernel C for the deferred ones)
// ---------------------------------------------------------------------------
//
// On entry ph[e] = p_half and grad[e] = d logp / dq' for the E elements idx[e] of this thread (act[e] false:
// the element is not this thread's business).  Writes P, V (and PS / PSUM where the reference adds p_sums),
// returns per-thread partial dot products in dd[NDOT] slots that the caller reduces:
//   dd[0]               p'.v'                                   (integration.py:133)
//   dd[1+6l .. 1+6l+5]  the six U-turn dots of the level-l merge  (nuts.py:454-463)
//   dd[DOT_TOP..+5]     the six dots of `extend`                  (nuts.py:380-390)
// Reductions are done by the caller through `red`, a [NDOT][waves] LDS scratch.
// Operands of the first MERGE_PF merge levels of one element.  They belong to earlier leaves of the trajectory, so
// their addresses are known when the kernel starts: kernel B issues these loads with its very first ones and a merge
// level then costs arithmetic only (measured before: ~3.5 k cycles per level, mostly the first-touch latency of six
// dependent-free but late loads).  Slots a level does not use are loaded from valid addresses and ignored.
#define MERGE_PF 3
struct MergePrefetch { double v[MERGE_PF][6]; };

__device__ __forceinline__ void merge_prefetch(const ArenaDev& A, const Leaf& lf, int j, int i, MergePrefetch& pf) {
  const int dir = lf.dir, edge = lf.edge;
#pragma unroll
  for (int l = 0; l < MERGE_PF; ++l) {
    const int t1_left = edge + dir * (j - (2 << l) + 2);
    const int t1_right = edge + dir * (j - (1 << l) + 1);
    const int t2_left = t1_right + dir;
    const int64_t o1l = slot_off(A, t1_left), o1r = slot_off(A, t1_right), o2l = slot_off(A, t2_left);
    const double* ps1 = (l == 0) ? (A.P + o1r) : (A.PS + (int64_t)l * A.n);
    pf.v[l][0] = ps1[i]; pf.v[l][1] = A.V[o1l + i];
    if (l >= 1) { pf.v[l][2] = A.P[o2l + i]; pf.v[l][3] = A.V[o2l + i]; pf.v[l][4] = A.P[o1r + i]; pf.v[l][5] = A.V[o1r + i]; }
    else { pf.v[l][2] = pf.v[l][3] = pf.v[l][4] = pf.v[l][5] = 0.0; }   // level 0 uses two operands (the other slots alias the new leaf)
  }
}

// FROM_ARENA (dense mass matrix, k_tree_vec): p' and v' = C p' of the new leaf were stored by B/C and k_dense_mv and are read
// back instead of being produced here; everything after that is the same code.
// L8: only lanes 0 .. 7 of the wave hold elements (E == 1, one group of D <= 8 elements per wave): the reductions stop after three steps
template <int E, bool FROM_ARENA = false, bool L8 = false>
__device__ __forceinline__ void leaf_post(const ArenaDev& A, const Leaf& lf, int j, int d, bool tree,
                                          const int (&idx)[E], const bool (&act)[E], const double (&grad)[E],
                                          const double (&ph)[E], double* red, int nwaves, int& m_out, bool& last_out,
                                          const MergePrefetch* pf = nullptr, int wsel = -1, bool pf_on = true) {
  // `wsel` >= 0: `red` is this wave's own scratch (rows_gb_kernel.h: every wave finishes a group of its own)
  const int lane = threadIdx.x & (WAVE - 1), w = wsel >= 0 ? wsel : (int)(threadIdx.x >> 6);
  const int dir = lf.dir, edge = lf.edge, t = lf.t;
  const int64_t to = lf.d_o;
  double acc[E], vt[E], pt[E];
  double kin = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    acc[e] = 0.0; vt[e] = 0.0; pt[e] = 0.0;
    if (act[e]) {
      const int i = idx[e];
      double p, v;
      if (FROM_ARENA) { p = A.P[to + i]; v = A.V[to + i]; }
      else {
        p = fma(lf.half, grad[e], ph[e]);  // p' = p_half + eps/2 g'   (integration.py:131)
        v = A.var[i] * p;                  // v' = M^-1 p'
        A.P[to + i] = p; A.V[to + i] = v;
      }
      acc[e] = p; vt[e] = v; pt[e] = p;
      kin = fma(p, v, kin);
    }
  }
  {
    const double s = (L8 ? wave_sum8 : wave_sum)(kin);
    if (lane == 0) red[0 * nwaves + w] = s;
  }
  int m = 0;
  bool last = false;
  if (tree) {
    while (((j >> m) & 1) && m < d) ++m;
    last = (j + 1 == (1 << d));
    // merges: level l joins leaves [j-2^(l+1)+1, j-2^l] (t1) with [j-2^l+1, j] (t2)   (nuts.py:452-463)
    auto level = [&](int l, const double* pfl) {   // pfl: prefetched operands of this level (E == 1 only)
      const int t1_left = edge + dir * (j - (2 << l) + 2);
      const int t1_right = edge + dir * (j - (1 << l) + 1);
      const int t2_left = t1_right + dir;
      const int64_t o1l = slot_off(A, t1_left), o1r = slot_off(A, t1_right), o2l = slot_off(A, t2_left);
      const double* ps1 = (l == 0) ? (A.P + o1r) : (A.PS + (int64_t)l * A.n);  // a single leaf's p_sum is its p
      double dd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (act[e]) {
          const int i = idx[e];
          double s1, v1l, p2l, v2l, p1r, v1r;
          if (pfl) { s1 = pfl[0]; v1l = pfl[1]; p2l = pfl[2]; v2l = pfl[3]; p1r = pfl[4]; v1r = pfl[5]; }
          else {
            // all six operands are loaded unconditionally (valid slots at every level) so they are in flight together
            s1 = ps1[i]; v1l = A.V[o1l + i]; p2l = A.P[o2l + i]; v2l = A.V[o2l + i]; p1r = A.P[o1r + i]; v1r = A.V[o1r + i];
          }
          const double s2 = acc[e];
          const double rho = s1 + s2;                  // tree1.p_sum + tree2.p_sum
          dd[0] = fma(rho, v1l, dd[0]);
          dd[1] = fma(rho, vt[e], dd[1]);
          if (l >= 1) {
            const double rho1 = s1 + p2l;              // tree1.p_sum + tree2.left.p
            dd[2] = fma(rho1, v1l, dd[2]);
            dd[3] = fma(rho1, v2l, dd[3]);
            const double rho2 = p1r + s2;              // tree1.right.p + tree2.p_sum
            dd[4] = fma(rho2, v1r, dd[4]);
            dd[5] = fma(rho2, vt[e], dd[5]);
          }
          acc[e] = rho;
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) { const double s = (L8 ? wave_sum8 : wave_sum)(dd[k]); if (lane == 0) red[(1 + 6 * l + k) * nwaves + w] = s; }
    };
    if (E == 1 && pf && pf_on) {
#pragma unroll
      for (int l = 0; l < MERGE_PF; ++l) if (l < m) level(l, pf->v[l]);
      for (int l = MERGE_PF; l < m; ++l) level(l, nullptr);
    } else {
      for (int l = 0; l < m; ++l) level(l, nullptr);
    }
    if (!last) {
      // subtree not complete: park the merged p_sum as the pending left sibling of level m
      if (m >= 1) {
        double* ps = A.PS + (int64_t)m * A.n;
#pragma unroll
        for (int e = 0; e < E; ++e) if (act[e]) ps[idx[e]] = acc[e];
      }
    } else {
      // subtree complete: top-level merge of `extend` (nuts.py:346-390), speculative
      const int first = edge + dir;  // first leaf of the new subtree
      int lm_begin, lm_end, rm_begin, rm_end, new_left, new_right;
      if (dir > 0) { lm_begin = lf.left; lm_end = lf.right; rm_begin = first; rm_end = t; new_left = lf.left; new_right = t; }
      else         { lm_begin = t; lm_end = first; rm_begin = lf.left; rm_end = lf.right; new_left = t; new_right = lf.right; }
      const int64_t onl = slot_off(A, new_left), onr = slot_off(A, new_right);
      const int64_t olb = slot_off(A, lm_begin), ole = slot_off(A, lm_end), orb = slot_off(A, rm_begin), ore = slot_off(A, rm_end);
      double dd[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (act[e]) {
          const int i = idx[e];
          const double old = A.PSUM[i], sub = acc[e];
          const double tot = old + sub;                       // p_sum[:] += tree.p_sum
          A.PSUM[i] = tot;
          const double lm_sum = dir > 0 ? old : sub, rm_sum = dir > 0 ? sub : old;
          // the new edge state is this leaf: its v is in registers (the store above may not be visible yet)
          const double v_nl = (new_left == t) ? vt[e] : A.V[onl + i];
          const double v_nr = (new_right == t) ? vt[e] : A.V[onr + i];
          dd[0] = fma(tot, v_nl, dd[0]);
          dd[1] = fma(tot, v_nr, dd[1]);
          const double p_rb = (rm_begin == t) ? pt[e] : A.P[orb + i];
          const double v_rb = (rm_begin == t) ? vt[e] : A.V[orb + i];
          const double v_lb = (lm_begin == t) ? vt[e] : A.V[olb + i];
          const double r1 = lm_sum + p_rb;                    // leftmost_p_sum + rightmost_begin.p
          dd[2] = fma(r1, v_lb, dd[2]);
          dd[3] = fma(r1, v_rb, dd[3]);
          const double p_le = (lm_end == t) ? pt[e] : A.P[ole + i];
          const double v_le = (lm_end == t) ? vt[e] : A.V[ole + i];
          const double v_re = (rm_end == t) ? vt[e] : A.V[ore + i];
          const double r2 = p_le + rm_sum;                    // leftmost_end.p + rightmost_p_sum
          dd[4] = fma(r2, v_le, dd[4]);
          dd[5] = fma(r2, v_re, dd[5]);
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) { const double s = (L8 ? wave_sum8 : wave_sum)(dd[k]); if (lane == 0) red[(DOT_TOP + k) * nwaves + w] = s; }
    }
  }
  m_out = m; last_out = last;
}

__device__ __forceinline__ bool dot_needed(int k, int m, bool last) {
  return (k == 0) || (k >= 1 && k < 1 + 6 * m) || (last && k >= DOT_TOP);
}

// ---------------------------------------------------------------------------
// gathered adjoints (model_dev.h GSlot): one forward + reverse sweep per element of the factors that read variables through index
// vectors, ahead of kernel B / C, whose gathers add the stored adjoints up
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gsweep(ModelDev md, ArenaDev A, EvalIO io, int j) {
  if (load_aborted(io, A)) return;
  __shared__ __attribute__((aligned(16))) char s_prog[PROG_LDS_MAX];
  __shared__ double s_bacc[MAX_BTERMS][256];
  __shared__ double s_w[256 / WAVE];
  ProgRegs pregs;
  prog_issue(md, pregs);
  const Prog pg = load_prog(md, s_prog, pregs);
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  const int tid = threadIdx.x;
  for (int b = 0; b < md.n_bterms; ++b) s_bacc[b][tid] = 0.0;
  double lp = 0.0;
  for (int e = blockIdx.x * 256 + tid; e < md.n_gs_elems; e += gridDim.x * 256) lp += gsweep_element<true>(pg, qv, e, &s_bacc[0][tid], 256);
  // this workgroup's share of the orphan factors' log-density and of their scalars' adjoints (kernel B's workgroup 0 adds the records up)
  double* rec = md.gs_part + (int64_t)blockIdx.x * (1 + MAX_BTERMS);
  const double t = block_sum<false>(lp, s_w);
  if (tid == 0) rec[0] = t;
  for (int b = 0; b < md.n_bterms; ++b) {
    const double tb = block_sum<false>(s_bacc[b][tid], s_w);
    if (tid == 0) rec[1 + b] = tb;
  }
}

// The same sweep with the instructions' values, their adjoints and the slots' totals in LDS instead of scratch (model_dev.h LdsVec): one
// WAVE per workgroup, (2 x longest program + most slots) KB/2 of dynamic LDS; chosen when that fits 40 KB (engine.hip).  Same
// arithmetic in the same order: the bits of the scratch version.
#define GSL_THREADS 64
__global__ __launch_bounds__(GSL_THREADS) void k_gsweep_lds(ModelDev md, ArenaDev A, EvalIO io, int j) {
  if (load_aborted(io, A)) return;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // [gs_lds_rows][64] doubles, then the interpreter's tables
  const int tid = threadIdx.x;
  char* s_prog = reinterpret_cast<char*>(s_dyn + (size_t)md.gs_lds_rows * GSL_THREADS);
  {   // (one wave: the blob is copied 64 x 16 B at a time)
    const int n16 = (md.prog_bytes + 15) >> 4;
    for (int i = tid; i < n16; i += GSL_THREADS) reinterpret_cast<uint4*>(s_prog)[i] = reinterpret_cast<const uint4*>(md.prog)[i];
    __syncthreads();
  }
  const Prog pg = prog_view(md, s_prog);
#ifdef NUTS_KTIMING
  const bool tk = blockIdx.x == 0 && tid == 0;
  const long long tk_a = tk ? tick_now() : 0;
#endif
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  int max_instr = 0;
  for (int t = 0; t < pg.n_gsf; ++t) max_instr = max(max_instr, pg.factors[pg.gsf[t].f].n_instr);
  double* s_bacc = s_dyn + tid;                                           // rows [0, n_bterms): the broadcast accumulators
  for (int b = 0; b < md.n_bterms; ++b) s_bacc[b * GSL_THREADS] = 0.0;
  double* col = s_dyn + (size_t)md.n_bterms * GSL_THREADS + tid;
  const LdsVec tv{col, GSL_THREADS}, ta{col + (size_t)max_instr * GSL_THREADS, GSL_THREADS}, gadj{col + (size_t)2 * max_instr * GSL_THREADS, GSL_THREADS};
  double lp = 0.0;
  for (int e = blockIdx.x * GSL_THREADS + tid; e < md.n_gs_elems; e += gridDim.x * GSL_THREADS)
    lp += gsweep_element_lds<true>(pg, qv, e, s_bacc, GSL_THREADS, tv, ta, gadj);
#ifdef NUTS_KTIMING
  if (tk) md.ticks[39] += tick_now() - tk_a;   // the wave's whole sweep (set-up of the views + element loop)
#endif
  double* rec = md.gs_part + (int64_t)blockIdx.x * (1 + MAX_BTERMS);
  const double t = wave_sum(lp);
  if (tid == 0) rec[0] = t;
  for (int b = 0; b < md.n_bterms; ++b) {
    const double tb = wave_sum(s_bacc[b * GSL_THREADS]);
    if (tid == 0) rec[1 + b] = tb;
  }
}

// The adjoint sweep with resolved operands, driven by the SCALAR unit (model_dev.h SwFactor; round 6).  One wave per workgroup and per
// 64-element block of ONE factor, so everything about the program is the same in all lanes -- and the compiler is told so: the
// sweep's tables are read through the constant address space (s_load into SGPRs: opcode, operand kinds and references, the leaves'
// pointers), the 47-way opcode switch and the operand kinds are scalar compares and branches, and the vector unit is left with the
// arithmetic and the LDS columns.  (Read per lane, as the generic sweeps read them, every `switch` is a tree of exec-mask tests
// with the merged values of all its cases alive: 432 vector registers, one wave per SIMD, ~2 000 cycles per interpreted instruction.)
// Dynamic LDS = [sw_rows][64] doubles: broadcast accumulators, leaf values, slot adjoints, instruction values, instruction adjoints.
#define CONSTAS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const CONSTAS T* as_const(const T* p) { return (const CONSTAS T*)(uintptr_t)p; }
typedef int sw_v16i __attribute__((ext_vector_type(16)));
typedef int sw_v8i __attribute__((ext_vector_type(8)));
typedef int sw_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double sw_f64(int lo, int hi) { return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo)); }
// one instruction of a swept program in scalar registers: a single 64-byte s_load (nuts_instr is 64 bytes: op, pad, k, then three
// operands of (kind, ref, c)), issued one instruction AHEAD of its use -- a dependent chain of separate scalar loads (opcode, then
// the operand's kind, then its reference ...) costs a trip to the scalar cache each, 150 - 200 cycles, five or six times per instruction
struct SwOp { int kind, ref; double c; };
struct SwInstr { int op; double k; SwOp x, y, z; };
__device__ __forceinline__ SwInstr sw_load_instr(const CONSTAS nuts_instr* I) {
  const sw_v16i w = *reinterpret_cast<const CONSTAS sw_v16i*>(I);
  SwInstr r;
  r.op = w[0]; r.k = sw_f64(w[2], w[3]);
  r.x = SwOp{w[4], w[5], sw_f64(w[6], w[7])};
  r.y = SwOp{w[8], w[9], sw_f64(w[10], w[11])};
  r.z = SwOp{w[12], w[13], sw_f64(w[14], w[15])};
  return r;
}
__device__ __forceinline__ SwOp sw_load_op(const CONSTAS nuts_operand* q) {
  const sw_v4i w = *reinterpret_cast<const CONSTAS sw_v4i*>(q);
  return SwOp{w[0], w[1], sw_f64(w[2], w[3])};
}

__global__ __launch_bounds__(GSL_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gsweep_fast(ModelDev md, ArenaDev A, EvalIO io, int j) {
  if (load_aborted(io, A)) return;
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  const int tid = threadIdx.x;
  constexpr int S = GSL_THREADS;
  const CONSTAS SwFactor* swf = as_const(reinterpret_cast<const SwFactor*>(md.sw_blob));
  const CONSTAS SwLeaf* leaves = as_const(reinterpret_cast<const SwLeaf*>(md.sw_blob + md.sw_po_leaf));
  const CONSTAS nuts_instr* instrs = as_const(reinterpret_cast<const nuts_instr*>(md.sw_blob + md.sw_po_instr));
  const CONSTAS int64_t* slot_off = as_const(reinterpret_cast<const int64_t*>(md.sw_blob + md.sw_po_slot));
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  double* s_bacc = s_dyn + tid;
  for (int b = 0; b < md.n_bterms; ++b) s_bacc[b * S] = 0.0;
  double* lv = s_dyn + (size_t)md.n_bterms * S + tid;      // leaf values
  double* la = lv + (size_t)md.sw_max_leaves * S;           // the slots' adjoints
  double* tv = la + (size_t)md.sw_max_slots * S;            // instruction values
  double* ta = tv + (size_t)md.sw_max_instr * S;            // instruction adjoints
  Prog pgk; pgk.fdead_mode = md.fdead_mode; pgk.fdead = md.fdead;
  double lp_acc = 0.0;
#ifdef NUTS_KTIMING
  // phases of the first block's sweep, wave 0: ticks[32 ..] += {forward, arguments + density, arguments' adjoints, reverse, stores, leaves}; [38]: sweeps
  const bool tk_ = blockIdx.x == 0;
  long long tk0_ = tk_ ? tick_now() : 0;
  const long long tk_a = tk0_;
#define SWF_TICK(slot) do { if (tk_ && tid == 0) { const long long t_ = tick_now(); md.ticks[slot] += t_ - tk0_; tk0_ = t_; } } while (0)
#else
#define SWF_TICK(slot) do { } while (0)
#endif
  for (int blk = blockIdx.x; blk < md.sw_blocks; blk += gridDim.x) {
    int t = 0;
    while (t + 1 < md.n_swf && blk >= swf[t + 1].blk0) ++t;
    const CONSTAS SwFactor* F = swf + t;
    // (the factor's header: two 32-byte loads)
    const sw_v8i h0 = *reinterpret_cast<const CONSTAS sw_v8i*>(F), h1 = *(reinterpret_cast<const CONSTAS sw_v8i*>(F) + 1);
    const int ff = h0[0], fsize = h0[2], orphan = h0[3], leaf0 = h0[4], nl = h0[5], n2 = h0[6], instr0 = h0[7];
    const int n_instr = h1[0], nargs = h1[1], dist = h1[2], slot0 = h1[3], n_slots = h1[4], blk0 = h1[5];
    const double konst = sw_f64(h1[6], h1[7]);
    const int li = (blk - blk0) * S + tid;
    if (li >= fsize) continue;                          // (the factor's last block)
    const CONSTAS SwLeaf* L = leaves + leaf0;
    // ---- every leaf, up front: first the direct loads (data, predictor columns, gather indices), four leaves in flight ----
    for (int l0 = 0; l0 < nl; l0 += 4) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const sw_v8i w = *reinterpret_cast<const CONSTAS sw_v8i*>(L + min(l0 + u, nl - 1));      // kind, bcast, voff, transform, ptr
        const double* p = reinterpret_cast<const double*>(((unsigned long long)(unsigned)w[5] << 32) | (unsigned)w[4]);
        v[u] = w[0] == SWL_VAR ? 0.0 : p[w[1] ? 0 : li];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (l0 + u < nl) lv[(l0 + u) * S] = v[u];
    }
    // ... then the position's elements behind the gathers and the direct variables (the first n2 leaves)
    for (int l0 = 0; l0 < n2; l0 += 4) {
      double v[4];
      int tr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int l = min(l0 + u, n2 - 1);
        const sw_v4i w = *reinterpret_cast<const CONSTAS sw_v4i*>(L + l);
        tr[u] = w[3];
        const int i2 = w[2] + (w[0] == SWL_GATHER ? (int)lv[l * S] : (w[1] ? 0 : li));
        v[u] = qv.at(i2);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (l0 + u < n2) {
        double x = v[u];
        if (tr[u] != NUTS_TR_NONE) { const CONSTAS SwLeaf* Lf = L + l0 + u; x = transform_x_ol(tr[u], Lf->lower, Lf->upper, x); }
        lv[(l0 + u) * S] = x;
      }
    }
    for (int sl = 0; sl < n_slots; ++sl) la[sl * S] = 0.0;
    SWF_TICK(37);
    auto val = [&](const SwOp& q) -> double { return q.kind == NUTS_OP_TMP ? tv[q.ref * S] : (q.kind == SW_LEAF ? lv[q.ref * S] : q.c); };
    // a leaf operand's `c`: >= 0 the factor's slot that collects its adjoint; <= -2 the broadcast accumulator -2 - c of a scalar
    // (credited when the sweep accounts the factor); -1: nothing is kept (data)
    auto push = [&](const SwOp& q, double g) {
      if (q.kind == NUTS_OP_TMP) { ta[q.ref * S] += g; return; }
      if (q.kind != SW_LEAF) return;
      const int sl = (int)q.c;
      if (sl >= 0) la[sl * S] += g;
      else if (sl <= -2 && orphan) s_bacc[(-2 - sl) * S] += g;
    };
    // ---- forward sweep (model_dev.h prog_forward, on LDS only) ----
    const CONSTAS nuts_instr* ins = instrs + instr0;
    int pdead0 = 0;
    {
      SwInstr nx = sw_load_instr(ins);
#pragma unroll 1
      for (int i = 0; i < n_instr; ++i) {
        const SwInstr I = nx;
        nx = sw_load_instr(ins + min(i + 1, n_instr - 1));
        const int op = I.op;
        const double x = val(I.x);
        const bool has_y = op <= NUTS_E_DIV || (op >= NUTS_E_GT && op <= NUTS_E_OR) || op == NUTS_E_SWITCH || op == NUTS_E_MAXIMUM || op == NUTS_E_MINIMUM ||
                           op == NUTS_E_POW || op == NUTS_E_LOGADDEXP || op == NUTS_E_CLIP || op == NUTS_E_CHECK;
        const double y = has_y ? val(I.y) : 0.0;
        const double z = (op == NUTS_E_SWITCH || op == NUTS_E_CLIP) ? val(I.z) : 0.0;
        if (op == NUTS_E_CHECK && y == 0.0) pdead0 = 1;
        tv[i * S] = prog_op_value(op, I.k, x, y, z);
        ta[i * S] = 0.0;
      }
    }
    SWF_TICK(32);
    SwOp arg[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int u = 0; u < 3; ++u) arg[k][u] = sw_load_op(&F->arg[k][u]);
    double a[4] = {0.0, 0.0, 0.0, 0.0}, bv[4] = {0.0, 0.0, 0.0, 0.0}, cv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nargs) {
        bv[k] = val(arg[k][1]); cv[k] = val(arg[k][2]);
        a[k] = val(arg[k][0]) + bv[k] * cv[k];
      }
    // ---- the factor's density and its partials ----
    double d[4];
    int pdead = pdead0;
    double lp = dist_eval_uniform(dist, konst, a, d, &pdead);
    if (pdead0) { lp = -INFINITY; d[0] = d[1] = d[2] = d[3] = 0.0; }
    factor_kill(pgk, ff, pdead, lp, d);
    SWF_TICK(33);
    // ---- reverse sweep (model_dev.h factor_prog_rev_t: the same rules, the same order) ----
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nargs && d[k] != 0.0) { push(arg[k][0], d[k]); push(arg[k][1], d[k] * cv[k]); push(arg[k][2], d[k] * bv[k]); }
    SWF_TICK(34);
    if (n_instr > 0) {
      SwInstr nx = sw_load_instr(ins + n_instr - 1);
#pragma unroll 1
      for (int i = n_instr - 1; i >= 0; --i) {
        const SwInstr I = nx;
        nx = sw_load_instr(ins + max(i - 1, 0));
        const double g = ta[i * S];
        if (g == 0.0) continue;
        double gx, gy, gz; bool px, py, pz;
        prog_op_adjoint(I.op, I.k, g, tv[i * S], val(I.x), val(I.y), val(I.z), gx, gy, gz, px, py, pz);
        if (px) push(I.x, gx);
        if (py) push(I.y, gy);
        if (pz) push(I.z, gz);
      }
    }
    SWF_TICK(35);
    // ---- the slots' adjoints to where the gathers / the transposed mat-vec read them ----
    const CONSTAS int64_t* so = slot_off + slot0;
    for (int sl = 0; sl < n_slots; ++sl) md.adj[so[sl] + li] = la[sl * S];
    SWF_TICK(36);
    if (orphan) lp_acc += lp;
  }
#ifdef NUTS_KTIMING
  if (tk_ && tid == 0) { md.ticks[39] += tick_now() - tk_a; md.ticks[38] += 1; }
#endif
#undef SWF_TICK
  double* rec = md.gs_part + (int64_t)blockIdx.x * (1 + MAX_BTERMS);
  const double tsum = wave_sum(lp_acc);
  if (tid == 0) rec[0] = tsum;
  for (int b = 0; b < md.n_bterms; ++b) {
    const double tb = wave_sum(s_bacc[b * S]);
    if (tid == 0) rec[1 + b] = tb;
  }
}

// ... and the totals of the long inverse-index lists (model_dev.h GLong): one workgroup per chunk of a list
__global__ __launch_bounds__(256) void k_gadj_reduce(ModelDev md, ArenaDev A, EvalIO io) {
  if (load_aborted(io, A)) return;
  __shared__ double s_w[256 / WAVE];
  const double tot = gadj_long_total(md, md.glong[blockIdx.x], s_w);
  if (threadIdx.x == 0) md.adj_red[blockIdx.x] = tot;
}

// ---------------------------------------------------------------------------
// B: the O(n) kernel
// ---------------------------------------------------------------------------
template <int EPT, bool PROG>
__global__ __launch_bounds__(VEC_THREADS) void k_vector(ModelDev md, ArenaDev A, EvalIO io, int j, int d) {
  Leaf lf; QView qv;
  const int aborted = load_aborted(io, A);
  resolve_leaf(io, A, j, lf, qv);
  constexpr int NW = VEC_THREADS / WAVE;
  __shared__ __attribute__((aligned(16))) char s_prog[PROG_LDS_MAX];
  __shared__ double s_bacc[MAX_BTERMS][VEC_THREADS];
  __shared__ double s_red[NDOT * NW];
  __shared__ double s_dz[2][VEC_THREADS];
  __shared__ double s_w[NW];
  __shared__ double s_bw[MAX_BTERMS][NW];
  const int tid = threadIdx.x;
  const bool leaf = io.mode != MODE_PLAIN;
  const RowsDev& lg = md.lg;
  // XCD affinity (speed only, never correctness): workgroup b is observed to run on XCD b % 8, each XCD has its own
  // L2.  When the launch is 8x oversubscribed only every 8th workgroup works, so the whole O(n) state of the chain
  // (arena slots, partials, control block) stays in ONE L2 and kernels B and C hit it instead of going to HBM.
  if ((int)gridDim.x != md.nblk && (blockIdx.x & 7) != 0) return;
  const int bid = (int)gridDim.x != md.nblk ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int nb = md.nblk;
  const int base = bid * VEC_THREADS * EPT;
  double* part = md.part + (int64_t)bid * md.part_stride;
  const bool tk = bid == nb / 2 && tid == 0 && (md.tick_j < 0 || j == md.tick_j);
  TICK(md, tk, 0);
  ProgRegs pregs;
  prog_issue(md, pregs);

  // ---- everything whose address is known up front is loaded before the model tables are needed ----
  int idx[EPT];
  bool act[EPT];
  double grad[EPT], ph[EPT], qn[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = base + e * VEC_THREADS + tid;
    idx[e] = i; act[e] = false; grad[e] = 0.0; ph[e] = 0.0; qn[e] = 0.0;
    if (i >= md.n) continue;
    // first half of the leapfrog for this element (integration.py:118-127)
    if (leaf) {
      if (io.explicit_pre) { ph[e] = A.P[lf.d_o + i]; qn[e] = A.Q[lf.d_o + i]; }
      else {
        ph[e] = fma(lf.half, A.G[lf.so + i], A.P[lf.so + i]);
        qn[e] = fma(lf.eps, A.var[i] * ph[e], A.Q[lf.so + i]);
      }
    } else qn[e] = io.q[i];
  }
  MergePrefetch mpf;
  const bool use_mpf = EPT == 1 && io.mode == MODE_TREE && !io.dense;
  if (use_mpf) merge_prefetch(A, lf, j, min(idx[0], md.n - 1), mpf);
  // the logit node's z elements: segment ranges (static tables) and the group's sigma, loaded up front
  int za0[EPT], za1[EPT], zb0[EPT], zb1[EPT];
  double zsig[EPT];
  const bool seg_fixed = EPT == 1 && lg.segK > 0;   // fixed-slot segment layout: addresses follow from the element index
  double sv[SEG_MAIN_MAX + 2];
#pragma unroll
  for (int s = 0; s < SEG_MAIN_MAX + 2; ++s) sv[s] = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    za0[e] = za1[e] = zb0[e] = zb1[e] = 0; zsig[e] = 0.0;
    const int zi = idx[e] - lg.off_z;
    if (md.has_logit && zi >= 0 && zi < lg.G * lg.D) {
      const int g = zi / lg.D, dd = zi - g * lg.D;
      if (seg_fixed) {
        const int K = lg.segK;
        const double* sp = lg.seg_part + (int64_t)g * K * lg.D + dd;
#pragma unroll
        for (int s = 0; s < SEG_MAIN_MAX; ++s) sv[s] = sp[min(s, K - 3) * lg.D];
        sv[SEG_MAIN_MAX] = sp[(K - 2) * lg.D]; sv[SEG_MAIN_MAX + 1] = sp[(K - 1) * lg.D];
      } else {
        za0[e] = lg.gseg_ptr[g]; za1[e] = lg.gseg_ptr[g + 1]; zb0[e] = lg.gmix_ptr[g]; zb1[e] = lg.gmix_ptr[g + 1];
      }
      zsig[e] = qv.at(lg.off_sigma + dd);
    }
  }
  // first slice of the row-pass log-likelihood partials (summed below): issued with the loads above
  double lp_first = 0.0;
  if (md.has_logit) {
    const int w0 = bid * VEC_THREADS + tid;
    if (w0 < lg.n_waves + lg.n_mixed) lp_first = lg.wave_lp[w0];
  }
  TICK(md, tk, 1);
  const Prog pg = load_prog(md, s_prog, pregs);   // -> LDS, ends with a barrier
  TICK(md, tk, 2);
  if (aborted) return;
  TICK(md, tk, 3);
  if (leaf && !io.explicit_pre) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) if (idx[e] < md.n) A.Q[lf.d_o + idx[e]] = qn[e];   // q' is stored for every element
  }
  // a slice of the row-pass log-likelihood partials rides along with this workgroup's logp partial
  double lp = 0.0;
  if (md.has_logit) {
    const int nlp = lg.n_waves + lg.n_mixed;
    lp += lp_first;
    for (int w = (bid + nb) * VEC_THREADS + tid; w < nlp; w += nb * VEC_THREADS) lp += lg.wave_lp[w];
  }
  if (md.has_mvn) for (int r = bid * VEC_THREADS + tid; r < md.mv.k; r += nb * VEC_THREADS) lp -= 0.5 * md.mv.rowq[r];
  if (md.has_mix && bid == 0 && tid == 0) lp += *md.mix.lp;   // (mixture_kernel.h: k_mix_reduce ran before this kernel)
  if (md.has_glm && bid == 0 && tid == 0) lp += *md.glm.lp;   // (glm_kernel.h: k_glm_reduce ran before this kernel)

  for (int b = 0; b < md.n_bterms; ++b) s_bacc[b][tid] = 0.0;
  double db_acc = 0.0, dbz_acc = 0.0;

#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = idx[e];
    if (i >= md.n) continue;
    const bool is_z = md.has_logit && i >= lg.off_z && i < lg.off_z + lg.G * lg.D;
    const int k = is_z ? lg.var_z : find_var(pg, i);
    const VarDev v = pg.vars[k];
    if (v.deferred) {
      // finished by the control kernel; on the lean path everything that does not need the cross-workgroup sums is
      // done here: value transform, own factors (their logp goes into this workgroup's partial), local gradient
      if (io.lean) {
        double x, dxdq, lj, dj, gx = 0.0;
        transform_full(v, qn[e], x, dxdq, lj, dj);
        lp += lj;
        gather_element<PROG>(pg, qv, k, i - v.offset, x, gx, lp, &s_bacc[0][tid], VEC_THREADS);
        double2* loc = reinterpret_cast<double2*>(md.def_loc) + 2 * (v.def_base + (i - v.offset));
        loc[0] = make_double2(gx, dxdq);
        loc[1] = make_double2(dj, ph[e]);
      }
      continue;
    }
    double gd = 0.0, db = 0.0;
    if (is_z) {
      // fixed-order sum of the group's segment partials; the loads of a batch are issued together
      const int dd = (i - lg.off_z) % lg.D;
      if (seg_fixed) {
        // same association as the pointer-table path: (main segments in wave order) + (start span + end span)
        const int km = lg.segK - 2;
#pragma unroll
        for (int s = 0; s < SEG_MAIN_MAX; ++s) db = s < km ? db + sv[s] : db;
        db += (0.0 + sv[SEG_MAIN_MAX]) + sv[SEG_MAIN_MAX + 1];
      } else {
        db = sum_strided(lg.seg_part + dd, lg.D, za0[e], za1[e]);
        db += sum_strided(lg.mixed_part + dd, lg.D, zb0[e], zb1[e]);
      }
    }
    double x, dxdq, lj, dj, gx = 0.0;
    if (v.normal_prior) {
      x = qn[e]; dxdq = 1.0; dj = 0.0;
      const double r = x - v.np_mu;
      gx = -r * v.np_inv_var;
      lp += -0.5 * r * r * v.np_inv_var - v.np_lognorm;
    } else {
      transform_full(v, qn[e], x, dxdq, lj, dj);
      lp += lj;
      gather_element<PROG>(pg, qv, k, i - v.offset, x, gx, lp, &s_bacc[0][tid], VEC_THREADS);
    }
    if (is_z) {
      const double sgd = lg.sigma_tr == NUTS_TR_LOG ? exp(zsig[e]) : zsig[e];
      gd = sgd * db;
      db_acc += db;            // (e*VEC_THREADS) % D == 0: every e of this thread has the same d
      dbz_acc += db * x;
    }
    if (md.has_mvn && i >= md.mv.off && i < md.mv.off + md.mv.k) gd += md.mv.gdense[i];
    if (md.has_mix) gd += md.mix.gdense[i];
    if (md.has_glm) gd += md.glm.gdense[i];
    if (md.n_lins > 0) gd += md.lin_gdense[i];   // the coefficients of linear predictors (lin_kernel.h); zero elsewhere
    grad[e] = (gx + gd) * dxdq + dj;
    act[e] = true;
    if (leaf) A.G[lf.d_o + i] = grad[e];
    else io.grad[i] = grad[e];
  }

  TICK(md, tk, 4);
  // factors without an owning variable (only scalars and data): grid-stride over their elements
  for (int o = 0; o < md.n_orphans_b; ++o) {
    const int fi = md.orphans[o];
    const int fsize = pg.factors[fi].size;
    for (int li = bid * VEC_THREADS + tid; li < fsize; li += nb * VEC_THREADS) lp += orphan_element<PROG>(pg, qv, fi, li, &s_bacc[0][tid], VEC_THREADS);
  }
  // ... the swept ones among them were accounted by k_gsweep (ModelDev.gs_part): its per-workgroup records, in order
  if (PROG && bid == 0 && md.n_orphans_b < md.n_orphans)
    for (int r = tid; r < md.n_gs_blocks; r += VEC_THREADS) {
      const double* rec = md.gs_part + (int64_t)r * (1 + MAX_BTERMS);
      lp += rec[0];
      for (int b = 0; b < md.n_bterms; ++b) s_bacc[b][tid] += rec[1 + b];
    }

  // ---- second half kick + tree-merge dot products (wave partials land in s_red) ----
  int m = 0; bool last = false;
  TICK(md, tk, 5);
  if (leaf && !io.dense) leaf_post<EPT>(A, lf, j, d, io.mode == MODE_TREE, idx, act, grad, ph, s_red, NW, m, last, &mpf, -1, use_mpf);
  if (leaf && io.pre_next) {
    // first half of the next leaf (integration.py:118-127) from this leaf's registers: the same arithmetic as k_leaf_pre
    const int64_t no = slot_off(A, lf.t + lf.dir);
#pragma unroll
    for (int e = 0; e < EPT; ++e) if (act[e]) {
      const int i = idx[e];
      const double p = fma(lf.half, grad[e], ph[e]);   // p' of this leaf, as leaf_post computed it
      const double phn = fma(lf.half, grad[e], p);
      A.P[no + i] = phn;
      A.Q[no + i] = fma(lf.eps, A.var[i] * phn, qn[e]);
    }
  }
  if (leaf && io.dense) {   // dense mass matrix: only the kick here; v' = C p' needs the mat-vec that follows
#pragma unroll
    for (int e = 0; e < EPT; ++e) if (act[e]) A.P[lf.d_o + idx[e]] = fma(lf.half, grad[e], ph[e]);
  }
  TICK(md, tk, 6);

  // ---- per-workgroup partials: logp, broadcast terms, hyper-parameter sums of the logit node, dots ----
  {
    const double w = wave_sum(lp);
    if ((tid & (WAVE - 1)) == 0) s_w[tid >> 6] = w;
  }
  for (int b = 0; b < md.n_bterms; ++b) {
    const double w = wave_sum(s_bacc[b][tid]);
    if ((tid & (WAVE - 1)) == 0) s_bw[b][tid >> 6] = w;
  }
  if (md.has_logit) { s_dz[0][tid] = db_acc; s_dz[1][tid] = dbz_acc; }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < NW; ++w) t += s_w[w];
    part[PART_LP] = t;
  }
  if (tid >= WAVE && tid < WAVE + md.n_bterms) {   // one thread per broadcast term: fixed-order sum of the wave partials
    const int b = tid - WAVE;
    double t = 0.0;
    for (int w = 0; w < NW; ++w) t += s_bw[b][w];
    part[PART_BT + b] = t;
  }
  if (md.has_logit && tid >= 2 * WAVE && tid < 2 * WAVE + 2 * lg.D) {
    const int D = lg.D, u = tid - 2 * WAVE;
    const int which = u / D, dd = u - which * D;
    // thread t' holds coordinate (base + t' - off_z) mod D
    int c0 = (base - lg.off_z) % D; if (c0 < 0) c0 += D;
    int t0 = dd - c0; if (t0 < 0) t0 += D;
    double sdz = 0.0;
    for (int t = t0; t < VEC_THREADS; t += D) sdz += s_dz[which][t];
    part[(which ? PART_DSG : PART_DMU) + dd] = sdz;
  }
  if (leaf && !io.dense) {
    for (int k = tid; k < NDOT; k += VEC_THREADS) {
      if (!dot_needed(k, m, last)) continue;
      double r = 0.0;
      for (int w = 0; w < NW; ++w) r += s_red[k * NW + w];
      part[PART_DOT + k] = r;
    }
  }
  TICK(md, tk, 7);
}

// ---------------------------------------------------------------------------
// dense mass matrix: tree vector work of one leaf, after v' = C p' (per-workgroup partial dot products)
// ---------------------------------------------------------------------------
template <int EPT>
__global__ __launch_bounds__(VEC_THREADS) void k_tree_vec(ModelDev md, ArenaDev A, EvalIO io, int j, int d) {
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  constexpr int NW = VEC_THREADS / WAVE;
  __shared__ double s_red[NDOT * NW];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * VEC_THREADS * EPT;
  double* part = md.part + (int64_t)blockIdx.x * md.part_stride;
  int idx[EPT];
  bool act[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) { idx[e] = base + e * VEC_THREADS + tid; act[e] = idx[e] < md.n; }
  int m = 0; bool last = false;
  double grad[EPT], ph[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) grad[e] = ph[e] = 0.0;
  leaf_post<EPT, true>(A, lf, j, d, io.mode == MODE_TREE, idx, act, grad, ph, s_red, NW, m, last);
  __syncthreads();
  for (int k = tid; k < NDOT; k += VEC_THREADS) {
    if (!dot_needed(k, m, last)) continue;
    double r = 0.0;
    for (int w = 0; w < NW; ++w) r += s_red[k * NW + w];
    part[PART_DOT + k] = r;
  }
}

// ---------------------------------------------------------------------------
// C: the control kernel (one workgroup)
// ---------------------------------------------------------------------------

// The uniforms a leaf's decisions may consume -- log u of its m merges and of `extend`, the raw u of the next direction -- sit at
// cursor, cursor + 1, ...: addresses known as soon as the control block is.  Read one after the other inside tree_decide each
// was a dependent miss on the single lane that decides (~1300 cycles per merge level, profiles/r03f_gb_round.txt: 26 -> 82 hundred
// cycles from m = 0 to m = 3); requested together, up front, they cost one.  Entries beyond `n` are read the old way.
#define UNI_PF 6
struct UniPrefetch { double lu0, lu1, lu2, lu3, lu4, lu5, un; int base, n, un_idx; };   // (scalars: an array member ended up in scratch)
__device__ __forceinline__ double uni_log_at(const ArenaDev& A, int i) { return A.log_uniforms ? A.log_uniforms[i] : log(A.uniforms[i]); }
__device__ __forceinline__ void uni_prefetch_none(UniPrefetch& pf) {
  pf.base = 0; pf.n = 0; pf.un_idx = -1; pf.un = 0.0;
  pf.lu0 = pf.lu1 = pf.lu2 = pf.lu3 = pf.lu4 = pf.lu5 = 0.0;
}
__device__ __forceinline__ void uni_prefetch(const ArenaDev& A, int cursor, int m, bool last, UniPrefetch& pf) {
  pf.base = cursor;
  const int want = m + (last ? 1 : 0);
  const int n = want < UNI_PF ? want : UNI_PF;
  pf.n = n;
  // (indices <= cursor + m + 1: what this leaf consumes if nothing stops it, inside the draw's worst-case budget)
  pf.lu0 = uni_log_at(A, cursor + min(0, max(n - 1, 0)));
  pf.lu1 = uni_log_at(A, cursor + min(1, max(n - 1, 0)));
  pf.lu2 = uni_log_at(A, cursor + min(2, max(n - 1, 0)));
  pf.lu3 = uni_log_at(A, cursor + min(3, max(n - 1, 0)));
  pf.lu4 = uni_log_at(A, cursor + min(4, max(n - 1, 0)));
  pf.lu5 = uni_log_at(A, cursor + min(5, max(n - 1, 0)));
  pf.un_idx = last ? cursor + m + 1 : -1;
  pf.un = A.uniforms[cursor + (last ? m + 1 : 0)];
}
// tree_decide picks the values by a run-time index: they are parked in LDS by the deciding lane itself just before it decides (a select
// chain over registers is turned into a scratch look-up table by the compiler).  `lds` holds UNI_PF doubles.
struct UniView { const double* lds; double un; int base, n, un_idx; };
__device__ __forceinline__ UniView uni_stash(const UniPrefetch& pf, double* lds) {
  lds[0] = pf.lu0; lds[1] = pf.lu1; lds[2] = pf.lu2; lds[3] = pf.lu3; lds[4] = pf.lu4; lds[5] = pf.lu5;
  return UniView{lds, pf.un, pf.base, pf.n, pf.un_idx};
}
__device__ __forceinline__ UniView uni_view_none() { return UniView{nullptr, 0.0, 0, 0, -1}; }
__device__ __forceinline__ double ctl_log_uniform(const ArenaDev& A, int cursor, const UniView& pf) {
  const int r = cursor - pf.base;
  if (r >= 0 && r < pf.n) return pf.lds[r];
  return uni_log_at(A, cursor);
}

// direction of the next doubling: `(rng.random() < 0.5) * 2 - 1` (nuts.py:215)
__device__ __forceinline__ void ctl_next_direction(Ctl* c, const double* uniforms, double u_pf = 0.0, int u_pf_idx = -1) {
  const double u = u_pf_idx == c->cursor ? u_pf : uniforms[c->cursor];
  c->cursor++;
  c->dir = (u < 0.5) ? 1 : -1;
  c->eps = c->dir > 0 ? c->eps_abs : -c->eps_abs;
  c->edge = c->dir > 0 ? c->right : c->left;
}

// Scalar decisions of one leaf (nuts.py:394-476 and, on the last leaf, `extend` 334-392), on the LDS copy of the
// control block.  `dot` = the reduced dot products of this leaf, E its energy.
__device__ __forceinline__ void tree_decide(Ctl* c, const ArenaDev& A, const Leaf& lf, const double* dot, double E, int m, bool last,
                                            double Emax, int max_depth, const UniView upf) {
  const int t = lf.t;
  const int dir = lf.dir;
  double dE = E - c->E0;                 // nuts.py:408-410
  if (isnan(dE)) dE = INFINITY;
  c->log_accept_sum = logaddexp_d(c->log_accept_sum, dE > 0 ? -dE : 0.0);  // nuts.py:412-414
  if (fabs(dE) > fabs(c->max_energy_change)) c->max_energy_change = dE;     // nuts.py:417-418
  c->n_proposals += 1;                                                      // nuts.py:436-437
  c->n_leaves_total += 1;
  if (!(dE < Emax)) {                                                       // nuts.py:419,433-435
    c->diverging = 1; c->aborted = 1; c->div_dE = dE; c->div_t = t;
    c->depth += 1;                                                          // extend: self.depth += 1 happens regardless
  } else {
    double cur_ls = -dE;
    int cur_prop = t;
    bool turning = false;
    for (int l = 0; l < m && !turning; ++l) {
      const double* dd = &dot[1 + 6 * l];
      turning = (dd[0] <= 0) || (dd[1] <= 0);
      if (!turning && l >= 1) {
        turning = (dd[2] <= 0) || (dd[3] <= 0);
        if (!turning) turning = (dd[4] <= 0) || (dd[5] <= 0);
      }
      const double ls = logaddexp_d(c->st_ls[l], cur_ls);                   // nuts.py:464
      const double logu = ctl_log_uniform(A, c->cursor, upf);
      c->cursor++;
      if (logu < cur_ls - ls) { /* keep tree2's proposal */ } else cur_prop = c->st_prop[l];
      cur_ls = ls;
    }
    if (turning) {
      c->turning = 1; c->aborted = 1; c->depth += 1;
    } else if (!last) {
      c->st_ls[m] = cur_ls; c->st_prop[m] = cur_prop;
    } else {
      // extend (nuts.py:365-392)
      if (dir > 0) c->right = t; else c->left = t;
      c->depth += 1;
      const double logu = ctl_log_uniform(A, c->cursor, upf);
      c->cursor++;
      if (logu < cur_ls - c->log_size) c->proposal = cur_prop;
      c->log_size = logaddexp_d(cur_ls, c->log_size);
      const double* dd = &dot[DOT_TOP];
      bool turn = (dd[0] <= 0) || (dd[1] <= 0);
      if (!turn) turn = (dd[2] <= 0) || (dd[3] <= 0);
      if (!turn) turn = (dd[4] <= 0) || (dd[5] <= 0);
      if (turn) { c->turning = 1; c->aborted = 1; }
      else if (c->depth < max_depth) ctl_next_direction(c, A.uniforms, upf.un, upf.un_idx);
    }
  }
}

template <bool PROG>
__global__ __launch_bounds__(VEC_THREADS) void k_control(ModelDev md, ArenaDev A, EvalIO io, int j, int d, double Emax,
                                                        int max_depth, HostStatus* st, int seq) {
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  constexpr int NW = VEC_THREADS / WAVE;
  __shared__ __attribute__((aligned(16))) char s_prog[PROG_LDS_MAX];
  __shared__ double s_sum[PART_STRIDE];
  __shared__ double s_chunk[CTL_CHUNKS][PART_STRIDE];
  __shared__ double s_bacc[MAX_BTERMS][VEC_THREADS];
  __shared__ double s_red[NDOT * NW];
  __shared__ double s_w[NW];
  __shared__ Ctl s_ctl;
  const int tid = threadIdx.x;
  const bool leaf = io.mode != MODE_PLAIN;
  const bool tree = io.mode == MODE_TREE;
  const RowsDev& lg = md.lg;

  int m = 0;
  bool last = false;
  if (tree) {
    while (((j >> m) & 1) && m < d) ++m;
    last = (j + 1 == (1 << d));
  }
  const bool tk = tid == 0 && (md.tick_j < 0 || j == md.tick_j);
  TICK(md, tk, 16);
  ProgRegs pregs;
  prog_issue(md, pregs);
  // control block -> LDS (first thing: everything else overlaps this round trip)
  if (leaf && tid < (int)(sizeof(Ctl) / sizeof(int))) reinterpret_cast<int*>(&s_ctl)[tid] = reinterpret_cast<const int*>(A.ctl)[tid];
  // the deferred elements' own inputs do not depend on the partials: fetch them now (element index from the global
  // copy of the list, then the four arena reads), so that both round trips overlap the partial sums below
  const bool mine = tid < md.n_deferred;
  int def_i = 0;
  double def_ph = 0.0, def_q = 0.0;
  if (mine) {
    def_i = md.deferred_g[2 * tid];
    if (leaf) {
      if (io.explicit_pre) { def_ph = A.P[lf.d_o + def_i]; def_q = A.Q[lf.d_o + def_i]; }
      else { def_ph = qv.p_half(def_i); def_q = qv.at(def_i); }   // q' itself was stored by kernel B
    } else def_q = io.q[def_i];
  }
  // ---- fixed-order sums of the per-workgroup partials this leaf needs: (slot, chunk) pairs in parallel ----
  const int nbt = md.n_bterms, nlg = md.has_logit ? lg.D : 0;
  const int nn = 1 + nbt + 2 * nlg + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
  auto need_slot = [&](int q) {
    if (q < 1) return PART_LP;
    q -= 1;
    if (q < nbt) return PART_BT + q;
    q -= nbt;
    if (q < nlg) return PART_DMU + q;
    q -= nlg;
    if (q < nlg) return PART_DSG + q;
    q -= nlg;
    if (q < 1 + 6 * m) return PART_DOT + q;
    return PART_DOT + DOT_TOP + (q - 1 - 6 * m);
  };
  {
    const int per = (md.nblk + CTL_CHUNKS - 1) / CTL_CHUNKS;
    for (int t = tid; t < nn * CTL_CHUNKS; t += VEC_THREADS) {
      const int c = t % CTL_CHUNKS, k = need_slot(t / CTL_CHUNKS);
      const int b0 = c * per, b1 = min(md.nblk, (c + 1) * per);
      s_chunk[c][k] = sum_strided(md.part + k, md.part_stride, b0, b1);
    }
  }
  TICK(md, tk, 17);
  const Prog pg = load_prog(md, s_prog, pregs);   // ends with a barrier (when the program fits in LDS)
  for (int b = 0; b < md.n_bterms; ++b) s_bacc[b][tid] = 0.0;
  __syncthreads();
  TICK(md, tk, 18);
  if (tree && s_ctl.aborted) {   // terminated earlier in this doubling: drain
    if (tid == 0 && st) publish_status(&s_ctl, st, seq);
    return;
  }
  for (int t = tid; t < nn; t += VEC_THREADS) {
    const int k = need_slot(t);
    double sacc = 0.0;
#pragma unroll
    for (int c = 0; c < CTL_CHUNKS; ++c) sacc += s_chunk[c][k];
    s_sum[k] = sacc;
  }
  __syncthreads();

  TICK(md, tk, 19);
  // ---- deferred elements: one thread each ----
  int idx[1] = {0};
  bool act[1] = {false};
  double grad[1] = {0.0}, ph[1] = {0.0};
  double lp = 0.0, gx = 0.0, dxdq = 1.0, dj = 0.0;
  int k = -1;
  if (mine) {
    const int i = def_i;
    k = pg.deferred[2 * tid + 1];
    const VarDev v = pg.vars[k];
    idx[0] = i;
    ph[0] = def_ph;
    const double qn = def_q;
    double x, lj;
    transform_full(v, qn, x, dxdq, lj, dj);
    lp += lj;
    gather_element<PROG>(pg, qv, k, i - v.offset, x, gx, lp, &s_bacc[0][tid], VEC_THREADS);
    if (md.has_logit) {
      if (k == lg.var_mu) gx += s_sum[PART_DMU + (i - lg.off_mu)];
      else if (k == lg.var_sigma) gx += s_sum[PART_DSG + (i - lg.off_sigma)];
    }
    if (md.has_glm) gx += md.glm.gdense[i];   // the GLM node's scalar parameters (intercept, sigma): glm_kernel.h; zero elsewhere
    if (md.n_lins > 0) gx += md.lin_gdense[i]; // a scalar coefficient of a linear predictor (lin_kernel.h)
  }
  TICK(md, tk, 20);
  // broadcast terms: the share of the ordinary elements (kernel B) + the share of deferred vector elements (here)
  if (md.n_bterms > 0) {
    __syncthreads();
    for (int b = 0; b < md.n_bterms; ++b) {
      const double t = block_sum<true>(s_bacc[b][tid], s_w);
      if (mine && pg.vars[k].size == 1 && pg.bterm_var[b] == k) gx += s_sum[PART_BT + b] + t;
    }
  }
  if (mine) {
    grad[0] = gx * dxdq + dj;
    act[0] = true;
    if (leaf) A.G[lf.d_o + idx[0]] = grad[0];
    else io.grad[idx[0]] = grad[0];
  }
  TICK(md, tk, 21);
  if (leaf && io.dense) {
    // dense mass matrix: kick the deferred elements, publish logp, stop -- energy and tree logic follow in
    // k_tree_ctl once v' = C p' is known
    if (mine) A.P[lf.d_o + idx[0]] = fma(lf.half, grad[0], ph[0]);
    const double lpd = block_sum<true>(lp, s_w);
    if (tid == 0) A.LOGP[lf.t & (A.S - 1)] = s_sum[PART_LP] + lpd + (md.has_mvn ? md.mv.konst : 0.0);
    return;
  }
  int m2 = 0; bool last2 = false;
  if (leaf) leaf_post<1>(A, lf, j, d, tree, idx, act, grad, ph, s_red, NW, m2, last2);
  TICK(md, tk, 22);
  const double lp_def = block_sum<true>(lp, s_w);   // barriers inside also publish s_red
  double logp = s_sum[PART_LP] + lp_def;
  if (md.has_mvn) logp += md.mv.konst;
  if (!leaf) {
    if (tid == 0) *io.logp = logp;
    return;
  }
  // totals: workgroup partials (in order) + the deferred elements' share
  for (int q = tid; q < NDOT; q += VEC_THREADS) {
    if (!dot_needed(q, m, last)) continue;
    double r = 0.0;
    for (int w = 0; w < NW; ++w) r += s_red[q * NW + w];
    s_sum[PART_DOT + q] += r;
  }
  __syncthreads();
  TICK(md, tk, 23);
  if (tid != 0) return;
  const double* dot = &s_sum[PART_DOT];
  const int t = lf.t, ts = t & (A.S - 1);
  A.LOGP[ts] = logp;
  const double E = 0.5 * dot[0] - logp;  // integration.py:133-134
  A.E[ts] = E;
  if (!tree) return;

  tree_decide(&s_ctl, A, lf, dot, E, m, last, Emax, max_depth, uni_view_none());
  Ctl* c = &s_ctl;
  TICK(md, tk, 24);
  *A.ctl = *c;
  if (st) publish_status(c, st, seq);
  TICK(md, tk, 25);
}

// ---------------------------------------------------------------------------
// C (lean): the control work when the only deferred elements are the logit node's mu / sigma
// ---------------------------------------------------------------------------
// Kernel B has already evaluated the local part of those elements (md.def_loc) and their logp, so what is left is
// arithmetic on the partial sums: no model tables, no interpreter.  That makes the control work small enough to ride
// in workgroup 0 of the NEXT leaf's kernel A ("folded control", k_rows below): kernel A of leaf j+1 needs from leaf j
// only the z gradients (kernel B) and mu' / sigma', which its waves finish themselves (rows_hyper_fold) -- the sums
// of the dot products, the energy and the tree decision of leaf j are off the critical path and overlap the row pass.
// What a folded launch may observe late is the `aborted` flag: a leaf that starts while its predecessor's control
// work decides to stop runs its row pass for nothing (it writes only scratch); kernel B of that leaf already sees
// the flag and the trajectory arena is never touched by a speculative leaf.
// `part` / `nblk`: the per-workgroup partial records of this leaf (kernel B's, or the block partials of the group-aligned row
// pass); `def_loc`: the local parts of its deferred elements; runs in any workgroup of 64 .. 256 threads.
// `slot_major` > 0 (group-block pass, rows_gb_kernel.h): the records are stored slot by slot, part[k * slot_major + r] with
// slot_major = the padded record count (a multiple of 64, padding zero), and a slot's total is `slot_sum` below instead of
// the chunked sum -- hundreds of records read as a few coalesced wave loads.
struct LeanSrc { const double* part; int stride, nblk; const double* def_loc; int slot_major = 0; };

// Total of one slot over all records of a slot-major table: lane l adds records l, l + 64, ... in order, then the wave's fixed
// DPP tree.  Used by everything that needs such a total (control_lean and the prologue of k_rows_gb), so they agree bit for bit.
#define SLOT_SUM_MAXR 8   // records per lane: ga_nblk <= 512
__device__ __forceinline__ void slot_sum_issue(const double* slot, int npad, int lane, double (&v)[SLOT_SUM_MAXR]) {
#pragma unroll
  for (int u = 0; u < SLOT_SUM_MAXR; ++u) v[u] = slot[min(lane + WAVE * u, npad - 1)];   // (clamped: unconditional loads -- a test per load measured 0.4 us slower per launch)
}
__device__ __forceinline__ double slot_sum_finish(const double (&v)[SLOT_SUM_MAXR], int npad, int lane) {
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < SLOT_SUM_MAXR; ++u) acc += (lane + WAVE * u < npad) ? v[u] : 0.0;
  return wave_sum(acc);
}

// AGENT: the records were written by other workgroups of THIS launch (persistent tree kernel, rows_ga_tree.h).
// `nt` (optional): the number of threads of the workgroup that take part (the others have left), if not all of them.
// BATCH: partial records in flight per thread while they are summed (sum_strided): 8 covers the paths with a few dozen records;
// the group-block pass (rows_gb_kernel.h) has hundreds and asks for more -- a template parameter so that the register needs of
// the larger batch stay out of the kernels that do not want it.
// PF: the operands of the first merge levels of the deferred elements are requested at the top, with everything else whose address
// is known (they belong to earlier leaves) -- otherwise every merge level costs the control workgroup a round of far loads.
// The control work's LDS (10 KB).  A kernel whose control workgroup needs none of the kernel's other LDS hands in bytes of its own
// (rows_gal_kernel.h: the tile ring of the row workgroups) -- static LDS of an inlined function is added to EVERY workgroup's footprint.
struct CtlLds {
  double s_sum[PART_STRIDE];
  double s_chunk[CTL_CHUNKS][PART_STRIDE];
  double s_red[NDOT * (VEC_THREADS / WAVE)];
  double s_lu[UNI_PF];
  Ctl s_ctl;
};

template <bool AGENT = false, int BATCH = 8, bool PF = false>
__device__ __forceinline__ void control_lean_in(const ModelDev& md, const ArenaDev& A, const EvalIO& io, int j, int d, double Emax,
                                                int max_depth, HostStatus* st, int seq, const LeanSrc src, int nt, bool have_upf,
                                                UniPrefetch upf, CtlLds& lds) {
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  constexpr int NWMAX = VEC_THREADS / WAVE;
  const int NT = nt ? nt : (int)blockDim.x, NW = NT / WAVE;
  double (&s_sum)[PART_STRIDE] = lds.s_sum;
  double (&s_chunk)[CTL_CHUNKS][PART_STRIDE] = lds.s_chunk;
  double (&s_red)[NDOT * NWMAX] = lds.s_red;
  double (&s_lu)[UNI_PF] = lds.s_lu;
  Ctl& s_ctl = lds.s_ctl;
  const int tid = threadIdx.x;
  const bool leaf = io.mode != MODE_PLAIN;
  const bool tree = io.mode == MODE_TREE;
  const RowsDev& lg = md.lg;
  const bool ctk = tid == 0 && (md.tick_j < 0 || j == md.tick_j);   // NUTS_KTIMING builds: slots 8 .. 13 (tools/gb_ticks.py)
  TICK(md, ctk, 8);
  int m = 0;
  bool last = false;
  if (tree) {
    while (((j >> m) & 1) && m < d) ++m;
    last = (j + 1 == (1 << d));
  }
  if (leaf) for (int t = tid; t < (int)(sizeof(Ctl) / sizeof(int)); t += NT) reinterpret_cast<int*>(&s_ctl)[t] = reinterpret_cast<const int*>(A.ctl)[t];
  const bool mine = tid < md.n_deferred;
  MergePrefetch mpf;
  int def_i = 0, def_k = 0;
  double2 l01 = make_double2(0.0, 1.0), l23 = make_double2(0.0, 0.0);
  if (mine) {
    def_i = md.deferred_g[2 * tid]; def_k = md.deferred_g[2 * tid + 1];
    if (PF && tree) merge_prefetch(A, lf, j, def_i, mpf);
    if (AGENT) {
      const double* dl = src.def_loc + 4 * tid;
      l01 = make_double2(ld_agent(dl), ld_agent(dl + 1));
      l23 = make_double2(ld_agent(dl + 2), ld_agent(dl + 3));
    } else {
      l01 = reinterpret_cast<const double2*>(src.def_loc)[2 * tid];
      l23 = reinterpret_cast<const double2*>(src.def_loc)[2 * tid + 1];
    }
  }
  // ---- fixed-order sums of the per-workgroup partials this leaf needs: (slot, chunk) pairs in parallel ----
  const int nlg = md.has_logit ? lg.D : 0;
  const int nn = 1 + 2 * nlg + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
  auto need_slot = [&](int q) {
    if (q < 1) return PART_LP;
    q -= 1;
    if (q < nlg) return PART_DMU + q;
    q -= nlg;
    if (q < nlg) return PART_DSG + q;
    q -= nlg;
    if (q < 1 + 6 * m) return PART_DOT + q;
    return PART_DOT + DOT_TOP + (q - 1 - 6 * m);
  };
  if (src.slot_major) {
    // slot-major records: wave w totals slots w, w + NW, ... four at a time (their loads in flight together)
    // (every wave of the workgroup helps here, also those beyond `nt`, which leave after the barrier below; a slot is totalled by ONE
    // wave in lane order + the fixed DPP tree, so its bits do not depend on which wave that is)
    const int lane = tid & (WAVE - 1), w = tid >> 6, NWS = (int)blockDim.x >> 6;
    for (int q0 = w; q0 < nn; q0 += 4 * NWS) {
      double v[4][SLOT_SUM_MAXR];
#pragma unroll
      for (int u = 0; u < 4; ++u) slot_sum_issue(src.part + (int64_t)need_slot(min(q0 + u * NWS, nn - 1)) * src.slot_major, src.slot_major, lane, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double tot = slot_sum_finish(v[u], src.slot_major, lane);
        if (q0 + u * NWS < nn && lane == 0) s_sum[need_slot(q0 + u * NWS)] = tot;
      }
    }
  } else {
    const int per = (src.nblk + CTL_CHUNKS - 1) / CTL_CHUNKS;
    for (int t = tid; t < nn * CTL_CHUNKS; t += NT) {
      const int c = t % CTL_CHUNKS, k = need_slot(t / CTL_CHUNKS);
      const int b0 = c * per, b1 = min(src.nblk, (c + 1) * per);
      s_chunk[c][k] = sum_strided<AGENT, BATCH>(src.part + k, src.stride, b0, b1);
    }
  }
  __syncthreads();
  if (tid >= NT) return;   // (helper waves of a larger workgroup: only the sums above)
  TICK(md, ctk, 9);
  if (tree && s_ctl.aborted) {   // terminated earlier in this doubling: drain
    if (tid == 0 && st) publish_status(&s_ctl, st, seq);
    return;
  }
  if (tree && tid == 0 && !have_upf) uni_prefetch(A, s_ctl.cursor, m, last, upf);   // lands while the elements below are finished
  if (!src.slot_major) {
    for (int t = tid; t < nn; t += NT) {
      const int k = need_slot(t);
      double sacc = 0.0;
#pragma unroll
      for (int c = 0; c < CTL_CHUNKS; ++c) sacc += s_chunk[c][k];
      s_sum[k] = sacc;
    }
  }
  __syncthreads();
  // ---- deferred elements: one thread each ----
  int idx[1] = {def_i};
  bool act[1] = {false};
  double grad[1] = {0.0}, ph[1] = {l23.y};
  if (mine) {
    // (a deferred element that is neither mu nor sigma of the logit node -- a scalar with factors of its own -- has no share in the
    // cross-workgroup sums: its gradient is its local part)
    const double S = !md.has_logit ? 0.0 : (def_k == lg.var_mu ? s_sum[PART_DMU + (def_i - lg.off_mu)]
                                           : (def_k == lg.var_sigma ? s_sum[PART_DSG + (def_i - lg.off_sigma)] : 0.0));
    grad[0] = deferred_finish(l01.x, S, l01.y, l23.x);
    act[0] = true;
    if (leaf) A.G[lf.d_o + def_i] = grad[0];
    else io.grad[def_i] = grad[0];
  }
  const double logp = s_sum[PART_LP] + (md.has_mvn ? md.mv.konst : 0.0);
  if (!leaf) {
    if (tid == 0) *io.logp = logp;
    return;
  }
  int m2 = 0; bool last2 = false;
  TICK(md, ctk, 10);
  if (md.n_deferred > 0) {   // (a model without deferred elements -- the MvNormal node -- has nothing to add to the dot products: no
                             // merge-level wave sums of zeros on the path the launch waits for)
    leaf_post<1>(A, lf, j, d, tree, idx, act, grad, ph, s_red, NW, m2, last2, (PF && tree) ? &mpf : nullptr);
    __syncthreads();
    TICK(md, ctk, 11);
    // totals: workgroup partials (in order) + the deferred elements' share
    for (int q = tid; q < NDOT; q += NT) {
      if (!dot_needed(q, m, last)) continue;
      double r = 0.0;
      for (int w = 0; w < NW; ++w) r += s_red[q * NW + w];
      s_sum[PART_DOT + q] += r;
    }
    __syncthreads();
  }
  if (tid != 0) return;
  const double* dot = &s_sum[PART_DOT];
  const int t = lf.t, ts = t & (A.S - 1);
  A.LOGP[ts] = logp;
  const double E = 0.5 * dot[0] - logp;  // integration.py:133-134
  A.E[ts] = E;
  if (!tree) return;
  TICK(md, ctk, 12);
  tree_decide(&s_ctl, A, lf, dot, E, m, last, Emax, max_depth, uni_stash(upf, s_lu));
  *A.ctl = s_ctl;
  if (st) publish_status(&s_ctl, st, seq);
  TICK(md, ctk, 13);
}

template <bool AGENT = false, int BATCH = 8, bool PF = false>
__device__ __forceinline__ void control_lean(const ModelDev& md, const ArenaDev& A, const EvalIO& io, int j, int d, double Emax,
                                             int max_depth, HostStatus* st, int seq, const LeanSrc src, int nt, bool have_upf,
                                             UniPrefetch upf) {
  __shared__ CtlLds lds;
  control_lean_in<AGENT, BATCH, PF>(md, A, io, j, d, Emax, max_depth, st, seq, src, nt, have_upf, upf, lds);
}

template <bool AGENT = false, int BATCH = 8, bool PF = false>
__device__ __forceinline__ void control_lean(const ModelDev& md, const ArenaDev& A, const EvalIO& io, int j, int d, double Emax,
                                             int max_depth, HostStatus* st, int seq, const LeanSrc src, int nt = 0) {
  UniPrefetch upf;
  uni_prefetch_none(upf);
  control_lean<AGENT, BATCH, PF>(md, A, io, j, d, Emax, max_depth, st, seq, src, nt, false, upf);
}

// `par`: launch parity of the leaf's row pass (group-aligned row pass only; its partials are double-buffered)
__device__ __forceinline__ LeanSrc lean_src(const ModelDev& md, int par) {
  if (md.lg.ga && md.lg.ga_gpw) {   // group-block pass: slot-major block partials, [2][PART_STRIDE][npad]
    const int npad = (md.lg.ga_nrec + WAVE - 1) / WAVE * WAVE;
    return LeanSrc{md.lg.ga_bpart + (int64_t)par * PART_STRIDE * npad, 1, md.lg.ga_nrec, md.def_loc + (int64_t)par * 4 * MAX_DEFERRED, npad};
  }
  // (ga_nrec = the block partials + the records of the auxiliary workgroups, rows_aux.h; equal to ga_nblk for the closed-form model)
  if (md.lg.ga) return LeanSrc{md.lg.ga_bpart + (int64_t)par * md.lg.ga_nrec * PART_STRIDE, PART_STRIDE, md.lg.ga_nrec, md.def_loc + (int64_t)par * 4 * MAX_DEFERRED};
  return LeanSrc{md.part, md.part_stride, md.nblk, md.def_loc};
}

__global__ __launch_bounds__(VEC_THREADS) void k_control_lean(ModelDev md, ArenaDev A, EvalIO io, int j, int d, double Emax,
                                                             int max_depth, HostStatus* st, int seq, int par) {
  // (standalone: the last leaf of a tree, whose control work has no next launch to ride in and sits between two draws -- the
  // operands of its merge levels are requested up front, PF)
  control_lean<false, 8, true>(md, A, io, j, d, Emax, max_depth, st, seq, lean_src(md, par));
}

// ---------------------------------------------------------------------------
// A: hierarchical Bernoulli-logit rows (the HBM-bound pass; body in rows_kernel.h)
// ---------------------------------------------------------------------------
// `fold` != 0: this is leaf j > 0 of a doubling on the lean path; workgroup 0 does the control work of leaf j-1
// (control_lean above) and every wave finishes the source state's mu / sigma itself.
template <int D, int RPL, int OCC>
__global__ __launch_bounds__(ROWS_BLOCK, OCC) void k_rows(ModelDev md, ArenaDev A, EvalIO io, int j, int rev, int fold, int d,
                                                        double Emax, int max_depth, HostStatus* st) {
  int b = (int)blockIdx.x;
  if (fold) {
    if (b == 0) { control_lean(md, A, io, j - 1, d, Emax, max_depth, st, 0, lean_src(md, 0)); return; }
    --b;
  }
  const RowsDev& R = md.lg;
  Leaf lf; QView qv;
  const int aborted = load_aborted(io, A);
  resolve_leaf(io, A, j, lf, qv);
  const int lane = threadIdx.x & (WAVE - 1);
  // workgroups [0, nb_mixed) take the mixed spans (dispatched first, so they overlap the streaming workgroups)
  const int nb_mixed = (R.n_mixed + (ROWS_BLOCK / WAVE) - 1) / (ROWS_BLOCK / WAVE);
  if (b >= nb_mixed) {
    int wave = (b - nb_mixed) * (ROWS_BLOCK / WAVE) + (threadIdx.x >> 6);
    // alternate the traversal direction between launches: the tail of the previous pass is still in the
    // 256 MiB Infinity Cache when the next pass starts from that end
    if (rev) wave = R.n_waves - 1 - wave;
    rows_main<D, RPL>(md, qv, wave, lane, aborted, fold);
  } else {
    const int mw = b * (ROWS_BLOCK / WAVE) + (threadIdx.x >> 6);
    if (mw < R.n_mixed) rows_mixed<D, RPL>(md, qv, mw, lane, aborted, fold);
  }
}

// ---------------------------------------------------------------------------
// A': MvNormal precision mat-vec (multivariate.py:165-185, 275-295): one workgroup per row of P
// ---------------------------------------------------------------------------
// The position is always materialised (every row needs all k coordinates).  P (8 k^2 bytes; 33.5 MB at k = 2048) is
// cache-resident between leapfrogs; one row per wave left only 8 waves per CU to cover the latency, so a row is
// spread over the four waves of a workgroup (16 B per lane per load, 8 KiB of the row per iteration), wave partials
// are combined in wave order.  `fold`: workgroup 0 does the control work of the previous leaf (control_lean).
#define MVN_BLOCK 256
__global__ __launch_bounds__(MVN_BLOCK) void k_mvn_matvec(ModelDev md, ArenaDev A, EvalIO io, int j, int fold, int d, double Emax,
                                                        int max_depth, HostStatus* st) {
  int row = (int)blockIdx.x;
  if (fold) {
    if (row == 0) { control_lean(md, A, io, j - 1, d, Emax, max_depth, st, 0, lean_src(md, 0)); return; }
    --row;
  }
  const MvnDev& mv = md.mv;
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  __shared__ double s_w[MVN_BLOCK / WAVE];
  const double* __restrict__ q = qv.q + mv.off;
  const double* __restrict__ mu = mv.mu;
  const int tid = threadIdx.x;
  if (row >= mv.k) return;
  const double* __restrict__ pr = mv.prec + (int64_t)row * mv.k;
  double s = 0.0;
  const int k2 = mv.k & ~1;
#pragma unroll 4
  for (int c = 2 * tid; c < k2; c += 2 * MVN_BLOCK) {
    const double2 p = *reinterpret_cast<const double2*>(pr + c);
    s = fma(p.x, q[c] - mu[c], s);
    s = fma(p.y, q[c + 1] - mu[c + 1], s);
  }
  if (tid == 0 && (mv.k & 1)) s = fma(pr[mv.k - 1], q[mv.k - 1] - mu[mv.k - 1], s);
  s = wave_sum(s);
  if ((tid & (WAVE - 1)) == 0) s_w[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < MVN_BLOCK / WAVE; ++w) t += s_w[w];
    mv.gdense[mv.off + row] = -t;
    mv.rowq[row] = (q[row] - mu[row]) * t;
  }
}

// Row-aligned pass for models that ARE one MvNormal node (C3): the second kernel of the leapfrog (k_vector: 8 workgroups, pure
// latency -- as long as the mat-vec itself, profiles/r02j_profile_c3_mvn2048.txt) disappears.  The workgroup that produced
// (P delta)_i for its R rows finishes those R elements in its own tail: gradient, second kick, v' = M^-1 p', the tree-merge dot
// products of `leaf_post` restricted to its elements, and the first half of the NEXT leaf (the position is always materialised
// for this node).  What crosses workgroups is one record per workgroup [logp share, dots], stored slot-major with plain
// stores; it is read after the kernel boundary by the control work (`mva_control`: workgroup 0 of the next leaf's launch, or a
// launch of its own behind the last leaf of a doubling), which sums the records in workgroup order -- fixed order, no
// floating-point atomics, and no hand-off inside a launch (a ticket + write-through records per workgroup, as the row pass of
// the logit node does it, cost 16 us here: the kernel is too short to hide them; measured).  Records are double-buffered by
// launch parity: the control work of leaf j rides in leaf j+1's launch, whose rows are already writing theirs.
#define MVA_RS 80   // doubles per record: logp share, p'.v', six dots per merge level (<= MAX_LEVELS), the six of `extend`
#define MVA_SW 16   // slots per load window (a wave's load covers 4 records x 16 slots)
#define MVA_U 32    // loads in flight per lane
__device__ __forceinline__ void mva_control(const ModelDev& md, const ArenaDev& A, const EvalIO& io, int j, int d, double Emax, int max_depth,
                                            HostStatus* st, int seq, int par, int nt, const double* al_part) {
  // (`al_part`: the records of THIS chain -- md.mv.al_part for a chain alone on its model, the chain's own buffer when several
  // chains of a group share one launch, mvn_multi_kernel.h)
  const MvnDev& mv = md.mv;
  __shared__ double s_rec[PART_STRIDE];
  __shared__ double s_wp[VEC_THREADS / WAVE][NDOT + 1];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6, NT = nt;   // (`nt` threads of the workgroup take part)
  const bool leaf = io.mode != MODE_PLAIN, tree = io.mode == MODE_TREE;
  int m = 0;
  bool last = false;
  if (tree) {
    while (((j >> m) & 1) && m < d) ++m;
    last = (j + 1 == (1 << d));
  }
  const int nn = 1 + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
  auto need_slot = [&](int qq) {
    if (qq < 1) return PART_LP;
    if (qq < 2 + 6 * m) return PART_DOT + (qq - 1);
    return PART_DOT + DOT_TOP + (qq - 2 - 6 * m);
  };
  const int nwg = mv.al_nwg;
  const double* rec = al_part + (int64_t)par * MVA_RS * nwg;
  // the uniforms this leaf may consume: the cursor is read straight from the control block with the very first loads
  UniPrefetch upf;
  uni_prefetch_none(upf);
  int cur0 = 0;
  if (tree && tid == 0) cur0 = A.ctl->cursor;
  // Records are RECORD-major and compact (slot qq of a record = the qq-th number this leaf needs): a wave reads MVA_SW slots of
  // four records with one coalesced load, lane = (record mod 4, slot), and keeps MVA_U such loads in flight -- the 512 records
  // of C3 cost every lane ONE trip to memory however many slots the leaf needs up to 16 (m <= 2: 7 of 8 leaves), where the
  // slot-major layout with a wave sum per slot cost a trip plus sixteen DPP trees per 16 slots.  Fixed order: a lane adds its records in
  // index order, the four record classes pair up (0 + 1) + (2 + 3), the waves are added in wave order.
  // (addresses = a wave-uniform base per load + ONE per-lane offset: 64 registers of data in flight, not 64 more of addresses --
  // the row waves of this kernel need the occupancy)
  constexpr int NW = VEC_THREADS / WAVE;
  const int rsub = lane >> 4, sl = lane & (MVA_SW - 1), wv = __builtin_amdgcn_readfirstlane(w);
  for (int win = 0; win * MVA_SW < nn; ++win) {
    const int qq = win * MVA_SW + sl;
    const bool on = qq < nn;
    const unsigned loff = (unsigned)(rsub * MVA_RS + qq);
    double acc = 0.0;
    for (int ib = 4 * wv; ib < nwg; ib += 4 * NW * MVA_U) {
      double v[MVA_U];
#pragma unroll
      for (int u = 0; u < MVA_U; ++u) {
        const int i = ib + 4 * NW * u;                          // (uniform)
        const double* bu = rec + (int64_t)min(i, nwg - 1) * MVA_RS;
        v[u] = (on && i + rsub < nwg) ? bu[loff] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < MVA_U; ++u) acc += v[u];
    }
    acc += __shfl_xor(acc, 16, WAVE);
    acc += __shfl_xor(acc, 32, WAVE);
    if (rsub == 0 && on) s_wp[w][qq] = acc;
  }
  if (tree && tid == 0) uni_prefetch(A, cur0, m, last, upf);
  __syncthreads();
  for (int qq = tid; qq < nn; qq += NT) {
    double t = 0.0;
    for (int ww = 0; ww < NW; ++ww) t += s_wp[ww][qq];
    s_rec[need_slot(qq)] = t;
  }
  __syncthreads();
  control_lean<false, 8, false>(md, A, io, j, d, Emax, max_depth, st, seq, LeanSrc{s_rec, PART_STRIDE, 1, md.def_loc}, nt, tree, upf);
}

__global__ __launch_bounds__(VEC_THREADS) void k_mva_control(ModelDev md, ArenaDev A, EvalIO io, int j, int d, double Emax, int max_depth,
                                                            HostStatus* st, int seq, int par) {
  mva_control(md, A, io, j, d, Emax, max_depth, st, seq, par, VEC_THREADS, md.mv.al_part);
}

// `fold`: workgroup 0 does the control work of the leaf of the PREVIOUS row-aligned launch -- leaf (cio, cj, cd): normally leaf
// j - 1 of this doubling (cseq = 0), or the last leaf of the previous doubling when this launch was queued by the look-ahead
// right behind it (cseq = that doubling's sequence number: it publishes the status word the host waits for).
//
// Five waves: four stream the rows; the fifth requests, while they stream, everything of the workgroup's R elements whose
// address is known at launch (p_half, the operands of the first merge levels: written by the previous launch on other XCDs,
// i.e. misses) and finishes the elements when the dot products arrive.  Loads return in order within a wave: with the requests
// in a streaming wave, that wave's first fma waited behind every one of those misses.
#define MVA_THREADS (MVN_BLOCK + WAVE)
template <int R>
__global__ __launch_bounds__(MVA_THREADS) void k_mvn_aligned(ModelDev md, ArenaDev A, EvalIO io, int j, int fold, int d, double Emax,
                                                           int max_depth, HostStatus* st, int par, EvalIO cio, int cj, int cd, int cseq) {
  const MvnDev& mv = md.mv;
  // Workgroup 0 is the control workgroup in EVERY launch (idle when there is nothing to fold): rows [R (b - 1), R b) then always
  // belong to workgroup b, hence to the same XCD, whose L2 keeps its eighth of P from launch to launch.
  if (blockIdx.x == 0) {
    if (!fold || threadIdx.x >= VEC_THREADS) return;   // (the control code is written for VEC_THREADS threads)
    mva_control(md, A, cio, cj, cd, Emax, max_depth, st, cseq, par ^ 1, VEC_THREADS, mv.al_part);
    return;
  }
  const int b = (int)blockIdx.x - 1;
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  __shared__ double s_w[R][MVN_BLOCK / WAVE];
  constexpr int TW = MVN_BLOCK / WAVE;      // index of the tail wave: leaf_post files a wave's sums under its index
  __shared__ double s_red[NDOT + TW];
  const double* __restrict__ q = qv.q;
  const double* __restrict__ mu = mv.mu;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6;
  const int K = mv.k, row0 = b * R;
  const bool leaf = io.mode != MODE_PLAIN, tree = io.mode == MODE_TREE;
  const bool tailwave = w == TW;
  const int my = min(row0 + min(lane, R - 1), K - 1);   // tail wave, lane l < R: element row0 + l
  MergePrefetch mpf;
  double phv = 0.0, qr = 0.0, mur = 0.0, var_r = 0.0;
  if (tailwave) {
    if (leaf) { phv = A.P[lf.d_o + my]; var_r = A.var[my]; }
    qr = q[my]; mur = mu[my];
    if (tree) merge_prefetch(A, lf, j, my, mpf);
  } else {
    const double* pr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = mv.prec + (int64_t)min(row0 + r, K - 1) * K;
    double s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] = 0.0;
    const int k2 = K & ~1;
#pragma unroll 4
    for (int c = 2 * tid; c < k2; c += 2 * MVN_BLOCK) {
      const double d0 = q[c] - mu[c], d1 = q[c + 1] - mu[c + 1];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double2 p = *reinterpret_cast<const double2*>(pr[r] + c);
        s[r] = fma(p.x, d0, s[r]);
        s[r] = fma(p.y, d1, s[r]);
      }
    }
    if (tid == 0 && (K & 1)) {
      const double dl = q[K - 1] - mu[K - 1];
#pragma unroll
      for (int r = 0; r < R; ++r) s[r] = fma(pr[r][K - 1], dl, s[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double t = wave_sum(s[r]);
      if (lane == 0) s_w[r][w] = t;
    }
  }
  __syncthreads();
  if (!tailwave) return;
  // ---- the tail wave: the R elements of this workgroup, lane = element ----
  const bool a0 = lane < R && row0 + lane < K;
  double t = 0.0;
#pragma unroll
  for (int ww = 0; ww < MVN_BLOCK / WAVE; ++ww) t += s_w[min(lane, R - 1)][ww];
  int idx[1] = {my};
  bool act[1] = {a0};
  double grad[1] = {-t}, ph[1] = {phv};
  if (a0) {
    if (leaf) A.G[lf.d_o + my] = -t;
    else io.grad[my] = -t;
  }
  int m = 0; bool last = false;
  // (only lanes 0 .. R - 1 hold elements: for R <= 8 the wave sums stop after three DPP steps -- the same bits, device_math.h wave_sum8)
  if (leaf) leaf_post<1, false, R <= 8>(A, lf, j, d, tree, idx, act, grad, ph, s_red, 1, m, last, tree ? &mpf : nullptr);   // -> s_red[k + TW]
  if (leaf && io.pre_next == 3 && a0) {
    // last leaf of a doubling, the next one (queued behind it) grows on the OTHER side: first half of its first leaf from the tree's
    // other edge state -- k_leaf_pre's arithmetic (signed step of the other direction) on this workgroup's rows
    const int e = lf.dir > 0 ? lf.left : lf.right;
    const int64_t eo = slot_off(A, e), no = slot_off(A, e - lf.dir);
    const double eps2 = -lf.eps, half2 = 0.5 * eps2;
    const double ph2 = fma(half2, A.G[eo + my], A.P[eo + my]);
    A.P[no + my] = ph2;
    A.Q[no + my] = fma(eps2, var_r * ph2, A.Q[eo + my]);
  } else if (leaf && io.pre_next && a0) {   // first half of the next leaf (integration.py:118-127): the arithmetic of k_leaf_pre
    const int64_t no = slot_off(A, lf.t + lf.dir);
    const double p = fma(lf.half, -t, phv);   // p' of this leaf, as leaf_post computed it
    const double phn = fma(lf.half, -t, p);
    A.P[no + my] = phn;
    A.Q[no + my] = fma(lf.eps, var_r * phn, qr);
  }
  const double lp = (R <= 8 ? wave_sum8(a0 ? -0.5 * (qr - mur) * t : 0.0) : wave_sum(a0 ? -0.5 * (qr - mur) * t : 0.0));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (s_red was written by lane 0 of this wave inside leaf_post)
  __builtin_amdgcn_wave_barrier();
  // ---- this workgroup's record: compact (the numbers this leaf needs, in mva_control's order), one coalesced store ----
  const int nwg = mv.al_nwg;
  double* rec = mv.al_part + ((int64_t)par * nwg + b) * MVA_RS;
  const int nn = 1 + (leaf ? 1 + 6 * m + (last ? 6 : 0) : 0);
  for (int qq = lane; qq < nn; qq += WAVE) {
    const int k = qq < 2 + 6 * m ? max(qq - 1, 0) : DOT_TOP + (qq - 2 - 6 * m);
    rec[qq] = qq == 0 ? lp : s_red[k + TW];
  }
}

// "cholesky" solver of the MvNormal node: delta = q - mu before the two mat-vecs with W = chol(cov)^-1 (k_dense_mv), and the
// node's outputs from P delta = W^T (W delta) after them
__global__ __launch_bounds__(256) void k_mvn_delta(MvnDev mv, ArenaDev A, EvalIO io, int j) {
  Leaf lf; QView qv;
  if (load_aborted(io, A)) return;
  resolve_leaf(io, A, j, lf, qv);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < mv.k) mv.wy[i] = qv.q[mv.off + i] - mv.mu[i];
}
__global__ __launch_bounds__(256) void k_mvn_finish(MvnDev mv, ArenaDev A, EvalIO io) {
  if (load_aborted(io, A)) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < mv.k) {
    const double t = mv.wy[2 * mv.k + i];
    mv.gdense[mv.off + i] = -t;
    mv.rowq[i] = mv.wy[i] * t;
  }
}

// ---------------------------------------------------------------------------
// dense mass matrix: energy + tree logic of one leaf (one workgroup), after k_tree_vec
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(VEC_THREADS) void k_tree_ctl(ModelDev md, ArenaDev A, EvalIO io, int j, int d, double Emax,
                                                         int max_depth, HostStatus* st, int seq) {
  Leaf lf; QView qv;
  resolve_leaf(io, A, j, lf, qv);
  __shared__ double s_dot[NDOT];
  __shared__ double s_chunk[CTL_CHUNKS][NDOT];
  __shared__ Ctl s_ctl;
  const int tid = threadIdx.x;
  const bool tree = io.mode == MODE_TREE;
  int m = 0;
  bool last = false;
  if (tree) {
    while (((j >> m) & 1) && m < d) ++m;
    last = (j + 1 == (1 << d));
  }
  if (tid < (int)(sizeof(Ctl) / sizeof(int))) reinterpret_cast<int*>(&s_ctl)[tid] = reinterpret_cast<const int*>(A.ctl)[tid];
  const int nn = 1 + 6 * m + (last ? 6 : 0);
  auto need_dot = [&](int q) { return q < 1 + 6 * m ? q : DOT_TOP + (q - 1 - 6 * m); };
  {
    const int per = (md.nblk + CTL_CHUNKS - 1) / CTL_CHUNKS;
    for (int t = tid; t < nn * CTL_CHUNKS; t += VEC_THREADS) {
      const int c = t % CTL_CHUNKS, k = need_dot(t / CTL_CHUNKS);
      const int b0 = c * per, b1 = min(md.nblk, (c + 1) * per);
      s_chunk[c][k] = sum_strided(md.part + PART_DOT + k, md.part_stride, b0, b1);
    }
  }
  const int ts = lf.t & (A.S - 1);
  const double logp = A.LOGP[ts];
  __syncthreads();
  if (tree && s_ctl.aborted) {   // terminated earlier in this doubling: drain
    if (tid == 0 && st) publish_status(&s_ctl, st, seq);
    return;
  }
  for (int t = tid; t < nn; t += VEC_THREADS) {
    const int k = need_dot(t);
    double sacc = 0.0;
#pragma unroll
    for (int c = 0; c < CTL_CHUNKS; ++c) sacc += s_chunk[c][k];
    s_dot[k] = sacc;
  }
  __syncthreads();
  if (tid != 0) return;
  const double E = (A.kin_user ? *A.kin_user : 0.5 * s_dot[0]) - logp;  // integration.py:133-134
  A.E[ts] = E;
  if (!tree) return;
  tree_decide(&s_ctl, A, lf, s_dot, E, m, last, Emax, max_depth, uni_view_none());
  *A.ctl = s_ctl;
  if (st) publish_status(&s_ctl, st, seq);
}

// ---------------------------------------------------------------------------
// start / end of a draw
// ---------------------------------------------------------------------------

// p0 = z / sigma, v0 = var * p0, PSUM = p0 (base_hmc.py:201-202, quadpotential.py:323-326)
// `q_src` / `g_src` (optional): position and gradient of the start state when it is the previous draw's proposal
// (the values the model pass would reproduce bit for bit; the reference recomputes them, base_hmc.py:202)
__global__ __launch_bounds__(VEC_THREADS) void k_draw_start(ArenaDev A, const double* __restrict__ normals,
                                                            const double* __restrict__ p_exact, double* __restrict__ kin_part,
                                                            const double* __restrict__ q_src, const double* __restrict__ g_src,
                                                            int dense) {
  __shared__ double sm[VEC_THREADS / WAVE];
  double kin = 0.0;
  const int base = blockIdx.x * VEC_THREADS * A.ept;
  for (int e = 0; e < A.ept; ++e) {
    const int i = base + e * VEC_THREADS + threadIdx.x;
    if (i < A.n) {
      double p, v;
      if (dense) { p = A.P[i]; v = A.V[i]; }   // p0 = W z and v0 = C p0 were produced by k_dense_mv
      else {
        p = p_exact ? p_exact[i] : normals[i] * A.inv_stds[i];
        v = A.var[i] * p;
        A.P[i] = p; A.V[i] = v;
      }
      A.PSUM[i] = p;
      if (q_src) { A.Q[i] = q_src[i]; A.G[i] = g_src[i]; }
      kin = fma(p, v, kin);
    }
  }
  const double tot = block_sum<false>(kin, sm);
  if (threadIdx.x == 0) kin_part[blockIdx.x] = tot;
}

// gather the proposal and tree statistics (nuts.py:478-489)
struct DrawOut {
  double energy, logp, E0, log_accept_sum, max_energy_change, div_dE;
  int depth, n_proposals, proposal, cursor, turning, diverging, bad_energy, div_t;
};

__global__ void k_draw_ctl_start(ArenaDev A, const double* __restrict__ kin_part, double step_size, int dir_forced, int max_depth,
                                 HostStatus* st, int use_cached_logp, double cached_logp, const DrawOut* prev) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < A.nblk; ++b) s += kin_part[b];
    Ctl* c = A.ctl;
    if (use_cached_logp) A.LOGP[0] = prev ? prev->logp : cached_logp;   // (`prev`: the record of the previous draw, still on the device)
    const double logp = A.LOGP[0];
    const double E = (A.kin_user ? *A.kin_user : 0.5 * s) - logp;  // integration.py:72-74
    A.E[0] = E;
    c->E0 = E;
    c->log_size = 0.0;
    c->log_accept_sum = -INFINITY;
    c->max_energy_change = 0.0;
    c->div_dE = 0.0;
    c->n_proposals = 0; c->depth = 0; c->left = 0; c->right = 0; c->proposal = 0; c->cursor = 0;
    c->turning = 0; c->diverging = 0;
    c->bad_energy = !isfinite(E);
    c->aborted = c->bad_energy;
    c->eps_abs = step_size;
    c->n_leaves_total = 0;
    c->dir = 1; c->edge = 0; c->eps = step_size;
    if (dir_forced != 0) { c->dir = dir_forced; c->eps = dir_forced > 0 ? step_size : -step_size; }
    else if (!c->aborted && max_depth > 0) ctl_next_direction(c, A.uniforms);
    if (st) publish_fields(c, st);
  }
  if (A.ga_sync && threadIdx.x < GA_SYNC_WORDS) A.ga_sync[threadIdx.x] = 0u;
}


// `trace_q` (optional): the proposal's position also goes into this row of a device-resident trace buffer (multi-draw
// calls); `mapped` (optional): the record is also written to pinned, device-mapped host memory and `seq` is published
// behind it with a system-scope store, so the host learns the outcome of the draw without a stream synchronisation.
struct DrawOutMapped { DrawOut o; unsigned long long seq; };

__global__ __launch_bounds__(VEC_THREADS) void k_draw_finish(ArenaDev A, double* __restrict__ q_out,
                                                             double* __restrict__ g_out, DrawOut* out,
                                                             double* __restrict__ trace_q, DrawOutMapped* mapped, unsigned seq) {
  const Ctl* c = A.ctl;
  const int prop = c->proposal;
  const int64_t po = slot_off(A, prop);
  // a launch that was in flight while the tree stopped may have left some arrival counters of the group-aligned row pass
  // half-way (the `aborted` flag flips under it): every draw ends with the counters at zero
  if (blockIdx.x == 0) for (int t = threadIdx.x; t < A.ga_nticket; t += blockDim.x) A.ga_ticket[t] = 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) {
    const double qi = A.Q[po + i];
    q_out[i] = qi;
    g_out[i] = A.G[po + i];
    if (trace_q) trace_q[i] = qi;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int ps = prop & (A.S - 1);
    DrawOut o;
    o.energy = A.E[ps]; o.logp = A.LOGP[ps]; o.E0 = c->E0;
    o.log_accept_sum = c->log_accept_sum; o.max_energy_change = c->max_energy_change; o.div_dE = c->div_dE;
    o.depth = c->depth; o.n_proposals = c->n_proposals; o.proposal = prop; o.cursor = c->cursor;
    o.turning = c->turning; o.diverging = c->diverging; o.bad_energy = c->bad_energy; o.div_t = c->div_t;
    *out = o;
    if (mapped) {
      mapped->o = o;
      __threadfence_system();
      __hip_atomic_store(&mapped->seq, (unsigned long long)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- mass-matrix adaptation (quadpotential.py:328-355, 431-437) ----
//   flags bit0: add sample to fg and bg;  bit1: var = clip(fg.m2 / fg.count)
__global__ __launch_bounds__(VEC_THREADS) void k_potential_update(int n, const double* __restrict__ x,
                                                                  double* fg_mean, double* fg_m2, double fg_count_new,
                                                                  double* bg_mean, double* bg_m2, double bg_count_new,
                                                                  double* var, double* stds, double* inv_stds, int flags) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double m2 = fg_m2[i];
    if (flags & 1) {
      const double xi = x[i];
      double mean = fg_mean[i];
      double od = xi - mean;
      mean += od / fg_count_new;
      m2 += od * (xi - mean);
      fg_mean[i] = mean; fg_m2[i] = m2;
      double bm = bg_mean[i], b2 = bg_m2[i];
      od = xi - bm;
      bm += od / bg_count_new;
      b2 += od * (xi - bm);
      bg_mean[i] = bm; bg_m2[i] = b2;
    }
    if (flags & 2) {
      double v = m2 / fg_count_new;
      v = fmin(fmax(v, 1e-12), 1e12);   // np.clip(var, 1e-12, 1e12)
      if (isnan(m2 / fg_count_new)) v = m2 / fg_count_new;
      const double s = sqrt(v);
      var[i] = v; stds[i] = s; inv_stds[i] = 1.0 / s;
    }
  }
}

// ---- QuadPotentialDiagAdaptExp.update (quadpotential.py:534-579) with `_ExpWeightedVariance.add_sample` (:466-470) ----
//   flags bit0: start the estimators at this sample (mean = x, variance = 0)   bit1: add the sample   bit2: new variance
// Every operation is a separate IEEE operation in NumPy's order (`__d*_rn`: no contraction into fma), so the device estimator is
// bit for bit the host's -- which is what lets the reference-run fixture of `init="jitter+adapt_diag_grad"` pass unchanged.
__global__ __launch_bounds__(VEC_THREADS) void k_potential_update_exp(int n, const double* __restrict__ x, const double* __restrict__ g,
                                                                      double* ms, double* vs, double* mg, double* vg, double alpha,
                                                                      double one_m_alpha, int use_grads, double* var, double* stds,
                                                                      double* inv_stds, int flags) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (flags & 1) {
      ms[i] = x[i]; vs[i] = 0.0;
      if (use_grads) { mg[i] = g[i]; vg[i] = 0.0; }
    }
    if (flags & 2) {
      {
        const double delta = __dsub_rn(x[i], ms[i]);
        ms[i] = __dadd_rn(ms[i], __dmul_rn(alpha, delta));                                              // mean += alpha * delta
        vs[i] = __dmul_rn(one_m_alpha, __dadd_rn(vs[i], __dmul_rn(alpha, __dmul_rn(delta, delta))));    // (1 - a) (var + a delta**2)
      }
      if (use_grads) {
        const double delta = __dsub_rn(g[i], mg[i]);
        mg[i] = __dadd_rn(mg[i], __dmul_rn(alpha, delta));
        vg[i] = __dmul_rn(one_m_alpha, __dadd_rn(vg[i], __dmul_rn(alpha, __dmul_rn(delta, delta))));
      }
    }
    if (flags & 4) {
      double v;
      if (use_grads) v = __dsqrt_rn(__ddiv_rn(vs[i], vg[i]));        // _update_from_variances: sqrt(var / inv_var), not clipped
      else { v = vs[i]; v = fmin(fmax(v, 1e-12), 1e12); if (isnan(vs[i])) v = vs[i]; }   // _update_from_weightvar: np.clip
      const double sd = __dsqrt_rn(v);
      var[i] = v; stds[i] = sd; inv_stds[i] = __ddiv_rn(1.0, sd);
    }
  }
}

#include "rows_ga_kernel.h"
#include "rows_ga_tree.h"
#include "rows_gb_kernel.h"
#include "dense_adapt.h"
#include "glm_kernel.h"
#include "mvn_multi_kernel.h"
#include "mvn_mfma_kernel.h"
#include "rows_ga_multi_kernel.h"
#include "rows_gal_kernel.h"
#include "rows_gb_multi_kernel.h"
#include "lin_kernel.h"

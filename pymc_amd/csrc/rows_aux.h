// Auxiliary workgroups of the one-launch row passes (rows_ga_kernel.h, rows_gb_kernel.h): everything of the model that is NOT the
// row stream and the per-group finish of the z elements.
//
// The one-launch passes used to evaluate exactly one model in closed form -- mu ~ Normal, sigma ~ HalfNormal, z ~ Normal with
// constant parameters (ga_def_local) -- and every other hierarchical-logit model fell back to the two- or three-launch paths.  The
// reference differentiates any sum of factors (model/core.py:612-695): a HalfCauchy / Exponential / LogNormal / Gamma hyper-prior
// (continuous.py:2383-2390, 1478-1486, 1807-1819), parameters that are data vectors, further scalar or vector variables with
// priors and observed factors of their own.  Here such a model keeps the row stream and the per-group finish as they are, and
// ga_naux EXTRA workgroups of the same launch run the element-wise interpreter (model_dev.h: gather_element, the same code kernel
// B runs) over every element that is not a z element -- thread t of auxiliary workgroup a owns element number a * blockDim + t of
// that list:
//
//   * ordinary elements (vector variables): first half of the leapfrog, transform, gradient as a gather over the variable's
//     factors, second half kick, v' = M^-1 p', q' store, the tree-merge dot products of `leaf_post`;
//   * deferred elements (mu / sigma of the logit node, whose gradient needs the cross-workgroup sums of the row pass; scalars): the
//     local part {d logp / dx, dx / dq, dlog|J| / dq, p_half} into `def_loc` -- what kernel B leaves on the lean path and what
//     ga_def_local leaves in closed form -- finished by the control work (control_lean) as before;
//   * factors without an owning variable (data only): their logp, grid-stride over the auxiliary threads.
//
// Each auxiliary workgroup leaves ONE record {logp share, dots} behind the ga_nblk block partials of the launch (ga_nrec =
// ga_nblk + ga_naux records: lean_src), written with plain stores -- the kernel boundary publishes it, the next launch's prologue
// and control work total it with the block partials in record order.  Nothing of it is on the launch's critical path: the
// auxiliary workgroups are done microseconds into a stream that takes tens, and what they produce is read after the boundary.
//
// A launch runs next to the control work of the PREVIOUS leaf (workgroup 0, GA_FOLD_CTL), which is still writing the source
// state's gradient and momentum of the deferred elements.  So the auxiliary workgroups compose q' and p_half of every deferred
// element themselves -- mu / sigma by the prologue every row workgroup runs (the caller passes them in), the others from the
// previous launch's `def_loc` (their cross-workgroup share is zero) -- into an LDS table the interpreter reads through
// QView::defq whenever a factor refers to such an element.
//
// No expression programs / gathers here (the PROG = false interpreter: a model that carries them stays on the general path), and
// the register budget is the row pass's (128 VGPRs on the group-aligned pass): tools/kernel_resources.py shows no scratch.
#pragma once

#define GA_AUX_MAXDEF 64   // deferred elements of a model on a one-launch pass (one LDS slot each; also: one control thread each)
#define GA_AUX_PROG_LDS 6144   // LDS the row kernels set aside for the interpreter's tables (a model whose tables are larger reads them from
                               // global memory: correct, but every table look-up is then a dependent trip to L2, and at ~60 us the auxiliary
                               // workgroup of a 300-element extra variable outlasted the 57 us row stream of C2-L: profiles/r04c_variants_ab.txt)

struct AuxScratch {   // LDS the caller lends (aliased onto arrays the row workgroups use for something else)
  double* red;        // [NDOT * waves]
  double* defq;       // [GA_AUX_MAXDEF]
  double* defph;      // [GA_AUX_MAXDEF]
  double* w;          // [waves]
};
#define GA_AUX_SCRATCH_DOUBLES(NW) (NDOT * (NW) + 2 * GA_AUX_MAXDEF + (NW))

// (included by rows_ga_kernel.h behind the definition of GaArgs)
// hval0 / hph0: lane l of every wave holds q' and p_half of hyper-parameter element l mod 2D (mu[0..D), sigma[0..D)) of THIS leaf.
// rec / rs: slot k of this workgroup's record is rec[k * rs] (record-major: rs = 1; slot-major: rs = padded record count).
// `ap`: the kernel's argument struct where it lies (the kernarg segment).  `lds`: GA_AUX_SCRATCH_DOUBLES(waves) doubles of LDS.
//
// ONE out-of-line function for every instantiation of both row passes: the interpreter's code (tens of thousands of instructions
// once gather_element and leaf_post are inlined) exists once in the library, and its register allocation -- spills included, it is
// off the critical path -- is its own: the streaming kernels that call it keep the code and the registers they had without it.
// BUDGET: one copy per register budget of the callers (the compiler gives an out-of-line function the LOOSEST limit among its
// callers: a copy shared by the 128-register streaming kernel and the 256-register group-block kernel took 248 and halved the
// occupancy of the former) -- 4 / 3: group-aligned pass at that many waves per SIMD, 0: group-block pass.
template <int BUDGET>
__device__ __noinline__ void ga_aux(const GaArgs* ap, int aux_id, double hval0, double hph0, double* lds, int nw_lds, char* lds_prog, double* rec, int64_t rs) {
  const GaArgs& a = *ap;
  const AuxScratch sc{lds, lds + NDOT * nw_lds, lds + NDOT * nw_lds + GA_AUX_MAXDEF, lds + NDOT * nw_lds + 2 * GA_AUX_MAXDEF};
  const ModelDev& md = a.md;
  const ArenaDev& A = a.A;
  const EvalIO& io = a.io;
  const RowsDev& R = md.lg;
  const int D = R.D;
  const int j = a.j, d = a.d, par = a.par;
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid >> 6, NT = (int)blockDim.x, NW = NT >> 6;
  Leaf lf; QView qv;
  const int aborted = load_aborted(io, A);
  resolve_leaf(io, A, j, lf, qv);
  const bool leaf = io.mode != MODE_PLAIN, tree = io.mode == MODE_TREE;
  // The auxiliary workgroup is a chain of dependent trips to memory (tables -> element -> operands -> merge operands), and under a
  // row stream that saturates HBM every trip costs microseconds: the tables go to LDS first (constant data, L2 hits).  Measured
  // and declined (profiles/r04e_variants_ab.txt ... r04g): requesting every thread's source state and merge operands up front
  // (most threads of a small model own a deferred element or nothing, and the tables then waited behind loads nobody needed), and
  // prefetching the operands of the first three merge levels as the row workgroups do (18 far loads per element and leaf, where
  // half the leaves complete no merge at all: +4.7 us per launch on a 300-element extra variable at C2-L).
  const int ai = aux_id * NT + tid;
  const bool valid = ai < R.ga_auxel;
  const int i = valid ? (ai < R.off_z ? ai : ai + R.G * D) : 0;   // (the z elements are the row workgroups')
  // (loads return in order: first what is needed first -- the tables and the deferred list, constant data that sits in L2 --
  // then, in one batch, everything that was written by the previous launch on other XCDs and comes from far away)
  const bool has_def = tid < md.n_deferred;                      // (n_deferred <= GA_AUX_MAXDEF <= the threads of a workgroup)
  int def_i = 0, def_k = 0;
  if (has_def) { def_i = md.deferred_g[2 * tid]; def_k = md.deferred_g[2 * tid + 1]; }
  const bool prog_in_lds = md.prog_bytes <= GA_AUX_PROG_LDS;
  if (prog_in_lds) {
    const int n16 = (md.prog_bytes + 15) >> 4;
    for (int t = tid; t < n16; t += NT) reinterpret_cast<uint4*>(lds_prog)[t] = reinterpret_cast<const uint4*>(md.prog)[t];
  }
  const Prog pg = prog_view(md, prog_in_lds ? lds_prog : md.prog);   // (first used behind the barrier below)

  // ---- q' and p_half of every deferred element of this leaf -> LDS ----
  const bool from_prev = (a.fold & GA_FOLD_SRC) != 0;
  const double2* prev_loc = reinterpret_cast<const double2*>(md.def_loc + (int64_t)(par ^ 1) * 4 * MAX_DEFERRED);
  if (has_def && def_k != R.var_mu && def_k != R.var_sigma) {   // (mu / sigma below: the caller's prologue values)
    double qn, ph;
    if (from_prev) {
      // the source state is the leaf of the previous launch: its gradient is the local part that launch left (no cross-workgroup
      // share), its p' the second half kick of it -- the arithmetic of control_lean / rows_hyper_fold_elem, the same bits
      const double2 l01 = prev_loc[2 * tid], l23 = prev_loc[2 * tid + 1];
      const double g = deferred_finish(l01.x, 0.0, l01.y, l23.x);
      const double p_src = fma(qv.half, g, l23.y);
      ph = fma(qv.half, g, p_src);
      qn = fma(qv.eps, qv.var[def_i] * ph, qv.q[def_i]);
    } else if (qv.composed) { ph = qv.p_half(def_i); qn = fma(qv.eps, qv.var[def_i] * ph, qv.q[def_i]); }
    else { ph = 0.0; qn = qv.q[def_i]; }
    sc.defq[tid] = qn; sc.defph[tid] = ph;
  }
  if (w == 0 && lane < 2 * D) {
    const int slot = lane < D ? R.def_mu + lane : R.def_sigma + (lane - D);
    sc.defq[slot] = hval0; sc.defph[slot] = hph0;
  }
  __syncthreads();
  if (aborted) return;   // (a launch queued behind a tree that has terminated: nothing may be written)
  QView qd = qv;
  qd.defq = sc.defq;

  // ---- this thread's element ----
  int idx[1] = {i};
  bool act[1] = {false};
  double grad[1] = {0.0}, ph[1] = {0.0};
  double lp = 0.0;
  if (valid) {
    const int k = find_var(pg, i);
    const VarDev v = pg.vars[k];
    double qn;
    if (v.deferred) { const int slot = v.def_base + (i - v.offset); qn = sc.defq[slot]; ph[0] = sc.defph[slot]; }
    else if (leaf) {
      const double own_g = A.G[lf.so + i], own_p = A.P[lf.so + i], own_q = A.Q[lf.so + i], own_var = A.var[i];
      ph[0] = fma(lf.half, own_g, own_p); qn = fma(lf.eps, own_var * ph[0], own_q);   // integration.py:118-127
    } else qn = io.q[i];
    double x, dxdq, lj, dj, gx = 0.0;
    transform_full(v, qn, x, dxdq, lj, dj);
    lp += lj;
    gather_element<false, true>(pg, qd, k, i - v.offset, x, gx, lp, nullptr, 0);   // (lean path: no broadcast terms, the accumulators are never touched)
    if (leaf) A.Q[lf.d_o + i] = qn;
    if (v.deferred) {
      double2* loc = reinterpret_cast<double2*>(md.def_loc + (int64_t)par * 4 * MAX_DEFERRED) + 2 * (v.def_base + (i - v.offset));
      loc[0] = make_double2(gx, dxdq);
      loc[1] = make_double2(dj, ph[0]);
    } else {
      grad[0] = gx * dxdq + dj;
      act[0] = true;
      if (leaf) A.G[lf.d_o + i] = grad[0];
      else io.grad[i] = grad[0];
    }
  }
  // factors without an owning variable (data only on this path): their logp, grid-stride over the auxiliary threads
  for (int o = 0; o < md.n_orphans; ++o) {
    const int fi = md.orphans[o];
    const int fsize = pg.factors[fi].size;
    for (int li = ai; li < fsize; li += R.ga_naux * NT) lp += orphan_element<false, true>(pg, qd, fi, li, nullptr, 0);
  }

  // ---- second half kick + tree-merge dot products of the ordinary elements, then this workgroup's record ----
  int m = 0; bool last = false;
  if (leaf) leaf_post<1>(A, lf, j, d, tree, idx, act, grad, ph, sc.red, NW, m, last);
  {
    const double s = wave_sum(lp);
    if (lane == 0) sc.w[w] = s;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int ww = 0; ww < NW; ++ww) t += sc.w[ww];
    rec[PART_LP * rs] = t;
  }
  for (int k = tid; k < 2 * LOGIT_MAXD; k += NT) rec[(PART_DMU + k) * rs] = 0.0;   // (no share in d logp / d mu, d sigma of the logit node)
  if (leaf) {
    for (int k = tid; k < NDOT; k += NT) {
      if (!dot_needed(k, m, last)) continue;
      double r = 0.0;
      for (int ww = 0; ww < NW; ++ww) r += sc.red[k * NW + ww];
      rec[(PART_DOT + k) * rs] = r;
    }
  }
}

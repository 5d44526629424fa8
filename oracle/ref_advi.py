"""CPU oracle of full-rank minibatch ADVI on a GLM (TEST INFRASTRUCTURE; see oracle/__init__.py).

Pinned by EXECUTING the reference: tests/golden/refrun_advi.py loads `FullRankGroup`, the normalised terms of `Group` /
`Approximation`, `KL.apply`, `get_scaling`, `adagrad_window`, `rho2sigma` and the distributions' `logp` bodies from the reference
checkout and runs them eagerly on torch float64 tensors (`pytensor.grad` = torch autograd); a step of this file must reproduce
the loss, both gradients, both parameters and both adagrad rings of that execution to 1e-13, step after step through a wrap of
the window (tests/test_advi.py, committed fixture tests/golden/advi_reference_steps.npz).  That execution is what showed the
normalising constant missing here in round 2 (`scale_cost_to_minibatch`, below).  What this file restates, formula by formula:

  * `FullRankGroup` (variational/approximations.py:118-188): parameters mu [d] and L_tril [d (d + 1) / 2] in `np.tril_indices`
    order, initialised to `start` and `eye(d)[tril]`; the diagonal of L goes through `rho2sigma` = softplus (`L`, :141-147);
    z = z0 L^T + mu (:184-188); logq = sum(-z0^2 / 2 - log sqrt(2 pi)) - sum log diag(L) (:175-182).
  * `KL.apply` (variational/operators.py:64-65): loss = -datalogp + (logq - varlogp) for one Monte-Carlo sample (`obj_n_mc=1`).
  * `MinibatchRandomVariable` logp (variational/minibatch_rv.py:87-106): the likelihood of the B drawn rows times N / B.
  * `adagrad_window` (variational/updates.py:542-585): per-parameter window of the last `n_win` squared gradients.

  * `symbolic_normalizing_constant` (variational/opvi.py:1314-1332) with `scale_cost_to_minibatch` on: every term divided by N / B.

Also checked: the gradient against torch float64 autograd of the same loss, and the closed-form Gaussian posterior of the
linear-Gaussian GLM that the fit must converge to (tests/test_advi.py).  The random inputs of a step (minibatch row
indices, z0) are ARGUMENTS here: the reference draws them with PyTensor RNG ops whose streams are not reproducible outside it.
"""

from __future__ import annotations

import numpy as np


def softplus(x):
    return np.logaddexp(0.0, x)


def sigmoid(x):
    return 0.5 * (1.0 + np.tanh(0.5 * x))


class FullRankState:
    def __init__(self, d, start=None):
        self.d = d
        self.mu = np.zeros(d) if start is None else np.array(start, dtype="float64", copy=True)
        self.tril = np.tril_indices(d)
        self.L_tril = np.eye(d)[self.tril].astype("float64")          # approximations.py:138-141
        self.acc_mu = np.zeros((d, 0))
        self.acc_L = np.zeros((len(self.L_tril), 0))
        self.i = 0

    def L(self):                                                      # approximations.py:143-149
        L = np.zeros((self.d, self.d))
        L[self.tril] = self.L_tril
        idx = np.arange(self.d)
        L[idx, idx] = softplus(L[idx, idx])
        return L


class GLM:
    """y_i ~ family(eta_i), eta_i = x_i . beta;  beta ~ Normal(0, prior_sd);  family "normal" (known sigma) or "bernoulli" (logit)."""

    def __init__(self, X, y, family="normal", sigma=1.0, prior_sd=1.0):
        self.X, self.y, self.family, self.sigma, self.prior_sd = np.asarray(X, float), np.asarray(y, float), family, float(sigma), float(prior_sd)
        self.N, self.d = self.X.shape

    def datalogp_grad(self, beta, idx):
        """(N / B) sum_b log p(y_b | x_b, beta) over the drawn rows and its gradient (minibatch_rv.py:102-106)."""
        xb, yb = self.X[idx], self.y[idx]
        eta = xb @ beta
        scale = self.N / len(idx)
        if self.family == "normal":
            r = (yb - eta) / self.sigma
            ll = -0.5 * r * r - np.log(self.sigma) - 0.5 * np.log(2 * np.pi)
            res = r / self.sigma
        else:
            ll = yb * eta - softplus(eta)
            res = yb - sigmoid(eta)
        return scale * ll.sum(), scale * (xb.T @ res)

    def varlogp_grad(self, beta):
        z = beta / self.prior_sd
        return np.sum(-0.5 * z * z - np.log(self.prior_sd) - 0.5 * np.log(2 * np.pi)), -z / self.prior_sd


def advi_step(glm: GLM, st: FullRankState, idx, z0, learning_rate=0.001, epsilon=0.1, n_win=10, scale_cost_to_minibatch=True):
    """One call of the compiled step function (opvi.py:318-404): returns the loss; updates `st` in place.

    `scale_cost_to_minibatch` (opvi.py:1264, default on in the reference): every term of the objective is divided by the
    normalising constant = the largest minibatch scaling N / B (`symbolic_normalizing_constant`, opvi.py:1314-1332;
    `datalogp_norm`, `varlogp_norm`, `logq_norm`, :1344-1421) -- the loss is on the scale of ONE minibatch, and so are the
    gradients adagrad_window sees (its epsilon is not scale free, so this changes the path of the fit)."""
    L = st.L()
    z = z0 @ L.T + st.mu                                              # approximations.py:184-188
    dlp, dg = glm.datalogp_grad(z, idx)
    vlp, vg = glm.varlogp_grad(z)
    diag = np.diag(L)
    logq = np.sum(-0.5 * z0**2 - np.log(np.sqrt(2 * np.pi))) - np.sum(np.log(diag))
    nc = (glm.N / len(idx)) if scale_cost_to_minibatch else 1.0       # opvi.py:1314-1332
    loss = -(dlp / nc) + (logq / nc - vlp / nc)                       # operators.py:64-65 on the normalised terms
    g = dg + vg                                                       # d logp / dz
    grad_mu = -g
    GL = -np.outer(g, z0)                                             # d loss / dL (lower triangle used)
    idx_d = np.arange(st.d)
    GL[idx_d, idx_d] += -1.0 / diag
    grad_tril = GL[st.tril]
    rho = st.L_tril[[i * (i + 1) // 2 + i for i in range(st.d)]]
    dpos = np.array([i * (i + 1) // 2 + i for i in range(st.d)])
    grad_tril[dpos] *= sigmoid(rho)                                   # through rho2sigma
    grad_mu = grad_mu / nc
    grad_tril = grad_tril / nc
    # adagrad_window (updates.py:571-584)
    if st.acc_mu.shape[1] != n_win:
        st.acc_mu = np.zeros((st.d, n_win))
        st.acc_L = np.zeros((len(st.L_tril), n_win))
        st.i = 0
    st.acc_mu[:, st.i] = grad_mu**2
    st.acc_L[:, st.i] = grad_tril**2
    st.i = st.i + 1 if st.i + 1 < n_win else 0
    st.mu = st.mu - learning_rate * grad_mu / np.sqrt(st.acc_mu.sum(axis=-1) + epsilon)
    st.L_tril = st.L_tril - learning_rate * grad_tril / np.sqrt(st.acc_L.sum(axis=-1) + epsilon)
    return loss, grad_mu, grad_tril


def advi_step_logp(logp_grad, st: FullRankState, z0, learning_rate=0.001, epsilon=0.1, n_win=10):
    """The same step for ANY model: `logp_grad(z) -> (logp, d logp / dz)` over the raveled unconstrained vector (the joint
    log-density `Model.logp` assembles, model/core.py:612-695; no minibatch, so every scaling is 1 and the normalising constant of
    opvi.py:1314-1332 is 1).  loss = logq - logp (`KL.apply`, operators.py:64-65, with datalogp + varlogp = logp).  With the GLM's
    full data as one batch this is `advi_step` (tests/test_advi.py pins one against the other)."""
    L = st.L()
    z = z0 @ L.T + st.mu                                              # approximations.py:184-188
    lp, g = logp_grad(z)
    g = np.asarray(g, dtype="float64")
    diag = np.diag(L)
    logq = np.sum(-0.5 * z0**2 - np.log(np.sqrt(2 * np.pi))) - np.sum(np.log(diag))
    loss = logq - lp
    grad_mu = -g
    GL = -np.outer(g, z0)
    idx_d = np.arange(st.d)
    GL[idx_d, idx_d] += -1.0 / diag
    grad_tril = GL[st.tril]
    dpos = np.array([i * (i + 1) // 2 + i for i in range(st.d)])
    grad_tril[dpos] *= sigmoid(st.L_tril[dpos])                      # through rho2sigma
    if st.acc_mu.shape[1] != n_win:
        st.acc_mu = np.zeros((st.d, n_win))
        st.acc_L = np.zeros((len(st.L_tril), n_win))
        st.i = 0
    st.acc_mu[:, st.i] = grad_mu**2
    st.acc_L[:, st.i] = grad_tril**2
    st.i = st.i + 1 if st.i + 1 < n_win else 0
    st.mu = st.mu - learning_rate * grad_mu / np.sqrt(st.acc_mu.sum(axis=-1) + epsilon)
    st.L_tril = st.L_tril - learning_rate * grad_tril / np.sqrt(st.acc_L.sum(axis=-1) + epsilon)
    return loss, grad_mu, grad_tril

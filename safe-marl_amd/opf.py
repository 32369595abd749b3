"""Multi-period OPF comparator on the device: ``utils/opf.py:13-192`` re-designed as a batched solve.

The reference builds one Pyomo MIQCP per day (T = episode_limit periods x {DistFlow state of 33 buses, 20 controls,
5 storage binaries}) and hands it to Gurobi.  Here the same program is solved for MANY days at once:

  * the network state is not a variable: for given controls the DistFlow equalities opf.py:96-129 ARE a power flow,
    so every period's state comes from the HIP power-flow kernel (``pf_solve_batch``, one lane per bus), B*T solves
    per launch;
  * sensitivities of Vsqr, Isqr and the losses to the 4*n controls of a period come from the same kernel by central
    differences — 40 more solves per period in the SAME launch (a control of period t only moves period t), which
    costs microseconds on this hardware and needs no adjoint code;
  * with those, each outer iteration solves a convex QP in the controls of the whole horizon (concave separable
    revenue/discomfort terms, a positive-semidefinite loss model 2*Re(Zbus) around the exact loss gradient,
    linearised voltage/current limits, box limits and the storage energy chain opf.py:139-148) by a batched dense
    primal-dual interior-point method: one Cholesky of an (n x n) matrix per instance and iteration
    (n = 20 T = 1920 at T = 96; 29.5 MB per instance — 288 GB of HBM hold thousands of days), LP-like bang-bang
    structure included;
  * the outer loop (sequential convex programming) repeats until the controls stop moving; at that point the
    linearisation is exact, so the result satisfies the reference's nonlinear constraints to solver tolerance.

The storage binaries opf.py:150-156 are relaxed: with ess_cost > 0 simultaneous charging and discharging only burns
energy and money; the solution reports ``min(Pesc, Pesd)`` and the indicator ``Pesc > Pesd``.

Boundary: ``opf_model(network_data, flex_price, active_power_demand, reactive_power_demand, pv_active_power,
initial_ess_energy)`` has the reference's signature and returns its solution dict (opf.py:160-189);
``BatchedOPF.solve`` is the tensor interface underneath.  Needs the HIP library (no CPU path).
"""
from __future__ import annotations

from math import acos, tan

import numpy as np
import torch as th

from .flex_env import pf_solve_batch
from .network import build_tables

NATIVE_QP = True          # the QP of every outer iteration through csrc/opf.hip (tests switch it off to compare with qp_ipm)

ENV_DEFAULTS = dict(episode_limit=96, v_min=0.9, v_max=1.1, pv_cost=0.05, ess_cost=0.03, discomfort_coeff=0.15,
                    eta_ch=0.9, eta_dis=0.9, max_power_reduction=0.5, e_min=0.0, e_max=0.025, p_ch_max=0.005,
                    p_dis_max=0.005, cos_phi_max=0.95)           # flex_provision.yaml:4-27, read at opf.py:10-11


# ---------------------------------------------------------------------------------------------------------------
# batched dense primal-dual interior point for   min 1/2 x'Qx + c'x   s.t.  rows of three structured blocks
# ---------------------------------------------------------------------------------------------------------------
class _Rows:
    """One block of inequality rows  A x <= u  (and/or  A x >= l) with a structured A."""

    def apply(self, x):            # [B, n] -> [B, m]
        raise NotImplementedError

    def apply_t(self, y):          # [B, m] -> [B, n]
        raise NotImplementedError

    def add_gram(self, N, d):      # N[B, n, n] += A' diag(d) A
        raise NotImplementedError


class _Identity(_Rows):
    def apply(self, x):
        return x

    def apply_t(self, y):
        return y

    def add_gram(self, N, d):
        N.diagonal(dim1=1, dim2=2).add_(d)


class _PeriodBlocks(_Rows):
    """Rows that only see the controls of their own period: J[B, T, R, w], x viewed as [B, T, w]."""

    def __init__(self, J):
        self.J = J
        self.B, self.T, self.R, self.w = J.shape

    def apply(self, x):
        return th.einsum("btrw,btw->btr", self.J, x.view(self.B, self.T, self.w)).reshape(self.B, -1)

    def apply_t(self, y):
        return th.einsum("btrw,btr->btw", self.J, y.view(self.B, self.T, self.R)).reshape(self.B, -1)

    def add_gram(self, N, d):
        blk = th.einsum("btrv,btr,btrw->btvw", self.J, d.view(self.B, self.T, self.R), self.J)
        # the diagonal (t, t) blocks of N as one strided view [B, w, w, T]
        N.view(self.B, self.T, self.w, self.T, self.w).diagonal(dim1=1, dim2=3).add_(blk.permute(0, 2, 3, 1))


class _Shared(_Rows):
    """The same dense matrix C[m, n] for every instance (the storage energy chain)."""

    def __init__(self, C):
        self.C = C

    def apply(self, x):
        return x @ self.C.T

    def apply_t(self, y):
        return y @ self.C

    def add_gram(self, N, d):
        N += th.einsum("mi,bm,mj->bij", self.C, d, self.C)


class _EnergyChain(_Rows):
    """The storage energy chain opf.py:139-148 as an operator: row (t, k) is E[t, k] - E_init[k] =
    sum_{1 <= s <= t} (a Pesc[s, k] - b Pesd[s, k]) with a = dt eta_ch, b = dt / eta_dis (period 0 contributes nothing).
    Same rows as ``_Shared(C)`` with the explicit matrix, but apply / transpose are cumulative sums and the Gram
    matrix is a gather of suffix sums: C' diag(d) C [s, s'] = coef coef' sum_{t >= max(s, s')} d[t]."""

    def __init__(self, T, na, a, b):
        self.T, self.na, self.a, self.b = T, na, a, b

    def apply(self, x):
        x = x.view(x.shape[0], self.T, 4, self.na)
        inc = self.a * x[:, :, 2] - self.b * x[:, :, 3]
        inc = th.cat([th.zeros_like(inc[:, :1]), inc[:, 1:]], 1)
        return inc.cumsum(1).reshape(x.shape[0], -1)

    def _suffix(self, y):
        y = y.view(y.shape[0], self.T, self.na)
        s = y.flip(1).cumsum(1).flip(1)
        return th.cat([th.zeros_like(s[:, :1]), s[:, 1:]], 1)           # period 0 has no coefficient

    def apply_t(self, y):
        s = self._suffix(y)
        out = th.zeros(y.shape[0], self.T, 4, self.na, dtype=y.dtype, device=y.device)
        out[:, :, 2] = self.a * s
        out[:, :, 3] = -self.b * s
        return out.reshape(y.shape[0], -1)

    def _gram_core(self, d):
        s = self._suffix(d)                                              # [B, T, na]
        idx = th.arange(self.T, device=d.device)
        g = s[:, th.maximum(idx[:, None], idx[None, :])]                 # [B, T, T, na]: suffix sum at max(s, s')
        return g * ((idx[:, None] > 0) & (idx[None, :] > 0)).to(d.dtype)[None, :, :, None]

    def add_gram(self, N, d):
        B = d.shape[0]
        g = self._gram_core(d)
        blk = N.view(B, self.T, 4, self.na, self.T, 4, self.na).diagonal(dim1=3, dim2=6)     # [B, T, 4, T, 4, na]
        blk[:, :, 2, :, 2] += (self.a * self.a) * g
        blk[:, :, 2, :, 3] -= (self.a * self.b) * g
        blk[:, :, 3, :, 2] -= (self.a * self.b) * g
        blk[:, :, 3, :, 3] += (self.b * self.b) * g

    def add_gram_coupled(self, E, d):
        """The same Gram matrix restricted to the columns it touches: E[B, T*2na, T*2na], ordered (t, Pesc|Pesd, k)."""
        B = d.shape[0]
        g = self._gram_core(d)
        blk = E.view(B, self.T, 2, self.na, self.T, 2, self.na).diagonal(dim1=3, dim2=6)     # [B, T, 2, T, 2, na]
        blk[:, :, 0, :, 0] += (self.a * self.a) * g
        blk[:, :, 0, :, 1] -= (self.a * self.b) * g
        blk[:, :, 1, :, 0] -= (self.a * self.b) * g
        blk[:, :, 1, :, 1] += (self.b * self.b) * g


def _chol_retry(M, what):
    """Cholesky with a growing diagonal bump where round-off cost positive definiteness (z/s spans twenty decades late
    in the iteration); an infeasible program makes the iterates diverge until even that fails — the reference raises
    'Solver failed to find a solution' there (opf.py:155-157)."""
    L, fail = th.linalg.cholesky_ex(M)
    bump = 1e-10
    diag = M.diagonal(dim1=-2, dim2=-1)
    while bool((fail > 0).any()) and bump < 1e-3:
        diag.add_((bump * diag.amax(-1, keepdim=True)) * (fail > 0).to(M.dtype).unsqueeze(-1))
        L, fail = th.linalg.cholesky_ex(M)
        bump *= 100.0
    if bool((fail > 0).any()):
        raise RuntimeError(f"Solver failed to find a solution (QP infeasible, or its {what} lost definiteness)")
    return L


def _tri_solve(L, rhs):
    """(L L')^-1 rhs by two batched triangular solves (rocBLAS strided-batched trsm).  ``torch.cholesky_solve`` on the same
    operands — L [32, 1920, 1920] fp64 from ``cholesky_ex`` and rhs [32, 1920, 1], both contiguous with standard strides —
    ended in a GPU memory fault in round 1 (for batch > 1 and one right-hand side ATen routes it to hipSOLVER's
    potrsBatched, array-of-pointers).  The operands are well-formed (this route and ``cholesky_ex`` consume the very same
    tensors, and the steps pass the KKT / SLSQP checks); the CAUSE is unconfirmed — no log or fault address of that run
    was kept (tools/potrs_repro.py).  Nothing here depends on that call."""
    y = th.linalg.solve_triangular(L, rhs, upper=False)
    return th.linalg.solve_triangular(L.transpose(-1, -2), y, upper=True)


def _factor_dense(Qblk, sets, s, z, fmask, reg):
    """Normal matrix Q + sum A' diag(z/s) A as one dense [n, n] matrix per instance; the pinned variables' rows and
    columns are replaced by identity.  Returns the solve closure."""
    B, T, w, _ = Qblk.shape
    n = T * w
    N = th.zeros(B, n, n, dtype=Qblk.dtype, device=Qblk.device)
    N.view(B, T, w, T, w).diagonal(dim1=1, dim2=3).add_(Qblk.permute(0, 2, 3, 1))
    for (rows, sg, h), si, zi in zip(sets, s, z):
        rows.add_gram(N, zi / si)
    N *= fmask.unsqueeze(1) * fmask.unsqueeze(2)
    diag = N.diagonal(dim1=1, dim2=2)
    diag.add_(reg * diag.amax(1, keepdim=True) + (1.0 - fmask))
    L = _chol_retry(N, "normal matrix")
    return lambda rhs: _tri_solve(L, rhs.unsqueeze(-1)).squeeze(-1)


def _factor_structured(Qblk, sets, s, z, fmask, reg):
    """The same system by block elimination.  Only the storage controls (Pesc, Pesd: the columns the energy chain
    touches) are coupled across periods; the other half of every period's controls only meets its own period, so it is
    eliminated with T small Cholesky factorisations per instance and the dense factorisation shrinks from (4 n T)^2
    to the Schur complement of size (2 n T)^2 — an eighth of the flops, half the panel steps."""
    B, T, w, _ = Qblk.shape
    dt, dev = Qblk.dtype, Qblk.device
    chain = next(rows for rows, _, _ in sets if isinstance(rows, _EnergyChain))
    m = 2 * chain.na                                     # local controls per period (Pred, Qpv) = coupled ones (Pesc, Pesd)
    P = Qblk.clone()                                     # [B, T, w, w] period blocks
    E = th.zeros(B, T * m, T * m, dtype=dt, device=dev)  # the chain's Gram matrix on the coupled controls
    for (rows, sg, h), si, zi in zip(sets, s, z):
        d = zi / si
        if isinstance(rows, _Identity):
            P.diagonal(dim1=2, dim2=3).add_(d.view(B, T, w))
        elif isinstance(rows, _PeriodBlocks):
            P += th.einsum("btrv,btr,btrw->btvw", rows.J, d.view(B, T, rows.R), rows.J)
        else:
            rows.add_gram_coupled(E, d)
    fm = fmask.view(B, T, w)
    P *= fm.unsqueeze(-1) * fm.unsqueeze(-2)
    pd_ = P.diagonal(dim1=2, dim2=3)
    pd_.add_(reg * pd_.amax(dim=(1, 2), keepdim=True) + (1.0 - fm))
    fc = fm[:, :, m:].reshape(B, T * m)
    E *= fc.unsqueeze(1) * fc.unsqueeze(2)
    A, Bm, Cb = P[:, :, :m, :m], P[:, :, :m, m:], P[:, :, m:, m:]
    LA = _chol_retry(A.contiguous(), "period block")
    X = _tri_solve(LA, Bm.contiguous())                  # A^-1 B   [B, T, m, m]
    S = E
    S.view(B, T, m, T, m).diagonal(dim1=1, dim2=3).add_((Cb - Bm.transpose(2, 3) @ X).permute(0, 2, 3, 1))
    LS = _chol_retry(S, "Schur complement")

    def solve(rhs):
        r = rhs.view(B, T, w)
        tl = _tri_solve(LA, r[:, :, :m].unsqueeze(-1)).squeeze(-1)                              # A^-1 r_L
        rc = r[:, :, m:] - th.einsum("btlc,btl->btc", Bm, tl)
        yc = _tri_solve(LS, rc.reshape(B, T * m, 1)).view(B, T, m)
        yl = tl - th.einsum("btlc,btc->btl", X, yc)
        return th.cat([yl, yc], -1).reshape(B, T * w)

    return solve


def _factor_riccati(Qblk, sets, s, z, fmask, reg):
    """The same system by a Riccati recursion over the periods — the algorithm of csrc/opf.hip in torch (CPU-runnable; the tests
    pin the kernel's algebra with it).  N = blockdiag(P_t) + C' diag(D) C is the optimality condition of a linear-quadratic
    problem with the n_agents energy offsets as state: per period (in parallel) L_t = chol(P_t), W_t = L_t^-1 G', M_t = W_t' W_t;
    backward S_t = L_S L_S', B_t = I + L_S' M_t L_S, S_{t-1} = D_{t-1} + L_S B_t^-1 L_S'; every product stays factored (the
    textbook forms I - H M and S e cancel catastrophically once z / s spans twenty decades), and each solve is followed by one
    step of iterative refinement against N applied through its operators."""
    B, T, w, _ = Qblk.shape
    dt = Qblk.dtype
    chain = next(rows for rows, _, _ in sets if isinstance(rows, _EnergyChain))
    na = chain.na
    P = Qblk.clone()
    D = th.zeros(B, T, na, dtype=dt, device=Qblk.device)
    for (rows, sg, h), si, zi in zip(sets, s, z):
        d = zi / si
        if isinstance(rows, _Identity):
            P.diagonal(dim1=2, dim2=3).add_(d.view(B, T, w))
        elif isinstance(rows, _PeriodBlocks):
            P += th.einsum("btrv,btr,btrw->btvw", rows.J, d.view(B, T, rows.R), rows.J)
        else:
            D += d.view(B, T, na)
    fm = fmask.view(B, T, w)
    P *= fm.unsqueeze(-1) * fm.unsqueeze(-2)
    pd_ = P.diagonal(dim1=2, dim2=3)
    regv = reg * pd_.amax(dim=(1, 2), keepdim=True)             # [B, 1, 1]
    pd_.add_(regv + (1.0 - fm))
    G = th.zeros(T, na, w, dtype=dt, device=Qblk.device)
    idx = th.arange(na, device=Qblk.device)
    G[1:, idx, 2 * na + idx] = chain.a                          # (period 0 has no coefficient: opf.py:140-142)
    G[1:, idx, 3 * na + idx] = -chain.b
    G = G.unsqueeze(0) * fm.unsqueeze(2)                        # [B, T, na, w]
    L = th.linalg.cholesky(P)
    W = th.linalg.solve_triangular(L, G.transpose(2, 3), upper=False)           # [B, T, w, na]
    M = W.transpose(2, 3) @ W
    eye = th.eye(na, dtype=dt, device=Qblk.device)
    LS, LB = [None] * T, [None] * T
    S = th.diag_embed(D[:, T - 1])
    for t in range(T - 1, 0, -1):
        LS[t] = th.linalg.cholesky(S)
        LB[t] = th.linalg.cholesky(eye + LS[t].transpose(1, 2) @ M[:, t] @ LS[t])
        Z = th.linalg.solve_triangular(LB[t], LS[t].transpose(1, 2), upper=False)
        S = th.diag_embed(D[:, t - 1]) + Z.transpose(1, 2) @ Z

    def mv(A, v):
        return (A @ v.unsqueeze(-1)).squeeze(-1)

    def lsolve(Lm, v, transpose=False):
        if transpose:
            return th.linalg.solve_triangular(Lm.transpose(1, 2), v.unsqueeze(-1), upper=True).squeeze(-1)
        return th.linalg.solve_triangular(Lm, v.unsqueeze(-1), upper=False).squeeze(-1)

    def once(rhs):
        y = th.linalg.solve_triangular(L, rhs.view(B, T, w, 1), upper=False)
        g = (W.transpose(2, 3) @ y).squeeze(-1)                                 # [B, T, na]
        sv = [None] * T
        sv[T - 1] = th.zeros(B, na, dtype=dt, device=Qblk.device)
        for t in range(T - 1, 0, -1):
            a = mv(LS[t].transpose(1, 2), g[:, t]) + lsolve(LS[t], sv[t])
            sv[t - 1] = mv(LS[t], lsolve(LB[t], lsolve(LB[t], a), transpose=True))
        e = th.zeros(B, na, dtype=dt, device=Qblk.device)
        nu = th.zeros(B, T, na, dtype=dt, device=Qblk.device)
        for t in range(1, T):
            v = e + g[:, t] - mv(M[:, t], sv[t])
            q = lsolve(LB[t], lsolve(LB[t], mv(LS[t].transpose(1, 2), v)), transpose=True)
            nu[:, t] = -(mv(LS[t], q) + sv[t])
            e = lsolve(LS[t], q, transpose=True)
        dx = th.linalg.solve_triangular(L.transpose(2, 3), y + W @ nu.unsqueeze(-1), upper=True)
        return dx.reshape(B, T * w)

    def apply_n(v):
        out = th.einsum("btvw,btw->btv", Qblk, v.view(B, T, w)).reshape(B, -1) + regv.view(B, 1) * v
        for (rows, sg, h), si, zi in zip(sets, s, z):
            out = out + rows.apply_t((zi / si) * rows.apply(v))
        return out * fmask

    def solve(rhs):
        x = once(rhs)
        return x + once((rhs - apply_n(x)) * fmask)

    return solve


def qp_ipm(Qblk, c, blocks, x0, free=None, max_iter=80, tol=1e-11, reg=1e-12, verbose=False, factor=None):
    """Mehrotra predictor-corrector on a batch of convex QPs.

    Qblk [B, T, w, w]: block-diagonal Hessian;  c [B, n];  blocks: list of (rows, lower [B, m] or None,
    upper [B, m] or None);  x0 [B, n] start;  free [B, n] bool: variables that may move (the others stay at x0 —
    their rows and columns leave the Newton system).  Returns x [B, n] and a dict with the duality measures.
    Instances that have converged are frozen (zero step) while the slowest ones finish: pushing a converged
    instance further only ruins the conditioning of its normal matrix.  ``factor="riccati"`` solves the Newton system by
    :func:`_factor_riccati` (the kernel's algorithm) instead of a Cholesky factorisation."""
    B, T, w, _ = Qblk.shape
    n = T * w
    dev, dt = c.device, c.dtype
    # one-sided row sets: (rows, sign, bound) meaning  sign * (A x) <= sign * bound
    sets = []
    for rows, lo, hi in blocks:
        if hi is not None:
            sets.append((rows, 1.0, hi))
        if lo is not None:
            sets.append((rows, -1.0, -lo))
    fmask = th.ones(B, n, dtype=dt, device=dev) if free is None else free.to(dt)
    # every row block period-local except ONE energy chain: the controls it does not touch are eliminated per period
    structured = (sum(isinstance(r, _EnergyChain) for r, _, _ in blocks) == 1
                  and all(isinstance(r, (_Identity, _PeriodBlocks, _EnergyChain)) for r, _, _ in blocks))

    def Qx(x):
        return th.einsum("btvw,btw->btv", Qblk, x.view(B, T, w)).reshape(B, n)

    x = x0.clone()
    s = [th.clamp(h - sg * rows.apply(x), min=1e-3) for rows, sg, h in sets]
    z = [th.ones_like(si) for si in s]
    m_tot = sum(si.shape[1] for si in s)
    info = {}
    done = th.zeros(B, dtype=th.bool, device=dev)
    for it in range(max_iter):
        r_d = Qx(x) + c
        for (rows, sg, h), zi in zip(sets, z):
            r_d = r_d + sg * rows.apply_t(zi)
        r_d = r_d * fmask
        r_p = [sg * rows.apply(x) + si - h for (rows, sg, h), si in zip(sets, s)]
        mu = sum((si * zi).sum(1) for si, zi in zip(s, z)) / m_tot                      # [B]
        res_d = r_d.abs().amax(1)
        res_p = th.stack([rp.abs().amax(1) for rp in r_p]).amax(0)
        # the dual residual floors near 1e-8 once z/s spans twenty decades (conditioning of the normal matrix)
        done = done | ((mu < tol) & (res_p < 1e-8) & (res_d < 1e-6)) | (mu < 1e-4 * tol)
        info = dict(iters=it, mu=mu, res_d=res_d, res_p=res_p, converged=done)
        if verbose:
            print(f"  ipm {it:2d} mu {mu.max().item():.2e} rd {res_d.max().item():.2e} rp {res_p.max().item():.2e} "
                  f"done {int(done.sum())}/{B}")
        if bool(done.all()):
            break
        if factor == "riccati":
            if not structured:
                raise ValueError("the Riccati recursion needs period-local rows plus one energy chain")
            solve = _factor_riccati(Qblk, sets, s, z, fmask, reg)
        else:
            solve = (_factor_structured if structured else _factor_dense)(Qblk, sets, s, z, fmask, reg)

        def newton(r_c):
            rhs = -r_d
            for (rows, sg, h), si, zi, rp, rc in zip(sets, s, z, r_p, r_c):
                rhs = rhs - sg * rows.apply_t((zi * rp - rc) / si)
            dx = solve(rhs * fmask) * fmask
            ds, dz = [], []
            for (rows, sg, h), si, zi, rp, rc in zip(sets, s, z, r_p, r_c):
                dsi = -rp - sg * rows.apply(dx)
                ds.append(dsi)
                dz.append((-rc - zi * dsi) / si)
            return dx, ds, dz

        def step_len(v, dv):
            ratio = th.where(dv < 0, -v / dv, th.full_like(v, float("inf")))
            return ratio.amin(1)

        # predictor
        dx_a, ds_a, dz_a = newton([si * zi for si, zi in zip(s, z)])
        a_p = th.stack([step_len(si, d) for si, d in zip(s, ds_a)]).amin(0).clamp(max=1.0)
        a_d = th.stack([step_len(zi, d) for zi, d in zip(z, dz_a)]).amin(0).clamp(max=1.0)
        mu_a = sum(((si + a_p[:, None] * d1) * (zi + a_d[:, None] * d2)).sum(1)
                   for si, zi, d1, d2 in zip(s, z, ds_a, dz_a)) / m_tot
        sigma = (mu_a / mu).clamp(min=0.0, max=1.0) ** 3
        # corrector
        r_c = [si * zi + d1 * d2 - (sigma * mu)[:, None] for si, zi, d1, d2 in zip(s, z, ds_a, dz_a)]
        dx, ds, dz = newton(r_c)
        live = (~done).to(dt)
        a_p = (0.995 * th.stack([step_len(si, d) for si, d in zip(s, ds)]).amin(0)).clamp(max=1.0) * live
        a_d = (0.995 * th.stack([step_len(zi, d) for zi, d in zip(z, dz)]).amin(0)).clamp(max=1.0) * live
        x = x + a_p[:, None] * dx
        s = [si + a_p[:, None] * d for si, d in zip(s, ds)]
        z = [zi + a_d[:, None] * d for zi, d in zip(z, dz)]
    info["duals"] = z
    if not bool(done.all()) and bool((info["res_p"][~done] > 1e-6).any()):
        raise RuntimeError("Solver failed to find a solution (constraints cannot be met)")           # opf.py:155-157
    return x, info


def qp_ipm_native(Qblk, c, lo, hi, free, jv, v_lo, v_hi, ji, i_hi, chain_a, chain_b, e_lo, e_hi, x0, max_iter=80, tol=1e-11,
                  reg=1e-12):
    """The OPF's QP through ``flexopf_qp_solve`` (include/flexopf.h, csrc/opf.hip): the same predictor-corrector iteration as
    :func:`qp_ipm` with the row blocks [box | Jv (two-sided) | Ji (upper) | energy chain (two-sided)], one persistent
    work-group per instance for the WHOLE iteration (no launch, no host decision per iteration) and the Newton system by a
    Riccati recursion over the periods instead of a dense factorisation.  Qblk [B, T, w, w]; c, lo, hi, x0 [B, T w]; free
    [B, T w] bool; jv, ji [B, T, R, w]; v_lo, v_hi, i_hi [B, T R]; e_lo, e_hi [B, T na].  Returns (x, info) like qp_ipm;
    ``info["duals"]`` in qp_ipm's order (upper, lower of every block that has them)."""
    import ctypes as C
    from . import _lib
    B, T, w, _ = Qblk.shape
    R, na = jv.shape[2], w // 4
    dev = Qblk.device
    if dev.type != "cuda":
        raise RuntimeError("qp_ipm_native needs the HIP library and a GPU (qp_ipm is the torch form)")
    lib = _lib.load()
    per = lib.flexopf_qp_work_doubles(T, na, R)
    if per < 0:
        raise ValueError(f"flexopf_qp_solve: sizes out of range (T {T}, agents {na}, rows {R}; include/flexopf.h)")
    f64 = dict(dtype=th.float64, device=dev)
    mp = 2 * w + 3 * R + 2 * na
    keep = [t.to(th.float64).contiguous() for t in (Qblk, c, lo, hi, jv, v_lo, v_hi, ji, i_hi, e_lo, e_hi, x0)]
    fm = free.to(th.uint8).contiguous()
    x = th.empty(B, T * w, **f64)
    duals = th.empty(B, T, mp, **f64)
    info = th.empty(B, _lib.FLEXOPF_INFO, **f64)
    work = th.empty(B * per, **f64)
    a = _lib.FlexQpArgs()
    a.batch, a.periods, a.n_agents, a.rows, a.max_iter = B, T, na, R, int(max_iter)
    a.tol, a.reg, a.chain_a, a.chain_b = float(tol), float(reg), float(chain_a), float(chain_b)
    (a.q, a.c, a.lo, a.hi, a.jv, a.v_lo, a.v_hi, a.ji, a.i_hi, a.e_lo, a.e_hi, a.x0) = [t.data_ptr() for t in keep]
    a.free_mask, a.x, a.duals, a.info, a.work = fm.data_ptr(), x.data_ptr(), duals.data_ptr(), info.data_ptr(), work.data_ptr()
    _lib.check(lib.flexopf_qp_solve(C.byref(a), C.c_void_p(th.cuda.current_stream().cuda_stream)), "flexopf_qp_solve")
    res = info.cpu()                                      # (synchronises: the scratch and the inputs outlive the launch)
    conv = res[:, 4] > 0.5
    out = dict(iters=int(res[:, 0].max().item()), mu=res[:, 1].to(dev), res_d=res[:, 2].to(dev), res_p=res[:, 3].to(dev),
               converged=conv.to(dev), floored_pivots=res[:, 5])
    o = [0, w, 2 * w, 2 * w + R, 2 * w + 2 * R, 2 * w + 3 * R, 2 * w + 3 * R + na, mp]
    out["duals"] = [duals[:, :, o[i]:o[i + 1]].reshape(B, -1) for i in range(7)]
    # an infeasible program drives the iterates off until the factorisations break down (pivots floored, then non-finite
    # numbers): the reference raises 'Solver failed to find a solution' there (opf.py:155-157)
    bad = ~conv & (~(res[:, 3] <= 1e-6) | ~th.isfinite(res[:, 1:4]).all(1))
    if bool(bad.any()) or not bool(th.isfinite(x).all()):
        raise RuntimeError("Solver failed to find a solution (constraints cannot be met)")
    return x, out


# ---------------------------------------------------------------------------------------------------------------
# the OPF itself
# ---------------------------------------------------------------------------------------------------------------
class BatchedOPF:
    """B independent days x T periods of opf.py's program on one device."""

    CTRL = 4                                   # Pred, Qpv, Pesc, Pesd per building (opf.py:54-58)

    def __init__(self, net, env_args=None, device="cuda:0"):
        self.net = net
        self.cfg = dict(ENV_DEFAULTS)
        self.cfg.update({k: v for k, v in (env_args or {}).items() if k in ENV_DEFAULTS})
        self.device = th.device(device)
        t = build_tables(net)
        if not all(t.line_forward[i] for i in range(t.n_bus) if t.line_of_bus[i] is not None):
            raise ValueError("opf.py:96-129 assume every line is keyed (from, to) away from the substation")
        self.tables = t
        self.n_bus = t.n_bus
        buses = list(net["bus_numbers"])
        self.bld = [buses.index(b) for b in net["buildings"]]
        if list(net["PVs_at_buildings"]) != list(net["buildings"]) or list(net["ESSs_at_buildings"]) != list(net["buildings"]):
            raise ValueError("PVs and ESSs are expected at the buildings (flex_provision.yaml:28-30)")
        self.na = len(self.bld)
        self.w = self.CTRL * self.na
        f64 = dict(dtype=th.float64, device=self.device)
        self.nonslack = th.tensor([i for i in range(t.n_bus) if i != t.slack], device=self.device)
        self.r_line = th.tensor(t.r, **f64)                                     # by child bus
        imax = np.array([net["max_line_currents"][k] if k is not None else np.inf for k in t.line_of_bus])
        self.imax2 = th.tensor(imax[[i for i in range(t.n_bus) if i != t.slack]] ** 2, **f64)
        self.tanphi = tan(acos(self.cfg["cos_phi_max"]))
        self.dt = 24.0 / self.cfg["episode_limit"]                              # opf.py:27-28
        # loss model Hessian: losses ~ p' Re(Z) p + q' Re(Z) q in the net loads (V ~ 1), Z = path-intersection matrix
        rbb = np.zeros((self.na, self.na))
        paths = []
        for b in self.bld:
            p, i = set(), b
            while t.parent[i] >= 0:
                p.add(i)
                i = t.parent[i]
            paths.append(p)
        for a in range(self.na):
            for b in range(self.na):
                rbb[a, b] = sum(t.r[k] for k in paths[a] & paths[b])
        # controls -> net-load change of the buildings: dp = -Pred + Pesc - Pesd, dq = -Qpv   (opf.py:96-116)
        M = np.zeros((2 * self.na, self.w))
        for k in range(self.na):
            M[k, 0 * self.na + k] = -1.0
            M[k, 2 * self.na + k] = 1.0
            M[k, 3 * self.na + k] = -1.0
            M[self.na + k, 1 * self.na + k] = -1.0
        H = np.zeros((2 * self.na, 2 * self.na))
        H[:self.na, :self.na] = 2 * rbb
        H[self.na:, self.na:] = 2 * rbb
        self.loss_hess = th.tensor(M.T @ H @ M, **f64)                           # [w, w], PSD

    # -- network response --------------------------------------------------------------------------------------
    def _net_loads(self, pd, qd, ppv, x):
        """x [..., 4, na] -> (pnet, qnet) [..., n_bus] (pd, qd, ppv broadcast against x's leading dimensions)"""
        lead = th.broadcast_shapes(x.shape[:-2], pd.shape[:-1])
        pnet, qnet = pd.expand(*lead, self.n_bus).clone(), qd.expand(*lead, self.n_bus).clone()
        pnet[..., self.bld] += -x[..., 0, :] - ppv + x[..., 2, :] - x[..., 3, :]
        qnet[..., self.bld] += -x[..., 1, :]
        return pnet, qnet

    def _pf(self, pnet, qnet, want_branch=False):
        shape = pnet.shape[:-1]
        # Newton on the tree, not the sweeps the env steps with: the central differences of linearise() divide the solver's
        # error by h = 5e-5, and a sweep solve stops wherever its mismatch estimate passes the tolerance — two neighbouring trial
        # points may stop a sweep apart, a kink of ~1e-12 / h in one sensitivity that held one of 128 days at a control move
        # of 4.8e-5 pu for four outer iterations (10 instead of 6: tools/opf_outer_probe.py, profiles/r05bm_opf_outer.txt).
        # Newton's last step lands orders below the tolerance; the power flow is 5 % of a solve either way.
        from . import _lib
        out = pf_solve_batch(self.net, pnet.reshape(-1, self.n_bus), qnet.reshape(-1, self.n_bus), want_branch=True,
                             solver=_lib.FLEX_SOLVER_TREE)
        if bool(out["failed"].any()):
            raise RuntimeError("power flow failed inside the OPF (voltage collapse at a trial point)")
        v2 = (out["v"] ** 2).reshape(*shape, self.n_bus)
        i2 = out["isqr"].reshape(*shape, self.n_bus)
        res = dict(v2=v2, i2=i2, loss=(i2 * self.r_line).sum(-1))
        if want_branch:
            res.update(pl=out["pl"].reshape(*shape, self.n_bus), ql=out["ql"].reshape(*shape, self.n_bus))
        return res

    def linearise(self, pd, qd, ppv, x, h=5e-5):
        """State at x [B, T, 4, na] and central-difference sensitivities, all in ONE power-flow launch of
        (1 + 2 w) B T solves.  h balances the O(h^2) truncation (~1e-9) against the power-flow tolerance (1e-12 in
        the state, /h in the derivative)."""
        B, T = x.shape[:2]
        w = self.w
        xf = x.reshape(B, T, w)
        eye = th.eye(w, dtype=x.dtype, device=x.device) * h
        trial = th.cat([xf.unsqueeze(2), xf.unsqueeze(2) + eye, xf.unsqueeze(2) - eye], 2)     # [B, T, 1+2w, w]
        pnet, qnet = self._net_loads(pd.unsqueeze(2), qd.unsqueeze(2), ppv.unsqueeze(2), trial.view(B, T, 1 + 2 * w, 4, self.na))
        s = self._pf(pnet, qnet)
        v2 = s["v2"][..., self.nonslack]
        i2 = s["i2"][..., self.nonslack]
        jv = ((v2[:, :, 1:1 + w] - v2[:, :, 1 + w:]) / (2 * h)).transpose(2, 3)               # [B, T, rows, w]
        ji = ((i2[:, :, 1:1 + w] - i2[:, :, 1 + w:]) / (2 * h)).transpose(2, 3)
        gl = (s["loss"][:, :, 1:1 + w] - s["loss"][:, :, 1 + w:]) / (2 * h)                   # [B, T, w]
        # measured diagonal curvature of the losses -> one scale per period for the model Hessian 2 Re(Zbus)
        # (voltages below 1 pu and the losses' own feedback make the true curvature 10-40 % larger)
        d2 = (s["loss"][:, :, 1:1 + w] - 2 * s["loss"][:, :, :1] + s["loss"][:, :, 1 + w:]) / (h * h)
        kappa = (d2.sum(-1) / self.loss_hess.diagonal().sum()).clamp(0.5, 3.0)                # [B, T]
        return dict(v2=v2[:, :, 0], i2=i2[:, :, 0], loss=s["loss"][:, :, 0], jv=jv, ji=ji, gloss=gl, kappa=kappa)

    # -- objective and chain -----------------------------------------------------------------------------------
    def energy(self, e0, x):
        """opf.py:139-148: E[1] = E_init; E[t] = E[t-1] + dt (eta_ch Pesc[t] - Pesd[t]/eta_dis)."""
        inc = self.dt * (self.cfg["eta_ch"] * x[:, :, 2] - x[:, :, 3] / self.cfg["eta_dis"])
        inc = th.cat([th.zeros_like(inc[:, :1]), inc[:, 1:]], 1)
        return e0[:, None, :] + inc.cumsum(1)

    def objective(self, price, x, loss):
        c = self.cfg
        per = (price[:, :, None] * x[:, :, 0] - c["pv_cost"] * x[:, :, 1] - c["ess_cost"] * (x[:, :, 2] + x[:, :, 3])
               - c["discomfort_coeff"] * x[:, :, 0] ** 2).sum(-1) - loss
        return self.dt * per.sum(1)                                               # opf.py:80-93

    def bounds(self, pd, ppv):
        c = self.cfg
        B, T = pd.shape[:2]
        lo = th.zeros(B, T, 4, self.na, dtype=pd.dtype, device=pd.device)
        hi = th.zeros_like(lo)
        hi[:, :, 0] = pd[..., self.bld] * c["max_power_reduction"]                # opf.py:46,54
        lo[:, :, 1], hi[:, :, 1] = -self.tanphi * ppv, self.tanphi * ppv          # opf.py:118-120
        hi[:, :, 2], hi[:, :, 3] = c["p_ch_max"], c["p_dis_max"]                  # opf.py:56-57
        return lo, hi

    def _energy_matrix(self, T, dtype):
        """E (stacked [T, na]) = e0 + C x."""
        C = th.zeros(T, self.na, T, 4, self.na, dtype=dtype, device=self.device)
        for k in range(self.na):
            for s in range(1, T):
                C[s:, k, s, 2, k] = self.dt * self.cfg["eta_ch"]
                C[s:, k, s, 3, k] = -self.dt / self.cfg["eta_dis"]
        return C.view(T * self.na, T * self.w)

    # -- the solve ---------------------------------------------------------------------------------------------
    def solve(self, price, pd, qd, ppv, e0, max_outer=12, tol=1e-6, verbose=False):
        """price [B, T], pd/qd [B, T, n_bus], ppv [B, T, na], e0 [B, na] (device f64).  Returns a dict of tensors.
        Stops when no control moved by more than ``tol`` pu (1e-6 pu = 1 W on the 1 MVA base) in an outer iteration."""
        c = self.cfg
        f64 = dict(dtype=th.float64, device=self.device)
        price, pd, qd, ppv, e0 = (th.as_tensor(a, **f64) for a in (price, pd, qd, ppv, e0))
        B, T = price.shape
        w, n = self.w, T * self.w
        lo, hi = self.bounds(pd, ppv)
        x = lo.clone()
        x[:, :, 0] = th.minimum(price[:, :, None] / (2 * c["discomfort_coeff"]), hi[:, :, 0])
        history = []
        for outer in range(max_outer):
            lin = self.linearise(pd, qd, ppv, x)
            xk = x.reshape(B, n)
            # model:  minimise  -dt [ price Pred - pv Qpv - ess (ch + dis) - disc Pred^2 - loss_k - g'(x-xk) - 1/2 (x-xk)' H (x-xk) ]
            hess = lin["kappa"][:, :, None, None] * self.loss_hess                             # [B, T, w, w]
            Qblk = self.dt * hess
            idx = th.arange(self.na, device=self.device)
            Qblk[:, :, idx, idx] += 2 * self.dt * c["discomfort_coeff"]
            lin_c = th.zeros(B, T, 4, self.na, **f64)
            lin_c[:, :, 0] = -price[:, :, None]
            lin_c[:, :, 1] = c["pv_cost"]
            lin_c[:, :, 2] = c["ess_cost"]
            lin_c[:, :, 3] = c["ess_cost"]
            hk = th.einsum("btvw,btw->btv", hess, x.reshape(B, T, w))
            cvec = (self.dt * (lin_c.reshape(B, T, w) + lin["gloss"] - hk)).reshape(B, n)
            # linearised network limits in x:  J x <= u - g_k + J x_k
            jvx = th.einsum("btrw,btw->btr", lin["jv"], x.reshape(B, T, w))
            jix = th.einsum("btrw,btw->btr", lin["ji"], x.reshape(B, T, w))
            v_lo = (c["v_min"] ** 2 - lin["v2"] + jvx).reshape(B, -1)
            v_hi = (c["v_max"] ** 2 - lin["v2"] + jvx).reshape(B, -1)
            i_hi = (self.imax2 - lin["i2"] + jix).reshape(B, -1)
            e_lo = (c["e_min"] - e0)[:, None, :].expand(B, T, self.na).reshape(B, -1)
            e_hi = (c["e_max"] - e0)[:, None, :].expand(B, T, self.na).reshape(B, -1)
            # a control whose box is a point (Qpv at night) is pinned: it leaves the Newton system, and its box rows are
            # widened so that they stay inactive (the QP keeps an interior)
            free = ((hi - lo) >= 1e-9).reshape(B, n)
            pin = (~free).to(lo.dtype).reshape(B, T, 4, self.na)
            blocks = [(_Identity(), (lo - pin).reshape(B, n), (hi + pin).reshape(B, n)),
                      (_PeriodBlocks(lin["jv"]), v_lo, v_hi),
                      (_PeriodBlocks(lin["ji"]), None, i_hi),
                      (_EnergyChain(T, self.na, self.dt * c["eta_ch"], self.dt / c["eta_dis"]), e_lo, e_hi)]
            x0 = th.where(free, 0.5 * (lo + hi).reshape(B, n), lo.reshape(B, n))
            native = self.device.type == "cuda" and NATIVE_QP
            if native:
                from . import _lib
                if _lib.load().flexopf_qp_work_doubles(T, self.na, lin["jv"].shape[2]) < 0:
                    # sizes beyond the kernel's compile-time limits (include/flexopf.h: periods, rows, agents): the torch
                    # iteration solves them as it did before the native kernel existed — slower, same programme;
                    # counted and reported once (util.note_fallback)
                    from .util import note_fallback
                    note_fallback("opf.qp_ipm_native", f"T = {T}, rows = {lin['jv'].shape[2]}, agents = {self.na} outside "
                                  "flexopf_qp_solve's limits")
                    native = False
            if native:
                xn, info = qp_ipm_native(Qblk, cvec, (lo - pin).reshape(B, n), (hi + pin).reshape(B, n), free, lin["jv"], v_lo, v_hi,
                                         lin["ji"], i_hi, self.dt * c["eta_ch"], self.dt / c["eta_dis"], e_lo, e_hi, x0)
            else:
                xn, info = qp_ipm(Qblk, cvec, blocks, x0, free=free, verbose=verbose)
            xn = th.minimum(th.maximum(xn.view(B, T, 4, self.na), lo), hi)
            move = (xn - x).abs().amax().item()                                    # pu
            obj = self.objective(price, x, lin["loss"])
            history.append(dict(outer=outer, move=move, objective=obj.clone(), ipm_iters=info["iters"]))
            if verbose:
                print(f"outer {outer}: objective {obj.mean().item():+.9f}  max move {move:.2e}  ipm iters {info['iters']}")
            x = xn
            if move < tol:
                break
        fin_p, fin_q = self._net_loads(pd, qd, ppv, x)
        st = self._pf(fin_p, fin_q, want_branch=True)
        e = self.energy(e0, x)
        return dict(x=x, Pred=x[:, :, 0], Qpv=x[:, :, 1], Pesc=x[:, :, 2], Pesd=x[:, :, 3], E=e, Vsqr=st["v2"],
                    Isqr=st["i2"], Pl=st["pl"], Ql=st["ql"], loss=st["loss"], objective=self.objective(price, x, st["loss"]),
                    outer_iters=len(history), history=history, duals=info.get("duals"))


def opf_model(network_data, flex_price, active_power_demand, reactive_power_demand, pv_active_power,
              initial_ess_energy, env_args=None, device="cuda:0"):
    """Drop-in for ``utils/opf.py:13`` (same arguments, same solution dict; ``run_opf.py:71`` is the caller).
    ``flex_price`` {t: price} with t = 1..T; demands {bus: [T]}; PV {bus: [T]}; initial energy {bus: E}."""
    buses = list(network_data["bus_numbers"])
    T = len(flex_price)
    price = np.array([flex_price[t + 1] for t in range(T)])
    pd = np.array([[active_power_demand[b][t] for b in buses] for t in range(T)])
    qd = np.array([[reactive_power_demand[b][t] for b in buses] for t in range(T)])
    ppv = np.array([[pv_active_power[g][t] for g in network_data["PVs_at_buildings"]] for t in range(T)])
    e0 = np.array([initial_ess_energy[k] for k in network_data["ESSs_at_buildings"]])
    solver = BatchedOPF(network_data, env_args, device)
    r = solver.solve(price[None], pd[None], qd[None], ppv[None], e0[None])
    tb = solver.tables
    B_, G_, K_ = network_data["buildings"], network_data["PVs_at_buildings"], network_data["ESSs_at_buildings"]
    cpu = {k: v[0].cpu().numpy() for k, v in r.items() if isinstance(v, th.Tensor)}
    lines = [(i, tb.line_of_bus[i]) for i in range(tb.n_bus) if tb.line_of_bus[i] is not None]
    sol = {key: {} for key in ("Power Reduction", "PV Reactive Power", "ESS Charging", "ESS Discharging", "Voltage Squared",
                               "Active Power Flow", "Reactive Power Flow", "Current Squared", "ESS Energy",
                               "Charging Indicator", "Active Power Load", "Reactive Power Load", "PV Active Power")}
    for t in range(T):
        sol["Power Reduction"][t + 1] = {b: float(cpu["Pred"][t, i]) for i, b in enumerate(B_)}
        sol["PV Reactive Power"][t + 1] = {g: float(cpu["Qpv"][t, i]) for i, g in enumerate(G_)}
        sol["ESS Charging"][t + 1] = {k: float(cpu["Pesc"][t, i]) for i, k in enumerate(K_)}
        sol["ESS Discharging"][t + 1] = {k: float(cpu["Pesd"][t, i]) for i, k in enumerate(K_)}
        sol["Voltage Squared"][t + 1] = {b: float(cpu["Vsqr"][t, i]) for i, b in enumerate(buses)}
        sol["Active Power Flow"][t + 1] = {key: float(cpu["Pl"][t, i]) for i, key in lines}
        sol["Reactive Power Flow"][t + 1] = {key: float(cpu["Ql"][t, i]) for i, key in lines}
        sol["Current Squared"][t + 1] = {key: float(cpu["Isqr"][t, i]) for i, key in lines}
        sol["ESS Energy"][t + 1] = {k: float(cpu["E"][t, i]) for i, k in enumerate(K_)}
        sol["Charging Indicator"][t + 1] = {k: float(cpu["Pesc"][t, i] > cpu["Pesd"][t, i]) for i, k in enumerate(K_)}
        sol["Active Power Load"][t + 1] = {b: float(pd[t, i]) for i, b in enumerate(buses)}
        sol["Reactive Power Load"][t + 1] = {b: float(qd[t, i]) for i, b in enumerate(buses)}
        sol["PV Active Power"][t + 1] = {g: float(ppv[t, i]) for i, g in enumerate(G_)}
    return sol

"""Simulate pf_sweep (flex_device.h) in numpy to count sweeps, and try two-step extrapolations."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
import safe_marl_amd
from safe_marl_amd.network import create_network
from safe_marl_amd.series import make_synthetic_series
from oracle.pf_oracle import build_ybus
net = create_network()
s = make_synthetic_series(net, n_days=8)
Y = build_ybus(net)
Z = np.linalg.inv(Y[1:, 1:])
Zf = Z.astype(np.complex64)
COARSE, REANCHOR, MAXS = 1e-9, 8, 40

def cur(S, V): return np.conj(S / V)

def solve(S, V0, tol=2.5e-13, mode="base", mu=None, stats=None):
    """returns (V, sweeps).  mode: base | x1 (extrapolate at k=2 of each phase with mu) | est (estimate mu in phase 1)"""
    V = V0.copy(); it = 0; fine = False; mine = None
    Sf = S.astype(np.complex64)
    mu_hat = mu
    while it < MAXS:
        Ia = cur(S, V)
        V = 1.0 + Z @ Ia; it += 1
        Ib = cur(S, V)
        dI = Ia - Ib
        m = np.maximum(abs((V * np.conj(dI)).real), abs((V * np.conj(dI)).imag)).max()
        if m < tol: return V, it
        if not fine and m < COARSE: fine = True
        Va = V.copy(); Vaf = Va.astype(np.complex64)
        c = Zf @ (Ib - Ia).astype(np.complex64); it += 1
        d = c.copy(); pd = np.zeros(32, np.complex64)
        xs = [np.zeros(32, np.complex64)]   # x_0 = 0
        done = False
        k = 0
        while k < REANCHOR and it < MAXS:
            Vk = Vaf + d
            P = Vk * Vaf
            x = -np.conj(Sf * d / P)                  # delta(d)
            g = pd - x
            mm = np.maximum(abs((Vk * np.conj(g)).real), abs((Vk * np.conj(g)).imag)).max()
            if mm < np.float32(tol): done = True; break
            if not fine and mm < np.float32(COARSE): fine = True; break
            pd = x
            zx = Zf @ x; it += 1
            xs.append(zx)
            if mode in ("x1", "est") and k == 1 and mu_hat is not None:
                om = np.float32(mu_hat / (1 - mu_hat))
                zx = zx * (1 + om); pd = pd * (1 + om)
            if mode == "est" and k == 1 and mu_hat is None:
                # u0 = c, u2 = x_2 - x_1
                u0 = c.astype(np.complex128); u2 = (xs[2] - xs[1]).astype(np.complex128)
                mu_hat = float((np.vdot(u0, u2)).real / np.vdot(u0, u0).real)
                if stats is not None: stats.append(mu_hat)
                if 0 < mu_hat < 0.05:
                    om = np.float32(mu_hat / (1 - mu_hat))
                    zx = zx * (1 + om); pd = pd * (1 + om)
            d = c + zx
            k += 1
        V = Va + d.astype(np.complex128)
        if done: return V, it
    return V, it

def true_mu(S, V):
    M = -Z @ np.diag(np.conj(S) / np.conj(V) ** 2)
    ev = np.linalg.eigvals(M @ np.conj(M)); ev = ev[np.argsort(-abs(ev))]
    return ev[0].real, ev[1].real

rng = np.random.default_rng(1)
bld = [4, 9, 14, 19, 24]   # bus idx 5,10,.. minus slack -> pq index = bus-2 ... approx
def run(mode, n_env=40, n_step=12, use_true=False):
    tot = 0; cnt = 0; st = []
    for e in range(n_env):
        row = rng.integers(0, s.active.shape[0] - 100)
        V = np.ones(32, complex)
        for t in range(n_step):
            P = np.array(s.active[row + t])[1:].copy(); Q = np.array(s.reactive[row + t])[1:].copy()
            for b in bld:   # random actions: reduction, pv, ess
                P[b] += -P[b] * 0.5 * rng.uniform(0.5, 1) - 0.15 * rng.uniform(0, 1) * rng.uniform(0.7, 1) + rng.uniform(-0.005, 0.005)
                Q[b] -= rng.uniform(-0.03, 0.03)
            S = -(P + 1j * Q)
            # fp16 warm start of offsets
            V0 = 1 + (V.real - 1).astype(np.float16).astype(float) + 1j * V.imag.astype(np.float16).astype(float)
            mu = None
            if use_true:
                Vt, _ = solve(S, V0); mu = true_mu(S, Vt)[0]
            V, it = solve(S, V0, mode=mode, mu=mu, stats=st)
            # check
            res = V - (1 + Z @ cur(S, V))
            assert abs(res).max() < 1e-11, abs(res).max()
            if t > 0: tot += it; cnt += 1
    return tot / cnt, (np.mean(st), np.std(st)) if st else None

for mode, ut in (("base", False), ("x1", True), ("est", False)):
    rng = np.random.default_rng(1)
    print(mode, ut, run(mode, use_true=ut))

print("---- variants")
def solve2(S, V0, tol=2.5e-13, mus=(), sched=(1,), coarse=COARSE, reanchor=REANCHOR):
    """general: at increments k in `sched` apply two-step extrapolation with mus[0]; if len(mus)==2 and k>=3 apply the
    two-eigenvalue (4-step) formula instead at k==3"""
    V = V0.copy(); it = 0; fine = False
    Sf = S.astype(np.complex64)
    while it < MAXS:
        Ia = cur(S, V); V = 1.0 + Z @ Ia; it += 1
        Ib = cur(S, V); dI = Ia - Ib
        m = np.maximum(abs((V * np.conj(dI)).real), abs((V * np.conj(dI)).imag)).max()
        if m < tol: return V, it
        if not fine and m < coarse: fine = True
        Va = V.copy(); Vaf = Va.astype(np.complex64)
        c = Zf @ (Ib - Ia).astype(np.complex64); it += 1
        d = c.copy(); pd = np.zeros(32, np.complex64)
        hist_x = {0: np.zeros(32, np.complex64)}; hist_pd = {0: np.zeros(32, np.complex64)}   # x_j, delta_j with d_{j+1} = c + x_j
        done = False; k = 0
        while k < reanchor and it < MAXS:
            Vk = Vaf + d; P = Vk * Vaf
            x = -np.conj(Sf * d / P); g = pd - x
            mm = np.maximum(abs((Vk * np.conj(g)).real), abs((Vk * np.conj(g)).imag)).max()
            if mm < np.float32(tol): done = True; break
            if not fine and mm < np.float32(coarse): fine = True; break
            pd = x; zx = Zf @ x; it += 1
            j = k + 1                      # this is x_j: d_{j+1} = c + x_j
            if j in [q + 1 for q in sched] and len(mus) >= 1:
                if len(mus) == 2 and (j - 4) in hist_x:
                    a, b = mus; den = (1 - a) * (1 - b)
                    zx2 = (zx - (a + b) * hist_x[j - 2] + a * b * hist_x[j - 4]) / den + 0 * zx
                    # d* = (d_{j+1} - (a+b) d_{j-1} + ab d_{j-3})/den ; with d = c + x: c*(1-(a+b)+ab)/den = c
                    pd = ((pd - (a + b) * hist_pd[j - 2] + a * b * hist_pd[j - 4]) / den).astype(np.complex64)
                    zx = zx2.astype(np.complex64)
                else:
                    om = np.float32(mus[0] / (1 - mus[0]))
                    zx = (zx + om * (zx - hist_x[j - 2])).astype(np.complex64)
                    pd = (pd + om * (pd - hist_pd[j - 2])).astype(np.complex64)
            hist_x[j] = zx; hist_pd[j] = pd
            d = c + zx; k += 1
        V = Va + d.astype(np.complex128)
        if done: return V, it
    return V, it

def run2(label, n_env=40, n_step=12, **kw):
    rng = np.random.default_rng(1)
    tot = cnt = 0; prev_mu = None
    for e in range(n_env):
        row = rng.integers(0, s.active.shape[0] - 100)
        V = np.ones(32, complex)
        for t in range(n_step):
            P = np.array(s.active[row + t])[1:].copy(); Q = np.array(s.reactive[row + t])[1:].copy()
            for b in bld:
                P[b] += -P[b] * 0.5 * rng.uniform(0.5, 1) - 0.15 * rng.uniform(0, 1) * rng.uniform(0.7, 1) + rng.uniform(-0.005, 0.005)
                Q[b] -= rng.uniform(-0.03, 0.03)
            S = -(P + 1j * Q)
            V0 = 1 + (V.real - 1).astype(np.float16).astype(float) + 1j * V.imag.astype(np.float16).astype(float)
            Vt, _ = solve(S, V0); tm = true_mu(S, Vt)
            which = kw.get("which", "true")
            if which == "true": mus = tm[:kw.get("nmu", 1)]
            elif which == "lag": mus = (prev_mu if prev_mu is not None else tm)[:kw.get("nmu", 1)]
            elif which == "const": mus = kw["const"]
            elif which == "none": mus = ()
            prev_mu = tm
            V, it = solve2(S, V0, mus=mus, sched=kw.get("sched", (1,)), coarse=kw.get("coarse", COARSE), reanchor=kw.get("reanchor", REANCHOR))
            res = V - (1 + Z @ cur(S, V)); assert abs(res).max() < 1e-11, abs(res).max()
            if t > 0: tot += it; cnt += 1
    print(label, round(tot / cnt, 3))

run2("none", which="none")
run2("true k=1", which="true")
run2("true k=1,3", which="true", sched=(1, 3))
run2("true k=1,3,5", which="true", sched=(1, 3, 5))
run2("lag k=1", which="lag")
run2("lag k=1,3", which="lag", sched=(1, 3))
run2("const .0024 k=1,3", which="const", const=(0.0024,), sched=(1, 3))
run2("true 2mu k=1,3", which="true", nmu=2, sched=(1, 3))
run2("lag 2mu k=1,3", which="lag", nmu=2, sched=(1, 3))
run2("const 2mu k=1,3", which="const", const=(0.0024, 0.0004), sched=(1, 3))

print("---- proxy / mixed schedule")
def solve3(S, V0, tol=2.5e-13, plan=None, kappa=None, lane=31, coarse=COARSE, reanchor=REANCHOR, mu_exact=None):
    """plan: {k: which}  which in {'mu1','mu2'}; mu from proxy kappa*|1-Va[lane]|^2 (mu1) or mu_exact"""
    V = V0.copy(); it = 0; fine = False
    Sf = S.astype(np.complex64)
    while it < MAXS:
        Ia = cur(S, V); V = 1.0 + Z @ Ia; it += 1
        Ib = cur(S, V); dI = Ia - Ib
        m = np.maximum(abs((V * np.conj(dI)).real), abs((V * np.conj(dI)).imag)).max()
        if m < tol: return V, it
        if not fine and m < coarse: fine = True
        Va = V.copy(); Vaf = Va.astype(np.complex64)
        if mu_exact is not None: mus = {"mu1": mu_exact[0], "mu2": mu_exact[1]}
        else:
            base = abs(1 - Va[lane]) ** 2
            mus = {"mu1": kappa[0] * base, "mu2": kappa[1] * base}
        c = Zf @ (Ib - Ia).astype(np.complex64); it += 1
        d = c.copy(); pd = np.zeros(32, np.complex64)
        hx = {0: np.zeros(32, np.complex64)}; hp = {0: np.zeros(32, np.complex64)}
        done = False; k = 0
        while k < reanchor and it < MAXS:
            Vk = Vaf + d; P = Vk * Vaf
            x = -np.conj(Sf * d / P); g = pd - x
            mm = np.maximum(abs((Vk * np.conj(g)).real), abs((Vk * np.conj(g)).imag)).max()
            if mm < np.float32(tol): done = True; break
            if not fine and mm < np.float32(coarse): fine = True; break
            pd = x; zx = Zf @ x; it += 1
            j = k + 1
            if plan and k in plan:
                mu = mus[plan[k]]; om = np.float32(mu / (1 - mu))
                zx = (zx + om * (zx - hx[j - 2])).astype(np.complex64)
                pd = (pd + om * (pd - hp[j - 2])).astype(np.complex64)
            hx[j] = zx; hp[j] = pd
            d = c + zx; k += 1
        V = Va + d.astype(np.complex128)
        if done: return V, it
    return V, it

def run3(label, n_env=40, n_step=12, exact=False, **kw):
    rng = np.random.default_rng(1)
    tot = cnt = 0; hist = {}
    for e in range(n_env):
        row = rng.integers(0, s.active.shape[0] - 100)
        V = np.ones(32, complex)
        for t in range(n_step):
            P = np.array(s.active[row + t])[1:].copy(); Q = np.array(s.reactive[row + t])[1:].copy()
            for b in bld:
                P[b] += -P[b] * 0.5 * rng.uniform(0.5, 1) - 0.15 * rng.uniform(0, 1) * rng.uniform(0.7, 1) + rng.uniform(-0.005, 0.005)
                Q[b] -= rng.uniform(-0.03, 0.03)
            S = -(P + 1j * Q)
            V0 = 1 + (V.real - 1).astype(np.float16).astype(float) + 1j * V.imag.astype(np.float16).astype(float)
            me = None
            if exact:
                Vt, _ = solve(S, V0); me = true_mu(S, Vt)
            V, it = solve3(S, V0, mu_exact=me, **kw)
            res = V - (1 + Z @ cur(S, V)); assert abs(res).max() < 1e-11, abs(res).max()
            if t > 0: tot += it; cnt += 1; hist[it] = hist.get(it, 0) + 1
    print(label, round(tot / cnt, 3), dict(sorted(hist.items())))

run3("none", plan=None, kappa=(0.784, 0.13))
run3("exact mu1@1", exact=True, plan={1: "mu1"})
run3("exact mu1@1 mu2@3", exact=True, plan={1: "mu1", 3: "mu2"})
run3("exact mu1@1 mu2@2", exact=True, plan={1: "mu1", 2: "mu2"})
run3("proxy mu1@1", plan={1: "mu1"}, kappa=(0.784, 0.13))
run3("proxy mu1@1 mu2@3", plan={1: "mu1", 3: "mu2"}, kappa=(0.784, 0.13))
run3("proxy mu1@1, tol 5e-13", plan={1: "mu1"}, kappa=(0.784, 0.13), tol=5e-13)
run3("proxy mu1@1, coarse 1e-8", plan={1: "mu1"}, kappa=(0.784, 0.13), coarse=1e-8)
run3("proxy mu1@1, coarse 1e-10", plan={1: "mu1"}, kappa=(0.784, 0.13), coarse=1e-10)

print("---- projected estimate")
# reference eigenvectors from the base loads (no actions), row 0
P0 = np.array(s.active[500])[1:]; Q0 = np.array(s.reactive[500])[1:]
S0 = -(P0 + 1j * Q0); Vb, _ = solve(S0, np.ones(32, complex))
M0 = -Z @ np.diag(np.conj(S0) / np.conj(Vb) ** 2); A2 = M0 @ np.conj(M0)
ev, vr = np.linalg.eig(A2); i0 = np.argmax(abs(ev)); v1 = vr[:, i0]
evl, vl = np.linalg.eig(A2.conj().T); j0 = np.argmax(abs(evl)); w1 = vl[:, j0]
w1 = w1 / np.vdot(w1, v1).conj()
w1f = w1.astype(np.complex64)

def solve4(S, V0, tol=2.5e-13, coarse=COARSE, reanchor=REANCHOR, stats=None, mu_true=None, second=True):
    V = V0.copy(); it = 0; fine = False; mu_hat = None
    Sf = S.astype(np.complex64)
    while it < MAXS:
        Ia = cur(S, V); V = 1.0 + Z @ Ia; it += 1
        Ib = cur(S, V); dI = Ia - Ib
        m = np.maximum(abs((V * np.conj(dI)).real), abs((V * np.conj(dI)).imag)).max()
        if m < tol: return V, it
        if not fine and m < coarse: fine = True
        Va = V.copy(); Vaf = Va.astype(np.complex64)
        c = Zf @ (Ib - Ia).astype(np.complex64); it += 1
        d = c.copy(); pd = np.zeros(32, np.complex64)
        hx = {0: np.zeros(32, np.complex64)}; hp = {0: np.zeros(32, np.complex64)}
        done = False; k = 0
        while k < reanchor and it < MAXS:
            Vk = Vaf + d; P = Vk * Vaf
            x = -np.conj(Sf * d / P); g = pd - x
            mm = np.maximum(abs((Vk * np.conj(g)).real), abs((Vk * np.conj(g)).imag)).max()
            if mm < np.float32(tol): done = True; break
            if not fine and mm < np.float32(coarse): fine = True; break
            pd = x; zx = Zf @ x; it += 1
            j = k + 1
            if k == 1:
                if mu_hat is None:
                    u0 = c; u2 = zx - hx[1]
                    num = np.vdot(w1f, u2); den = np.vdot(w1f, u0)
                    mu_hat = float((num / den).real)
                    if stats is not None and mu_true is not None: stats.append(mu_hat / mu_true)
                    if not (0 < mu_hat < 0.02): mu_hat = 0.0
                if mu_hat > 0 and (second or it < 6):
                    om = np.float32(mu_hat / (1 - mu_hat))
                    zx = (zx + om * (zx - hx[j - 2])).astype(np.complex64)
                    pd = (pd + om * (pd - hp[j - 2])).astype(np.complex64)
            hx[j] = zx; hp[j] = pd
            d = c + zx; k += 1
        V = Va + d.astype(np.complex128)
        if done: return V, it
    return V, it

def run4(label, n_env=40, n_step=12, **kw):
    rng = np.random.default_rng(1)
    tot = cnt = 0; hist = {}; st = []
    for e in range(n_env):
        row = rng.integers(0, s.active.shape[0] - 100)
        V = np.ones(32, complex)
        for t in range(n_step):
            P = np.array(s.active[row + t])[1:].copy(); Q = np.array(s.reactive[row + t])[1:].copy()
            for b in bld:
                P[b] += -P[b] * 0.5 * rng.uniform(0.5, 1) - 0.15 * rng.uniform(0, 1) * rng.uniform(0.7, 1) + rng.uniform(-0.005, 0.005)
                Q[b] -= rng.uniform(-0.03, 0.03)
            S = -(P + 1j * Q)
            V0 = 1 + (V.real - 1).astype(np.float16).astype(float) + 1j * V.imag.astype(np.float16).astype(float)
            Vt, _ = solve(S, V0); mt = true_mu(S, Vt)[0]
            V, it = solve4(S, V0, stats=st, mu_true=mt, **kw)
            res = V - (1 + Z @ cur(S, V)); assert abs(res).max() < 1e-11, abs(res).max()
            if t > 0: tot += it; cnt += 1; hist[it] = hist.get(it, 0) + 1
    st = np.array(st)
    print(label, round(tot / cnt, 3), dict(sorted(hist.items())), "mu_hat/mu: mean %.3f std %.3f" % (st.mean(), st.std()))

run4("projected est, both phases")
run4("projected est, tol 5e-13", tol=5e-13)

/*
 * TEST INFRASTRUCTURE — plain-C restatement of the environment step and power flow, used as the
 * large-N checker and as bench.py's cpu_baseline ("port").  Not product code: nothing under
 * safe-marl_amd/ may link or call it.
 *
 * PINNED (round 4) to runs of the reference's own environment code for everything but the numerical
 * solve: tests/test_env_golden_cpu.py holds it to tests/golden/env_golden.npz — whole episodes of the
 * reference's flexibility_provision_env.py executed from /root/reference with its power_flow_solver call
 * rebound to oracle/pf_oracle.py (pyomo / IPOPT absent; tests/golden/make_env_golden.py) — at 1e-11.
 * PARITY UNPINNED for the solve itself against IPOPT: there it is pinned against oracle/pf_oracle.py
 * (tests/test_oracle_cpu.py), in turn by two-algorithm agreement, the reference's own constraint
 * residuals and its NLP statement through SciPy (oracle/pf_nlp_oracle.py).
 *
 * Algorithm: Newton-Raphson in POLAR form on the dense Ybus with a dense partial-pivot LU — the
 * textbook formulation, deliberately different from the HIP kernel's rectangular current-mismatch /
 * tree-elimination form so that agreement between the two means something.
 *
 * Restates (paths under /root/reference):
 *   utils/pf.py:58-98            DistFlow equations -> same fixed point as AC power flow on the Ybus
 *   flexibility_provision_env.py:241-356 step, :370-403 get_obs, :609-706 helpers/reward
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OMAXB 64
#define OMAXA 8

typedef struct {
    int32_t n_bus, n_agents, history, episode_limit, raw_actions, pf_max_iter, slack, n_lines;
    int32_t pf_solver, pad0;      /* 0: dense polar Newton-Raphson (default, the checker); 1: DistFlow backward/forward sweep */
    double v_min, v_max, e_min, e_max, p_ch_max, p_dis_max, eta_ch, eta_dis, tan_phi, max_power_reduction;
    double pv_cost, ess_cost, discomfort_coeff, voltage_coeff, dt, fail_penalty, pf_tol;
    int32_t agent_bus[OMAXA];
    int32_t line_from[OMAXB], line_to[OMAXB];
    double line_r[OMAXB], line_x[OMAXB];
} OCfg;

typedef struct {
    double G[OMAXB][OMAXB], B[OMAXB][OMAXB];
} OYbus;

static void build_ybus(const OCfg* c, OYbus* y) {
    memset(y, 0, sizeof(*y));
    for (int l = 0; l < c->n_lines; ++l) {
        const int i = c->line_from[l], j = c->line_to[l];
        const double r = c->line_r[l], x = c->line_x[l], z2 = r * r + x * x;
        const double g = r / z2, b = -x / z2;
        y->G[i][i] += g; y->B[i][i] += b; y->G[j][j] += g; y->B[j][j] += b;
        y->G[i][j] -= g; y->B[i][j] -= b; y->G[j][i] -= g; y->B[j][i] -= b;
    }
}

/* dense LU with partial pivoting, solves A x = b in place (n <= 2*OMAXB) */
static int lu_solve(int n, double A[2 * OMAXB][2 * OMAXB], double* b) {
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = fabs(A[k][k]);
        for (int i = k + 1; i < n; ++i) if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); p = i; }
        if (!(best > 0.0)) return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        const double inv = 1.0 / A[k][k];
        for (int i = k + 1; i < n; ++i) {
            const double m = A[i][k] * inv;
            if (m == 0.0) continue;
            for (int j = k + 1; j < n; ++j) A[i][j] -= m * A[k][j];
            b[i] -= m * b[k];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < n; ++j) s -= A[i][j] * b[j];
        b[i] = s / A[i][i];
    }
    return 0;
}

/* returns iterations (>=0) on convergence, -1 on failure; vm out */
static int pf_polar(const OCfg* c, const OYbus* y, const double* pnet, const double* qnet, double* vm) {
    const int n = c->n_bus, m = n - 1;
    int pq[OMAXB];
    for (int i = 0, k = 0; i < n; ++i) if (i != c->slack) pq[k++] = i;
    double va[OMAXB], v[OMAXB], P[OMAXB], Q[OMAXB];
    static _Thread_local double J[2 * OMAXB][2 * OMAXB];
    double rhs[2 * OMAXB];
    for (int i = 0; i < n; ++i) { v[i] = 1.0; va[i] = 0.0; }
    for (int it = 0; it <= c->pf_max_iter; ++it) {
        double err = 0.0;
        for (int i = 0; i < n; ++i) {
            double p = 0.0, q = 0.0;
            for (int k = 0; k < n; ++k) {
                const double g = y->G[i][k], b = y->B[i][k];
                if (g == 0.0 && b == 0.0) continue;
                const double th = va[i] - va[k], cs = cos(th), sn = sin(th);
                p += v[i] * v[k] * (g * cs + b * sn);
                q += v[i] * v[k] * (g * sn - b * cs);
            }
            P[i] = p; Q[i] = q;
        }
        for (int a = 0; a < m; ++a) {
            const int i = pq[a];
            rhs[a] = -pnet[i] - P[i];
            rhs[m + a] = -qnet[i] - Q[i];
            const double e1 = fabs(rhs[a]), e2 = fabs(rhs[m + a]);
            if (!(e1 == e1) || !(e2 == e2)) return -1;
            if (e1 > err) err = e1;
            if (e2 > err) err = e2;
        }
        if (err < c->pf_tol) { memcpy(vm, v, n * sizeof(double)); return it; }
        if (it == c->pf_max_iter) break;
        for (int a = 0; a < m; ++a) {
            const int i = pq[a];
            for (int bb = 0; bb < m; ++bb) {
                const int k = pq[bb];
                const double g = y->G[i][k], b = y->B[i][k];
                if (i == k) {
                    J[a][bb] = -Q[i] - b * v[i] * v[i];
                    J[a][m + bb] = P[i] / v[i] + g * v[i];
                    J[m + a][bb] = P[i] - g * v[i] * v[i];
                    J[m + a][m + bb] = Q[i] / v[i] - b * v[i];
                } else if (g == 0.0 && b == 0.0) {
                    J[a][bb] = J[a][m + bb] = J[m + a][bb] = J[m + a][m + bb] = 0.0;
                } else {
                    const double th = va[i] - va[k], cs = cos(th), sn = sin(th);
                    J[a][bb] = v[i] * v[k] * (g * sn - b * cs);
                    J[a][m + bb] = v[i] * (g * cs + b * sn);
                    J[m + a][bb] = -v[i] * v[k] * (g * cs + b * sn);
                    J[m + a][m + bb] = v[i] * (g * sn - b * cs);
                }
            }
        }
        if (lu_solve(2 * m, J, rhs) != 0) return -1;
        for (int a = 0; a < m; ++a) { va[pq[a]] += rhs[a]; v[pq[a]] += rhs[m + a]; }
    }
    return -1;
}

/* ---- second solver (pf_solver = 1): backward/forward sweep on the DistFlow recursion in the reference's own variables
 * (utils/pf.py:65-94: Pl, Ql, Isqr, Vsqr; same iteration as oracle/pf_oracle.py distflow_sweep) — O(n) per
 * iteration like the HIP kernel's sweeps, where the dense Newton-Raphson above is O(n^3): bench.py times both so that the
 * CPU figure beside the GPU's is like for like.  Cold start (Vsqr = 1, Isqr = 0), iterated until no
 * Vsqr / Isqr moves by 1e-13. */
typedef struct { int32_t order[OMAXB], parent[OMAXB]; double R[OMAXB], X[OMAXB]; } OTree;
typedef struct { OYbus y; OTree t; } ONet;

static void build_tree(const OCfg* c, OTree* t) {
    const int n = c->n_bus;
    int seen[OMAXB] = {0};
    int head = 0, tail = 0;
    t->order[tail++] = c->slack; t->parent[c->slack] = -1; seen[c->slack] = 1;
    t->R[c->slack] = t->X[c->slack] = 0.0;
    while (head < tail) {
        const int u = t->order[head++];
        for (int l = 0; l < c->n_lines; ++l) {
            const int a = c->line_from[l], b = c->line_to[l];
            const int v = a == u ? b : (b == u ? a : -1);
            if (v < 0 || seen[v]) continue;
            seen[v] = 1; t->parent[v] = u; t->R[v] = c->line_r[l]; t->X[v] = c->line_x[l];
            t->order[tail++] = v;
        }
    }
    for (int i = tail; i < n; ++i) t->order[i] = -1;     /* unreachable buses (not a feeder): the solve reports failure */
}

static int pf_distflow(const OCfg* c, const OTree* t, const double* pnet, const double* qnet, double* vm) {
    const int n = c->n_bus;
    double vs[OMAXB], isq[OMAXB], P[OMAXB], Q[OMAXB];
    if (t->order[n - 1] < 0) return -1;
    for (int i = 0; i < n; ++i) { vs[i] = 1.0; isq[i] = 0.0; }
    for (int it = 1; it <= 500; ++it) {
        for (int i = 0; i < n; ++i) { P[i] = pnet[i]; Q[i] = qnet[i]; }
        for (int k = n - 1; k >= 1; --k) {                /* leaf -> root: power received + losses of the children's lines */
            const int v = t->order[k], u = t->parent[v];
            P[u] += P[v] + t->R[v] * isq[v];
            Q[u] += Q[v] + t->X[v] * isq[v];
        }
        double delta = 0.0;
        for (int k = 1; k < n; ++k) {                     /* root -> leaf: pf.py:85-94 for one line given Vsqr[parent] */
            const int v = t->order[k], u = t->parent[v];
            const double s2 = P[v] * P[v] + Q[v] * Q[v];
            const double a = vs[u] - 2.0 * (t->R[v] * P[v] + t->X[v] * Q[v]);
            const double cc = (t->R[v] * t->R[v] + t->X[v] * t->X[v]) * s2;
            const double disc = a * a - 4.0 * cc;
            if (!(disc >= 0.0)) return -1;                /* no real root: voltage collapse (or NaN) */
            const double vn = 0.5 * (a + sqrt(disc)), in = s2 / vn;
            const double d1 = fabs(vn - vs[v]), d2 = fabs(in - isq[v]);
            if (d1 > delta) delta = d1;
            if (d2 > delta) delta = d2;
            vs[v] = vn; isq[v] = in;
        }
        if (!(delta == delta)) return -1;
        if (delta < 1e-13) {
            for (int i = 0; i < n; ++i) vm[i] = sqrt(vs[i]);
            return it;
        }
    }
    return -1;
}

static int pf_any(const OCfg* c, const ONet* net, const double* pnet, const double* qnet, double* vm) {
    return c->pf_solver == 1 ? pf_distflow(c, &net->t, pnet, qnet, vm) : pf_polar(c, &net->y, pnet, qnet, vm);
}
static ONet* build_net(const OCfg* c) {
    ONet* net = (ONet*)malloc(sizeof(ONet));
    build_ybus(c, &net->y);
    build_tree(c, &net->t);
    return net;
}

int oracle_pf_batch(const OCfg* c, int n, const double* pnet, const double* qnet, double* vm, int32_t* iters) {
    ONet* y = build_net(c);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        iters[i] = pf_any(c, y, pnet + (size_t)i * c->n_bus, qnet + (size_t)i * c->n_bus, vm + (size_t)i * c->n_bus);
    free(y);
    return 0;
}

static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* env:628-661 */
static void clip_charge(const OCfg* c, double* ch, double* dis, double e) {
    *ch = clipd(*ch, 0, c->p_ch_max);
    *dis = clipd(*dis, 0, c->p_dis_max);
    double e_next = e + c->eta_ch * *ch - (1 / c->eta_dis) * *dis;
    if (e_next > c->e_max) {
        double excess = e_next - c->e_max;
        if (*ch > excess / c->eta_ch) *ch -= excess / c->eta_ch;
        else { *dis += (excess - *ch * c->eta_ch) * c->eta_dis; *ch = 0; }
    } else if (e_next < c->e_min) {
        double lack = c->e_min - e_next;
        if (*dis > lack * c->eta_dis) *dis -= lack * c->eta_dis;
        else { *ch += (lack - *dis / c->eta_dis) / c->eta_ch; *dis = 0; }
    }
    *ch = clipd(*ch, 0, c->p_ch_max);
    *dis = clipd(*dis, 0, c->p_dis_max);
}

/* Per-env state, struct-of-arrays owned by the caller (numpy):
 *   V[N][n_bus] E[N][na] Einit[N][na] act[N][4][na] (pred,ch,dis,q) cum[N]
 *   steps[N] start[N] row[N] obscnt[N]; hist[N][na][H][6] (double)
 * series: [rows][2*n_bus+na+1] */
typedef struct {
    double *V, *E, *Einit, *act, *cum, *hist;
    int32_t *steps, *start, *row, *obscnt;
    const double* series;
    int64_t rows;
    int32_t cols;
} OState;

static void parse(const OCfg* c, int raw, const double* a, double pd, double ppv, double e_clip,
                  double* pred, double* ch, double* dis, double* q) {
    double pr, cc, dd, qq;
    if (raw) { pr = a[0]; cc = a[1]; dd = a[2]; qq = a[3]; }
    else {
        pr = c->max_power_reduction * a[0]; cc = c->p_ch_max * a[1]; dd = c->p_dis_max * a[2];
        double lim = c->tan_phi * ppv;
        qq = clipd(-lim + a[3] * (lim - (-lim)), -lim, lim);
    }
    pr = clipd(pr, 0, c->max_power_reduction);
    if (cc > 0 && dd > 0) { if (cc > dd) { cc -= dd; dd = 0; } else { dd -= cc; cc = 0; } }
    clip_charge(c, &cc, &dd, e_clip);
    *pred = pd * pr; *ch = cc; *dis = dd; *q = qq;
}

static void push_obs(const OCfg* c, OState* s, int i, float* obs) {
    const int nb = c->n_bus, na = c->n_agents, H = c->history;
    const double* sr = s->series + (size_t)s->row[i] * s->cols;
    const int k = s->obscnt[i];
    double* hist = s->hist + (size_t)i * na * H * 6;
    for (int a = 0; a < na; ++a) {
        const int b = c->agent_bus[a];
        double* slot = hist + ((size_t)a * H + (k % H)) * 6;
        slot[0] = sr[b]; slot[1] = sr[nb + b]; slot[2] = sr[2 * nb + a];
        slot[3] = s->V[(size_t)i * nb + b]; slot[4] = sr[2 * nb + na]; slot[5] = s->E[(size_t)i * na + a];
        if (obs) {
            float* o = obs + ((size_t)i * na + a) * H * 6;
            for (int h = 0; h < H; ++h) {
                const int src = k - (H - 1) + h;
                for (int f = 0; f < 6; ++f)
                    o[h * 6 + f] = src < 0 ? 0.0f : (float)hist[((size_t)a * H + (src % H)) * 6 + f];
            }
        }
    }
    s->obscnt[i] = k + 1;
}

#define ODOMAIN_EPS 1e-8
static int solve_env(const OCfg* c, const ONet* y, const OState* s, int i, int row, const double* pred,
                     const double* ch, const double* dis, const double* q, const double* e_init, double* vm) {
    const int nb = c->n_bus, na = c->n_agents;
    const double* sr = s->series + (size_t)row * s->cols;
    double pnet[OMAXB], qnet[OMAXB];
    for (int b = 0; b < nb; ++b) { pnet[b] = sr[b]; qnet[b] = sr[nb + b]; }
    for (int a = 0; a < na; ++a) {
        const int b = c->agent_bus[a];
        pnet[b] += -pred[a] - sr[2 * nb + a] + ch[a] - dis[a];
        qnet[b] -= q[a];
    }
    (void)i;
    /* pf.py:41-45: E_next is declared NonNegativeReals and pinned by the equality pf.py:96-98 -> a negative value makes the
       NLP infeasible -> pf.py:104-105 raises.  ODOMAIN_EPS = IPOPT's default bound relaxation (oracle/pf_oracle.py). */
    for (int a = 0; a < na; ++a) {
        const double en = e_init[a] + c->dt * (c->eta_ch * ch[a] - (1 / c->eta_dis) * dis[a]);
        if (!(en >= -ODOMAIN_EPS)) return -1;
    }
    return pf_any(c, y, pnet, qnet, vm);
}

/* reset with injected draws: start[N], e0[N][na], a0[N][4na]; returns number of failed envs */
int oracle_env_reset_batch(const OCfg* c, OState* s, int n, const int32_t* start, const double* e0,
                           const double* a0, float* obs, uint8_t* failed) {
    ONet* y = build_net(c);
    const int nb = c->n_bus, na = c->n_agents;
    int nfail = 0;
#pragma omp parallel for schedule(static) reduction(+ : nfail)
    for (int i = 0; i < n; ++i) {
        s->steps[i] = 1; s->cum[i] = 0.0; s->obscnt[i] = 0; s->start[i] = start[i];
        s->row[i] = start[i] + 1;
        const double* sr = s->series + (size_t)s->row[i] * s->cols;
        double pred[OMAXA], ch[OMAXA], dis[OMAXA], q[OMAXA], vm[OMAXB];
        for (int a = 0; a < na; ++a)
            parse(c, 0, a0 + ((size_t)i * na + a) * 4, sr[c->agent_bus[a]], sr[2 * nb + a], e0[(size_t)i * na + a],
                  &pred[a], &ch[a], &dis[a], &q[a]);
        const int it = solve_env(c, y, s, i, s->row[i], pred, ch, dis, q, e0 + (size_t)i * na, vm);
        failed[i] = it < 0;
        nfail += it < 0;
        if (it >= 0) memcpy(s->V + (size_t)i * nb, vm, nb * sizeof(double));
        for (int a = 0; a < na; ++a) {
            s->Einit[(size_t)i * na + a] = e0[(size_t)i * na + a];
            s->E[(size_t)i * na + a] = e0[(size_t)i * na + a] + c->dt * (c->eta_ch * ch[a] - (1 / c->eta_dis) * dis[a]);
            double* act = s->act + (size_t)i * 4 * na;
            act[0 * na + a] = pred[a]; act[1 * na + a] = ch[a]; act[2 * na + a] = dis[a]; act[3 * na + a] = q[a];
        }
        push_obs(c, s, i, obs);
    }
    free(y);
    return nfail;
}

/* step + get_obs for n envs; actions [n][na][4] double; info [n][7] */
int oracle_env_step_batch(const OCfg* c, OState* s, int n, const double* actions, double* reward, uint8_t* done,
                          double* info, uint8_t* failed, float* obs) {
    ONet* y = build_net(c);
    const int nb = c->n_bus, na = c->n_agents;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const double* sr = s->series + (size_t)s->row[i] * s->cols;
        double pred[OMAXA], ch[OMAXA], dis[OMAXA], q[OMAXA], vm[OMAXB];
        double* act = s->act + (size_t)i * 4 * na;
        for (int a = 0; a < na; ++a)
            parse(c, c->raw_actions, actions + ((size_t)i * na + a) * 4, sr[c->agent_bus[a]], sr[2 * nb + a],
                  s->E[(size_t)i * na + a], &pred[a], &ch[a], &dis[a], &q[a]);
        const int it = solve_env(c, y, s, i, s->row[i], pred, ch, dis, q, s->Einit + (size_t)i * na, vm);
        const int ok = it >= 0;
        if (ok) {
            memcpy(s->V + (size_t)i * nb, vm, nb * sizeof(double));
            for (int a = 0; a < na; ++a) {
                s->E[(size_t)i * na + a] = s->Einit[(size_t)i * na + a] + c->dt * (c->eta_ch * ch[a] - (1 / c->eta_dis) * dis[a]);
                act[0 * na + a] = pred[a]; act[1 * na + a] = ch[a]; act[2 * na + a] = dis[a]; act[3 * na + a] = q[a];
            }
        }
        const double price = sr[2 * nb + na];
        double revenue = 0, der = 0, ess = 0, disc = 0, vpen = 0;
        for (int a = 0; a < na; ++a) {
            revenue += price * act[0 * na + a];
            der += c->pv_cost * act[3 * na + a];
            ess += c->ess_cost * (act[1 * na + a] + act[2 * na + a]);
            disc += c->discomfort_coeff * act[0 * na + a] * act[0 * na + a];
        }
        for (int b = 0; b < nb; ++b) {
            const double v = s->V[(size_t)i * nb + b];
            const double over = v - c->v_max, under = c->v_min - v;
            const double mx = over > under ? over : under;
            vpen += c->voltage_coeff * (mx > 0 ? mx : 0);
        }
        double r = revenue - der - ess - disc - vpen;
        if (info) {
            double* io = info + (size_t)i * 7;
            io[0] = r; io[1] = revenue; io[2] = der; io[3] = ess; io[4] = disc; io[5] = vpen; io[6] = s->cum[i];
        }
        if (!ok) r -= c->fail_penalty;
        int64_t nr = (int64_t)s->start[i] + s->steps[i];
        if (nr >= s->rows) nr = s->rows - 1;
        s->row[i] = (int32_t)nr;
        s->steps[i] += 1;
        s->cum[i] += r;
        reward[i] = r;
        done[i] = (s->steps[i] >= c->episode_limit) || !ok;
        if (failed) failed[i] = !ok;
        for (int a = 0; a < na; ++a) s->Einit[(size_t)i * na + a] = s->E[(size_t)i * na + a];
        push_obs(c, s, i, obs);
    }
    free(y);
    return 0;
}

/* thread count of the OpenMP team the batch calls run on (bench.py's cpu_baseline pins it and reports what it got) */
#include <omp.h>
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int oracle_get_threads(void) {
    int t = 1;
#pragma omp parallel
    {
#pragma omp single
        t = omp_get_num_threads();
    }
    return t;
}

int oracle_sizeof_cfg(void) { return (int)sizeof(OCfg); }
int oracle_sizeof_state(void) { return (int)sizeof(OState); }

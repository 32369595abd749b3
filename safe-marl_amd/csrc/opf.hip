// opf.hip — the OPF comparator's QP solve (include/flexopf.h): one persistent work-group per day runs the whole Mehrotra
// predictor-corrector iteration of safe-marl_amd/opf.py: qp_ipm; the Newton system is solved by a Riccati recursion over the
// periods.  Reference: utils/opf.py:13-192 (the program), run_opf.py:71 (the caller).
//
// Newton system.  N dx = r with N = blockdiag(P_t) + C' diag(D) C, where P_t [w, w] collects everything period-local (Hessian
// block, box and network barrier terms) and C is the cumulative-sum operator of the storage energy chain (opf.py:139-148):
// (C x)[t, k] = sum_{1 <= s <= t} G x_s, G[k, :] = a on Pesc[k], -b on Pesd[k].  That is the optimality condition of the
// linear-quadratic problem  min sum_t 1/2 dx_t' P_t dx_t - r_t' dx_t + 1/2 e_t' D_t e_t,  e_t = e_{t-1} + G dx_t, e_0 = 0.
// Eliminating dx_t for a given increment u_t = G dx_t (in parallel over the periods: Cholesky of P_t, W_t = L_t^-1 G',
// M_t = W_t' W_t, g_t = W_t' L_t^-1 r_t) leaves a recursion in the n_agents-vector e_t:
//     backward   S_{T-1} = D_{T-1};  S_t = L_S L_S',  B_t = I + L_S' M_t L_S = L_B L_B',  H_t = (M_t + S_t^-1)^-1 = L_S B_t^-1 L_S';
//                S_{t-1} = D_{t-1} + H_t;      s_{t-1} = L_S B_t^-1 (L_S' g_t + L_S^-1 s_t)
//     forward    v = e_{t-1} + g_t - M_t s_t;  q = B_t^-1 L_S' v;  e_t = L_S^-T q;  nu_t = -(L_S q + s_t);
//                dx_t = L_t^-T (L_t^-1 r_t + W_t nu_t)
// O(T w^3) per factorisation instead of O((T w)^3 / 24) for the dense Schur complement.  Every matrix that is factored is a
// sum of positive (semi-)definite terms, and nothing is formed by subtraction: late in the iteration z / s spans twenty decades
// (S_t ~ 1e10 where an energy bound is active), and the textbook forms I - H M and S e lose the small quantities they are after
// (measured: the iteration stalls from its tenth step on).  The recursion is still not backward stable as a solver for N, so
// every Newton solve is followed by ONE step of iterative refinement against N applied through its operators — with it the
// iteration needs as many steps as with a dense Cholesky factorisation of N and ends with a smaller dual residual.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "flexopf.h"

#define QP_W (4 * FLEXOPF_MAX_AGENTS)      // controls per period
#define QP_NA FLEXOPF_MAX_AGENTS
#define QP_WAVES 8
#define QP_THREADS (64 * QP_WAVES)
#define QP_CHUNK 32                         // network rows staged per pass

struct QpCtx {
    int T, na, w, R, mp, n, mt;
    int o_blo, o_vhi, o_vlo, o_ihi, o_ehi, o_elo;
    double ca, cb, regv;                     // regv: the diagonal regularisation of the current factorisation
    const double *q, *c, *lo, *hi, *jv, *vlo, *vhi, *ji, *ihi, *elo, *ehi;
    const uint8_t* fr;
    double *x, *s, *z, *rp, *ds, *dz, *qr, *rd, *rhs, *dx, *dxa, *y, *atz, *av, *ai, *ae;
    double *L, *Li, *Wm, *M, *D, *nu;
};

static __host__ __device__ inline int64_t qp_rows_per_period(int na, int R) { return 2 * 4 * na + 3 * R + 2 * na; }
static __host__ __device__ inline int64_t qp_work_doubles(int T, int na, int R) {
    const int64_t w = 4 * na, n = T * w, mt = T * qp_rows_per_period(na, R);
    return 6 * mt + 6 * n + 2 * (int64_t)T * R + (int64_t)T * QP_NA            // rows, variables, A v
           + (int64_t)T * w * w + n + (int64_t)T * w * QP_NA                      // L, 1 / diag(L), W
           + (int64_t)T * QP_NA * QP_NA + 2 * (int64_t)T * QP_NA;                 // M, D nu
}

#define QP_PK (QP_NA * (QP_NA + 1) / 2)      // packed lower triangle, element (a, b <= a) at a (a + 1) / 2 + b
#define QP_SEQ (3 * QP_PK + 2 * QP_NA)       // per period: L_S, L_B, M packed, 1 / diag(L_S), 1 / diag(L_B)
#define PK(a, b) ((a) * ((a) + 1) / 2 + (b))
struct __attribute__((aligned(16))) QpLds {
    // js / pm are per-period work space of the parallel phases; the recursion's matrices (seq(): QP_SEQ doubles per period,
    // written by the factorisation's serial phase, read by the solves) live in the same bytes — the two never overlap in time
    double js[QP_WAVES][QP_CHUNK * QP_W];       // a chunk of Jacobian rows per wavefront
    double pm[QP_WAVES][QP_W * (QP_W + 1)];     // P_t being assembled / W_t for its Gram matrix
    __device__ __forceinline__ double* seq(int t) { return &js[0][0] + (int64_t)t * QP_SEQ; }
    double va[QP_WAVES][FLEXOPF_MAX_ROWS];      // per-wavefront broadcast vectors
    double vb[QP_WAVES][FLEXOPF_MAX_ROWS];
    double scan[FLEXOPF_MAX_PERIODS * QP_NA];
    double scan2[FLEXOPF_MAX_PERIODS * QP_NA];
    double red[QP_WAVES];
};

// what one wavefront wrote to LDS is there for its other lanes (LDS operations of a wavefront execute in order)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane64(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int OP> __device__ __forceinline__ double red_op(double a, double b) {
    if (OP == 0) return a + b;
    if (OP == 1) return fmax(a, b);
    return fmin(a, b);
}
// all threads; same value (same bits) in every thread; fixed order
template <int OP> __device__ double block_reduce(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = red_op<OP>(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int i = 1; i < QP_WAVES; ++i) r = red_op<OP>(r, red[i]);
    __syncthreads();
    return r;
}

// row idx = t * mp + o of the one-sided row sets: sign, A v of its row (v = x or dx, network / chain parts from av / ai / ae), bound
__device__ __forceinline__ void row_decode(const QpCtx& c, int idx, const double* v, double& sg, double& ax, double& h) {
    const int t = idx / c.mp, o = idx - t * c.mp;
    if (o < c.o_blo) { sg = 1.0; ax = v[t * c.w + o]; h = c.hi[t * c.w + o]; }
    else if (o < c.o_vhi) { const int j = o - c.o_blo; sg = -1.0; ax = v[t * c.w + j]; h = -c.lo[t * c.w + j]; }
    else if (o < c.o_vlo) { const int r = o - c.o_vhi; sg = 1.0; ax = c.av[t * c.R + r]; h = c.vhi[t * c.R + r]; }
    else if (o < c.o_ihi) { const int r = o - c.o_vlo; sg = -1.0; ax = c.av[t * c.R + r]; h = -c.vlo[t * c.R + r]; }
    else if (o < c.o_ehi) { const int r = o - c.o_ihi; sg = 1.0; ax = c.ai[t * c.R + r]; h = c.ihi[t * c.R + r]; }
    else if (o < c.o_elo) { const int k = o - c.o_ehi; sg = 1.0; ax = c.ae[t * QP_NA + k]; h = c.ehi[t * c.na + k]; }
    else { const int k = o - c.o_elo; sg = -1.0; ax = c.ae[t * QP_NA + k]; h = -c.elo[t * c.na + k]; }
}

// av = Jv v, ai = Ji v per period; ae = the chain's cumulative sums.  Ends with a block barrier.
__device__ void apply_A(const QpCtx& c, QpLds& s, const double* v) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    for (int t = wv; t < c.T; t += QP_WAVES) {
        if (l < c.w) s.va[wv][l] = v[t * c.w + l];
        wave_sync();
        if (l < c.R) {
            const double* rv = c.jv + ((int64_t)t * c.R + l) * c.w;
            const double* ri = c.ji + ((int64_t)t * c.R + l) * c.w;
            double a = 0.0, b = 0.0;
            for (int j = 0; j < c.w; ++j) { const double xj = s.va[wv][j]; a = fma(rv[j], xj, a); b = fma(ri[j], xj, b); }
            c.av[t * c.R + l] = a;
            c.ai[t * c.R + l] = b;
        }
        wave_sync();
    }
    for (int i = tid; i < c.T * c.na; i += QP_THREADS) {
        const int t = i / c.na, k = i - t * c.na;
        s.scan[t * QP_NA + k] = t >= 1 ? c.ca * v[t * c.w + 2 * c.na + k] - c.cb * v[t * c.w + 3 * c.na + k] : 0.0;
    }
    __syncthreads();
    if (tid < c.na) {
        double acc = 0.0;
        for (int t = 0; t < c.T; ++t) { acc += s.scan[t * QP_NA + tid]; c.ae[t * QP_NA + tid] = acc; }
    }
    __syncthreads();
}

// atz = A' q for row weights q (layout of the row arrays): box and network parts per period, the chain through suffix sums.
// Ends with a block barrier.
__device__ void apply_At(const QpCtx& c, QpLds& s, const double* q) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    for (int i = tid; i < c.T * c.na; i += QP_THREADS) {
        const int t = i / c.na, k = i - t * c.na;
        s.scan[t * QP_NA + k] = q[t * c.mp + c.o_ehi + k] - q[t * c.mp + c.o_elo + k];
    }
    __syncthreads();
    if (tid < c.na) {
        double acc = 0.0;
        for (int t = c.T - 1; t >= 1; --t) { acc += s.scan[t * QP_NA + tid]; s.scan2[t * QP_NA + tid] = acc; }
        s.scan2[tid] = 0.0;                                  // period 0 has no coefficient in the chain (opf.py:140-142)
    }
    __syncthreads();
    for (int t = wv; t < c.T; t += QP_WAVES) {
        const double* qt = q + (int64_t)t * c.mp;
        if (l < c.R) { s.va[wv][l] = qt[c.o_vhi + l] - qt[c.o_vlo + l]; s.vb[wv][l] = qt[c.o_ihi + l]; }
        wave_sync();
        if (l < c.w) {
            double acc = qt[l] - qt[c.o_blo + l];
            const double* cv = c.jv + (int64_t)t * c.R * c.w + l;
            const double* ci = c.ji + (int64_t)t * c.R * c.w + l;
            for (int r = 0; r < c.R; ++r) acc = fma(cv[r * c.w], s.va[wv][r], fma(ci[r * c.w], s.vb[wv][r], acc));
            const int grp = l / c.na, k = l - grp * c.na;
            if (grp == 2) acc += c.ca * s.scan2[t * QP_NA + k];
            if (grp == 3) acc -= c.cb * s.scan2[t * QP_NA + k];
            c.atz[t * c.w + l] = acc;
        }
        wave_sync();
    }
    __syncthreads();
}

// Cholesky of an n x n matrix held as a local array (fully unrolled); pivots at or below `floor_` are raised to it
template <int N> __device__ __forceinline__ int chol_small(double (&a)[N][N], double floor_) {
    int floored = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = a[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d = fma(-a[j][k], a[j][k], d);
        if (!(d > floor_)) { d = floor_; floored = 1; }
        const double ljj = sqrt(d), inv = 1.0 / ljj;
        a[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = a[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = fma(-a[i][k], a[j][k], v);
            a[i][j] = v * inv;
        }
    }
    return floored;
}

// The factorisation of the Newton matrix for the current s, z.  Returns the number of pivots that had to be floored (block-wide).
__device__ double qp_factor(QpCtx& c, QpLds& s, double reg) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int w = c.w, R = c.R, na = c.na;
    double dmax = 0.0;
    // ---- P_t = Q_t + diag(box terms) + Jv' dv Jv + Ji' di Ji, masked; lane (i, kh): row i, columns [10 kh, 10 kh + 10)
    const int i = l & 31, kh = l >> 5;
    for (int t = wv; t < c.T; t += QP_WAVES) {
        double p[QP_W / 2];
#pragma unroll
        for (int kk = 0; kk < QP_W / 2; ++kk) p[kk] = 0.0;
        const double *st = c.s + (int64_t)t * c.mp, *zt = c.z + (int64_t)t * c.mp;
        for (int set = 0; set < 2; ++set) {
            const double* J = (set == 0 ? c.jv : c.ji) + (int64_t)t * R * w;
            for (int r0 = 0; r0 < R; r0 += QP_CHUNK) {
                const int nr = min(QP_CHUNK, R - r0);
                for (int e = l; e < nr * w; e += 64) s.js[wv][e] = J[r0 * w + e];
                if (l < nr) {
                    const int r = r0 + l;
                    s.va[wv][l] = set == 0 ? zt[c.o_vhi + r] / st[c.o_vhi + r] + zt[c.o_vlo + r] / st[c.o_vlo + r]
                                           : zt[c.o_ihi + r] / st[c.o_ihi + r];
                }
                wave_sync();
                if (i < w) {
                    for (int r = 0; r < nr; ++r) {
                        const double a = s.js[wv][r * w + i] * s.va[wv][r];
#pragma unroll
                        for (int kk = 0; kk < QP_W / 2; ++kk) {
                            const int k = (QP_W / 2) * kh + kk;
                            if (k < w) p[kk] = fma(a, s.js[wv][r * w + k], p[kk]);
                        }
                    }
                }
                wave_sync();
            }
        }
        if (i < w) {
            const double fi = c.fr[t * w + i] ? 1.0 : 0.0;
#pragma unroll
            for (int kk = 0; kk < QP_W / 2; ++kk) {
                const int k = (QP_W / 2) * kh + kk;
                if (k < w) {
                    double v = p[kk] + c.q[((int64_t)t * w + i) * w + k];
                    if (k == i) v += zt[i] / st[i] + zt[c.o_blo + i] / st[c.o_blo + i];
                    v *= fi * (c.fr[t * w + k] ? 1.0 : 0.0);
                    s.pm[wv][i * (QP_W + 1) + k] = v;
                }
            }
        }
        wave_sync();
        if (l < w) dmax = fmax(dmax, s.pm[wv][l * (QP_W + 1) + l]);
        for (int e = l; e < w * w; e += 64) c.L[(int64_t)t * w * w + e] = s.pm[wv][(e / w) * (QP_W + 1) + (e % w)];
        wave_sync();
    }
    for (int e = tid; e < c.T * na; e += QP_THREADS) {
        const int t = e / na, k = e - t * na;
        const double *st = c.s + (int64_t)t * c.mp, *zt = c.z + (int64_t)t * c.mp;
        c.D[t * QP_NA + k] = zt[c.o_ehi + k] / st[c.o_ehi + k] + zt[c.o_elo + k] / st[c.o_elo + k];
    }
    for (int e = tid; e < c.T * (QP_NA - na); e += QP_THREADS) {        // padding up to QP_NA units: decoupled, D = 1
        const int t = e / (QP_NA - na), k = na + e - t * (QP_NA - na);
        c.D[t * QP_NA + k] = 1.0;
    }
    const double gmax = block_reduce<1>(dmax, s.red);       // (barrier: P_t and D are in memory)
    const double regv = reg * gmax, pfloor = fmax(gmax * 1e-20, 1e-300);
    c.regv = regv;
    double floored = 0.0;
    // ---- L_t = chol(P_t + regularisation), W_t = L_t^-1 G_t', M_t = W_t' W_t; lane i: row i
    for (int t = wv; t < c.T; t += QP_WAVES) {
        double p[QP_W], invd[QP_W];
        const bool row = l < w;
        const double fi = row && c.fr[t * w + l] ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < QP_W; ++k) p[k] = (row && k < w) ? c.L[((int64_t)t * w + l) * w + k] : 0.0;
#pragma unroll
        for (int k = 0; k < QP_W; ++k) if (k == l) p[k] += regv + (1.0 - fi);
#pragma unroll
        for (int j = 0; j < QP_W; ++j) {
            invd[j] = 0.0;
            if (j < w) {
                double djj = readlane64(p[j], j);
                if (!(djj > pfloor)) { djj = pfloor; floored += 1.0; }
                const double ljj = sqrt(djj), inv = 1.0 / ljj;
                invd[j] = inv;
                const double lij = (l == j) ? ljj : p[j] * inv;
                p[j] = lij;
#pragma unroll
                for (int k = j + 1; k < QP_W; ++k)
                    if (k < w) p[k] = fma(-lij, readlane64(lij, k), p[k]);          // (rows i >= k use it)
            }
        }
        if (row) {
#pragma unroll
            for (int k = 0; k < QP_W; ++k) if (k < w) c.L[((int64_t)t * w + l) * w + k] = (k <= l) ? p[k] : 0.0;
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < QP_W; ++k) if (k == l) mine = invd[k];
            c.Li[t * w + l] = mine;
        }
        // W: forward substitution on the n_agents columns of G_t' (zero above row 2 na; G_0 = 0)
        double acc[QP_NA], wr[QP_NA];
#pragma unroll
        for (int k = 0; k < QP_NA; ++k) {
            acc[k] = 0.0; wr[k] = 0.0;
            if (row && t >= 1 && k < na) {
                if (l == 2 * na + k) acc[k] = c.ca * fi;
                if (l == 3 * na + k) acc[k] = -c.cb * fi;
            }
        }
#pragma unroll
        for (int j = 0; j < QP_W; ++j) {
            if (j < w && j >= 2 * na) {
#pragma unroll
                for (int k = 0; k < QP_NA; ++k) {
                    const double wjk = readlane64(acc[k], j) * invd[j];
                    if (l == j) wr[k] = wjk;
                    if (l > j) acc[k] = fma(-p[j], wjk, acc[k]);
                }
            }
        }
        if (row) {
#pragma unroll
            for (int k = 0; k < QP_NA; ++k) { c.Wm[((int64_t)t * w + l) * QP_NA + k] = wr[k]; s.pm[wv][l * QP_NA + k] = wr[k]; }
        }
        wave_sync();
        if (l < QP_NA * QP_NA) {
            const int a = l / QP_NA, b = l - a * QP_NA;
            double m = 0.0;
            for (int r = 0; r < w; ++r) m = fma(s.pm[wv][r * QP_NA + a], s.pm[wv][r * QP_NA + b], m);
            c.M[(int64_t)t * QP_NA * QP_NA + l] = m;
        }
        wave_sync();
    }
    floored = block_reduce<0>(floored, s.red) / 64.0;       // (barrier: L, W, M are in memory; every lane counted the same pivots)
    // ---- the recursion over the periods: one wavefront, every lane the same arithmetic on uniform data; its matrices stay in
    // LDS (seq(t)) for the solves.  M_{t-1} is fetched while step t is worked on.
    static_assert(sizeof(((QpLds*)0)->js) + sizeof(((QpLds*)0)->pm) >= sizeof(double) * FLEXOPF_MAX_PERIODS * QP_SEQ, "seq() fits");
    if (wv == 0) {
        double S[QP_PK], Mn[QP_PK];
#pragma unroll
        for (int a = 0; a < QP_NA; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                S[PK(a, b)] = a == b ? c.D[(c.T - 1) * QP_NA + a] : 0.0;
                Mn[PK(a, b)] = c.M[(int64_t)(c.T - 1) * QP_NA * QP_NA + a * QP_NA + b];
            }
        for (int t = c.T - 1; t >= 1; --t) {
            double M[QP_NA][QP_NA], Ls[QP_NA][QP_NA], T1[QP_NA][QP_NA], Bm[QP_NA][QP_NA], Z[QP_NA][QP_NA], Dn[QP_NA];
#pragma unroll
            for (int a = 0; a < QP_NA; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) { M[a][b] = Mn[PK(a, b)]; M[b][a] = Mn[PK(a, b)]; Ls[a][b] = S[PK(a, b)]; }
            if (t >= 2) {
#pragma unroll
                for (int a = 0; a < QP_NA; ++a)
#pragma unroll
                    for (int b = 0; b <= a; ++b) Mn[PK(a, b)] = c.M[(int64_t)(t - 1) * QP_NA * QP_NA + a * QP_NA + b];
            }
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) Dn[a] = c.D[(t - 1) * QP_NA + a];
            chol_small<QP_NA>(Ls, 1e-300);
#pragma unroll
            for (int a = 0; a < QP_NA; ++a)
#pragma unroll
                for (int b = 0; b < QP_NA; ++b) {              // T1 = M Ls
                    double v = 0.0;
#pragma unroll
                    for (int k = b; k < QP_NA; ++k) v = fma(M[a][k], Ls[k][b], v);
                    T1[a][b] = v;
                }
#pragma unroll
            for (int a = 0; a < QP_NA; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {                 // Bm = I + Ls' T1 (lower triangle)
                    double v = a == b ? 1.0 : 0.0;
#pragma unroll
                    for (int k = a; k < QP_NA; ++k) v = fma(Ls[k][a], T1[k][b], v);
                    Bm[a][b] = v;
                }
            chol_small<QP_NA>(Bm, 1e-300);
            double ib[QP_NA], is_[QP_NA];
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) { ib[a] = 1.0 / Bm[a][a]; is_[a] = 1.0 / Ls[a][a]; }
#pragma unroll
            for (int b = 0; b < QP_NA; ++b)                    // Z = Lb^-1 Ls'
#pragma unroll
                for (int a = 0; a < QP_NA; ++a) {
                    double v = b >= a ? Ls[b][a] : 0.0;
#pragma unroll
                    for (int k = 0; k < a; ++k) v = fma(-Bm[a][k], Z[k][b], v);
                    Z[a][b] = v * ib[a];
                }
            if (l == 0) {
                double* q = s.seq(t);
#pragma unroll
                for (int a = 0; a < QP_NA; ++a) {
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        q[PK(a, b)] = Ls[a][b]; q[QP_PK + PK(a, b)] = Bm[a][b]; q[2 * QP_PK + PK(a, b)] = M[a][b];
                    }
                    q[3 * QP_PK + a] = is_[a];
                    q[3 * QP_PK + QP_NA + a] = ib[a];
                }
            }
#pragma unroll
            for (int a = 0; a < QP_NA; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {                 // S_{t-1} = D_{t-1} + Z' Z
                    double v = a == b ? Dn[a] : 0.0;
#pragma unroll
                    for (int k = 0; k < QP_NA; ++k) v = fma(Z[k][a], Z[k][b], v);
                    S[PK(a, b)] = v;
                }
        }
    }
    __syncthreads();
    return floored;
}

// v <- L^-1 v, v <- L^-T v, L v, L' v for a packed lower-triangular L in LDS (id = 1 / diag)
__device__ __forceinline__ void pk_solve(const double* L, const double* id, double (&v)[QP_NA]) {
#pragma unroll
    for (int a = 0; a < QP_NA; ++a) {
        double acc = v[a];
#pragma unroll
        for (int k = 0; k < a; ++k) acc = fma(-L[PK(a, k)], v[k], acc);
        v[a] = acc * id[a];
    }
}
__device__ __forceinline__ void pk_solve_t(const double* L, const double* id, double (&v)[QP_NA]) {
#pragma unroll
    for (int aa = 0; aa < QP_NA; ++aa) {
        const int a = QP_NA - 1 - aa;
        double acc = v[a];
#pragma unroll
        for (int k = a + 1; k < QP_NA; ++k) acc = fma(-L[PK(k, a)], v[k], acc);
        v[a] = acc * id[a];
    }
}
__device__ __forceinline__ void pk_mul(const double* L, const double (&v)[QP_NA], double (&o)[QP_NA]) {
#pragma unroll
    for (int a = 0; a < QP_NA; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k <= a; ++k) acc = fma(L[PK(a, k)], v[k], acc);
        o[a] = acc;
    }
}
__device__ __forceinline__ void pk_mul_t(const double* L, const double (&v)[QP_NA], double (&o)[QP_NA]) {
#pragma unroll
    for (int a = 0; a < QP_NA; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int k = a; k < QP_NA; ++k) acc = fma(L[PK(k, a)], v[k], acc);
        o[a] = acc;
    }
}

// dx = N^-1 rhs with the factorisation above (rhs already zero on pinned variables); dx masked.  Ends with a block barrier.
__device__ void qp_solve(const QpCtx& c, QpLds& s) {
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int w = c.w;
    // y_t = L_t^-1 rhs_t, g_t = W_t' y_t (g in LDS: scan2)
    for (int t = wv; t < c.T; t += QP_WAVES) {
        const bool row = l < w;
        double p[QP_W];
#pragma unroll
        for (int k = 0; k < QP_W; ++k) p[k] = (row && k < w) ? c.L[((int64_t)t * w + l) * w + k] : 0.0;
        const double myinv = row ? c.Li[t * w + l] : 0.0;
        double acc = row ? c.rhs[t * w + l] : 0.0, y = 0.0;
#pragma unroll
        for (int j = 0; j < QP_W; ++j) {
            if (j < w) {
                const double yj = readlane64(acc, j) * readlane64(myinv, j);
                if (l == j) y = yj;
                if (l > j) acc = fma(-p[j], yj, acc);
            }
        }
        if (row) { c.y[t * w + l] = y; s.va[wv][l] = y; }
        wave_sync();
        if (l < QP_NA) {
            double g = 0.0;
            for (int r = 0; r < w; ++r) g = fma(c.Wm[((int64_t)t * w + r) * QP_NA + l], s.va[wv][r], g);
            s.scan2[t * QP_NA + l] = g;
        }
        wave_sync();
    }
    __syncthreads();
    if (wv == 0) {
        double sv[QP_NA], e[QP_NA];
#pragma unroll
        for (int a = 0; a < QP_NA; ++a) { sv[a] = 0.0; e[a] = 0.0; }
        for (int t = c.T - 1; t >= 1; --t) {                   // s_{t-1} = L_S B^-1 (L_S' g_t + L_S^-1 s_t)
            const double *q = s.seq(t), *LS = q, *LB = q + QP_PK, *iS = q + 3 * QP_PK, *iB = iS + QP_NA;
            if (l == 0) {
#pragma unroll
                for (int a = 0; a < QP_NA; ++a) s.scan[t * QP_NA + a] = sv[a];
            }
            double g[QP_NA], a1[QP_NA];
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) g[a] = s.scan2[t * QP_NA + a];
            pk_mul_t(LS, g, a1);
            pk_solve(LS, iS, sv);
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) a1[a] += sv[a];
            pk_solve(LB, iB, a1);
            pk_solve_t(LB, iB, a1);
            pk_mul(LS, a1, sv);
        }
        wave_sync();
        if (l == 0) {
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) c.nu[a] = 0.0;
        }
        for (int t = 1; t < c.T; ++t) {                        // v = e + g - M s;  q = B^-1 L_S' v;  e = L_S^-T q;  nu = -(L_S q + s)
            const double *q = s.seq(t), *LS = q, *LB = q + QP_PK, *Mp = q + 2 * QP_PK, *iS = q + 3 * QP_PK, *iB = iS + QP_NA;
            double st[QP_NA], v[QP_NA], u[QP_NA], o[QP_NA];
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) st[a] = s.scan[t * QP_NA + a];
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) {
                double m = 0.0;
#pragma unroll
                for (int k = 0; k < QP_NA; ++k) m = fma(Mp[k <= a ? PK(a, k) : PK(k, a)], st[k], m);
                v[a] = e[a] + s.scan2[t * QP_NA + a] - m;
            }
            pk_mul_t(LS, v, u);
            pk_solve(LB, iB, u);
            pk_solve_t(LB, iB, u);
            pk_mul(LS, u, o);
            if (l == 0) {
#pragma unroll
                for (int a = 0; a < QP_NA; ++a) c.nu[t * QP_NA + a] = -(o[a] + st[a]);
            }
            pk_solve_t(LS, iS, u);
#pragma unroll
            for (int a = 0; a < QP_NA; ++a) e[a] = u[a];
        }
    }
    __syncthreads();
    // dx_t = L_t^-T (y_t + W_t nu_t); lane i holds column i of L_t
    for (int t = wv; t < c.T; t += QP_WAVES) {
        const bool row = l < w;
        double col[QP_W];
#pragma unroll
        for (int j = 0; j < QP_W; ++j) col[j] = (row && j < w) ? c.L[((int64_t)t * w + j) * w + l] : 0.0;
        const double myinv = row ? c.Li[t * w + l] : 0.0;
        double acc = 0.0, dx = 0.0;
        if (row) {
            acc = c.y[t * w + l];
#pragma unroll
            for (int k = 0; k < QP_NA; ++k) acc = fma(c.Wm[((int64_t)t * w + l) * QP_NA + k], c.nu[t * QP_NA + k], acc);
        }
#pragma unroll
        for (int jj = 0; jj < QP_W; ++jj) {
            const int j = QP_W - 1 - jj;
            if (j < w) {
                const double dj = readlane64(acc, j) * readlane64(myinv, j);
                if (l == j) dx = dj;
                if (l < j) acc = fma(-col[j], dj, acc);
            }
        }
        if (row) c.dx[t * w + l] = c.fr[t * w + l] ? dx : 0.0;
    }
    __syncthreads();
}

// One Newton solve: qp_solve on c.rhs, then one step of iterative refinement against N applied through its operators
// (N dx = Q dx + reg dx + sum A' (z / s) A dx on the free variables).  Leaves dx in c.dx and A dx in av / ai / ae.
__device__ void qp_newton_solve(const QpCtx& c, QpLds& s) {
    const int tid = threadIdx.x;
    qp_solve(c, s);
    apply_A(c, s, c.dx);
    for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
        double sg, ax, h;
        row_decode(c, idx, c.dx, sg, ax, h);
        c.qr[idx] = sg * (c.z[idx] / c.s[idx]) * ax;           // (apply_At forms upper - lower: + A' d A dx for both)
    }
    __syncthreads();
    apply_At(c, s, c.qr);
    for (int i = tid; i < c.n; i += QP_THREADS) {
        const int t = i / c.w, j = i - t * c.w;
        const double* qrow = c.q + ((int64_t)t * c.w + j) * c.w;
        double acc = fma(c.regv, c.dx[i], c.atz[i]);
        for (int k = 0; k < c.w; ++k) acc = fma(qrow[k], c.dx[t * c.w + k], acc);
        c.dxa[i] = c.dx[i];
        c.rhs[i] = c.fr[i] ? c.rhs[i] - acc : 0.0;
    }
    __syncthreads();
    qp_solve(c, s);
    for (int i = tid; i < c.n; i += QP_THREADS) c.dx[i] += c.dxa[i];
    __syncthreads();
    apply_A(c, s, c.dx);
}

__global__ __launch_bounds__(QP_THREADS) void qp_ipm_kernel(FlexQpArgs a) {
    __shared__ QpLds s;
    const int tid = threadIdx.x, b = blockIdx.x;
    QpCtx c;
    c.T = a.periods; c.na = a.n_agents; c.w = 4 * a.n_agents; c.R = a.rows;
    c.mp = (int)qp_rows_per_period(c.na, c.R); c.n = c.T * c.w; c.mt = c.T * c.mp;
    c.o_blo = c.w; c.o_vhi = 2 * c.w; c.o_vlo = c.o_vhi + c.R; c.o_ihi = c.o_vlo + c.R; c.o_ehi = c.o_ihi + c.R; c.o_elo = c.o_ehi + c.na;
    c.ca = a.chain_a; c.cb = a.chain_b; c.regv = 0.0;
    const int64_t bn = (int64_t)b * c.n, bR = (int64_t)b * c.T * c.R, be = (int64_t)b * c.T * c.na;
    c.q = a.q + bn * c.w; c.c = a.c + bn; c.lo = a.lo + bn; c.hi = a.hi + bn; c.fr = a.free_mask + bn;
    c.jv = a.jv + bR * c.w; c.vlo = a.v_lo + bR; c.vhi = a.v_hi + bR; c.ji = a.ji + bR * c.w; c.ihi = a.i_hi + bR;
    c.elo = a.e_lo + be; c.ehi = a.e_hi + be;
    c.x = a.x + bn;
    double* wk = a.work + (int64_t)b * qp_work_doubles(c.T, c.na, c.R);
    c.s = wk; wk += c.mt; c.z = wk; wk += c.mt; c.rp = wk; wk += c.mt; c.ds = wk; wk += c.mt; c.dz = wk; wk += c.mt; c.qr = wk; wk += c.mt;
    c.rd = wk; wk += c.n; c.rhs = wk; wk += c.n; c.dx = wk; wk += c.n; c.dxa = wk; wk += c.n; c.y = wk; wk += c.n; c.atz = wk; wk += c.n;
    c.av = wk; wk += c.T * c.R; c.ai = wk; wk += c.T * c.R; c.ae = wk; wk += c.T * QP_NA;
    c.L = wk; wk += (int64_t)c.T * c.w * c.w; c.Li = wk; wk += c.n; c.Wm = wk; wk += (int64_t)c.T * c.w * QP_NA;
    c.M = wk; wk += c.T * QP_NA * QP_NA; c.D = wk; wk += c.T * QP_NA; c.nu = wk;
    const double* x0 = a.x0 + bn;
    const double m_tot = (double)c.mt;

    for (int i = tid; i < c.n; i += QP_THREADS) c.x[i] = x0[i];
    // (the padded rows of the chain arrays read by the recursion)
    for (int i = tid; i < c.T * QP_NA; i += QP_THREADS) { c.nu[i] = 0.0; c.ae[i] = 0.0; }
    __syncthreads();
    apply_A(c, s, c.x);
    for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
        double sg, ax, h;
        row_decode(c, idx, c.x, sg, ax, h);
        c.s[idx] = fmax(h - sg * ax, 1e-3);
        c.z[idx] = 1.0;
    }
    __syncthreads();
    int it = 0, done = 0;
    double mu = 0.0, res_d = 0.0, res_p = 0.0, floored = 0.0;
    for (it = 0; it < a.max_iter; ++it) {
        if (it > 0) apply_A(c, s, c.x);
        double sz = 0.0, rpm = 0.0;
        for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
            double sg, ax, h;
            row_decode(c, idx, c.x, sg, ax, h);
            const double si = c.s[idx], rp = sg * ax + si - h;
            c.rp[idx] = rp;
            sz = fma(si, c.z[idx], sz);
            rpm = fmax(rpm, fabs(rp));
        }
        apply_At(c, s, c.z);
        double rdm = 0.0;
        for (int i = tid; i < c.n; i += QP_THREADS) {
            const int t = i / c.w, j = i - t * c.w;
            const double* qrow = c.q + ((int64_t)t * c.w + j) * c.w;
            double acc = c.c[i] + c.atz[i];
            for (int k = 0; k < c.w; ++k) acc = fma(qrow[k], c.x[t * c.w + k], acc);
            acc = c.fr[i] ? acc : 0.0;
            c.rd[i] = acc;
            rdm = fmax(rdm, fabs(acc));
        }
        mu = block_reduce<0>(sz, s.red) / m_tot;
        res_p = block_reduce<1>(rpm, s.red);
        res_d = block_reduce<1>(rdm, s.red);
        // the dual residual floors near 1e-8 once z / s spans twenty decades (conditioning of the Newton matrix)
        if ((mu < a.tol && res_p < 1e-8 && res_d < 1e-6) || mu < 1e-4 * a.tol) { done = 1; break; }
        if (!(fabs(mu) < 1e300) || !(res_p < 1e300) || !(res_d < 1e300)) break;      // broken down (infeasible program): not converged
        floored += qp_factor(c, s, a.reg);

        double sigma_mu = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            // right-hand side: -r_d - sum sg A' ((z r_p - r_c) / s)
            for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
                const double si = c.s[idx], zi = c.z[idx];
                const double rc = pass == 0 ? si * zi : fma(c.ds[idx], c.dz[idx], si * zi) - sigma_mu;
                c.qr[idx] = (zi * c.rp[idx] - rc) / si;            // (apply_At applies the sets' signs: upper - lower)
            }
            __syncthreads();
            apply_At(c, s, c.qr);
            for (int i = tid; i < c.n; i += QP_THREADS) c.rhs[i] = c.fr[i] ? -c.rd[i] - c.atz[i] : 0.0;
            __syncthreads();
            qp_newton_solve(c, s);
            double rs = INFINITY, rz = INFINITY;
            for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
                double sg, ax, h;
                row_decode(c, idx, c.dx, sg, ax, h);
                const double si = c.s[idx], zi = c.z[idx];
                const double rc = pass == 0 ? si * zi : fma(c.ds[idx], c.dz[idx], si * zi) - sigma_mu;
                const double dsi = -c.rp[idx] - sg * ax, dzi = (-rc - zi * dsi) / si;
                c.ds[idx] = dsi;
                c.dz[idx] = dzi;
                if (dsi < 0.0) rs = fmin(rs, -si / dsi);
                if (dzi < 0.0) rz = fmin(rz, -zi / dzi);
            }
            rs = block_reduce<2>(rs, s.red);
            rz = block_reduce<2>(rz, s.red);
            if (pass == 0) {
                const double ap = fmin(rs, 1.0), ad = fmin(rz, 1.0);
                double acc = 0.0;
                for (int idx = tid; idx < c.mt; idx += QP_THREADS)
                    acc = fma(fma(ap, c.ds[idx], c.s[idx]), fma(ad, c.dz[idx], c.z[idx]), acc);
                const double mu_a = block_reduce<0>(acc, s.red) / m_tot;
                const double sigma = fmin(fmax(mu_a / mu, 0.0), 1.0);
                sigma_mu = sigma * sigma * sigma * mu;
            } else {
                const double ap = fmin(0.995 * rs, 1.0), ad = fmin(0.995 * rz, 1.0);
                for (int i = tid; i < c.n; i += QP_THREADS) c.x[i] = fma(ap, c.dx[i], c.x[i]);
                for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
                    c.s[idx] = fma(ap, c.ds[idx], c.s[idx]);
                    c.z[idx] = fma(ad, c.dz[idx], c.z[idx]);
                }
                __syncthreads();
            }
        }
    }
    if (it >= a.max_iter) it = a.max_iter - 1;
    double* du = a.duals + (int64_t)b * c.mt;
    for (int idx = tid; idx < c.mt; idx += QP_THREADS) du[idx] = c.z[idx];
    if (tid == 0) {
        double* info = a.info + (int64_t)b * FLEXOPF_INFO;
        info[0] = (double)it; info[1] = mu; info[2] = res_d; info[3] = res_p; info[4] = (double)done; info[5] = floored;
    }
}

extern "C" int64_t flexopf_qp_work_doubles(int32_t periods, int32_t n_agents, int32_t rows) {
    if (periods < 1 || periods > FLEXOPF_MAX_PERIODS || n_agents < 1 || n_agents > FLEXOPF_MAX_AGENTS || rows < 1 ||
        rows > FLEXOPF_MAX_ROWS)
        return -1;
    return qp_work_doubles(periods, n_agents, rows);
}

extern "C" int flexopf_qp_solve(const FlexQpArgs* a, void* stream) {
    if (!a || a->batch < 0 || flexopf_qp_work_doubles(a->periods, a->n_agents, a->rows) < 0 || a->max_iter < 1) return FLEXOPF_EINVAL;
    if (!a->q || !a->c || !a->lo || !a->hi || !a->free_mask || !a->jv || !a->v_lo || !a->v_hi || !a->ji || !a->i_hi || !a->e_lo ||
        !a->e_hi || !a->x0 || !a->x || !a->duals || !a->info || !a->work)
        return FLEXOPF_EINVAL;
    if (!(a->tol > 0.0) || !(a->reg >= 0.0)) return FLEXOPF_EINVAL;
    if (a->batch == 0) return FLEXOPF_OK;
    hipLaunchKernelGGL(qp_ipm_kernel, dim3(a->batch), dim3(QP_THREADS), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? FLEXOPF_OK : FLEXOPF_EHIP;
}

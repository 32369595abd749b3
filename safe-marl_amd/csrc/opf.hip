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
#define QP_GROUP 4                          // chunks a wavefront keeps in flight
#define QP_REFINE_BELOW 1e-4                // duality measure below which a Newton solve gets its refinement step: above,
                                            // z / s spans few decades and the recursion alone is accurate to 1e-12

// diagnostic build (-DQP_STAMPS, tools/opf_qp_stamps.py): cycles per phase of block 0, printed at the end; never in the product
#ifdef QP_STAMPS
#define QP_NSTAMP 12
__device__ long long qp_stamp_acc[QP_NSTAMP];
__device__ long long qp_stamp_last;
__device__ __forceinline__ void qp_stamp(int slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const long long now = wall_clock64();
        if (slot >= 0) qp_stamp_acc[slot] += now - qp_stamp_last;
        qp_stamp_last = now;
    }
}
#define QSTAMP(slot) qp_stamp(slot)
#else
#define QSTAMP(slot) do { } while (0)
#endif

// Every pointer the kernel walks is a GLOBAL-memory pointer by type: with generic pointers the loads come out as flat_load,
// which counts on the LDS counter as well — every wait for an LDS read then also waits for the prefetch that is meant to stay
// in flight (seen in the ISA of the first build).
typedef __attribute__((address_space(1))) double gdbl;
typedef __attribute__((address_space(1))) const double gcdbl;
typedef __attribute__((address_space(1))) const uint8_t gcu8;
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const dbl2 gcdbl2;
struct QpCtx {
    int T, na, w, R, mp, n, mt;
    int o_blo, o_vhi, o_vlo, o_ihi, o_ehi, o_elo;
    double ca, cb, regv;                     // regv: the diagonal regularisation of the current factorisation
    gcdbl *q, *c, *lo, *hi, *jv, *vlo, *vhi, *ji, *ihi, *elo, *ehi;
    gcu8* fr;
    gdbl *x, *s, *z, *rp, *ds, *dz, *qr, *rd, *rhs, *dx, *dxa, *y, *atz, *ar, *hs;          // ar: sg (A v) per row of the row sets; hs: sg bound
    gdbl *L, *Li, *Wm, *M, *D, *nu;
};

static __host__ __device__ inline int64_t qp_rows_per_period(int na, int R) { return 2 * 4 * na + 3 * R + 2 * na; }
static __host__ __device__ inline int64_t qp_even(int64_t x) { return (x + 1) & ~(int64_t)1; }     // (arrays start 16-byte aligned)
static __host__ __device__ inline int64_t qp_work_doubles(int T, int na, int R) {
    const int64_t w = 4 * na, n = T * w, mt = qp_even(T * qp_rows_per_period(na, R));
    return 8 * mt + 6 * n                                                                      // rows (incl. A v, bounds), variables
           + (int64_t)T * w * w + n + (int64_t)T * w * QP_NA                                      // L, 1 / diag(L), W
           + qp_even((int64_t)T * QP_NA * QP_NA) + 2 * qp_even((int64_t)T * QP_NA);               // M, D nu
}

#define QP_PK (QP_NA * (QP_NA + 1) / 2)      // packed lower triangle, element (a, b <= a) at a (a + 1) / 2 + b
#define QP_SEQ (3 * QP_PK + 2 * QP_NA)       // per period: L_S, L_B, M packed, 1 / diag(L_S), 1 / diag(L_B)
#define PK(a, b) ((a) * ((a) + 1) / 2 + (b))
struct __attribute__((aligned(16))) QpLds {
    // js / pm: per-wavefront work space of the parallel phases; sq: the recursion's matrices (QP_SEQ doubles per period, written
    // by the factorisation's serial phase, read by every solve until the next factorisation)
    double js[QP_WAVES][QP_CHUNK * QP_W];       // a chunk of Jacobian rows per wavefront
    double pm[QP_WAVES][QP_W * (QP_W + 1)];     // P_t being assembled / W_t for its Gram matrix
    double sq[FLEXOPF_MAX_PERIODS * QP_SEQ];
    __device__ __forceinline__ double* seq(int t) { return sq + t * QP_SEQ; }
    double va[QP_WAVES][FLEXOPF_MAX_ROWS];      // per-wavefront broadcast vectors
    double vb[QP_WAVES][FLEXOPF_MAX_ROWS];
    double scan[FLEXOPF_MAX_PERIODS * QP_NA];
    double scan2[FLEXOPF_MAX_PERIODS * QP_NA];
    double red[QP_WAVES];
};

// What one wavefront wrote to LDS is there for its other lanes (the LDS operations of a wavefront execute in order); the
// compiler may not move memory operations across.  NOT a fence: a release fence waits for every outstanding global load too,
// i.e. for the prefetch that is meant to stay in flight across the compute (measured: it doubled the streaming phases).
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
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

// row idx = t * mp + o of the one-sided row sets  sg (A x)_row <= sg bound: the signed bound (the signed products sg (A v) are
// what apply_A leaves in c.ar, row by row, so that every pass over the rows is a plain elementwise loop)
__device__ __forceinline__ double row_bound(const QpCtx& c, int idx) {
    const int t = idx / c.mp, o = idx - t * c.mp;
    if (o < c.o_blo) return c.hi[t * c.w + o];
    if (o < c.o_vhi) return -c.lo[t * c.w + o - c.o_blo];
    if (o < c.o_vlo) return c.vhi[t * c.R + o - c.o_vhi];
    if (o < c.o_ihi) return -c.vlo[t * c.R + o - c.o_vlo];
    if (o < c.o_ehi) return c.ihi[t * c.R + o - c.o_ihi];
    if (o < c.o_elo) return c.ehi[t * c.na + o - c.o_ehi];
    return -c.elo[t * c.na + o - c.o_elo];
}

// ---- the Jacobian blocks stream through LDS -------------------------------------------------------------------------------
// Every parallel phase that needs Jv_t / Ji_t walks the same sequence of chunks (a wavefront's periods t = wv, wv + 8, ...;
// per period the chunks of Jv, then of Ji; 32 rows each): QP_GROUP chunks are fetched at once with 16-byte loads coalesced
// over the wavefront into registers (with the few per-row operands the phase needs), then handed to LDS one after the other
// for the compute, whose inner loops run on LDS only.  (The first form — 8-byte loads with the arithmetic waiting on each,
// the row passes decoding their row type per element — took 3.1 ms per interior-point step; this one 1.7 ms.  What is left is
// the bandwidth one CU draws with everything else of the day's state streaming past as well: DESIGN.md §9.)
struct QpChunk { int t, set, r0, nr; };
__device__ __forceinline__ bool qp_chunk_at(const QpCtx& c, int wv, int idx, QpChunk& k) {
    const int cps = (c.R + QP_CHUNK - 1) / QP_CHUNK, per = 2 * cps;
    k.t = wv + QP_WAVES * (idx / per);
    const int rem = idx % per;
    k.set = rem / cps;
    k.r0 = QP_CHUNK * (rem - k.set * cps);
    k.nr = min(QP_CHUNK, c.R - k.r0);
    return k.t < c.T;
}
template <int W> struct QpPre { dbl2 v[(QP_CHUNK * W / 2 + 63) / 64]; };
template <int W> __device__ __forceinline__ void qp_chunk_load(const QpCtx& c, const QpChunk& k, int l, QpPre<W>& pre) {
    gcdbl2* src = (gcdbl2*)((k.set == 0 ? c.jv : c.ji) + ((int64_t)k.t * c.R + k.r0) * W);
    const int cnt = k.nr * W / 2;
#pragma unroll
    for (int i = 0; i < (QP_CHUNK * W / 2 + 63) / 64; ++i) {
        const int e = l + 64 * i;
        if (e < cnt) pre.v[i] = src[e];
    }
}
template <int W> __device__ __forceinline__ void qp_chunk_store(const QpChunk& k, int l, const QpPre<W>& pre, double* js) {
    dbl2* dst = (dbl2*)js;
    const int cnt = k.nr * W / 2;
#pragma unroll
    for (int i = 0; i < (QP_CHUNK * W / 2 + 63) / 64; ++i) {
        const int e = l + 64 * i;
        if (e < cnt) dst[e] = pre.v[i];
    }
}

// c.ar = sg (A v) for every row of the row sets (box: +-v; network rows: +-Jv v, Ji v; chain: +- cumulative sums).  Ends
// with a block barrier.
template <int NA> __device__ void apply_A(const QpCtx& c, QpLds& s, gcdbl* v) {
    constexpr int W = 4 * NA;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    for (int base = 0;; base += QP_GROUP) {
        QpChunk ck[QP_GROUP];
        QpPre<W> pre[QP_GROUP];
        double xp[QP_GROUP];
        bool hv[QP_GROUP];
#pragma unroll
        for (int g = 0; g < QP_GROUP; ++g) {
            hv[g] = qp_chunk_at(c, wv, base + g, ck[g]);
            xp[g] = 0.0;
            if (hv[g]) { qp_chunk_load<W>(c, ck[g], l, pre[g]); if (l < W) xp[g] = v[ck[g].t * W + l]; }
        }
        if (!hv[0]) break;
#pragma unroll
        for (int g = 0; g < QP_GROUP; ++g) {
            if (!hv[g]) break;
            const QpChunk cur = ck[g];
            qp_chunk_store<W>(cur, l, pre[g], s.js[wv]);
            gdbl* art = c.ar + (int64_t)cur.t * c.mp;
            if (l < W) {
                s.va[wv][l] = xp[g];
                if (cur.set == 0 && cur.r0 == 0) { art[l] = xp[g]; art[c.o_blo + l] = -xp[g]; }
            }
            wave_sync();
            if (l < cur.nr) {
                const double* row = s.js[wv] + l * W;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < W; ++j) acc = fma(row[j], s.va[wv][j], acc);
                if (cur.set == 0) { art[c.o_vhi + cur.r0 + l] = acc; art[c.o_vlo + cur.r0 + l] = -acc; }
                else art[c.o_ihi + cur.r0 + l] = acc;
            }
            wave_sync();
        }
    }
    for (int i = tid; i < c.T * NA; i += QP_THREADS) {
        const int t = i / NA, k = i - t * NA;
        s.scan[t * QP_NA + k] = t >= 1 ? c.ca * v[t * W + 2 * NA + k] - c.cb * v[t * W + 3 * NA + k] : 0.0;
    }
    __syncthreads();
    if (tid < NA) {
        double acc = 0.0;
        for (int t = 0; t < c.T; ++t) {
            acc += s.scan[t * QP_NA + tid];
            c.ar[(int64_t)t * c.mp + c.o_ehi + tid] = acc;
            c.ar[(int64_t)t * c.mp + c.o_elo + tid] = -acc;
        }
    }
    __syncthreads();
    QSTAMP(6);
}

// atz = A' q for row weights q (layout of the row arrays): box and network parts per period, the chain through suffix sums.
// Ends with a block barrier.
template <int NA> __device__ void apply_At(const QpCtx& c, QpLds& s, gcdbl* q) {
    constexpr int W = 4 * NA;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    for (int i = tid; i < c.T * NA; i += QP_THREADS) {
        const int t = i / NA, k = i - t * NA;
        s.scan[t * QP_NA + k] = q[t * c.mp + c.o_ehi + k] - q[t * c.mp + c.o_elo + k];
    }
    __syncthreads();
    if (tid < NA) {
        double acc = 0.0;
        for (int t = c.T - 1; t >= 1; --t) { acc += s.scan[t * QP_NA + tid]; s.scan2[t * QP_NA + tid] = acc; }
        s.scan2[tid] = 0.0;                                  // period 0 has no coefficient in the chain (opf.py:140-142)
    }
    __syncthreads();
    {
        double acc = 0.0;
        for (int base = 0;; base += QP_GROUP) {
            QpChunk ck[QP_GROUP];
            QpPre<W> pre[QP_GROUP];
            double wa[QP_GROUP], wb[QP_GROUP], b0[QP_GROUP], b1[QP_GROUP];      // row weight wa - wb; box part b0 - b1
            bool hv[QP_GROUP];
#pragma unroll
            for (int g = 0; g < QP_GROUP; ++g) {
                hv[g] = qp_chunk_at(c, wv, base + g, ck[g]);
                wa[g] = 0.0; wb[g] = 0.0; b0[g] = 0.0; b1[g] = 0.0;
                if (hv[g]) {
                    const QpChunk& k = ck[g];
                    qp_chunk_load<W>(c, k, l, pre[g]);
                    gcdbl* qt = q + (int64_t)k.t * c.mp;
                    if (l < k.nr) {
                        if (k.set == 0) { wa[g] = qt[c.o_vhi + k.r0 + l]; wb[g] = qt[c.o_vlo + k.r0 + l]; }
                        else wa[g] = qt[c.o_ihi + k.r0 + l];
                    }
                    if (k.set == 0 && k.r0 == 0 && l < W) { b0[g] = qt[l]; b1[g] = qt[c.o_blo + l]; }
                }
            }
            if (!hv[0]) break;
#pragma unroll
            for (int g = 0; g < QP_GROUP; ++g) {
                if (!hv[g]) break;
                const QpChunk cur = ck[g];
                qp_chunk_store<W>(cur, l, pre[g], s.js[wv]);
                if (l < QP_CHUNK) s.va[wv][l] = wa[g] - wb[g];
                if (cur.set == 0 && cur.r0 == 0) {
                    acc = b0[g] - b1[g];
                    if (l < W) {
                        const int grp = l / NA, k = l - grp * NA;
                        if (grp == 2) acc += c.ca * s.scan2[cur.t * QP_NA + k];
                        if (grp == 3) acc -= c.cb * s.scan2[cur.t * QP_NA + k];
                    }
                }
                wave_sync();
                if (l < W) {
                    const double* col = s.js[wv] + l;
                    double a0 = 0.0, a1 = 0.0;                               // (two chains: half the dependent latency)
                    int r = 0;
                    for (; r + 1 < cur.nr; r += 2) {
                        a0 = fma(col[r * W], s.va[wv][r], a0);
                        a1 = fma(col[(r + 1) * W], s.va[wv][r + 1], a1);
                    }
                    if (r < cur.nr) a0 = fma(col[r * W], s.va[wv][r], a0);
                    acc += a0 + a1;
                    if (cur.set == 1 && cur.r0 + cur.nr == c.R) c.atz[cur.t * W + l] = acc;
                }
                wave_sync();
            }
        }
    }
    __syncthreads();
    QSTAMP(7);
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
template <int NA> __device__ double qp_factor(QpCtx& c, QpLds& s, double reg) {
    constexpr int W = 4 * NA, HALF = W / 2;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const int R = c.R;
    double dmax = 0.0;
    // ---- P_t = Q_t + diag(box terms) + Jv' dv Jv + Ji' di Ji, masked.  The Jacobian part streams (lane (i, kh): row i, columns
    // [HALF kh, HALF kh + HALF)) into c.L; one elementwise pass adds the rest
    {
        const int i = l & 31, kh = l >> 5;
        double p[HALF];
        for (int base = 0;; base += QP_GROUP) {
            QpChunk ck[QP_GROUP];
            QpPre<W> pre[QP_GROUP];
            double w0[QP_GROUP], w1[QP_GROUP], w2[QP_GROUP], w3[QP_GROUP];       // row weight w0 / w1 + w2 / w3
            bool hv[QP_GROUP];
#pragma unroll
            for (int g = 0; g < QP_GROUP; ++g) {
                hv[g] = qp_chunk_at(c, wv, base + g, ck[g]);
                w0[g] = 0.0; w1[g] = 1.0; w2[g] = 0.0; w3[g] = 1.0;
                if (hv[g]) {
                    const QpChunk& k = ck[g];
                    qp_chunk_load<W>(c, k, l, pre[g]);
                    gcdbl *st = c.s + (int64_t)k.t * c.mp, *zt = c.z + (int64_t)k.t * c.mp;
                    if (l < k.nr) {
                        const int r = k.r0 + l;
                        if (k.set == 0) { w0[g] = zt[c.o_vhi + r]; w1[g] = st[c.o_vhi + r]; w2[g] = zt[c.o_vlo + r]; w3[g] = st[c.o_vlo + r]; }
                        else { w0[g] = zt[c.o_ihi + r]; w1[g] = st[c.o_ihi + r]; }
                    }
                }
            }
            if (!hv[0]) break;
#pragma unroll
            for (int g = 0; g < QP_GROUP; ++g) {
                if (!hv[g]) break;
                const QpChunk cur = ck[g];
                qp_chunk_store<W>(cur, l, pre[g], s.js[wv]);
                if (l < QP_CHUNK) s.va[wv][l] = w0[g] / w1[g] + w2[g] / w3[g];
                if (cur.set == 0 && cur.r0 == 0) {
#pragma unroll
                    for (int kk = 0; kk < HALF; ++kk) p[kk] = 0.0;
                }
                wave_sync();
                if (i < W) {
                    const double* js = s.js[wv];
#pragma unroll 4
                    for (int r = 0; r < cur.nr; ++r) {
                        const double a = js[r * W + i] * s.va[wv][r];
#pragma unroll
                        for (int kk = 0; kk < HALF; ++kk) p[kk] = fma(a, js[r * W + HALF * kh + kk], p[kk]);
                    }
                    if (cur.set == 1 && cur.r0 + cur.nr == R) {
#pragma unroll
                        for (int kk = 0; kk < HALF; ++kk) c.L[((int64_t)cur.t * W + i) * W + HALF * kh + kk] = p[kk];
                    }
                }
                wave_sync();
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < c.T * W * W; e += QP_THREADS) {
        const int t = e / (W * W), ik = e - t * W * W, i = ik / W, k = ik - i * W;
        gcdbl *st = c.s + (int64_t)t * c.mp, *zt = c.z + (int64_t)t * c.mp;
        double v = c.L[(int64_t)t * W * W + ik] + c.q[(int64_t)t * W * W + ik];
        if (k == i) v += zt[i] / st[i] + zt[c.o_blo + i] / st[c.o_blo + i];
        v = (c.fr[t * W + i] && c.fr[t * W + k]) ? v : 0.0;
        c.L[(int64_t)t * W * W + ik] = v;
        if (k == i) dmax = fmax(dmax, v);
    }
    for (int e = tid; e < c.T * NA; e += QP_THREADS) {
        const int t = e / NA, k = e - t * NA;
        gcdbl *st = c.s + (int64_t)t * c.mp, *zt = c.z + (int64_t)t * c.mp;
        c.D[t * QP_NA + k] = zt[c.o_ehi + k] / st[c.o_ehi + k] + zt[c.o_elo + k] / st[c.o_elo + k];
    }
    if (NA < QP_NA) {
        for (int e = tid; e < c.T * (QP_NA - NA); e += QP_THREADS) {    // padding up to QP_NA units: decoupled, D = 1
            const int t = e / (QP_NA - NA > 0 ? QP_NA - NA : 1), k = NA + e - t * (QP_NA - NA);
            c.D[t * QP_NA + k] = 1.0;
        }
    }
    const double gmax = block_reduce<1>(dmax, s.red);       // (barrier: P_t and D are in memory)
    QSTAMP(0);
    const double regv = reg * gmax, pfloor = fmax(gmax * 1e-20, 1e-300);
    c.regv = regv;
    double floored = 0.0;
    // ---- L_t = chol(P_t + regularisation), W_t = L_t^-1 G_t', M_t = W_t' W_t; lane i: row i.  The next period's row is
    // fetched while this one is factored.
    {
        const bool row = l < W;
        double p[W], pn[W];
        uint8_t fb = 0, fbn = 0;
        auto fetch = [&](int t, double (&dst)[W], uint8_t& f) {
            if (row) {
                gcdbl2* src = (gcdbl2*)(c.L + ((int64_t)t * W + l) * W);
#pragma unroll
                for (int k = 0; k < W / 2; ++k) { const dbl2 v = src[k]; dst[2 * k] = v.x; dst[2 * k + 1] = v.y; }
                f = c.fr[t * W + l];
            }
        };
#pragma unroll
        for (int k = 0; k < W; ++k) { p[k] = 0.0; pn[k] = 0.0; }
        if (wv < c.T) fetch(wv, p, fb);
        for (int t = wv; t < c.T; t += QP_WAVES) {
            if (t + QP_WAVES < c.T) fetch(t + QP_WAVES, pn, fbn);
            const double fi = row && fb ? 1.0 : 0.0;
            double invd[W];
#pragma unroll
            for (int k = 0; k < W; ++k) if (k == l) p[k] += regv + (1.0 - fi);
#pragma unroll
            for (int j = 0; j < W; ++j) {
                double djj = readlane64(p[j], j);
                if (!(djj > pfloor)) { djj = pfloor; floored += 1.0; }
                const double inv = 1.0 / sqrt(djj), ljj = djj * inv;
                invd[j] = inv;
                const double lij = (l == j) ? ljj : p[j] * inv;
                p[j] = lij;
#pragma unroll
                for (int k = j + 1; k < W; ++k) p[k] = fma(-lij, readlane64(lij, k), p[k]);       // (rows i >= k use it)
            }
            if (row) {
#pragma unroll
                for (int k = 0; k < W; ++k) c.L[((int64_t)t * W + l) * W + k] = (k <= l) ? p[k] : 0.0;
                double mine = 0.0;
#pragma unroll
                for (int k = 0; k < W; ++k) if (k == l) mine = invd[k];
                c.Li[t * W + l] = mine;
            }
            // W: forward substitution on the n_agents columns of G_t' (zero above row 2 na; G_0 = 0)
            double acc[NA], wr[NA];
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                acc[k] = 0.0; wr[k] = 0.0;
                if (row && t >= 1) {
                    if (l == 2 * NA + k) acc[k] = c.ca * fi;
                    if (l == 3 * NA + k) acc[k] = -c.cb * fi;
                }
            }
#pragma unroll
            for (int j = 2 * NA; j < W; ++j) {
#pragma unroll
                for (int k = 0; k < NA; ++k) {
                    const double wjk = readlane64(acc[k], j) * invd[j];
                    if (l == j) wr[k] = wjk;
                    if (l > j) acc[k] = fma(-p[j], wjk, acc[k]);
                }
            }
            if (row) {
#pragma unroll
                for (int k = 0; k < QP_NA; ++k) {
                    const double v = k < NA ? wr[k < NA ? k : 0] : 0.0;
                    c.Wm[((int64_t)t * W + l) * QP_NA + k] = v;
                    s.pm[wv][l * QP_NA + k] = v;
                }
            }
            wave_sync();
            if (l < QP_NA * QP_NA) {
                const int a = l / QP_NA, b = l - a * QP_NA;
                double m = 0.0;
#pragma unroll
                for (int r = 0; r < W; ++r) m = fma(s.pm[wv][r * QP_NA + a], s.pm[wv][r * QP_NA + b], m);
                c.M[(int64_t)t * QP_NA * QP_NA + l] = m;
            }
            wave_sync();
#pragma unroll
            for (int k = 0; k < W; ++k) p[k] = pn[k];
            fb = fbn;
        }
    }
    floored = block_reduce<0>(floored, s.red) / 64.0;       // (barrier: L, W, M are in memory; every lane counted the same pivots)
    QSTAMP(1);
    // ---- the recursion over the periods: one wavefront, every lane the same arithmetic on uniform data; its matrices stay in
    // LDS (seq(t)) for the solves.  M_{t-1} is fetched while step t is worked on.
    static_assert(sizeof(QpLds) <= 160 * 1024, "one work-group's LDS");
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
    QSTAMP(2);
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
template <int NA> __device__ void qp_solve(const QpCtx& c, QpLds& s) {
    constexpr int W = 4 * NA;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
    const bool row = l < W;
    // y_t = L_t^-1 rhs_t, g_t = W_t' y_t (g in LDS: scan2); the next period's operands are fetched during the substitution
    {
        double p[W], pn[W], inv = 0.0, invn = 0.0, rh = 0.0, rhn = 0.0, wm[QP_NA], wmn[QP_NA];
        auto fetch = [&](int t, double (&dst)[W], double& iv, double& r, double (&wd)[QP_NA]) {
            if (row) {
                gcdbl2* src = (gcdbl2*)(c.L + ((int64_t)t * W + l) * W);
#pragma unroll
                for (int k = 0; k < W / 2; ++k) { const dbl2 v = src[k]; dst[2 * k] = v.x; dst[2 * k + 1] = v.y; }
                iv = c.Li[t * W + l];
                r = c.rhs[t * W + l];
#pragma unroll
                for (int k = 0; k < QP_NA; ++k) wd[k] = c.Wm[((int64_t)t * W + l) * QP_NA + k];
            }
        };
#pragma unroll
        for (int k = 0; k < W; ++k) { p[k] = 0.0; pn[k] = 0.0; }
#pragma unroll
        for (int k = 0; k < QP_NA; ++k) { wm[k] = 0.0; wmn[k] = 0.0; }
        if (wv < c.T) fetch(wv, p, inv, rh, wm);
        for (int t = wv; t < c.T; t += QP_WAVES) {
            if (t + QP_WAVES < c.T) fetch(t + QP_WAVES, pn, invn, rhn, wmn);
            double acc = rh, y = 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const double yj = readlane64(acc, j) * readlane64(inv, j);
                if (l == j) y = yj;
                if (l > j) acc = fma(-p[j], yj, acc);
            }
            if (row) c.y[t * W + l] = y;
            // g = W' y: sum over the rows (lanes) of wm[k] y
#pragma unroll
            for (int k = 0; k < QP_NA; ++k) {
                double v = row ? wm[k] * y : 0.0;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);          // rows live in lanes 0..W-1 < 32
                if (l == 0) s.scan2[t * QP_NA + k] = v;
            }
#pragma unroll
            for (int k = 0; k < W; ++k) p[k] = pn[k];
#pragma unroll
            for (int k = 0; k < QP_NA; ++k) wm[k] = wmn[k];
            inv = invn; rh = rhn;
        }
    }
    __syncthreads();
    QSTAMP(3);
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
    QSTAMP(4);
    // dx_t = L_t^-T (y_t + W_t nu_t); lane i holds column i of L_t
    {
        double col[W], coln[W], inv = 0.0, invn = 0.0, a0 = 0.0, a0n = 0.0;
        uint8_t fb = 0, fbn = 0;
        auto fetch = [&](int t, double (&dst)[W], double& iv, double& acc0, uint8_t& f) {
            if (row) {
#pragma unroll
                for (int j = 0; j < W; ++j) dst[j] = c.L[((int64_t)t * W + j) * W + l];
                iv = c.Li[t * W + l];
                double acc = c.y[t * W + l];
#pragma unroll
                for (int k = 0; k < NA; ++k) acc = fma(c.Wm[((int64_t)t * W + l) * QP_NA + k], c.nu[t * QP_NA + k], acc);
                acc0 = acc;
                f = c.fr[t * W + l];
            }
        };
#pragma unroll
        for (int k = 0; k < W; ++k) { col[k] = 0.0; coln[k] = 0.0; }
        if (wv < c.T) fetch(wv, col, inv, a0, fb);
        for (int t = wv; t < c.T; t += QP_WAVES) {
            if (t + QP_WAVES < c.T) fetch(t + QP_WAVES, coln, invn, a0n, fbn);
            double acc = a0, dx = 0.0;
#pragma unroll
            for (int jj = 0; jj < W; ++jj) {
                const int j = W - 1 - jj;
                const double dj = readlane64(acc, j) * readlane64(inv, j);
                if (l == j) dx = dj;
                if (l < j) acc = fma(-col[j], dj, acc);
            }
            if (row) c.dx[t * W + l] = fb ? dx : 0.0;
#pragma unroll
            for (int k = 0; k < W; ++k) col[k] = coln[k];
            inv = invn; a0 = a0n; fb = fbn;
        }
    }
    __syncthreads();
    QSTAMP(5);
}

// One Newton solve: qp_solve on c.rhs, then (late in the iteration) one step of iterative refinement against N applied through its operators
// (N dx = Q dx + reg dx + sum A' (z / s) A dx on the free variables).  Leaves dx in c.dx and A dx in av / ai / ae.
template <int NA> __device__ void qp_newton_solve(const QpCtx& c, QpLds& s, bool refine) {
    constexpr int W = 4 * NA;
    const int tid = threadIdx.x;
    qp_solve<NA>(c, s);
    if (refine) {
        apply_A<NA>(c, s, c.dx);
#pragma unroll 4
        for (int idx = tid; idx < c.mt; idx += QP_THREADS)
            c.qr[idx] = (c.z[idx] / c.s[idx]) * c.ar[idx];     // (apply_At forms upper - lower: + A' d A dx for both)
        __syncthreads();
        apply_At<NA>(c, s, c.qr);
        for (int i = tid; i < c.n; i += QP_THREADS) {
            const int t = i / W, j = i - t * W;
            gcdbl* qrow = c.q + ((int64_t)t * W + j) * W;
            double acc = fma(c.regv, c.dx[i], c.atz[i]);
#pragma unroll
            for (int k = 0; k < W; ++k) acc = fma(qrow[k], c.dx[t * W + k], acc);
            c.dxa[i] = c.dx[i];
            c.rhs[i] = c.fr[i] ? c.rhs[i] - acc : 0.0;
        }
        __syncthreads();
        qp_solve<NA>(c, s);
        for (int i = tid; i < c.n; i += QP_THREADS) c.dx[i] += c.dxa[i];
        __syncthreads();
    }
    apply_A<NA>(c, s, c.dx);
}

template <int NA> __global__ __launch_bounds__(QP_THREADS) void qp_ipm_kernel(FlexQpArgs a) {
    constexpr int W = 4 * NA;
    __shared__ QpLds s;
    const int tid = threadIdx.x, b = blockIdx.x;
    QpCtx c;
    c.T = a.periods; c.na = NA; c.w = W; c.R = a.rows;
    c.mp = (int)qp_rows_per_period(c.na, c.R); c.n = c.T * c.w; c.mt = c.T * c.mp;
    c.o_blo = c.w; c.o_vhi = 2 * c.w; c.o_vlo = c.o_vhi + c.R; c.o_ihi = c.o_vlo + c.R; c.o_ehi = c.o_ihi + c.R; c.o_elo = c.o_ehi + c.na;
    c.ca = a.chain_a; c.cb = a.chain_b; c.regv = 0.0;
    const int64_t bn = (int64_t)b * c.n, bR = (int64_t)b * c.T * c.R, be = (int64_t)b * c.T * c.na;
    c.q = (gcdbl*)a.q + bn * c.w; c.c = (gcdbl*)a.c + bn; c.lo = (gcdbl*)a.lo + bn; c.hi = (gcdbl*)a.hi + bn; c.fr = (gcu8*)a.free_mask + bn;
    c.jv = (gcdbl*)a.jv + bR * c.w; c.vlo = (gcdbl*)a.v_lo + bR; c.vhi = (gcdbl*)a.v_hi + bR; c.ji = (gcdbl*)a.ji + bR * c.w; c.ihi = (gcdbl*)a.i_hi + bR;
    c.elo = (gcdbl*)a.e_lo + be; c.ehi = (gcdbl*)a.e_hi + be;
    c.x = (gdbl*)a.x + bn;
    gdbl* wk = (gdbl*)a.work + (int64_t)b * qp_work_doubles(c.T, c.na, c.R);
    const int64_t mte = qp_even(c.mt), tne = qp_even((int64_t)c.T * QP_NA);
    c.s = wk; wk += mte; c.z = wk; wk += mte; c.rp = wk; wk += mte; c.ds = wk; wk += mte; c.dz = wk; wk += mte; c.qr = wk; wk += mte;
    c.rd = wk; wk += c.n; c.rhs = wk; wk += c.n; c.dx = wk; wk += c.n; c.dxa = wk; wk += c.n; c.y = wk; wk += c.n; c.atz = wk; wk += c.n;
    c.ar = wk; wk += mte; c.hs = wk; wk += mte;
    c.L = wk; wk += (int64_t)c.T * W * W; c.Li = wk; wk += c.n; c.Wm = wk; wk += (int64_t)c.T * W * QP_NA;
    c.M = wk; wk += qp_even((int64_t)c.T * QP_NA * QP_NA); c.D = wk; wk += tne; c.nu = wk;
    gcdbl* x0 = (gcdbl*)a.x0 + bn;
    const double m_tot = (double)c.mt;

    QSTAMP(-1);
    for (int i = tid; i < c.n; i += QP_THREADS) c.x[i] = x0[i];
    // (the padded rows of the chain arrays read by the recursion)
    for (int i = tid; i < c.T * QP_NA; i += QP_THREADS) c.nu[i] = 0.0;
    for (int idx = tid; idx < c.mt; idx += QP_THREADS) c.hs[idx] = row_bound(c, idx);
    __syncthreads();
    apply_A<NA>(c, s, c.x);
    for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
        c.s[idx] = fmax(c.hs[idx] - c.ar[idx], 1e-3);
        c.z[idx] = 1.0;
    }
    __syncthreads();
    int it = 0, done = 0;
    double mu = 0.0, res_d = 0.0, res_p = 0.0, floored = 0.0;
    for (it = 0; it < a.max_iter; ++it) {
        if (it > 0) apply_A<NA>(c, s, c.x);
        double sz = 0.0, rpm = 0.0;
#pragma unroll 4
        for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
            const double si = c.s[idx], rp = c.ar[idx] + si - c.hs[idx];
            c.rp[idx] = rp;
            sz = fma(si, c.z[idx], sz);
            rpm = fmax(rpm, fabs(rp));
        }
        QSTAMP(9);
        apply_At<NA>(c, s, c.z);
        double rdm = 0.0;
        for (int i = tid; i < c.n; i += QP_THREADS) {
            const int t = i / W, j = i - t * W;
            gcdbl* qrow = c.q + ((int64_t)t * W + j) * W;
            double acc = c.c[i] + c.atz[i];
#pragma unroll
            for (int k = 0; k < W; ++k) acc = fma(qrow[k], c.x[t * W + k], acc);
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
        QSTAMP(8);
        floored += qp_factor<NA>(c, s, a.reg);

        double sigma_mu = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            // right-hand side: -r_d - sum sg A' ((z r_p - r_c) / s)
#pragma unroll 4
            for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
                const double si = c.s[idx], zi = c.z[idx];
                const double rc = pass == 0 ? si * zi : fma(c.ds[idx], c.dz[idx], si * zi) - sigma_mu;
                c.qr[idx] = (zi * c.rp[idx] - rc) / si;            // (apply_At applies the sets' signs: upper - lower)
            }
            __syncthreads();
            QSTAMP(9);
            apply_At<NA>(c, s, c.qr);
            for (int i = tid; i < c.n; i += QP_THREADS) c.rhs[i] = c.fr[i] ? -c.rd[i] - c.atz[i] : 0.0;
            __syncthreads();
            QSTAMP(8);
            qp_newton_solve<NA>(c, s, mu < QP_REFINE_BELOW);
            double rs = INFINITY, rz = INFINITY;
#pragma unroll 4
            for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
                const double si = c.s[idx], zi = c.z[idx];
                const double rc = pass == 0 ? si * zi : fma(c.ds[idx], c.dz[idx], si * zi) - sigma_mu;
                const double dsi = -c.rp[idx] - c.ar[idx], dzi = (-rc - zi * dsi) / si;
                c.ds[idx] = dsi;
                c.dz[idx] = dzi;
                if (dsi < 0.0) rs = fmin(rs, -si / dsi);
                if (dzi < 0.0) rz = fmin(rz, -zi / dzi);
            }
            rs = block_reduce<2>(rs, s.red);
            rz = block_reduce<2>(rz, s.red);
            QSTAMP(9);
            if (pass == 0) {
                const double ap = fmin(rs, 1.0), ad = fmin(rz, 1.0);
                double acc = 0.0;
#pragma unroll 4
                for (int idx = tid; idx < c.mt; idx += QP_THREADS)
                    acc = fma(fma(ap, c.ds[idx], c.s[idx]), fma(ad, c.dz[idx], c.z[idx]), acc);
                const double mu_a = block_reduce<0>(acc, s.red) / m_tot;
                const double sigma = fmin(fmax(mu_a / mu, 0.0), 1.0);
                sigma_mu = sigma * sigma * sigma * mu;
            } else {
                const double ap = fmin(0.995 * rs, 1.0), ad = fmin(0.995 * rz, 1.0);
                for (int i = tid; i < c.n; i += QP_THREADS) c.x[i] = fma(ap, c.dx[i], c.x[i]);
#pragma unroll 4
                for (int idx = tid; idx < c.mt; idx += QP_THREADS) {
                    c.s[idx] = fma(ap, c.ds[idx], c.s[idx]);
                    c.z[idx] = fma(ad, c.dz[idx], c.z[idx]);
                }
                __syncthreads();
                QSTAMP(9);
            }
        }
    }
    QSTAMP(8);
#ifdef QP_STAMPS
    if (b == 0 && tid == 0) {
        printf("qp stamps (100 MHz ticks) it %d: P %lld chol %lld riccati %lld | solve: fwd %lld serial %lld back %lld | A %lld At %lld rest %lld rows %lld\n",
               it, qp_stamp_acc[0], qp_stamp_acc[1], qp_stamp_acc[2], qp_stamp_acc[3], qp_stamp_acc[4], qp_stamp_acc[5], qp_stamp_acc[6],
               qp_stamp_acc[7], qp_stamp_acc[8], qp_stamp_acc[9]);
        for (int i = 0; i < QP_NSTAMP; ++i) qp_stamp_acc[i] = 0;
    }
#endif
    if (it >= a.max_iter) it = a.max_iter - 1;
    gdbl* du = (gdbl*)a.duals + (int64_t)b * c.mt;
    for (int idx = tid; idx < c.mt; idx += QP_THREADS) du[idx] = c.z[idx];
    if (tid == 0) {
        gdbl* info = (gdbl*)a.info + (int64_t)b * FLEXOPF_INFO;
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
    const dim3 grid(a->batch), block(QP_THREADS);
    hipStream_t st = (hipStream_t)stream;
    switch (a->n_agents) {
        case 1: hipLaunchKernelGGL(qp_ipm_kernel<1>, grid, block, 0, st, *a); break;
        case 2: hipLaunchKernelGGL(qp_ipm_kernel<2>, grid, block, 0, st, *a); break;
        case 3: hipLaunchKernelGGL(qp_ipm_kernel<3>, grid, block, 0, st, *a); break;
        case 4: hipLaunchKernelGGL(qp_ipm_kernel<4>, grid, block, 0, st, *a); break;
        default: hipLaunchKernelGGL(qp_ipm_kernel<5>, grid, block, 0, st, *a); break;
    }
    return hipGetLastError() == hipSuccess ? FLEXOPF_OK : FLEXOPF_EHIP;
}

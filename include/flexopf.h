/* flexopf.h — C ABI of the OPF comparator's quadratic-program solve (libflexenv_hip.so, gfx950).
 *
 * The reference states one multi-period optimal power flow per day as a Pyomo MIQCP and hands it to Gurobi
 * (utils/opf.py:13-192; caller run_opf.py:71).  The MI355X re-design (safe-marl_amd/opf.py) solves MANY days at once by
 * sequential convex programming: every outer iteration linearises the network around the current controls with the batched HIP
 * power flow (include/flexenv.h: pf_solve_batch) and solves the convex QP below for the whole horizon.  This entry point is
 * that QP solve: the whole Mehrotra predictor-corrector interior-point iteration of one day runs inside ONE persistent
 * work-group (one launch for the batch, no host round trip per iteration), and its Newton system is solved by a Riccati
 * recursion over the periods — the storage energy chain opf.py:139-148 is the only coupling across periods, so the normal
 * matrix is (block diagonal) + (cumulative-sum Gram) and factors in O(T w^3) instead of O((T w)^3).
 *
 * The program, per instance b (T periods, w = 4 n_agents controls per period ordered [Pred | Qpv | Pesc | Pesd], opf.py:54-58):
 *     minimise   1/2 sum_t x_t' Q_t x_t + c' x
 *     subject to lo <= x <= hi                                     (opf.py:46,54-57,118-120; pinned controls widened)
 *                v_lo <= Jv_t x_t <= v_hi        rows per period   (linearised voltage band opf.py:131-133)
 *                        Ji_t x_t <= i_hi        rows per period   (linearised current limit opf.py:135-137)
 *                e_lo <= E_t <= e_hi,  E_t[k] = sum_{1 <= s <= t} (a Pesc[s, k] - b Pesd[s, k])      (opf.py:139-148: period 0
 *                                                                   contributes nothing)
 * Variables with free == 0 stay at x0.  Same iteration, stopping rule and step rule as safe-marl_amd/opf.py: qp_ipm (the pure
 * torch form the tests compare against).
 */
#ifndef FLEXOPF_H
#define FLEXOPF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FLEXOPF_OK 0
#define FLEXOPF_EINVAL (-1)
#define FLEXOPF_EHIP (-2)
#define FLEXOPF_MAX_PERIODS 128
#define FLEXOPF_MAX_AGENTS 5
#define FLEXOPF_MAX_ROWS 64          /* network rows per period and set (n_bus - 1) */
#define FLEXOPF_INFO 6               /* doubles per instance in `info` */

typedef struct {
    int32_t batch, periods, n_agents, rows;
    int32_t max_iter, pad0;
    double tol, reg;                 /* duality-measure tolerance (1e-11), relative diagonal regularisation (1e-12) */
    double chain_a, chain_b;         /* dt eta_ch, dt / eta_dis */
    const double* q;                 /* [batch, T, w, w] block-diagonal Hessian (symmetric positive semi-definite blocks) */
    const double* c;                 /* [batch, T, w] */
    const double* lo;                /* [batch, T, w] */
    const double* hi;                /* [batch, T, w] */
    const uint8_t* free_mask;        /* [batch, T, w] 1: the variable may move */
    const double* jv;                /* [batch, T, rows, w] */
    const double* v_lo;              /* [batch, T, rows] */
    const double* v_hi;              /* [batch, T, rows] */
    const double* ji;                /* [batch, T, rows, w] */
    const double* i_hi;              /* [batch, T, rows] */
    const double* e_lo;              /* [batch, T, n_agents] */
    const double* e_hi;              /* [batch, T, n_agents] */
    const double* x0;                /* [batch, T, w] start (strictly inside the box for free variables) */
    double* x;                       /* out [batch, T, w] */
    double* duals;                   /* out [batch, T, 2 w + 3 rows + 2 n_agents]: multipliers per period in the order
                                        [box upper | box lower | v upper | v lower | i upper | e upper | e lower] */
    double* info;                    /* out [batch, FLEXOPF_INFO]: iterations, mu, dual residual, primal residual,
                                        converged (0/1), pivots floored in the factorisations */
    double* work;                    /* scratch, flexopf_qp_work_doubles(...) doubles per instance */
} FlexQpArgs;

/* scratch doubles per instance for the given sizes (negative: sizes out of range) */
int64_t flexopf_qp_work_doubles(int32_t periods, int32_t n_agents, int32_t rows);
int flexopf_qp_solve(const FlexQpArgs* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif

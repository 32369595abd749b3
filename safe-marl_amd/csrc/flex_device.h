// Device-side building blocks of the flexibility-provision hot path (gfx950 only).
//
// Execution model: one LANE per PQ bus, one lane GROUP per environment.  The slack bus carries no
// unknown (V = 1∠0, pf.py:51-53) and gets no lane, so the 33-bus feeder needs 32 lanes and a 64-wide
// wavefront serves TWO environments (EPW = 2, group width LW = 32); feeders with more than 32 PQ buses
// run one environment per wavefront (EPW = 1, LW = 64).  Every cross-lane operation stays inside a
// group: DPP row operations never leave a 16-lane row, the row_bcast:15 step joins the two rows of a
// 32-lane group (row_bcast:31 is added only for LW = 64), and ds_bpermute sources are group-relative.
//
// Lanes are numbered in a depth-first preorder of the radial feeder (children of the slack first), so
// a subtree is a contiguous lane range and a chain bus has its only child in lane+1.  Everything an
// environment needs per step — its series row, its 20 actions, its voltage vector — is read with
// coalesced one-double-per-lane loads straight into registers; there is no cross-environment reuse,
// so nothing is staged through LDS (the network tables are <=64-entry arrays shared by every
// wavefront and live in L2/L1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "flexenv.h"

#define FLEX_WAVE 64
#if defined(FLEX_STAMPS) && !defined(FLEX_GROUP_SWEEP_STAT)
#define FLEX_GROUP_SWEEP_STAT 1
#endif
#ifndef FLEX_WAVES_PER_BLOCK
#define FLEX_WAVES_PER_BLOCK 4
#endif
#define FLEX_JUMP_ROUNDS 6           // 2^6 >= FLEX_MAX_BUS

// Network tables in group-local LANE order (device memory, one copy per handle).
struct DevNet {
    int32_t n_bus, n_pq, n_levels, max_children, n_agents, slack_bus, epw, pad0;
    int32_t bus_of_lane[FLEX_MAX_BUS];      // bus index of the PQ bus in this lane, -1 for idle lanes
    int32_t lane_of_bus[FLEX_MAX_BUS];      // -1 for the slack
    int32_t par_lane[FLEX_MAX_BUS];         // parent lane; own lane when the parent is the slack (or idle)
    int32_t par_slack[FLEX_MAX_BUS];        // 1 when the parent is the slack bus
    int32_t level[FLEX_MAX_BUS];            // distance from the slack (>= 1); -1 for idle lanes
    int32_t agent_of_lane[FLEX_MAX_BUS];    // -1 if the bus has no building
    int32_t lane_of_agent[FLEX_MAX_AGENTS];
    int32_t slots_at_level[FLEX_MAX_BUS];   // child slots in use by receivers at level L-1
    int32_t child_lane[FLEX_MAX_CHILDREN][FLEX_MAX_BUS];  // -1 padded
    double g[FLEX_MAX_BUS], b[FLEX_MAX_BUS];   // series admittance of the line to the parent: 1/(r+jx)
    double gd[FLEX_MAX_BUS], bd[FLEX_MAX_BUS]; // Ybus diagonal: own line + children's lines
    double r[FLEX_MAX_BUS], x[FLEX_MAX_BUS];
    // sweep solver: subtree(l) = lanes [l, sub_end[l]]; chain segments of the preorder (lane l continues the
    // segment of l-1 iff its parent is l-1): seg_start = first lane, seg_par = lane of the parent of the segment
    // head, seg_depth = segment hops to a segment hanging off the slack; anc[k][l] = 2^k-th ancestor or -1.
    int32_t sub_end[FLEX_MAX_BUS];
    int32_t seg_start[FLEX_MAX_BUS], seg_par[FLEX_MAX_BUS], seg_depth[FLEX_MAX_BUS];
    int32_t anc[FLEX_JUMP_ROUNDS][FLEX_MAX_BUS];
    int32_t n_jump_rounds, n_seg_rounds;
    // sweep acceleration (pf_sweep): the dominant eigenvalue of the TWO-sweep error map is estimated as
    // acc_kappa * |1 - V[acc_lane]|^2 — calibrated on the host when the handle is created (calibrate_sweep_accel);
    // acc_kappa = 0 switches the extrapolation off (bare pf_solve_batch, FlexCfg::no_sweep_accel)
    int32_t acc_lane;
    float acc_kappa;
    // the sweeps stop on their local mismatch estimate at sweep_tol_frac * pf_tol (pf_solve)
    float sweep_tol_frac, pad1;
};

// Per-lane registers: this bus's row of the Ybus and its place in the tree.
struct LaneNet {
    int lane, l, base, grp;      // wavefront lane, group-local lane, first lane of the group, group index
    int bus, par, lev, agent;
    bool pq, par_slack;
    double g, b, gd, bd, r, x;
    int cha[FLEX_MAX_CHILDREN];  // ds_bpermute byte addresses (4 x wavefront lane) of the children, -1: none.  Only the slots
                                 // the feeder uses (net->max_children: 2 on IEEE-33) are loaded; the others are never read
    int par_addr;                // the parent's, likewise
    int sub_end, seg_par, seg_depth;
    int mk[6];                   // high words (1.0 or 0.0) of the segmented-scan step masks
};

template <int EPW>
__device__ __forceinline__ void load_lane_net(const DevNet* __restrict__ net, int lane, LaneNet& ln) {
    constexpr int LW = FLEX_WAVE / EPW;
    const int l = lane & (LW - 1), base = lane - l;
    ln.lane = lane; ln.l = l; ln.base = base; ln.grp = lane / LW;
    ln.bus = net->bus_of_lane[l];
    ln.par = net->par_lane[l] + base;
    ln.par_slack = net->par_slack[l] != 0;
    ln.lev = net->level[l];
    ln.agent = net->agent_of_lane[l];
    ln.pq = ln.bus >= 0;
    ln.g = net->g[l]; ln.b = net->b[l]; ln.gd = net->gd[l]; ln.bd = net->bd[l];
    ln.r = net->r[l]; ln.x = net->x[l];
    ln.par_addr = ln.par << 2;
    // Child slots 0 and 1 unconditionally (a conditional load is a branch with a full s_waitcnt behind it: a memory round trip
    // of its own in the prologue); the other six in ONE uniform branch, which IEEE-33 (two children at most) never takes
#pragma unroll
    for (int k = 0; k < FLEX_MAX_CHILDREN; ++k) {
        if (k < 2) {
            const int c = net->child_lane[k][l];
            ln.cha[k] = c >= 0 ? (c + base) << 2 : -1;
        } else {
            ln.cha[k] = -1;
        }
    }
    if (net->max_children > 2) {
        int c[FLEX_MAX_CHILDREN];
#pragma unroll
        for (int k = 2; k < FLEX_MAX_CHILDREN; ++k) c[k] = net->child_lane[k][l];
#pragma unroll
        for (int k = 2; k < FLEX_MAX_CHILDREN; ++k) ln.cha[k] = c[k] >= 0 ? (c[k] + base) << 2 : -1;
    }
    ln.sub_end = net->sub_end[l] + base;
    ln.seg_par = net->seg_par[l] + base;
    ln.seg_depth = net->seg_depth[l];
    const int ss = net->seg_start[l], row = l >> 4;
    const int one = 0x3FF00000;
    ln.mk[0] = (l - 1 >= ss) ? one : 0;
    ln.mk[1] = (l - 2 >= ss) ? one : 0;
    ln.mk[2] = (l - 4 >= ss) ? one : 0;
    ln.mk[3] = (l - 8 >= ss) ? one : 0;
    ln.mk[4] = (ss <= 16 * row - 1) ? one : 0;     // row_bcast:15 into the odd rows
    ln.mk[5] = (ss <= 31) ? one : 0;               // row_bcast:31 into rows 2,3 (LW = 64 only)
}

// ---- group-wide scans and reductions on the DPP path (no LDS traffic) ------------------------------
// gfx9 DPP controls: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK, bool BOUND_CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, BOUND_CTRL);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, BOUND_CTRL);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the lanes of each group (lanes that must not contribute pass 0)
template <int EPW>
__device__ __forceinline__ double grp_scan_sum(double x) {
    x += dpp_mov_f64<0x111, 0xF, true>(x);
    x += dpp_mov_f64<0x112, 0xF, true>(x);
    x += dpp_mov_f64<0x114, 0xF, true>(x);
    x += dpp_mov_f64<0x118, 0xF, true>(x);
    x += dpp_mov_f64<0x142, 0xA, false>(x);   // lane 15 of rows 0,2 -> rows 1,3
    if constexpr (EPW == 1) x += dpp_mov_f64<0x143, 0xC, false>(x);   // lane 31 -> rows 2,3
    return x;
}
// the same, restarted at every chain segment: step masks are 1.0 where the source lane lies in the same
// segment, 0.0 elsewhere (precomputed per lane, off the critical path)
template <int EPW>
__device__ __forceinline__ double grp_segscan_sum(double x, const int (&mk)[6]) {
    x = fma(dpp_mov_f64<0x111, 0xF, true>(x), __hiloint2double(mk[0], 0), x);
    x = fma(dpp_mov_f64<0x112, 0xF, true>(x), __hiloint2double(mk[1], 0), x);
    x = fma(dpp_mov_f64<0x114, 0xF, true>(x), __hiloint2double(mk[2], 0), x);
    x = fma(dpp_mov_f64<0x118, 0xF, true>(x), __hiloint2double(mk[3], 0), x);
    x = fma(dpp_mov_f64<0x142, 0xA, false>(x), __hiloint2double(mk[4], 0), x);
    if constexpr (EPW == 1) x = fma(dpp_mov_f64<0x143, 0xC, false>(x), __hiloint2double(mk[5], 0), x);
    return x;
}
// Two independent scans advanced step by step: the DPP move -> add chain of one component fills the
// dependency bubbles of the other (issued back to back they would run one after the other).
template <int EPW>
__device__ __forceinline__ void grp_scan_sum2(double& x, double& y) {
#define FLEX_SCAN2_STEP(CTRL, RM, BC) { const double tx = dpp_mov_f64<CTRL, RM, BC>(x), ty = dpp_mov_f64<CTRL, RM, BC>(y); x += tx; y += ty; }
    FLEX_SCAN2_STEP(0x111, 0xF, true)
    FLEX_SCAN2_STEP(0x112, 0xF, true)
    FLEX_SCAN2_STEP(0x114, 0xF, true)
    FLEX_SCAN2_STEP(0x118, 0xF, true)
    FLEX_SCAN2_STEP(0x142, 0xA, false)
    if constexpr (EPW == 1) FLEX_SCAN2_STEP(0x143, 0xC, false)
#undef FLEX_SCAN2_STEP
}
template <int EPW>
__device__ __forceinline__ void grp_segscan_sum2(double& x, double& y, const int (&mk)[6]) {
#define FLEX_SEG2_STEP(CTRL, RM, BC, K) { const double m = __hiloint2double(mk[K], 0); \
    const double tx = dpp_mov_f64<CTRL, RM, BC>(x), ty = dpp_mov_f64<CTRL, RM, BC>(y); x = fma(tx, m, x); y = fma(ty, m, y); }
    FLEX_SEG2_STEP(0x111, 0xF, true, 0)
    FLEX_SEG2_STEP(0x112, 0xF, true, 1)
    FLEX_SEG2_STEP(0x114, 0xF, true, 2)
    FLEX_SEG2_STEP(0x118, 0xF, true, 3)
    FLEX_SEG2_STEP(0x142, 0xA, false, 4)
    if constexpr (EPW == 1) FLEX_SEG2_STEP(0x143, 0xC, false, 5)
#undef FLEX_SEG2_STEP
}
// fp32 flavours for the sweep solver's increments: one 32-bit DPP operand per step, which the compiler folds into
// the add / fma itself (v_add_f32_dpp, v_fmac_f32_dpp) — one instruction per component and step where the fp64
// scans need three.
template <int CTRL, int ROW_MASK, bool BOUND_CTRL>
__device__ __forceinline__ float dpp_mov_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xF, BOUND_CTRL));
}
// Round 5: written out as the instructions wanted.  From the C++ form the compiler (a) packed the two components' fma into
// v_pk_fma_f32 behind two v_mov_b32_dpp — three issue slots per step where two v_fmac_f32_dpp do — and (b) could not fold the
// row_bcast step (rows 1 and 3 only) into its add, because x + 0.0 is not an identity for x = -0.0: zero, v_mov_b32_dpp, add
// per component.  A DPP instruction with a partial row_mask leaves the destination of the other rows alone, which is the
// wanted result outright.  DPP reads of a VGPR need two wait states behind the VALU write: the other component's
// instruction is one, s_nop 0 the other (the assembler does not insert them in inline code).  11 issue slots less per sweep.
#define FLEX_DPP_ROW(n) " row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define FLEX_DPP_BC15 " row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define FLEX_DPP_BC31 " row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
template <int EPW>
__device__ __forceinline__ void grp_scan_sum2_f32(float& x, float& y) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0" FLEX_DPP_ROW(1) "v_add_f32_dpp %1, %1, %1" FLEX_DPP_ROW(1) "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0" FLEX_DPP_ROW(2) "v_add_f32_dpp %1, %1, %1" FLEX_DPP_ROW(2) "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0" FLEX_DPP_ROW(4) "v_add_f32_dpp %1, %1, %1" FLEX_DPP_ROW(4) "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0" FLEX_DPP_ROW(8) "v_add_f32_dpp %1, %1, %1" FLEX_DPP_ROW(8) "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0" FLEX_DPP_BC15 "v_add_f32_dpp %1, %1, %1" FLEX_DPP_BC15
                 : "+v"(x), "+v"(y));
    if constexpr (EPW == 1)
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0" FLEX_DPP_BC31 "v_add_f32_dpp %1, %1, %1" FLEX_DPP_BC31
                     : "+v"(x), "+v"(y));
}
template <int EPW>
__device__ __forceinline__ void grp_segscan_sum2_f32(float& x, float& y, const float (&m)[6]) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %0, %2" FLEX_DPP_ROW(1) "v_fmac_f32_dpp %1, %1, %2" FLEX_DPP_ROW(1) "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %3" FLEX_DPP_ROW(2) "v_fmac_f32_dpp %1, %1, %3" FLEX_DPP_ROW(2) "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %4" FLEX_DPP_ROW(4) "v_fmac_f32_dpp %1, %1, %4" FLEX_DPP_ROW(4) "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %5" FLEX_DPP_ROW(8) "v_fmac_f32_dpp %1, %1, %5" FLEX_DPP_ROW(8) "s_nop 0\n\t"
                 "v_fmac_f32_dpp %0, %0, %6" FLEX_DPP_BC15 "v_fmac_f32_dpp %1, %1, %6" FLEX_DPP_BC15
                 : "+v"(x), "+v"(y) : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]));
    if constexpr (EPW == 1)
        asm volatile("s_nop 1\n\t"
                     "v_fmac_f32_dpp %0, %0, %2" FLEX_DPP_BC31 "v_fmac_f32_dpp %1, %1, %2" FLEX_DPP_BC31
                     : "+v"(x), "+v"(y) : "v"(m[5]));
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                            __builtin_amdgcn_readlane(__double2loint(x), l));
}
// the value lane `byte_addr / 4` holds (ds_bpermute on both words; the address is precomputed: __shfl rebuilds it —
// mask, or, shift — at every call)
__device__ __forceinline__ double pull_f64(double x, int byte_addr) {
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(x)),
                            __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(x)));
}
// sum over the lanes of each group, returned in every lane of that group
template <int EPW>
__device__ __forceinline__ double grp_sum(double v, int grp) {
    const double s = grp_scan_sum<EPW>(v);
    if constexpr (EPW == 1) {
        return readlane_f64(s, 63);
    } else {
        const double t0 = readlane_f64(s, 31), t1 = readlane_f64(s, 63);
        return grp ? t1 : t0;
    }
}
// Five independent group sums advanced step by step (the reward's five terms): issued one after the other each
// scan is a chain of dependent DPP moves and adds; interleaved, the chains hide each other's latency.
// BROADCAST = false: the sums are left where the inclusive scan puts them — in the LAST lane of each group — and the caller
// lets that lane do the stores (round 5: handing them to every lane was four v_readlane, two moves and two selects per
// value, forty instructions for results one lane writes).
template <int EPW, bool BROADCAST = true>
__device__ __forceinline__ void grp_sum5(double (&v)[5], int grp) {
#define FLEX_SUM5_STEP(CTRL, RM, BC) { double t[5]; \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) t[i] = dpp_mov_f64<CTRL, RM, BC>(v[i]); \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) v[i] += t[i]; }
    FLEX_SUM5_STEP(0x111, 0xF, true)
    FLEX_SUM5_STEP(0x112, 0xF, true)
    FLEX_SUM5_STEP(0x114, 0xF, true)
    FLEX_SUM5_STEP(0x118, 0xF, true)
    FLEX_SUM5_STEP(0x142, 0xA, false)
    if constexpr (EPW == 1) FLEX_SUM5_STEP(0x143, 0xC, false)
#undef FLEX_SUM5_STEP
    if constexpr (!BROADCAST) return;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if constexpr (EPW == 1) {
            v[i] = readlane_f64(v[i], 63);
        } else {
            const double t0 = readlane_f64(v[i], 31), t1 = readlane_f64(v[i], 63);
            v[i] = grp ? t1 : t0;
        }
    }
}

// "does any lane of MY group raise the flag", plus the wavefront-wide answer for loop control
template <int EPW>
__device__ __forceinline__ bool grp_any(bool p, int grp, bool& wave_any) {
    const unsigned long long m = __ballot(p);
    wave_any = m != 0ull;
    if constexpr (EPW == 1) {
        return wave_any;
    } else {
        return (grp ? (unsigned)(m >> 32) : (unsigned)m) != 0u;
    }
}

// 1/d from v_rcp_f64 plus one Newton step: ~3 dependent instructions instead of the ~15 of an
// IEEE division.  The solvers are self-correcting iterations, so a last-bit error is immaterial.
__device__ __forceinline__ double fast_rcp(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    return fma(fma(-d, r, 1.0), r, r);
}

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// The local (per-bus) part of one Newton step, shared by the tree and the dense linear solvers:
// injected current from branch currents, power mismatch test, right-hand side I_spec - I_calc and the diagonal
// 2x2 Jacobian block  Yd - d(conj(S/V))/d(e,f).
struct NewtonLocal { double rhs0, rhs1, d11, d12, d21, d22; bool miss; };
__device__ __forceinline__ NewtonLocal newton_local(const LaneNet& ln, int maxc, double ps, double qs, double e,
                                                    double f, double tol) {
    NewtonLocal o;
    const double ep0 = pull_f64(e, ln.par_addr), fp0 = pull_f64(f, ln.par_addr);
    const double ep = ln.par_slack ? 1.0 : ep0, fp = ln.par_slack ? 0.0 : fp0;
    const double de = ep - e, df = fp - f;
    const double jr = ln.g * de - ln.b * df, ji = ln.b * de + ln.g * df;   // branch current parent -> bus
    double ir = -jr, ii = -ji;   // current injected at this bus = children's inflow - own inflow
    // (the child slots in use are 0 .. maxc - 1: leave at the first unused one — as eight independently guarded blocks the
    //  unused slots cost a scalar test and a branch each on every evaluation)
#pragma unroll
    for (int k = 0; k < FLEX_MAX_CHILDREN; ++k) {
        if (k >= maxc) break;
        const bool has = ln.cha[k] >= 0;
        const int src = has ? ln.cha[k] : (ln.lane << 2);
        const double tr = pull_f64(jr, src), ti = pull_f64(ji, src);
        ir += has ? tr : 0.0;
        ii += has ? ti : 0.0;
    }
    // power mismatch  S_calc - S_spec,  S_calc = V conj(I);  NaN counts as a miss
    const double dP = e * ir + f * ii - ps;
    const double dQ = f * ir - e * ii - qs;
    o.miss = ln.pq && !(fmax(fabs(dP), fabs(dQ)) < tol);
    // specified current conj(S/V) and its derivative wrt (e, f)
    const double inv_d = fast_rcp(e * e + f * f);
    const double isr = (ps * e + qs * f) * inv_d, isi = (ps * f - qs * e) * inv_d;
    o.rhs0 = isr - ir; o.rhs1 = isi - ii;
    o.d11 = ln.gd - (ps - 2.0 * e * isr) * inv_d;
    o.d12 = -ln.bd - (qs - 2.0 * f * isr) * inv_d;
    o.d21 = ln.bd - (-qs - 2.0 * e * isi) * inv_d;
    o.d22 = ln.gd - (ps - 2.0 * f * isi) * inv_d;
    return o;
}

// ---- Newton-Raphson on the Ybus, current-mismatch form, rectangular coordinates ---------------
// Unknowns per PQ bus: V = e + jf.  Residual per bus (utils/pf.py:65-94 restated on the Ybus,
// SURVEY.md App. B):   R_i = sum_k Y_ik V_k - conj(S_i / V_i),   S_i = -(Pnet_i + j Qnet_i).
// On a radial feeder row i of the Ybus couples bus i to its parent and its children only, so the
// injected current is formed from branch currents J_i = y_i (V_parent - V_i): one pull from the
// parent lane (the constant 1∠0 when the parent is the slack), one per child slot.  The Jacobian's
// off-diagonal 2x2 blocks are the constant [[g,-b],[b,g]] of each line; only the diagonal block
// depends on V.  The Newton step solves J dV = -R exactly by leaf->root block elimination (no
// fill-in on a tree) and root->leaf back-substitution.  Convergence is tested on the POWER mismatch
// inf-norm per environment: "does any lane of my group still miss the tolerance" (a ballot, no
// reduction).  A group that has converged keeps stepping harmlessly while its neighbour finishes.
//
// Returns (per group) true when converged; `iters` = Newton steps taken until then.
template <int EPW>
__device__ __forceinline__ bool pf_newton_tree(const DevNet* __restrict__ net, const LaneNet& ln,
                                               double pnet, double qnet, double& e, double& f,
                                               double tol, int max_iter, int& iters) {
    const int n_levels = net->n_levels, maxc = net->max_children;
    const double ps = -pnet, qs = -qnet;
    const int lane = ln.lane;
    bool ok = false;
    iters = max_iter;
    // The first evaluation stands apart: behind the sweep solver it is the whole job (the verification: 99.9 % of the solves
    // leave here), and as the first pass of the loop below it had the loop's invariants — child-slot predicates and addresses
    // for all eight slots, level masks: ~80 instructions — hoisted in front of it.
    NewtonLocal nl = newton_local(ln, maxc, ps, qs, e, f, tol);
    {
        bool wave_miss;
        const bool grp_miss = grp_any<EPW>(nl.miss, ln.grp, wave_miss);
        if (!grp_miss) { ok = true; iters = 0; }
        if (!wave_miss || max_iter <= 0) return ok;
    }
    for (int it = 0;; ++it) {
        if (it > 0) {
            nl = newton_local(ln, maxc, ps, qs, e, f, tol);
            bool wave_miss;
            const bool grp_miss = grp_any<EPW>(nl.miss, ln.grp, wave_miss);
            if (!grp_miss && !ok) { ok = true; iters = it; }
            if (!wave_miss) break;
            if (it >= max_iter) break;
        }
        double rhs0 = nl.rhs0, rhs1 = nl.rhs1, d11 = nl.d11, d12 = nl.d12, d21 = nl.d21, d22 = nl.d22;

        // leaf -> root: D_p -= Yb D_c^-1 Yb ; rhs_p += Yb D_c^-1 rhs_c   (Yb = [[g,-b],[b,g]])
        for (int L = n_levels - 1; L >= 2; --L) {
            const double idet = fast_rcp(d11 * d22 - d12 * d21);
            const double i11 = d22 * idet, i12 = -d12 * idet, i21 = -d21 * idet, i22 = d11 * idet;
            const double t11 = ln.g * i11 - ln.b * i21, t12 = ln.g * i12 - ln.b * i22;
            const double t21 = ln.b * i11 + ln.g * i21, t22 = ln.b * i12 + ln.g * i22;
            const double s11 = t11 * ln.g + t12 * ln.b, s12 = t12 * ln.g - t11 * ln.b;
            const double s21 = t21 * ln.g + t22 * ln.b, s22 = t22 * ln.g - t21 * ln.b;
            const double u0 = t11 * rhs0 + t12 * rhs1, u1 = t21 * rhs0 + t22 * rhs1;
            const int nslots = net->slots_at_level[L];
#pragma unroll
            for (int k = 0; k < FLEX_MAX_CHILDREN; ++k) {
                if (k < nslots) {
                    const bool has = (ln.cha[k] >= 0) && (ln.lev == L - 1);
                    const int src = (ln.cha[k] >= 0) ? ln.cha[k] : (lane << 2);
                    const double a11 = pull_f64(s11, src), a12 = pull_f64(s12, src);
                    const double a21 = pull_f64(s21, src), a22 = pull_f64(s22, src);
                    const double b0 = pull_f64(u0, src), b1 = pull_f64(u1, src);
                    if (has) {
                        d11 -= a11; d12 -= a12; d21 -= a21; d22 -= a22;
                        rhs0 += b0; rhs1 += b1;
                    }
                }
            }
        }
        // root -> leaf: dV_i = D_i^-1 (rhs_i + Yb dV_parent), dV_slack = 0
        const double idet = fast_rcp(d11 * d22 - d12 * d21);
        const double i11 = d22 * idet, i12 = -d12 * idet, i21 = -d21 * idet, i22 = d11 * idet;
        double dx0 = 0.0, dx1 = 0.0;
        for (int L = 1; L < n_levels; ++L) {
            const double q0 = pull_f64(dx0, ln.par_addr), q1 = pull_f64(dx1, ln.par_addr);
            const double p0 = ln.par_slack ? 0.0 : q0, p1 = ln.par_slack ? 0.0 : q1;
            const double w0 = rhs0 + ln.g * p0 - ln.b * p1, w1 = rhs1 + ln.b * p0 + ln.g * p1;
            if (ln.lev == L) {
                dx0 = i11 * w0 + i12 * w1;
                dx1 = i21 * w0 + i22 * w1;
            }
        }
        if (ln.pq) { e += dx0; f += dx1; }
    }
    return ok;
}

// ---- backward/forward sweep (Z-bus Gauss) on the radial feeder ------------------------------------
// The same equations as the Newton path, iterated as a fixed point:  V <- V_slack - Z * conj(S/V).
// On a tree Z is "subtree sum, times the line impedance, path sum".  With lanes in DFS preorder
//   * the subtree sum is one inclusive group scan (DPP) plus one pull at the subtree's last lane;
//   * the path sum is a scan restarted at every chain segment (DPP) plus one pull per level of
//     segment nesting (1 on the 33-bus feeder); trees with deep nesting use pointer jumping over
//     precomputed 2^k-th ancestors instead.
// ~16x fewer instructions and ~10x less dependent latency per iteration than a Newton step, linear
// convergence (~0.1 per sweep on this feeder).
// After a sweep the network equations hold exactly for (V_new, I_old), so the power mismatch at
// V_new is V_new * conj(I_old - I_new): a purely local quantity; "does any lane miss the tolerance"
// needs no reduction.  The caller always hands the result to pf_newton_tree, which re-evaluates the
// true Ybus mismatch and either confirms it (0 Newton steps) or finishes the job — so the
// convergence criterion and the failure semantics are those of the Newton path.
//
// Mixed precision.  Z (subtree sum, line impedance, path sum) is linear, so the iteration can be carried in
// increments against an ANCHOR: one fp64 sweep from the present voltages gives currents I_a and voltages
// V_a = V_slack + Z I_a exactly; with I_b = I(V_a) every later iterate is V_a + d,
//     d_1 = c = Z (I_b - I_a),      d_{k+1} = c + Z delta(d_k),      delta(d) = I(V_a + d) - I_b = -conj(S d / (V V_a)).
// delta is formed from d directly (no cancellation) and shrinks by ~rho every sweep, so the whole inner iteration —
// currents, scans, convergence test — runs in fp32: half the DPP traffic, one instruction per scan step
// (v_add_f32_dpp / packed fma), shorter dependent latencies.  fp32 leaves the iterate ~1e-7 |d| away from the fp64
// fixed point, so once the local mismatch is below FLEX_SWEEP_COARSE the iterate is re-anchored (one more fp64
// sweep); after that |d| ~ 1e-9 and the fp32 increments are exact to fp64 round-off.  Typical step: fp64 sweep,
// ~5 fp32, fp64 sweep, 2-3 fp32.  A re-anchor is also forced every FLEX_SWEEP_REANCHOR increments, so a run that
// cannot reach the coarse threshold (heavy loading, cold start) still ends on fp64-exact footing.
// After a sweep the network equations hold exactly for (V_new, I_old), so the power mismatch at V_new is
// V_new * conj(I_old - I_new) — between anchors I_old - I_new = delta_{k-1} - delta_k: a purely local quantity.
//
// Two-sweep extrapolation (round 5).  The error map of a sweep, e -> A e = -Z conj(S e / V^2), is ANTI-linear, so its real
// spectrum comes in pairs +-lambda and plain Aitken / Anderson(1) on consecutive iterates sees no single dominant mode; its
// square A^2 is complex-linear with real positive eigenvalues mu_1 = lambda_1^2 > mu_2 > ... (IEEE-33 at the bench's
// loading: 2.5e-3, 4e-4, 1e-4: sweeps contract by 0.05, 0.02, 0.01 per mode).  With e_{k+2} = mu e_k on the dominant mode,
//     d* = d_{k+2} + omega (d_{k+2} - d_k),   omega = mu / (1 - mu),
// removes that mode.  Applied to the increments once per anchor at k = 1, where d_1 = c exactly and d_3 = c + x_2, it is a
// scaling of x_2 and of the current increment delta_2 by (1 + omega): the pair (V, I_old) still satisfies the network
// equations exactly, so the local mismatch test stays the true mismatch.  The fixed point is untouched whatever omega is;
// a poor mu costs sweeps, never correctness (and pf_newton_tree has the last word as before).  mu is not measured in the
// solve (an inner product = a group reduction per solve costs what the extrapolation saves): it follows the feeder's
// loading, and kappa |1 - V_end|^2 with the host's calibration tracks it to ~4 % (tools/sweep_sim.py: 9.06 -> 7.96 sweeps
// per solve in the NumPy model of this loop; exact mu: 7.51; a second extrapolation for mu_2: no further gain).
// Returns (per group) the number of sweeps until its local test passed, or max_sweeps.
#define FLEX_SWEEP_COARSE 1e-9
#define FLEX_SWEEP_MU_MAX 0.02f
#ifndef FLEX_SWEEP_REANCHOR
#define FLEX_SWEEP_REANCHOR 8
#endif

// x <- Z x on the lanes of each group (x = per-bus current injections, result = voltage rise slack -> bus)
template <int EPW>
__device__ __forceinline__ void zbus_apply_f64(const DevNet* __restrict__ net, const LaneNet& ln, bool use_seg,
                                               int seg_rounds, int jump_rounds, double& xr, double& xi) {
    double sr = xr, si = xi;
    grp_scan_sum2<EPW>(sr, si);                                   // sum of injections over each subtree
    const double tr = __shfl(sr, ln.sub_end, FLEX_WAVE) - (sr - xr);
    const double ti = __shfl(si, ln.sub_end, FLEX_WAVE) - (si - xi);
    // voltage rise along the own line: -z*J with J = -(subtree injection)  =>  z * t
    double ar = ln.r * tr - ln.x * ti, ai = ln.r * ti + ln.x * tr;
    if (use_seg) {                                                // path sum slack -> bus
        grp_segscan_sum2<EPW>(ar, ai, ln.mk);
        if (seg_rounds >= 1) {                                    // (use_seg: at most two rounds; written out — no loop counter)
            const double br = __shfl(ar, ln.seg_par, FLEX_WAVE), bi = __shfl(ai, ln.seg_par, FLEX_WAVE);
            if (ln.seg_depth == 1) { ar += br; ai += bi; }
        }
        if (seg_rounds >= 2) {
            const double br = __shfl(ar, ln.seg_par, FLEX_WAVE), bi = __shfl(ai, ln.seg_par, FLEX_WAVE);
            if (ln.seg_depth == 2) { ar += br; ai += bi; }
        }
    } else {
        for (int k = 0; k < jump_rounds; ++k) {
            const int anc = net->anc[k][ln.l];
            const int src = anc >= 0 ? anc + ln.base : ln.lane;
            const double br = __shfl(ar, src, FLEX_WAVE), bi = __shfl(ai, src, FLEX_WAVE);
            if (anc >= 0) { ar += br; ai += bi; }
        }
    }
    xr = ar; xi = ai;
}
template <int EPW>
__device__ __forceinline__ void zbus_apply_f32(const LaneNet& ln, const float (&mkf)[6], float rf, float xf,
                                               int seg_rounds, float& xr, float& xi) {
    float sr = xr, si = xi;
    grp_scan_sum2_f32<EPW>(sr, si);
    const float tr = __shfl(sr, ln.sub_end, FLEX_WAVE) - (sr - xr);
    const float ti = __shfl(si, ln.sub_end, FLEX_WAVE) - (si - xi);
    float ar = rf * tr - xf * ti, ai = rf * ti + xf * tr;
    grp_segscan_sum2_f32<EPW>(ar, ai, mkf);
    if (seg_rounds >= 1) {                      // (laterals off the main line — IEEE-33 — need exactly this one round)
        const float br = __shfl(ar, ln.seg_par, FLEX_WAVE), bi = __shfl(ai, ln.seg_par, FLEX_WAVE);
        if (ln.seg_depth == 1) { ar += br; ai += bi; }
    }
    if (seg_rounds >= 2) {
        const float br = __shfl(ar, ln.seg_par, FLEX_WAVE), bi = __shfl(ai, ln.seg_par, FLEX_WAVE);
        if (ln.seg_depth == 2) { ar += br; ai += bi; }
    }
    xr = ar; xi = ai;
}

template <int EPW>
__device__ __forceinline__ int pf_sweep(const DevNet* __restrict__ net, const LaneNet& ln, double pnet,
                                        double qnet, double& e, double& f, double tol, int max_sweeps) {
    const int seg_rounds = net->n_seg_rounds, jump_rounds = net->n_jump_rounds;
    const bool use_seg = seg_rounds <= 2;       // deeper segment nesting: pointer jumping, fp64 sweeps only
    const double ps = ln.pq ? -pnet : 0.0, qs = ln.pq ? -qnet : 0.0;
    const float psf = (float)ps, qsf = (float)qs, rf = (float)ln.r, xf = (float)ln.x;
    const float tolf = (float)tol, coarsef = (float)FLEX_SWEEP_COARSE;
    float mkf[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) mkf[k] = ln.mk[k] ? 1.0f : 0.0f;
    // Statistics (PF_SWEEPS): the sweeps this wavefront EXECUTED — with two environments per wavefront both run until the slower
    // one's test has passed, and that is what either of them cost.  The sweep at which each environment's OWN test first passed
    // (7.75 on the bench's loading where 8.16 are executed) is kept by the diagnostic build only (FLEX_GROUP_SWEEP_STAT, implied
    // by FLEX_STAMPS): as two scalar counters it cost a dozen scalar instructions per sweep, 3.4 % of the launch
    // (profiles/r05ag_env_variants.txt).
    int it = 0;
#ifdef FLEX_GROUP_SWEEP_STAT
    int mine0 = max_sweeps, mine1 = max_sweeps;
#define FLEX_SWEEP_NOTE_PASS(miss, itv) do { \
        if constexpr (EPW == 1) { if ((miss) == 0ull && mine0 == max_sweeps) mine0 = (itv); } \
        else { if ((unsigned)(miss) == 0u && mine0 == max_sweeps) mine0 = (itv); \
               if ((unsigned)((miss) >> 32) == 0u && mine1 == max_sweeps) mine1 = (itv); } } while (0)
#else
#define FLEX_SWEEP_NOTE_PASS(miss, itv) do { } while (0)
#endif
    bool fine = false;                          // re-anchored below the coarse threshold already
    const float kappa = use_seg ? net->acc_kappa : 0.0f;
    const int acc_lane = net->acc_lane;
    float relax = 1.0f;                         // 1 + omega of the two-sweep extrapolation, set after the first anchor
    while (it < max_sweeps) {
        // ---- anchor: one fp64 sweep from (e, f)
        double inv_d = fast_rcp(e * e + f * f);
        const double iar = (ps * e + qs * f) * inv_d, iai = (ps * f - qs * e) * inv_d;      // conj(S/V)
        double ar = iar, ai = iai;
        zbus_apply_f64<EPW>(net, ln, use_seg, seg_rounds, jump_rounds, ar, ai);
        e = 1.0 + ar; f = ai;
        ++it;
        if (it == 1 && kappa > 0.0f) {          // mu_1 from the voltage drop at the feeder's end (see above)
            const float drop = (float)((1.0 - e) * (1.0 - e) + f * f);
            float dsel;
            if constexpr (EPW == 1) {
                dsel = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(drop), acc_lane));
            } else {
                const float d0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(drop), acc_lane));
                const float d1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(drop), acc_lane + 32));
                dsel = ln.grp ? d1 : d0;
            }
            const float mu = fminf(kappa * dsel, FLEX_SWEEP_MU_MAX);
            relax = 1.0f + mu * __builtin_amdgcn_rcpf(1.0f - mu);
        }
        // currents at the anchor, and the anchor's own mismatch
        inv_d = fast_rcp(e * e + f * f);
        const double ibr = (ps * e + qs * f) * inv_d, ibi = (ps * f - qs * e) * inv_d;
        {
            const double dr = iar - ibr, di = iai - ibi;
            const double m = fmax(fabs(e * dr + f * di), fabs(f * dr - e * di));
            const unsigned long long miss = __ballot(!(m < tol));
            const bool wave_miss = miss != 0ull;
            FLEX_SWEEP_NOTE_PASS(miss, it);
            if (!wave_miss) break;
            if (!use_seg) continue;
            if (!fine && __ballot(!(m < FLEX_SWEEP_COARSE)) == 0ull) fine = true;
        }
        // ---- increments against the anchor, fp32
        const double ea = e, fa = f;
        const float ear = (float)ea, eai = (float)fa;
        float cr = (float)(ibr - iar), ci = (float)(ibi - iai);
        zbus_apply_f32<EPW>(ln, mkf, rf, xf, seg_rounds, cr, ci);                           // c = Z (I_b - I_a)
        ++it;
        float dr = cr, di = ci, pdr = 0.0f, pdi = 0.0f;
        bool done = false;
        // ONE test per sweep against the threshold this phase can act on.  A threshold below the coarse one is only trusted on a
        // re-anchored iterate (with the extrapolation an increment can contract by more than coarse / tol in one step and pass
        // the fp32 test while still hanging on its first anchor): before that, passing the coarse threshold ends the phase and
        // re-anchors; afterwards (or when tol itself is not below coarse) passing tol ends the solve.  One counter bounds the
        // phase (re-anchor distance and what is left of max_sweeps).  Round 5: the two tests, two counters and their flags were
        // 28 scalar instructions and six branches per sweep — the launch is issue-bound and they are not free (HISTORY 15.5).
        const bool to_tol = fine || tolf >= coarsef;
        const float thr = to_tol ? tolf : coarsef;
        const int k_end = min((int)FLEX_SWEEP_REANCHOR, max_sweeps - it);
        int k = 0;
        for (; k < k_end; ++k) {
            const float vr = ear + dr, vi = eai + di;                                       // V = V_a + d
            const float pr = vr * ear - vi * eai, pi = vr * eai + vi * ear;                 // P = V V_a
            const float rr = __builtin_amdgcn_rcpf(pr * pr + pi * pi);
            const float tr = psf * dr - qsf * di, ti = psf * di + qsf * dr;                 // S d
            const float ur = tr * pr + ti * pi, ui = ti * pr - tr * pi;                     // S d conj(P)
            float xr = -ur * rr, xi = ui * rr;                                              // delta = -conj(S d / P)
            const float gr = pdr - xr, gi = pdi - xi;
            const float m = fmaxf(fabsf(vr * gr + vi * gi), fabsf(vi * gr - vr * gi));
#ifdef FLEX_GROUP_SWEEP_STAT
            { const unsigned long long miss = __ballot(!(m < tolf)); FLEX_SWEEP_NOTE_PASS(miss, it + k); }
#endif
            if (__ballot(!(m < thr)) == 0ull) { done = to_tol; fine = true; break; }       // (fine: re-anchor now, or moot)
            pdr = xr; pdi = xi;
            zbus_apply_f32<EPW>(ln, mkf, rf, xf, seg_rounds, xr, xi);
            const float fac = k == 1 ? relax : 1.0f;                                        // d_3 + omega (d_3 - d_1); x * 1.0f is exact
            xr *= fac; xi *= fac; pdr *= fac; pdi *= fac;
            dr = cr + xr; di = ci + xi;
        }
        it += k;                                    // (each completed pass of the loop is one sweep)
        e = ea + (double)dr; f = fa + (double)di;
        if (done) break;
    }
#undef FLEX_SWEEP_NOTE_PASS
#ifdef FLEX_GROUP_SWEEP_STAT
    if constexpr (EPW == 1) return mine0;
    else return ln.grp ? mine1 : mine0;
#else
    return it;
#endif
}

// ---- Newton-Raphson with a DENSE LU of the 2n x 2n Jacobian (the north-star's "small batched dense solve") ----
// Kept as the reference variant for the measurement table: O(n^3) work where the tree elimination does O(n).
// One environment per wavefront; row r = 2*bus + component lives in lane r, the matrix sits column-major in
// LDS (32 KB per wavefront: conflict-free column accesses, broadcast pivot-row reads), Gaussian elimination
// without pivoting in bus order (the 2x2 diagonal blocks dominate), rhs carried in registers.
// fp64 MFMA would run at the fp64 vector rate on gfx950 (MI355X_MICROARCH.md), so there is no matrix-core path.
__device__ __forceinline__ bool pf_newton_dense(const DevNet* __restrict__ net, const LaneNet& ln, double pnet,
                                                double qnet, double& e, double& f, double tol, int max_iter,
                                                int& iters, double* __restrict__ A /* LDS [64*64] */) {
    const int maxc = net->max_children, lane = ln.lane, nrow = 2 * net->n_pq;
    const double ps = -pnet, qs = -qnet;
    bool ok = false;
    iters = max_iter;
    for (int it = 0;; ++it) {
        const NewtonLocal nl = newton_local(ln, maxc, ps, qs, e, f, tol);
        if (!__any(nl.miss)) { ok = true; iters = it; break; }
        if (it >= max_iter) break;
        // assemble: zero, then every bus lane writes its diagonal block and the two blocks it shares with its parent
        for (int c = 0; c < 64; ++c) A[c * 64 + lane] = 0.0;
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        if (ln.pq) {
            const int r0 = 2 * ln.l;
            A[(r0) * 64 + r0] = nl.d11; A[(r0 + 1) * 64 + r0] = nl.d12;
            A[(r0) * 64 + r0 + 1] = nl.d21; A[(r0 + 1) * 64 + r0 + 1] = nl.d22;
            if (!ln.par_slack) {
                const int p0 = 2 * (ln.par - ln.base);
                // -Yb = -[[g,-b],[b,g]] couples (bus, parent) both ways
                A[(p0) * 64 + r0] = -ln.g;     A[(p0 + 1) * 64 + r0] = ln.b;
                A[(p0) * 64 + r0 + 1] = -ln.b; A[(p0 + 1) * 64 + r0 + 1] = -ln.g;
                A[(r0) * 64 + p0] = -ln.g;     A[(r0 + 1) * 64 + p0] = ln.b;
                A[(r0) * 64 + p0 + 1] = -ln.b; A[(r0 + 1) * 64 + p0 + 1] = -ln.g;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        // rhs of row `lane`: bus lane>>1, component lane&1
        const double r0v = __shfl(nl.rhs0, (lane >> 1) + ln.base, FLEX_WAVE), r1v = __shfl(nl.rhs1, (lane >> 1) + ln.base, FLEX_WAVE);
        double rhs = (lane & 1) ? r1v : r0v;
        if (lane >= nrow) rhs = 0.0;
        // forward elimination
        for (int k = 0; k < nrow - 1; ++k) {
            const double piv = A[k * 64 + k];
            const double m = (lane > k && lane < nrow) ? A[k * 64 + lane] * fast_rcp(piv) : 0.0;
            const double rk = readlane_f64(rhs, k);
            rhs -= m * rk;
            for (int j = k + 1; j < nrow; ++j) {
                const double akj = A[j * 64 + k];               // pivot row, broadcast
                A[j * 64 + lane] -= m * akj;                    // own row, conflict-free column access
            }
        }
        // back substitution
        double x = 0.0;
        for (int k = nrow - 1; k >= 0; --k) {
            const double xk = readlane_f64(rhs, k) * fast_rcp(A[k * 64 + k]);
            if (lane == k) x = xk;
            if (lane < k) rhs -= A[k * 64 + lane] * xk;
        }
        const double dx0 = __shfl(x, 2 * ln.l + ln.base, FLEX_WAVE), dx1 = __shfl(x, 2 * ln.l + 1 + ln.base, FLEX_WAVE);
        if (ln.pq) { e += dx0; f += dx1; }
    }
    return ok;
}

#define FLEX_MAX_SWEEPS 40
// One power-flow solve per group with the configured solver.  Returns converged?; iters = Newton steps, sweeps = sweeps.
template <int EPW>
__device__ __forceinline__ bool pf_solve(const DevNet* __restrict__ net, const LaneNet& ln, int solver, double pnet,
                                         double qnet, double& e, double& f, double tol, int max_iter, int& iters,
                                         int& sweeps) {
    sweeps = 0;
    if (solver == FLEX_SOLVER_SWEEP) {
        // sweeps stop on their LOCAL mismatch estimate at sweep_tol_frac * tol (0.5; 0.25 until round 4 — measured with 0.5 and
        // 0.8: still no solve of 2 M that the verification did not confirm, a quarter / half a sweep fewer per solve) so that
        // the Ybus re-evaluation below (different rounding) confirms it at tol instead of spending a full Newton step on a
        // borderline case
        // (Thresholds between 2.5e-11 and 2e-8 can be met by the fp32 increments while the iterate still hangs on its FIRST
        //  anchor, up to ~1e-9 from the fp64 fixed point: measured at pf_tol 1e-9 / 1e-8, 56 % / 31 % of the solves then failed
        //  the verification and paid a Newton step, 18.5 us per launch instead of 11.  Such tolerances sweep to the 1e-10
        //  level, which always re-anchors first; looser ones are met on the first anchor for real, tighter ones never were.
        //  Decided here, once per solve: a test inside the increment loop cost the default tolerance 3 %.)
        double sweep_tol = (double)net->sweep_tol_frac * tol;
        if (sweep_tol < 2e-8 && sweep_tol > 2.5e-11) sweep_tol = 2.5e-11;
        sweeps = pf_sweep<EPW>(net, ln, pnet, qnet, e, f, sweep_tol, FLEX_MAX_SWEEPS);
        if (sweeps >= FLEX_MAX_SWEEPS) { e = 1.0; f = 0.0; }   // sweeps stalled: Newton from a flat start
    }
    return pf_newton_tree<EPW>(net, ln, pnet, qnet, e, f, tol, max_iter, iters);
}

// ---- per-building action handling (env:262-293, 621-677) -----------------------------------------
struct FlexAct { double pct, pred, ch, dis, q; };

// _clip_power_charging_discharging, env:628-661 (note env:634 has no dt: SURVEY A4)
// `inv_ch`, `inv_dis` are 1/eta_ch and 1/eta_dis computed once on the host (the reference divides, env:634-655;
// the last-bit difference is far below the parity tolerance and saves four fp64 divisions per lane and step)
__device__ __forceinline__ void clip_charge(const FlexCfg& c, double inv_ch, double inv_dis, double& ch, double& dis,
                                            double e_now) {
    ch = clipd(ch, 0.0, c.p_ch_max);
    dis = clipd(dis, 0.0, c.p_dis_max);
    const double e_next = e_now + c.eta_ch * ch - inv_dis * dis;
    if (e_next > c.e_max) {
        const double excess = e_next - c.e_max;
        if (ch > excess * inv_ch) {
            ch -= excess * inv_ch;
        } else {
            dis += (excess - ch * c.eta_ch) * c.eta_dis;
            ch = 0.0;
        }
    } else if (e_next < c.e_min) {
        const double lack = c.e_min - e_next;
        if (dis > lack * c.eta_dis) {
            dis -= lack * c.eta_dis;
        } else {
            ch += (lack - dis * inv_dis) * inv_ch;
            dis = 0.0;
        }
    }
    ch = clipd(ch, 0.0, c.p_ch_max);
    dis = clipd(dis, 0.0, c.p_dis_max);
}

__device__ __forceinline__ FlexAct parse_actions(const FlexCfg& c, double inv_ch, double inv_dis, bool raw, double a0,
                                                 double a1, double a2, double a3, double pd, double ppv,
                                                 double e_clip) {
    FlexAct o;
    double pr, ch, dis, q;
    if (raw) {                                   // env:268-274
        pr = a0; ch = a1; dis = a2; q = a3;
    } else {                                     // env:276-281
        pr = c.max_power_reduction * a0;
        ch = c.p_ch_max * a1;
        dis = c.p_dis_max * a2;
        const double lim = c.tan_phi * ppv;      // env:621-626
        q = clipd(-lim + a3 * (lim - (-lim)), -lim, lim);
    }
    pr = clipd(pr, 0.0, c.max_power_reduction);  // env:284, 677
    if (ch > 0.0 && dis > 0.0) {                 // env:663-674
        if (ch > dis) { ch -= dis; dis = 0.0; }
        else { dis -= ch; ch = 0.0; }
    }
    clip_charge(c, inv_ch, inv_dis, ch, dis, e_clip);             // env:289-290
    o.pct = pr; o.ch = ch; o.dis = dis; o.q = q;
    o.pred = pd * pr;                            // env:293
    return o;
}

// ---- Philox4x32-10 reset stream (DESIGN.md "reset stream"; restated in oracle/env_oracle.py) ----
#define FLEX_PHILOX_M0 0xD2511F53u
#define FLEX_PHILOX_M1 0xCD9E8D57u
#define FLEX_PHILOX_W0 0x9E3779B9u
#define FLEX_PHILOX_W1 0xBB67AE85u
#define FLEX_RESET_TAG 0x5AFE0001u

__device__ __forceinline__ void philox_pair(uint32_t block, uint32_t episode, uint32_t env, uint64_t seed,
                                            double& u0, double& u1) {
    uint32_t c0 = block, c1 = episode, c2 = env, c3 = FLEX_RESET_TAG;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(FLEX_PHILOX_M0, c0), lo0 = FLEX_PHILOX_M0 * c0;
        const uint32_t hi1 = __umulhi(FLEX_PHILOX_M1, c2), lo1 = FLEX_PHILOX_M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += FLEX_PHILOX_W0; k1 += FLEX_PHILOX_W1;
    }
    const uint64_t a = ((uint64_t)c0 << 32) | c1, b = ((uint64_t)c2 << 32) | c3;
    u0 = (double)(a >> 11) * (1.0 / 9007199254740992.0);
    u1 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

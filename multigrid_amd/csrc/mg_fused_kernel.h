// The fused smoothing pass: kernel template and its launchers.  Included by mg_fused.hip (host entry
// points) and by the mg_fused_inst_*.hip units, which hold the explicit instantiations so that the
// kernels compile in parallel.  The design is described at the top of mg_fused.hip.
#pragma once
#include <type_traits>

#include "mg_common.h"

struct FusedArgs {
    const double *b;     // right-hand side of the level
    const double *xin;   // current x (unused when ZERO)
    double *xout;        // x after K sweeps (must not alias xin)
    double *r;           // residual (RESID)
    const double *crs;   // coarse x to prolong and add before the sweeps (PROLONG)
    i64 ny, nx, ld, cny, cnx, cld;
    i64 j0, j1;          // rows to produce
    i64 j2, j3;          // a second range of rows produced by the same launch (j2 == j3: none), with its coarse rows T2, T3
    i64 T2, T3;
    int nrc1;            // row chunks of the first range; the chunks from nrc1 on belong to the second
    i64 flo, fhi;        // rows <= flo and >= fhi are fixed (Dirichlet): passed through
    int rpb, ncb, nrc, ntc, oj, stagger;
    int lockstep;        // > 0: the waves of a workgroup meet at a barrier every `lockstep` groups of six rows
    double h2, hh;
    double om; // damped Jacobi: x + om * (t - x), t the undamped value (DAMPED instantiations only)
    u64 *norm;
    i64 norm_stride;     // words between the accumulators of consecutive grids of a batch; 0: one accumulator for all
    double *cb;          // coarse right-hand side produced by the fused restriction (RESTRICT)
    i64 cbnx, cbld, T0, T1; // its columns, leading dimension, and the coarse rows to produce
    i64 nbatch, bstride, cstride, cbstride; // number of grids; elements between consecutive grids of a batch: fine arrays, crs, cb
    int stream; // streaming (non-temporal) stores: set when the arrays written dwarf the caches
};

__device__ __forceinline__ double lane_bcast(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

// Whole-wave shifts by one lane whose end lane (0 resp. 63) receives 0: no fallback operand, hence no
// register copy in front of the DPP move.  Every use below either replaces the end lane's value or
// lets it fall into columns the strip does not own.
__device__ __forceinline__ double from_prev0(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next0(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int N>
using int_c = std::integral_constant<int, N>;

// The deep passes run at 2-3 waves per SIMD, so instruction issue and exposed latency count as much
// as memory (a row step of the 6-sweep pass with residual and restriction needs ~110 FP64 operations
// and 32 lane shifts per wave).  The row loop is therefore fully static:
//  * it is unrolled by six phases; windows are never shifted, the slot a new row goes into rotates;
//  * nothing in flight is ever copied: stage 0 keeps a ring of six rows (its window plus the three
//    rows being loaded ahead), b's queue slot is refilled after the row's last use, and the coarse
//    rows of the prolongation advance at compile-time phases.  A register copy of a load's
//    destination would wait for the whole in-order load queue and throw the prefetch away;
//  * a wave whose strip and rows touch no Dirichlet cell and no array edge (almost all of them) runs
//    a PLAIN instantiation without any of the per-cell selects and with advancing row pointers;
//  * row indices are 32-bit, so the uniform row tests are scalar compares.
// PS: the stage whose output row receives the prolongation.  0 = the input row, before the first
// sweep (x += P(c), then K sweeps).  PS > 0 goes with ZERO: PS sweeps from x = 0, += P(c), K - PS
// more sweeps -- a coarse level's pre-smoothing recomputed in front of its post-smoothing, so that
// the descent need not store x and the ascent need not read it.
// ST: x, r and the coarse right-hand side are written with streaming stores (see store_pair); a template
// parameter because the compiler folds a run-time choice between the two kinds of store into one plain store.
template <int K, bool PROLONG, bool ZERO, bool RESID, bool RESTRICT, int U, int PS = 0, bool DAMPED = false, bool ST = false>
__global__ __launch_bounds__(256) void mg_fused_kernel(const FusedArgs a0)
{
    FusedArgs a = a0; // blockIdx.y = grid of the batch
    {
        const i64 g = blockIdx.y;
        a.b += g * a.bstride;
        if (a.xout) a.xout += g * a.bstride;
        if (a.xin) a.xin += g * a.bstride;
        if (a.r) a.r += g * a.bstride;
        if (a.crs) a.crs += g * a.cstride;
        if (a.cb) a.cb += g * a.cbstride;
        if (a.norm) a.norm += g * a.norm_stride;
    }
    static_assert(!RESTRICT || RESID, "the restriction consumes the residual stage");
    constexpr int D = K + (RESID ? 1 : 0);   // pipeline stages that hold a three-row window
    constexpr int DC = D + (RESTRICT ? 1 : 0); // dependency depth: what validity shrinks by
    constexpr int Dv = (DC + 1) & ~1;
    constexpr int W = 128 * U;
    constexpr int STRIDE = W - 2 * Dv;
    constexpr bool BL = D >= 5;                         // b ring in LDS
    constexpr int NG = BL ? (D + 1) / 3 + 1 : 1;        // groups of three row slots: rows i-1 .. i-D plus the rotation slack
    // Scaled sweeps.  The reference's x' = 0.25 * (E + W + S + N - h2*b) costs six FP64 operations.  Keeping
    // stage s's values multiplied by 4^s costs four: with z = 4^(s-1) x and hb = fl(h2*b),
    //     z' = fma(-4^(s-1), hb, E_z + W_z + S_z + N_z) = 4^s x'
    // bit for bit, because a power-of-two factor commutes with every rounding (the product inside the fma
    // is exact) -- unless an intermediate leaves the normal range (|x| above 1e302 or below 1e-307), where
    // the reference rounds a subnormal 0.25*t that the scaled form keeps.  The last stage multiplies by
    // 4^-K.  Damped sweeps need the true values and keep the plain form.
    constexpr bool SC = !DAMPED;
    constexpr bool HB = SC && BL && !RESID;             // the ring holds hb instead of b (the residual wants b itself)
    static_assert(K <= 9, "stages are instantiated up to 9 sweeps");
    static_assert(PS == 0 || (PROLONG && ZERO && PS < K), "a prolongation inside the pipeline goes with a start from zero");
    static_assert(!(PROLONG && ZERO) || PS > 0, "x = 0 + P(c) before the first sweep is not a mode of the cycle");
    extern __shared__ double bring[]; // BL: (waves per workgroup) x NG x 3 row slots of W doubles
    const int wpb = blockDim.x >> 6;
    int cb, rc;
    mg_tile_of_block((int)blockIdx.x, a.ncb, a.nrc, cb, rc, a.stagger);
    const int lane = threadIdx.x & 63;
    const int tile = cb * wpb + (threadIdx.x >> 6);
    const bool active = tile < a.ntc;
    const int origin = (active ? tile : 0) * STRIDE;
    const int nx = (int)a.nx, ny = (int)a.ny, cnx = (int)a.cnx, cny = (int)a.cny;
    const int flo = (int)a.flo, fhi = (int)a.fhi;
    const i64 ld = a.ld;
    const int own_lo = tile == 0 ? 0 : origin + Dv;
    const int own_hi = origin + W - 1 >= nx - 1 ? ((nx - 1) | 1) : origin + W - 1 - Dv;
    const bool isR = lane == 63;

    int c0[U];
    unsigned cl[U];
    bool st[U], ic[U][2];
#pragma unroll
    for (int g = 0; g < U; ++g) {
        c0[g] = origin + g * 128 + 2 * lane;
        cl[g] = (unsigned)(c0[g] + 2 <= ld ? c0[g] : (int)ld - 2);
        st[g] = active && c0[g] >= own_lo && c0[g] + 1 <= own_hi && c0[g] + 2 <= ld;
        ic[g][0] = c0[g] >= 1 && c0[g] <= nx - 2;
        ic[g][1] = c0[g] + 1 >= 1 && c0[g] + 1 <= nx - 2;
    }
    // the launch may cover two row ranges (both ends of a slab in one launch): uniform per workgroup
    const bool second = rc >= a.nrc1;
    const int jhi = (int)(second ? a.j3 : a.j1);
    const int Tlo = (int)(second ? a.T2 : a.T0), Thi = (int)(second ? a.T3 : a.T1);
    const int rb = (int)(second ? a.j2 : a.j0) + (second ? rc - a.nrc1 : rc) * a.rpb;
    const int re = rb + a.rpb < jhi ? rb + a.rpb : jhi;
    (void)Tlo, (void)Thi;
    u64 nmax = 0;

    if (active && rb < re) { // uniform per wave
        // w[s]: stage-s values of consecutive rows, slots rotating (see step).  Stages >= 1 keep three
        // rows (slots 0..2).  Stage 0 keeps six: the three-row window and the three rows being loaded
        // ahead -- a loaded row is used where it landed, so no value in flight is ever copied, which
        // would make the copy wait for the whole load queue.
        double w[D][6][U][2];
        // b of the rows in flight: row i-s for stage s.  Shallow pipelines shift it through registers
        // (br[d]: b of row i-1-d).  Deep ones keep it in LDS instead, 28 VGPRs and 14 moves per step
        // less: a ring of NG groups of three row slots per wave; a step with rotation R writes row i-1
        // into slot R of the youngest group, so every slot a stage reads is a compile-time offset from
        // one of NG base pointers, which change places once per three steps.
        double br[BL ? 1 : D][U][2];
        double *G[NG];
        if constexpr (BL) {
#pragma unroll
            for (int k = 0; k < NG; ++k) G[k] = bring + ((threadIdx.x >> 6) * NG + k) * 3 * W + 2 * lane;
        }
#pragma unroll
        for (int s = 0; s < D; ++s)
#pragma unroll
            for (int q = 0; q < (s == 0 ? 6 : 3); ++q)
#pragma unroll
                for (int g = 0; g < U; ++g) w[s][q][g][0] = w[s][q][g][1] = 0.0;
#pragma unroll
        for (int d = 0; d < (BL ? 1 : D); ++d)
#pragma unroll
            for (int g = 0; g < U; ++g) br[d][g][0] = br[d][g][1] = 0.0;

        auto rowclamp = [&](int j) { return j < 0 ? 0 : (j > ny - 1 ? ny - 1 : j); };
        auto load_pairs = [&](const double *base, int j, double (&d)[U][2]) {
            const double *row = base + (i64)rowclamp(j) * ld;
#pragma unroll
            for (int g = 0; g < U; ++g) {
                double2 t = *reinterpret_cast<const double2 *>(row + cl[g]);
                d[g][0] = t.x, d[g][1] = t.y;
            }
        };
        // Coarse row J as loaded: the value at coarse column c0/2 and, in lane 63 only, the value one
        // coarse column beyond the strip.  Nothing is computed from them here, so the loads stay in
        // flight; coarse_next() makes the second operand when the row comes into use.
        auto load_coarse = [&](int J, double (&v0)[U], double &ve) {
            const int Jc = J < 0 ? 0 : (J > cny - 1 ? cny - 1 : J);
            const double *row = a.crs + (i64)Jc * a.cld;
#pragma unroll
            for (int g = 0; g < U; ++g) {
                const int q = c0[g] >> 1;
                v0[g] = row[(unsigned)(q > cnx - 1 ? cnx - 1 : q)];
            }
            const int q1 = (c0[U - 1] >> 1) + 1;
            ve = isR ? row[(unsigned)(q1 > cnx - 1 ? cnx - 1 : q1)] : 0.0;
        };
        // value at the next coarse column: the neighbouring lane's, across the seam the next group's or ve
        auto coarse_next = [&](const double (&v0)[U], const double ve, double (&v1)[U]) {
#pragma unroll
            for (int g = 0; g < U; ++g) {
                double nb = from_next0(v0[g]);
                if (g + 1 < U) {
                    const double e = lane_bcast(v0[g + 1 < U ? g + 1 : 0], 0);
                    nb = isR ? e : nb;
                } else {
                    nb = isR ? ve : nb;
                }
                v1[g] = nb;
            }
        };

        const int i0 = rb - DC, i1 = re + DC; // input rows streamed: i0 .. i1-1 (clamped into the array)
        double rw[RESTRICT ? 3 : 1][U][2];    // residual rows, slots rotating like w
#pragma unroll
        for (int q = 0; q < (RESTRICT ? 3 : 1); ++q)
#pragma unroll
            for (int g = 0; g < U; ++g) rw[q][g][0] = rw[q][g][1] = 0.0;
        // Loads run three row steps ahead of their use: with few waves per SIMD that is what keeps
        // enough bytes in flight.  bq[q]: b of row i-1+q at a step whose phase is a multiple of 3.
        double bq[3][U][2];
        // coarse rows for the prolongation: a ring of three rows that advances every other step, at
        // compile-time phases (the loop starts on a row that is even relative to the coarse grid);
        // C0 = value at coarse column c0/2, C1 = at the next column, Ce = the value beyond the strip
        // (lane 63 only)
        double C0[3][U], C1[3][U], Ce[3] = {0, 0, 0};
        const int oj = a.oj;
        const double h2 = a.h2, hh = a.hh;
        const double om = a.om;
        (void)om;

        // PL: no Dirichlet cell, no clamped row or column in anything this wave streams.
        auto run = [&](auto plain_c) {
            constexpr bool PL = decltype(plain_c)::value;
            // first row streamed; the coarse schedule wants it even relative to the coarse grid (one
            // more row of lead-in, whose values nothing valid depends on)
            const int ia = PROLONG ? i0 - ((i0 - PS - oj) & 1) : i0; // row ia - PS is the even one
            if constexpr (PROLONG) { // before the row loads: the loop's first wait is for these
                const int J = (ia - PS - oj) >> 1;
                load_coarse(J, C0[0], Ce[0]);
                load_coarse(J + 1, C0[1], Ce[1]);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if constexpr (!ZERO) load_pairs(a.xin, ia + q, w[0][q]);
                load_pairs(a.b, ia - 1 + q, bq[q]);
            }
            if constexpr (PROLONG) coarse_next(C0[0], Ce[0], C1[0]);
            // plain path: every row streamed lies inside the array, so the row pointers simply advance
            const double *px = ZERO ? nullptr : a.xin + (i64)(ia + 3) * ld; // next x row to load
            const double *pb = a.b + (i64)(ia + 2) * ld;                    // next b row to load
            auto load_next = [&](const double *&ptr, int j, const double *base, double (&d)[U][2]) {
                if constexpr (PL) {
#pragma unroll
                    for (int g = 0; g < U; ++g) {
                        double2 t = *reinterpret_cast<const double2 *>(ptr + cl[g]);
                        d[g][0] = t.x, d[g][1] = t.y;
                    }
                    ptr += ld;
                } else {
                    load_pairs(base, j, d);
                }
            };
            // One row step.  R = rotation: this step's new rows go into slot R of every window; after
            // that the window's rows, oldest first, are slots A0, A1, A2.
            auto step = [&](auto ph_c, const int i) {
                constexpr int PH = decltype(ph_c)::value; // phase 0..5 of the unrolled loop
                constexpr int R = PH % 3;
                constexpr int A0 = (R + 1) % 3, A1 = (R + 2) % 3, A2 = R;
                // stage 0: row i sits in slot PH, rows i-1 and i-2 in the two slots before it, and the
                // slot three ahead (row i-3, dead) takes the load of row i+3
                constexpr int Z2 = PH, Z1 = (PH + 5) % 6, Z0 = (PH + 4) % 6, ZL = (PH + 3) % 6;
                double bcur[U][2];
#pragma unroll
                for (int g = 0; g < U; ++g) {
                    if constexpr (ZERO) w[0][Z2][g][0] = w[0][Z2][g][1] = 0.0;
                    bcur[g][0] = bq[R][g][0], bcur[g][1] = bq[R][g][1];
                }
                if constexpr (!ZERO) load_next(px, i + 3, a.xin, w[0][ZL]);
                // Prolongation into row i - PS.  The loop starts on a step where that row is even
                // relative to the coarse grid, so it is even at even phases; rows outside the array follow
                // the same schedule, their coarse rows are clamped loads that nothing uses.
                constexpr int PC = PH;
                constexpr int CM = PC / 2;
                constexpr int SA = CM % 3, SB = (CM + 1) % 3, SN = (CM + 2) % 3;
                if constexpr (PROLONG && (PC & 1) == 0) {
                    const int J = (i - PS - oj) >> 1;
                    coarse_next(C0[SB], Ce[SB], C1[SB]);
                    load_coarse(J + 2, C0[SN], Ce[SN]); // becomes row J+1 two steps from now
                }
                auto prolong_into = [&](double (&v)[U][2]) {
                    const int rp = i - PS;
                    if (PL || (rp > flo && rp < fhi && rp >= 0 && rp <= ny - 1)) {
#pragma unroll
                        for (int g = 0; g < U; ++g) {
                            double add0, add1;
                            if constexpr ((PC & 1) == 1) {
                                add0 = (C0[SA][g] + C0[SB][g]) / 2;
                                add1 = (C1[SB][g] + C0[SB][g] + C1[SA][g] + C0[SA][g]) / 4;
                            } else {
                                add0 = C0[SA][g];
                                add1 = (C0[SA][g] + C1[SA][g]) / 2;
                            }
                            if constexpr (SC && PS > 0) { // the stage-PS values are held times 4^PS
                                constexpr double CP = (double)(1LL << (2 * PS));
                                add0 *= CP, add1 *= CP;
                            }
                            if (PL || ic[g][0]) v[g][0] += add0;
                            if (PL || ic[g][1]) v[g][1] += add1;
                        }
                    }
                };
                if constexpr (PROLONG && PS == 0) prolong_into(w[0][Z2]);
                // ---- push b (row i-1)
                if constexpr (BL) {
#pragma unroll
                    for (int g = 0; g < U; ++g) {
                        if constexpr (HB) bcur[g][0] = h2 * bcur[g][0], bcur[g][1] = h2 * bcur[g][1];
                        *reinterpret_cast<double2 *>(G[0] + R * W + g * 128) = make_double2(bcur[g][0], bcur[g][1]);
                    }
                } else {
#pragma unroll
                    for (int g = 0; g < U; ++g)
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
#pragma unroll
                            for (int d = D - 1; d > 0; --d) br[d][g][v] = br[d - 1][g][v];
                            br[0][g][v] = bcur[g][v];
                        }
                }
                // b of row i-s, for stage s
                auto b_of = [&](auto s_c, double (&bs)[U][2]) {
                    constexpr int s = decltype(s_c)::value;
                    if constexpr (s == 1 || !BL) {
#pragma unroll
                        for (int g = 0; g < U; ++g)
                            bs[g][0] = s == 1 ? bcur[g][0] : br[BL ? 0 : s - 1][g][0], bs[g][1] = s == 1 ? bcur[g][1] : br[BL ? 0 : s - 1][g][1];
                    } else {
                        constexpr int dl = R - s + 1;                     // slot relative to the youngest group's first
                        constexpr int grp = dl >= 0 ? 0 : (-dl + 2) / 3;
                        constexpr int sub = dl + 3 * grp;
                        static_assert(grp < NG && sub >= 0 && sub < 3, "b ring too short");
#pragma unroll
                        for (int g = 0; g < U; ++g) {
                            const double2 t = *reinterpret_cast<const double2 *>(G[grp] + sub * W + g * 128);
                            bs[g][0] = t.x, bs[g][1] = t.y;
                        }
                    }
                };
                // ---- stages 1..K: stage s produces row i-s from the stage s-1 window
                // b for stage s sits in bsq[s & 1]: each stage first issues the LDS read for the next one,
                // so that the read's latency passes under this stage's arithmetic
                double bsq[2][U][2];
                b_of(int_c<1>{}, bsq[1]);
                auto stage = [&](auto s_c) {
                    constexpr int s = decltype(s_c)::value;
                    const int row = i - s;
                    const bool rint = PL || (row > flo && row < fhi);
                    double o[U][2];
                    if constexpr (s + 1 <= D) {
                        b_of(int_c<s + 1>{}, bsq[(s + 1) & 1]);
                        if constexpr (BL) __builtin_amdgcn_sched_barrier(0);
                    }
                    double(&bs)[U][2] = bsq[s & 1];
                    // the window this stage reads, oldest row first
                    constexpr int Q0 = s == 1 ? Z0 : A0, Q1 = s == 1 ? Z1 : A1, Q2 = s == 1 ? Z2 : A2;
#pragma unroll
                    for (int g = 0; g < U; ++g) {
                        double west = from_prev0(w[s - 1][Q1][g][1]);
                        double east = from_next0(w[s - 1][Q1][g][0]);
                        if (g > 0) {
                            const double t = lane_bcast(w[s - 1][Q1][g > 0 ? g - 1 : 0][1], 63);
                            west = lane == 0 ? t : west;
                        }
                        if (g + 1 < U) {
                            const double t = lane_bcast(w[s - 1][Q1][g + 1 < U ? g + 1 : 0][0], 0);
                            east = isR ? t : east;
                        }
                        const double c0v = w[s - 1][Q1][g][0], c1v = w[s - 1][Q1][g][1];
                        if constexpr (SC) {
                            constexpr double CS = (double)(1LL << (2 * (s - 1))); // scale of the window this stage reads
                            const double hb0 = HB ? bs[g][0] : h2 * bs[g][0], hb1 = HB ? bs[g][1] : h2 * bs[g][1];
                            const double j0v = __builtin_fma(-CS, hb0, c1v + west + w[s - 1][Q2][g][0] + w[s - 1][Q0][g][0]);
                            const double j1v = __builtin_fma(-CS, hb1, east + c0v + w[s - 1][Q2][g][1] + w[s - 1][Q0][g][1]);
                            o[g][0] = (PL || (rint && ic[g][0])) ? j0v : 4.0 * c0v;
                            o[g][1] = (PL || (rint && ic[g][1])) ? j1v : 4.0 * c1v;
                            if constexpr (s == K) {
                                constexpr double INV = 1.0 / (double)(1LL << (2 * K));
                                o[g][0] *= INV, o[g][1] *= INV;
                            }
                        } else {
                            double j0v = 0.25 * (c1v + west + w[s - 1][Q2][g][0] + w[s - 1][Q0][g][0] - h2 * bs[g][0]);
                            double j1v = 0.25 * (east + c0v + w[s - 1][Q2][g][1] + w[s - 1][Q0][g][1] - h2 * bs[g][1]);
                            j0v = c0v + om * (j0v - c0v), j1v = c1v + om * (j1v - c1v);
                            o[g][0] = (PL || (rint && ic[g][0])) ? j0v : c0v;
                            o[g][1] = (PL || (rint && ic[g][1])) ? j1v : c1v;
                        }
                    }
                    if constexpr (PROLONG && PS > 0 && s == PS) prolong_into(o);
                    if (s == K && a.xout && row >= rb && row < re) {
                        double *orow = a.xout + (i64)row * ld;
#pragma unroll
                        for (int g = 0; g < U; ++g)
                            if (st[g]) store_pair<ST>(orow + (unsigned)c0[g], o[g][0], o[g][1]);
                    }
                    if constexpr (s < D) {
#pragma unroll
                        for (int g = 0; g < U; ++g) w[s][A2][g][0] = o[g][0], w[s][A2][g][1] = o[g][1];
                    }
                };
                stage(int_c<1>{});
                // b of row i-1 has had its last use from the register: its slot takes the load of row i+2
                // (issued here, not earlier, so that the old and the new value never coexist)
                __builtin_amdgcn_sched_barrier(0);
                load_next(pb, i + 2, a.b, bq[R]);
                if constexpr (K >= 2) stage(int_c<2>{});
                if constexpr (K >= 3) stage(int_c<3>{});
                if constexpr (K >= 4) stage(int_c<4>{});
                if constexpr (K >= 5) stage(int_c<5>{});
                if constexpr (K >= 6) stage(int_c<6>{});
                if constexpr (K >= 7) stage(int_c<7>{});
                if constexpr (K >= 8) stage(int_c<8>{});
                if constexpr (K >= 9) stage(int_c<9>{});
                // ---- residual of row i-K-1 from the stage-K window
                if constexpr (RESID) {
                    const int row = i - K - 1;
                    const bool mine = row >= rb && row < re;
                    const bool need = RESTRICT ? (row >= rb - 1 && row <= re) : mine; // uniform
                    if (need) {
                        const bool rint = PL || (row > flo && row < fhi); // uniform
                        double rv[U][2];
                        double(&bs)[U][2] = bsq[(K + 1) & 1];
                        if (rint) {
#pragma unroll
                            for (int g = 0; g < U; ++g) {
                                double west = from_prev0(w[K][A1][g][1]);
                                double east = from_next0(w[K][A1][g][0]);
                                if (g > 0) {
                                    const double t = lane_bcast(w[K][A1][g > 0 ? g - 1 : 0][1], 63);
                                    west = lane == 0 ? t : west;
                                }
                                if (g + 1 < U) {
                                    const double t = lane_bcast(w[K][A1][g + 1 < U ? g + 1 : 0][0], 0);
                                    east = isR ? t : east;
                                }
                                const double c0v = w[K][A1][g][0], c1v = w[K][A1][g][1];
                                rv[g][0] = bs[g][0] - west * hh - (c1v + w[K][A0][g][0] + w[K][A2][g][0]) * hh + (c0v * hh) * 4;
                                rv[g][1] = bs[g][1] - c0v * hh - (east + w[K][A0][g][1] + w[K][A2][g][1]) * hh + (c1v * hh) * 4;
                                // cells outside the interior keep what r holds (the reference never writes
                                // them); without an r array they are the zeros such an array would hold
                                if constexpr (!PL) {
                                    if (!(ic[g][0] && ic[g][1]) && (st[g] || RESTRICT)) {
                                        double2 keep = make_double2(0.0, 0.0);
                                        if (a.r) keep = *reinterpret_cast<const double2 *>(a.r + (i64)row * ld + cl[g]);
                                        if (!ic[g][0]) rv[g][0] = keep.x;
                                        if (!ic[g][1]) rv[g][1] = keep.y;
                                    }
                                }
                            }
                            if (mine) {
                                double *rrow = a.r + (i64)row * ld;
#pragma unroll
                                for (int g = 0; g < U; ++g) {
                                    if (a.r && st[g]) store_pair<ST>(rrow + (unsigned)c0[g], rv[g][0], rv[g][1]);
                                    // with one column pair per lane, lanes outside the owned columns are
                                    // dropped once, after the loop
                                    if ((U == 1 || st[g]) && (PL || ic[g][0])) {
                                        u64 key = (u64)__double_as_longlong(fabs(rv[g][0]));
                                        nmax = key > nmax ? key : nmax;
                                    }
                                    if ((U == 1 || st[g]) && (PL || ic[g][1])) {
                                        u64 key = (u64)__double_as_longlong(fabs(rv[g][1]));
                                        nmax = key > nmax ? key : nmax;
                                    }
                                }
                            }
                        } else if constexpr (RESTRICT) { // a Dirichlet row: r as stored
#pragma unroll
                            for (int g = 0; g < U; ++g) rv[g][0] = rv[g][1] = 0.0;
                            if (a.r && row >= 0 && row <= ny - 1) load_pairs(a.r, row, rv);
                        }
                        if constexpr (RESTRICT) {
#pragma unroll
                            for (int g = 0; g < U; ++g) rw[A2][g][0] = rv[g][0], rw[A2][g][1] = rv[g][1];
                            // full weighting of coarse row T, whose fine centre row is row-1 (owned by this
                            // tile); the rows it reads, row-2 .. row, were all produced in `need` steps
                            const int ctr = row - 1;
                            if (ctr >= rb && ctr < re && ((ctr - oj) & 1) == 0) {
                                const int T = (ctr - oj) >> 1;
                                if (T >= Tlo && T < Thi) {
#pragma unroll
                                    for (int g = 0; g < U; ++g) {
                                        const double a_right = rw[A1][g][1] / 8 + (rw[A2][g][1] + rw[A0][g][1]) / 16;
                                        double a_left = from_prev0(a_right);
                                        if (g > 0) {
                                            const double pr = rw[A1][g > 0 ? g - 1 : 0][1] / 8 +
                                                              (rw[A2][g > 0 ? g - 1 : 0][1] + rw[A0][g > 0 ? g - 1 : 0][1]) / 16;
                                            const double t = lane_bcast(pr, 63);
                                            a_left = lane == 0 ? t : a_left;
                                        }
                                        double v = rw[A1][g][0] / 4 + (rw[A2][g][0] + rw[A0][g][0]) * 1 / 8;
                                        v += a_right + a_left;
                                        const int I = c0[g] >> 1;
                                        if (st[g] && I >= 1 && I <= (int)a.cbnx - 2) store_one<ST>(&a.cb[(i64)T * a.cbld + I], v);
                                    }
                                }
                            }
                        }
                    }
                }
            };
            int i = ia;
            const int lockstep = a.lockstep;
            int since = 0;
            auto next_group = [&]() { // the oldest group of the b ring becomes the youngest
                if constexpr (BL) {
                    double *const t = G[NG - 1];
#pragma unroll
                    for (int k = NG - 1; k > 0; --k) G[k] = G[k - 1];
                    G[0] = t;
                }
            };
            for (; i + 6 <= i1; i += 6) {
                step(int_c<0>{}, i);
                step(int_c<1>{}, i + 1);
                step(int_c<2>{}, i + 2);
                next_group();
                step(int_c<3>{}, i + 3);
                step(int_c<4>{}, i + 4);
                step(int_c<5>{}, i + 5);
                next_group();
                // lockstep: the waves of the workgroup (neighbouring strips of the same rows) meet every `lockstep`
                // groups of six rows, so that they keep touching the same DRAM pages at about the same time
                if (lockstep && ++since == lockstep) {
                    since = 0;
                    __builtin_amdgcn_s_barrier();
                }
            }
            if (i < i1) step(int_c<0>{}, i);
            if (i + 1 < i1) step(int_c<1>{}, i + 1);
            if (i + 2 < i1) step(int_c<2>{}, i + 2);
            if (i + 3 < i1) {
                next_group();
                step(int_c<3>{}, i + 3);
                if (i + 4 < i1) step(int_c<4>{}, i + 4);
            }
        };
        // (the loop may start one row early and loads run three rows ahead)
        const bool plain = origin >= 1 && origin + W - 1 <= nx - 2 && origin + W <= ld && i0 - 2 > flo && i1 < fhi &&
                           i0 - 2 >= 0 && i1 + 2 <= ny - 1;
        if (plain)
            run(std::true_type{});
        else
            run(std::false_type{});
        if (U == 1 && !st[0]) nmax = 0; // only the owned columns count towards the norm
    } else if (a.lockstep && rb < re) {
        // a wave without a strip of its own (the last workgroup of a row of tiles) keeps its siblings' appointments
        const int i0 = rb - DC, i1 = re + DC;
        const int ia = PROLONG ? i0 - ((i0 - PS - a.oj) & 1) : i0;
        int since = 0;
        for (int i = ia; i + 6 <= i1; i += 6)
            if (++since == a.lockstep) {
                since = 0;
                __builtin_amdgcn_s_barrier();
            }
    }
    if constexpr (RESID) {
        if (a.norm) { // uniform
            __shared__ u64 wmax[4];
            nmax = wave_max_u64(nmax);
            if (lane == 0) wmax[threadIdx.x >> 6] = nmax;
            __syncthreads();
            if (threadIdx.x == 0) {
                u64 m = wmax[0];
                for (int k = 1; k < wpb; ++k) m = wmax[k] > m ? wmax[k] : m;
                if (m) atomicMax(&a.norm[blockIdx.x % MG_NORM_SLOTS], m);
            }
        }
    }
}

// ---- tuning knobs (defined in mg_fused.hip)
extern int g_fused_waves;  // waves (column strips) per workgroup: tuning knob "fused_waves"
extern int g_fused_unroll; // column pairs per lane of the shallow passes
extern int g_fused_rows;   // rows per tile, 0 = automatic
extern int g_fused_min_rows; // shortest tile the automatic choice goes down to on small levels
extern int g_fused_min_blocks; // ... as long as the launch has fewer workgroups than this
extern int g_fused_rows_shallow, g_fused_rows_deep; // default tile heights of the passes of depth <= 5 / >= 9
extern int g_fused_lockstep; // barrier among the waves of a workgroup every so many groups of six rows (0 = never)
extern int g_xcd_stagger;    // column blocks by which consecutive XCD bands are shifted; -1 = an eighth of the width

template <int K, bool PROLONG, bool ZERO, bool RESID, bool RESTRICT, int U, int PS = 0, bool DAMPED = false>
int launch_fused(FusedArgs a, hipStream_t st)
{
    constexpr int D = K + (RESID ? 1 : 0) + (RESTRICT ? 1 : 0), Dv = (D + 1) & ~1, W = 128 * U, STRIDE = W - 2 * Dv;
    constexpr int DK = K + (RESID ? 1 : 0); // window stages of the kernel: decides whether b lives in an LDS ring
    a.ntc = a.nx <= W ? 1 : (int)((a.nx - W + STRIDE - 1) / STRIDE) + 1;
    // column strips per workgroup: a narrow grid whose strip count does not fill the workgroups would leave
    // wave slots empty (three strips of a 257-column grid in a 4-wave workgroup: a quarter of the slots)
    int wpb = g_fused_waves;
    while (wpb > 1 && a.ntc % wpb != 0 && a.ntc < 4 * wpb) wpb >>= 1;
    a.ncb = (a.ntc + wpb - 1) / wpb;
    int rpb = g_fused_rows;
    if (rpb <= 0) {
        // measured at 32769 x 32769 (profiles/): 64-row tiles are as fast as or faster than taller ones at
        // every depth, 32-row tiles faster still for the shallow from-zero pass
        rpb = D <= 5 ? g_fused_rows_shallow : (D >= 9 ? g_fused_rows_deep : 64); // the 9-sweep pass streams 9 extra rows per tile end: taller tiles
        // a small level cannot fill the chip, and its pass lasts as long as one wave needs for one tile:
        // shorter tiles (down to g_fused_min_rows) cut that time; the extra overlap rows cost nothing there
        while (rpb > g_fused_min_rows &&
               (i64)a.ncb * ((a.j1 - a.j0 + rpb - 1) / rpb + (a.j3 - a.j2 + rpb - 1) / rpb) * a.nbatch < g_fused_min_blocks)
            rpb >>= 1;
    }
    a.rpb = rpb;
    a.stagger = g_xcd_stagger < 0 ? a.ncb / 8 : g_xcd_stagger;
    a.lockstep = g_fused_lockstep;
    a.nrc1 = (int)((a.j1 - a.j0 + rpb - 1) / rpb);
    a.nrc = a.nrc1 + (int)((a.j3 - a.j2 + rpb - 1) / rpb);
    const dim3 grid((unsigned)(a.ncb * a.nrc), (unsigned)a.nbatch), block(64 * wpb);
    const size_t lds = DK >= 5 ? (size_t)wpb * ((DK + 1) / 3 + 1) * 3 * W * sizeof(double) : 0;
    bool launched = false;
    if constexpr (U == 1) { // arrays that large are never run with two column pairs per lane
        if (a.stream) {
            hipLaunchKernelGGL((mg_fused_kernel<K, PROLONG, ZERO, RESID, RESTRICT, U, PS, DAMPED, true>), grid, block, lds, st, a);
            launched = true;
        }
    }
    if (!launched)
        hipLaunchKernelGGL((mg_fused_kernel<K, PROLONG, ZERO, RESID, RESTRICT, U, PS, DAMPED, false>), grid, block, lds, st, a);
    MG_LAUNCH_CHECK("mg_smooth_fused");
    return 0;
}

template <int K, bool PROLONG, bool ZERO, bool RESID, bool RESTRICT, bool DAMPED>
int pick_u(const FusedArgs &a, hipStream_t st)
{
    // deep pipelines keep D three-row windows per column pair in registers: one pair per lane then
    // two column pairs per lane halve the share of overlap columns but double the registers: worth it
    // only for the shallowest pipelines
    if constexpr (!PROLONG && (K + (RESID ? 1 : 0) + (RESTRICT ? 1 : 0) + (ZERO ? 0 : 1)) <= 3) {
        int u = g_fused_unroll;
        while (u > 1 && a.nx < 128 * u) u >>= 1;
        if (u == 2) return launch_fused<K, PROLONG, ZERO, RESID, RESTRICT, 2, 0, DAMPED>(a, st);
    }
    return launch_fused<K, PROLONG, ZERO, RESID, RESTRICT, 1, 0, DAMPED>(a, st);
}

template <int K, bool DAMPED = false>
int pick_mode(const FusedArgs &a, bool prolong, bool zero, bool resid, bool restr, hipStream_t st)
{
    if (restr) {
        if (zero) return pick_u<K, false, true, true, true, DAMPED>(a, st);
        if (prolong) return pick_u<K, true, false, true, true, DAMPED>(a, st);
        return pick_u<K, false, false, true, true, DAMPED>(a, st);
    }
    if (zero) return resid ? pick_u<K, false, true, true, false, DAMPED>(a, st) : pick_u<K, false, true, false, false, DAMPED>(a, st);
    if (prolong) return resid ? pick_u<K, true, false, true, false, DAMPED>(a, st) : pick_u<K, true, false, false, false, DAMPED>(a, st);
    return resid ? pick_u<K, false, false, true, false, DAMPED>(a, st) : pick_u<K, false, false, false, false, DAMPED>(a, st);
}


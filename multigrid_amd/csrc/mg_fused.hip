// Fused smoothing passes of the V-cycle on gfx950 (temporal blocking).
//
// One launch performs, on a whole level and in ONE pass over memory,
//     [x += prolongation(coarse)] -> K Jacobi sweeps -> [residual (+ max-norm)] -> [restriction]
// so that x and b are read once and x (and r, and the coarse right-hand side) written once for K
// sweeps, instead of once per operation: the finest-level pass of the V-cycle (prolongation, 6
// sweeps, residual, norm, restriction) moves 28 B per point where the separate kernels move 196.
// Results are bit-identical to the separate kernels: every point value is computed by the same
// expression from the same operands, only earlier.
//
// How: a wave owns a strip of W = 128*U columns and streams down the rows.  It keeps, per stage
// s = 0..D-1 (D = K, +1 with the residual), a three-row register window of stage-s values; when
// input row i arrives, stage 1 can produce row i-1, stage 2 row i-2, ... stage K row i-K, the
// residual row i-K-1, the restriction the coarse row centred on row i-K-2.  Sideways, the values a
// stage needs from the neighbouring lane come by DPP (v_readlane across the seam between a lane's
// column groups), and what lies beyond the strip is simply not available: validity shrinks by one
// column per stage from both strip edges, so strips overlap by 2*Dv columns (Dv = depth rounded up
// to even, keeping every owned range aligned to the 16-byte column pairs) and a strip stores only
// the columns it owns.  Dirichlet boundary cells are passed through every stage unchanged, so
// validity does not shrink from a fixed boundary.  The same shrinking applies to rows: a tile
// producing rows [rb, re) streams rows rb-depth .. re+depth-1.
// Redundant work: 2*Dv/W of the columns and 2*depth/rpb of the rows.
//
// Arithmetic follows the reference's expression order (see include/mg_hip.h); compiled with
// -ffp-contract=off.
#include "mg_fused_kernel.h"

// ------------------------------------------------------------------------------------------ host

int g_fused_waves = 4; // waves (column strips) per workgroup: tuning knob "fused_waves"
int g_fused_unroll = 2;
int g_fused_rows = 0;
int g_fused_min_rows = 8;
int g_fused_min_blocks = 2048;
int g_xcd_stagger = 0;
int g_fused_rows_shallow = 32, g_fused_rows_deep = 128;
int g_fused_lockstep = 1; // measured at 32769 x 32769: finest pass 5.57 -> 5.50 ms, 9-stage pass 4.23 -> 4.12 ms

extern "C" int mg_set_fused_tuning(int unroll, int rows)
{
    MG_REQUIRE(unroll == 1 || unroll == 2, "mg_set_fused_tuning: unroll must be 1 or 2");
    MG_REQUIRE(rows >= 0 && rows <= 8192, "mg_set_fused_tuning: rows must be 0..8192");
    g_fused_unroll = unroll, g_fused_rows = rows;
    return 0;
}

// A second row range for the next fused pass of the calling thread (mg_fused_also_rows): consumed by that call,
// whether it succeeds or not.  It is tied to the launch it was set for by that launch's row count and stream: a
// setting that was never consumed (its call was not made) is refused by the next fused call instead of adding rows
// to an unrelated launch.
static thread_local i64 t_also[4];
static thread_local i64 t_also_ny = 0;
static thread_local void *t_also_stream = nullptr;
static thread_local bool t_also_set = false;

extern "C" int mg_fused_also_rows(i64 j2, i64 j3, i64 T2, i64 T3, i64 ny, void *stream)
{
    t_also_set = false;
    MG_REQUIRE(j2 >= 0 && j2 <= j3 && T2 >= 0 && T2 <= T3, "mg_fused_also_rows: bad ranges [%lld, %lld), [%lld, %lld)", (long long)j2,
               (long long)j3, (long long)T2, (long long)T3);
    t_also[0] = j2, t_also[1] = j3, t_also[2] = T2, t_also[3] = T3;
    t_also_ny = ny, t_also_stream = stream;
    t_also_set = true;
    return 0;
}

// the instantiations live in mg_fused_inst_*.hip
extern template int pick_mode<1, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<2, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<3, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<4, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<5, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<6, false>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<1, true>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<3, true>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int pick_mode<6, true>(const FusedArgs &, bool, bool, bool, bool, hipStream_t);
extern template int launch_fused<9, true, true, true, false, 1, 3, false>(FusedArgs, hipStream_t);
extern template int launch_fused<9, true, true, false, false, 1, 3, false>(FusedArgs, hipStream_t);
extern template int launch_fused<9, true, true, true, false, 1, 3, true>(FusedArgs, hipStream_t);
extern template int launch_fused<9, true, true, false, false, 1, 3, true>(FusedArgs, hipStream_t);

static int smooth_fused_impl(const double *b, const double *x_in, double *x_out, double *r, const double *crs, i64 ny,
                             i64 nx, i64 ld, i64 cny, i64 cnx, i64 cld, i64 j0, i64 j1, i64 fix_lo, i64 fix_hi,
                             i64 ofst_j, int sweeps, int prolong_after, double h, double omega, uint64_t *norm_slots,
                             i64 norm_stride, bool want_resid, double *coarse_b,
                             i64 cbny, i64 cbnx, i64 cbld, i64 T0, i64 T1, i64 nbatch, i64 bstride, i64 cstride,
                             i64 cbstride, void *stream)
{
    i64 j2 = 0, j3 = 0, T2 = 0, T3 = 0;
    if (t_also_set) {
        t_also_set = false;
        MG_REQUIRE(t_also_ny == ny && t_also_stream == stream,
                   "mg_smooth_fused: a second row range was set (mg_fused_also_rows) for a launch on an array of %lld rows on "
                   "another stream or of another height and never consumed; dropped", (long long)t_also_ny);
        j2 = t_also[0], j3 = t_also[1], T2 = t_also[2], T3 = t_also[3];
    }
    MG_REQUIRE(b && (x_out || coarse_b), "mg_smooth_fused: null array");
    MG_REQUIRE(nbatch >= 1 && nbatch <= 65535, "mg_smooth_fused: nbatch %lld outside 1..65535", (long long)nbatch);
    MG_REQUIRE(nbatch == 1 || (bstride >= ny * ld && bstride % 2 == 0 && (!crs || cstride >= cny * cld) &&
                               (!coarse_b || cbstride >= cbny * cbld)),
               "mg_smooth_fused: batch strides overlap consecutive grids (or the fine stride is odd)");
    if (prolong_after > 0) {
        // the one combination the V-cycle uses: N1 = 3 sweeps from zero, + P(c), N2 = 6 more sweeps
        MG_REQUIRE(sweeps == 9 && prolong_after == 3 && !x_in && crs && !coarse_b && x_out,
                   "mg_smooth_fused: prolong_after is implemented for 9 sweeps from x = 0 with the prolongation after 3");
    } else {
        MG_REQUIRE(sweeps >= 1 && sweeps <= 6, "mg_smooth_fused: 1..6 sweeps per pass, got %d", sweeps);
    }
    MG_REQUIRE(omega > 0 && omega <= 1, "mg_smooth_fused: omega %g outside (0, 1]", omega);
    MG_REQUIRE(norm_stride == 0 || norm_stride >= MG_NORM_SLOTS, "mg_smooth_fused_restrict: norm_stride must be 0 or >= %d", MG_NORM_SLOTS);
    MG_REQUIRE(ny >= 3 && nx >= 3 && ld >= nx, "mg_smooth_fused: bad shape");
    MG_REQUIRE(ld % 2 == 0 && mg_aligned(b, 16) && (!x_out || mg_aligned(x_out, 16)) && (!x_in || mg_aligned(x_in, 16)) &&
                   (!r || mg_aligned(r, 16)),
               "mg_smooth_fused: arrays must be 16-byte aligned with an even leading dimension");
    MG_REQUIRE(j0 >= 0 && j1 <= ny && j0 <= j1, "mg_smooth_fused: row range [%lld, %lld) outside the array", (long long)j0,
               (long long)j1);
    MG_REQUIRE(j2 == j3 || (j2 >= j1 && j3 <= ny), "mg_smooth_fused: second row range [%lld, %lld) must follow the first inside the array",
               (long long)j2, (long long)j3);
    MG_REQUIRE((!x_out || (x_in != x_out && r != x_out)) && (!r || (const double *)r != x_in),
               "mg_smooth_fused: aliased arrays");
    MG_REQUIRE(nx < ((i64)1 << 30) && ny < ((i64)1 << 30), "mg_smooth_fused: grid side too large");
    if (crs) {
        MG_REQUIRE(x_in || prolong_after > 0, "mg_smooth_fused: prolongation needs x_in");
        MG_REQUIRE(cny >= 2 && cnx >= 2 && cld >= cnx, "mg_smooth_fused: bad coarse shape");
        MG_REQUIRE(ofst_j <= 0, "mg_smooth_fused: ofst_j must be <= 0");
        // every fine cell of the produced rows must have its coarse cells: 1 <= t - ofst_j <= 2*cny-3,
        // 1 <= c <= 2*cnx-3 (rows beyond, read only for the halo of the pipeline, are clamped)
        MG_REQUIRE(j0 - ofst_j >= 1 && (j3 > j2 ? j3 : j1) - 1 - ofst_j <= 2 * cny - 3 && nx - 2 <= 2 * cnx - 3,
                   "mg_smooth_fused: coarse grid %lld x %lld does not cover the fine rows", (long long)cny,
                   (long long)cnx);
    }
    if (coarse_b) {
        MG_REQUIRE((const double *)coarse_b != crs && (const double *)coarse_b != b && coarse_b != x_out && coarse_b != r &&
                       (const double *)coarse_b != x_in,
                   "mg_smooth_fused_restrict: the coarse right-hand side aliases another array of the pass");
        MG_REQUIRE(ofst_j <= 0 && cbny >= 3 && cbnx >= 3 && cbld >= cbnx, "mg_smooth_fused_restrict: bad coarse shape");
        MG_REQUIRE(T0 >= 1 && T1 <= cbny - 1 && T0 <= T1, "mg_smooth_fused_restrict: coarse rows [%lld, %lld) outside 1..%lld",
                   (long long)T0, (long long)T1, (long long)(cbny - 1));
        MG_REQUIRE(2 * (cbnx - 2) + 1 <= nx - 1, "mg_smooth_fused_restrict: fine grid too narrow for %lld coarse columns",
                   (long long)cbnx);
        // every coarse row's fine centre row must be one of the rows this pass produces
        MG_REQUIRE(T0 == T1 || (2 * T0 + ofst_j >= j0 && 2 * (T1 - 1) + ofst_j < j1 && 2 * T0 + ofst_j - 1 >= 0 &&
                                2 * (T1 - 1) + ofst_j + 1 <= ny - 1),
                   "mg_smooth_fused_restrict: coarse rows [%lld, %lld) are not centred on the produced fine rows",
                   (long long)T0, (long long)T1);
        MG_REQUIRE(T2 == T3 || (j2 < j3 && T2 >= T1 && T3 <= cbny - 1 && 2 * T2 + ofst_j >= j2 && 2 * (T3 - 1) + ofst_j < j3 &&
                                2 * (T3 - 1) + ofst_j + 1 <= ny - 1),
                   "mg_smooth_fused_restrict: coarse rows [%lld, %lld) of the second range are not centred on its fine rows",
                   (long long)T2, (long long)T3);
    } else {
        T2 = T3 = 0;
    }
    if (j0 == j1 && j2 < j3) j0 = j2, j1 = j3, T0 = T2, T1 = T3, j2 = j3 = 0, T2 = T3 = 0; // only the second range has rows
    if (j0 == j1) return 0;
    FusedArgs a;
    a.b = b, a.xin = x_in, a.xout = x_out, a.r = r, a.crs = crs;
    a.ny = ny, a.nx = nx, a.ld = ld, a.cny = cny, a.cnx = cnx, a.cld = cld;
    a.j2 = j2, a.j3 = j3, a.T2 = T2, a.T3 = T3, a.nrc1 = 0;
    a.j0 = j0, a.j1 = j1, a.flo = fix_lo, a.fhi = fix_hi, a.oj = (int)ofst_j;
    a.h2 = h * h, a.hh = 1 / (h * h), a.om = omega, a.norm = (u64 *)norm_slots, a.norm_stride = norm_stride;
    a.rpb = a.ncb = a.nrc = a.ntc = a.stagger = a.lockstep = 0;
    a.cb = coarse_b, a.cbnx = cbnx, a.cbld = cbld, a.T0 = T0, a.T1 = T1;
    a.nbatch = nbatch, a.bstride = bstride, a.cstride = cstride, a.cbstride = cbstride;
    // streaming stores when what the pass writes dwarfs the caches: x (and r), or the coarse right-hand side alone
    a.stream = mg_streams((x_out ? ny * ld : cbny * cbld) * nbatch * (i64)sizeof(double));
    const bool zero = x_in == nullptr, prolong = crs != nullptr, restr = coarse_b != nullptr;
    const bool resid = want_resid || restr;
    hipStream_t st = (hipStream_t)stream;
    if (omega != 1.0) {
        // damped sweeps: the pass depths the V-cycle uses
        if (prolong_after > 0)
            return resid ? launch_fused<9, true, true, true, false, 1, 3, true>(a, st)
                         : launch_fused<9, true, true, false, false, 1, 3, true>(a, st);
        switch (sweeps) {
        case 1: return pick_mode<1, true>(a, prolong, zero, resid, restr, st);
        case 3: return pick_mode<3, true>(a, prolong, zero, resid, restr, st);
        case 6: return pick_mode<6, true>(a, prolong, zero, resid, restr, st);
        default: mg_set_error("mg_smooth_fused: damped sweeps (omega != 1) are built for 1, 3 or 6 sweeps per pass, got %d", sweeps); return 1;
        }
    }
    if (prolong_after > 0)
        return resid ? launch_fused<9, true, true, true, false, 1, 3>(a, st) : launch_fused<9, true, true, false, false, 1, 3>(a, st);
    switch (sweeps) {
    case 1: return pick_mode<1>(a, prolong, zero, resid, restr, st);
    case 2: return pick_mode<2>(a, prolong, zero, resid, restr, st);
    case 3: return pick_mode<3>(a, prolong, zero, resid, restr, st);
    case 4: return pick_mode<4>(a, prolong, zero, resid, restr, st);
    case 5: return pick_mode<5>(a, prolong, zero, resid, restr, st);
    default: return pick_mode<6>(a, prolong, zero, resid, restr, st);
    }
}

extern "C" int mg_smooth_fused(const double *b, const double *x_in, double *x_out, double *r, const double *crs,
                               i64 ny, i64 nx, i64 ld, i64 cny, i64 cnx, i64 cld, i64 j0, i64 j1, i64 fix_lo,
                               i64 fix_hi, i64 ofst_j, int sweeps, int prolong_after, double h, double omega,
                               uint64_t *norm_slots, i64 nbatch, i64 bstride, i64 cstride, void *stream)
{
    return smooth_fused_impl(b, x_in, x_out, r, crs, ny, nx, ld, cny, cnx, cld, j0, j1, fix_lo, fix_hi, ofst_j, sweeps,
                             prolong_after, h, omega, norm_slots, 0, r != nullptr, nullptr, 0, 0, 0, 0, 0, nbatch, bstride,
                             cstride, 0, stream);
}

extern "C" int mg_smooth_fused_restrict(const double *b, const double *x_in, double *x_out, double *r,
                                        const double *crs, i64 ny, i64 nx, i64 ld, i64 cny, i64 cnx, i64 cld, i64 j0,
                                        i64 j1, i64 fix_lo, i64 fix_hi, i64 ofst_j, int sweeps, double h, double omega,
                                        uint64_t *norm_slots, i64 norm_stride, double *coarse_b, i64 cbny, i64 cbnx, i64 cbld, i64 T0,
                                        i64 T1, i64 nbatch, i64 bstride, i64 cstride, i64 cbstride, void *stream)
{
    MG_REQUIRE(coarse_b, "mg_smooth_fused_restrict: null coarse array");
    return smooth_fused_impl(b, x_in, x_out, r, crs, ny, nx, ld, cny, cnx, cld, j0, j1, fix_lo, fix_hi, ofst_j, sweeps,
                             0, h, omega, norm_slots, norm_stride, true, coarse_b, cbny, cbnx, cbld, T0, T1, nbatch, bstride, cstride,
                             cbstride, stream);
}

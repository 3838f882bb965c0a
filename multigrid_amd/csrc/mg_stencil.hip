// 5-point stencil kernels of the V-cycle on gfx950: Jacobi sweep, residual (+ fused max-norm),
// laplacian.  All three stream the arrays once:
//   * a workgroup (256 threads = 4 independent waves) owns a tile of 256*V columns x `rpb` rows
//     and marches down the rows with a three-row register window, so every x element is loaded
//     from memory once per tile (plus two halo rows per tile);
//   * each lane owns V consecutive columns (one 8*V-byte load per row and array); the east/west
//     neighbours that belong to the adjacent lane arrive by a whole-wave DPP shift, and only
//     lanes 0 and 63 fetch one extra element per row from the neighbouring wave's columns;
//   * no LDS, no barriers: the waves never wait for each other, and with ~20 VGPRs the CU holds
//     its maximum of 32 waves, which is what keeps enough loads in flight to cover HBM latency.
// Algorithmic traffic: 24 B per point (read x, read b, write out); 16 B for the laplacian.
//
// Arithmetic follows the reference's expression order (see include/mg_hip.h) and the file is
// compiled with -ffp-contract=off.
#include <string.h>

#include "mg_common.h"

enum { OP_JACOBI = 0, OP_JACOBI_ZERO = 1, OP_RESID = 2, OP_RESID_NORM = 3, OP_LAPLACE = 4 };

// V = doubles per memory access (2 = one aligned 16-byte access, 1 = scalar for unaligned arrays),
// STREAM = the output is written with streaming stores (arrays that dwarf the caches),
// U = accesses per lane and row.  Group k of a lane covers columns c0 + k*64*V .. +V-1, so each
// wave-level load/store instruction touches 64*V consecutive doubles (1 KiB when V = 2) and the
// wave's tile is 64*V*U columns wide.
template <int OP, int V, int U, bool STREAM>
__global__ __launch_bounds__(256) void mg_stencil5_kernel(const double *__restrict__ b, const double *__restrict__ x,
                                                          double *__restrict__ out, i64 nx, i64 ld, i64 j0, i64 j1,
                                                          int rpb, int ncb, int nrc, int linear_map, double s,
                                                          double om, i64 bstride, u64 *__restrict__ norm_slots)
{
    constexpr int WT = 64 * V * U; // wave tile width in columns
    constexpr int GS = 64 * V;     // column stride between the groups of a lane
    int cb, rc;
    if (linear_map) {
        cb = (int)blockIdx.x % ncb, rc = (int)blockIdx.x / ncb;
    } else {
        mg_tile_of_block((int)blockIdx.x, ncb, nrc, cb, rc);
    }
    const i64 boff = (i64)blockIdx.y * bstride;
    if (OP != OP_LAPLACE) b += boff;
    if (OP != OP_JACOBI_ZERO) x += boff;
    out += boff;

    const int lane = threadIdx.x & 63;
    const i64 wbase = ((i64)cb * 4 + (threadIdx.x >> 6)) * WT; // first column of this wave's tile
    i64 c0[U], cl[U];                                           // first column of each group / the one loaded
    bool full[U], any[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        c0[k] = wbase + k * GS + lane * V;
        cl[k] = c0[k] + V <= ld ? c0[k] : ld - V; // keep every access inside the row
        full[k] = c0[k] >= 1 && c0[k] + V - 1 <= nx - 2;
        any[k] = c0[k] + V - 1 >= 1 && c0[k] <= nx - 2 && c0[k] + V <= ld;
    }
    const i64 rb = j0 + (i64)rc * rpb;
    const i64 re = rb + rpb < j1 ? rb + rpb : j1;
    u64 nmax = 0;
    // damped Jacobi: x + om * (t - x) with t the undamped value; om == 1 keeps t itself (uniform)
    const bool damped = om != 1.0;
    (void)damped;

    // Writes a row of results.  Groups that straddle the interior (column 0, columns >= nx-1) blend
    // the cells outside it with what `out` already holds, so every store is a whole aligned vector.
    auto store_row = [&](double *orow, double (&o)[U][V]) {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (full[k]) {
                stv_stream<V, STREAM>(orow + c0[k], o[k]);
            } else if (any[k]) {
                double keep[V];
                ldv<V>(keep, orow + c0[k]);
#pragma unroll
                for (int v = 0; v < V; ++v)
                    if (!(c0[k] + v >= 1 && c0[k] + v <= nx - 2)) o[k][v] = keep[v];
                stv_stream<V, STREAM>(orow + c0[k], o[k]);
            }
            if constexpr (OP == OP_RESID_NORM) {
#pragma unroll
                for (int v = 0; v < V; ++v)
                    if (c0[k] + v >= 1 && c0[k] + v <= nx - 2) {
                        u64 key = (u64)__double_as_longlong(fabs(o[k][v]));
                        nmax = key > nmax ? key : nmax;
                    }
            }
        }
    };

    if (rb < re) { // uniform per workgroup
        if constexpr (OP == OP_JACOBI_ZERO) {
            for (i64 j = rb; j < re; ++j) {
                double o[U][V];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    double bb[V];
                    ldv<V>(bb, b + j * ld + cl[k]);
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const double t = 0.25 * (0.0 - s * bb[v]);
                        o[k][v] = damped ? 0.0 + om * (t - 0.0) : t;
                    }
                }
                store_row(out + j * ld, o);
            }
        } else {
            // Columns just outside the wave's tile: lane 0 fetches column wbase-1, lane 63 column
            // wbase+WT (one predicated 8-byte load per row for the two lanes together).
            const bool isL = lane == 0, isR = lane == 63;
            const i64 ce = isL ? wbase - 1 : wbase + WT;
            const bool pe = (isL && wbase >= 1 && wbase - 1 < nx) || (isR && wbase + WT < nx);
            auto load_row = [&](const double *row, double (&d)[U][V], double &e) {
#pragma unroll
                for (int k = 0; k < U; ++k) ldv<V>(d[k], row + cl[k]);
                e = pe ? row[ce] : 0.0;
            };
            auto load_b = [&](const double *row, double (&d)[U][V]) {
                if constexpr (OP != OP_LAPLACE) {
#pragma unroll
                    for (int k = 0; k < U; ++k) ldv<V>(d[k], row + cl[k]);
                }
            };
            // Software pipeline: the loads of row j+2 and of b[j+1] are issued before row j is
            // computed, so a wave always has one row of loads in flight while it computes.
            const double *xr = x + (rb - 1) * ld;
            double up[U][V], cur[U][V], dn[U][V], nxt[U][V], bb[U][V], bn[U][V];
            double eup, ecur, edn, enxt;
            load_row(xr, up, eup);
            load_row(xr + ld, cur, ecur);
            load_row(xr + 2 * ld, dn, edn);
            load_b(b + rb * ld, bb);
            (void)eup;
            for (i64 j = rb; j < re; ++j) {
                const i64 jn = j + 2 <= re ? j + 2 : re;         // row j+2, clamped into the tile's rows
                const i64 jb = j + 1 < re ? j + 1 : re - 1;
                load_row(x + jn * ld, nxt, enxt);
                load_b(b + jb * ld, bn);
                double o[U][V];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    // neighbours across the lane boundary: DPP from the adjacent lane; at the ends of
                    // the wave from the other group of the opposite end lane, or the fetched edge
                    double west = wave_from_prev(cur[k][V - 1]);
                    double east = wave_from_next(cur[k][0]);
                    if (k == 0) {
                        west = isL ? ecur : west;
                    } else {
                        const double w = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cur[k - 1][V - 1]), 63),
                                                          __builtin_amdgcn_readlane(__double2loint(cur[k - 1][V - 1]), 63));
                        west = isL ? w : west;
                    }
                    if (k == U - 1) {
                        east = isR ? ecur : east;
                    } else {
                        const double e = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cur[k + 1 < U ? k + 1 : 0][0]), 0),
                                                          __builtin_amdgcn_readlane(__double2loint(cur[k + 1 < U ? k + 1 : 0][0]), 0));
                        east = isR ? e : east;
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const double E = v + 1 < V ? cur[k][v + 1 < V ? v + 1 : 0] : east;
                        const double W = v > 0 ? cur[k][v > 0 ? v - 1 : 0] : west;
                        if constexpr (OP == OP_JACOBI) {
                            const double t = 0.25 * (E + W + dn[k][v] + up[k][v] - s * bb[k][v]);
                            o[k][v] = damped ? cur[k][v] + om * (t - cur[k][v]) : t;
                        } else if constexpr (OP == OP_LAPLACE) {
                            o[k][v] = (W + E + up[k][v] + dn[k][v] - 4 * cur[k][v]) * s;
                        } else {
                            const double left = W * s, cent = cur[k][v] * s;
                            o[k][v] = bb[k][v] - left - (E + up[k][v] + dn[k][v]) * s + cent * 4;
                        }
                    }
                }
                store_row(out + j * ld, o);
#pragma unroll
                for (int k = 0; k < U; ++k)
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        up[k][v] = cur[k][v], cur[k][v] = dn[k][v], dn[k][v] = nxt[k][v];
                        if constexpr (OP != OP_LAPLACE) bb[k][v] = bn[k][v];
                    }
                ecur = edn, edn = enxt;
            }
        }
    }
    if constexpr (OP == OP_RESID_NORM) {
        // |r| >= 0, so the IEEE bit patterns order like the values, and a NaN (0x7ff8...) beats
        // everything, which is how np.amax treats it.  One atomic per workgroup, spread over slots.
        __shared__ u64 wmax[4];
        nmax = wave_max_u64(nmax);
        if (lane == 0) wmax[threadIdx.x >> 6] = nmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 m = wmax[0];
            for (int w = 1; w < 4; ++w) m = wmax[w] > m ? wmax[w] : m;
            if (m) atomicMax(&norm_slots[(blockIdx.x + blockIdx.y) % MG_NORM_SLOTS], m);
        }
    }
}

// np.amax(np.abs(a)) of a whole array (boundary and halo cells included)
__global__ __launch_bounds__(256) void mg_maxabs_kernel(const double *__restrict__ a, i64 ny, i64 nx, i64 ld,
                                                        u64 *__restrict__ norm_slots)
{
    u64 nmax = 0;
    for (i64 j = blockIdx.y; j < ny; j += gridDim.y)
        for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nx; i += (i64)gridDim.x * 256) {
            u64 key = (u64)__double_as_longlong(fabs(a[j * ld + i]));
            nmax = key > nmax ? key : nmax;
        }
    __shared__ u64 wmax[4];
    nmax = wave_max_u64(nmax);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = nmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 m = wmax[0];
        for (int w = 1; w < 4; ++w) m = wmax[w] > m ? wmax[w] : m;
        if (m) atomicMax(&norm_slots[(blockIdx.x + blockIdx.y) % MG_NORM_SLOTS], m);
    }
}

// ------------------------------------------------------------------------------------------ host

// ---- tuning knobs (speed only; results never depend on them) -------------------------------
static int g_vec_cap = 2;   // doubles per access: 1 or 2
static int g_unroll = 1;    // accesses per lane and row: 1, 2 or 4
static int g_rows = 0;      // rows per tile, 0 = automatic
static int g_linear_map = 0;
i64 g_stream_min_bytes = (i64)256 << 20; // the Infinity Cache's size: what is larger cannot stay cached anyway
extern int g_sor_single_wave; // mg_sor.hip
extern int g_sor_transpose;   // mg_sor.hip
extern int g_fused_waves;     // mg_fused.hip
extern int g_fused_min_rows;  // mg_fused.hip
extern int g_fused_min_blocks; // mg_fused.hip
extern int g_xcd_stagger;      // mg_fused.hip
extern int g_fused_lockstep;   // mg_fused.hip
extern int g_fused_rows;       // mg_fused.hip
extern int g_fused_rows_shallow, g_fused_rows_deep; // mg_fused.hip

extern "C" int mg_set_tuning(const char *key, int value)
{
    MG_REQUIRE(key, "mg_set_tuning: null key");
    if (!strcmp(key, "vector")) {
        MG_REQUIRE(value == 1 || value == 2, "mg_set_tuning: vector must be 1 or 2");
        g_vec_cap = value;
    } else if (!strcmp(key, "unroll")) {
        MG_REQUIRE(value == 1 || value == 2 || value == 4, "mg_set_tuning: unroll must be 1, 2 or 4");
        g_unroll = value;
    } else if (!strcmp(key, "rows")) {
        MG_REQUIRE(value >= 0 && value <= 4096, "mg_set_tuning: rows must be 0..4096");
        g_rows = value;
    } else if (!strcmp(key, "linear_map")) {
        g_linear_map = value != 0;
    } else if (!strcmp(key, "fused_waves")) {
        MG_REQUIRE(value == 1 || value == 2 || value == 4, "mg_set_tuning: fused_waves must be 1, 2 or 4");
        g_fused_waves = value;
    } else if (!strcmp(key, "fused_min_rows")) {
        MG_REQUIRE(value >= 1 && value <= 128, "mg_set_tuning: fused_min_rows must be 1..128");
        g_fused_min_rows = value;
    } else if (!strcmp(key, "fused_min_blocks")) {
        MG_REQUIRE(value >= 1 && value <= (1 << 20), "mg_set_tuning: fused_min_blocks must be 1..2^20");
        g_fused_min_blocks = value;
    } else if (!strcmp(key, "fused_rows")) {
        MG_REQUIRE(value >= 0 && value <= 8192, "mg_set_tuning: fused_rows must be 0..8192");
        g_fused_rows = value;
    } else if (!strcmp(key, "fused_rows_shallow") || !strcmp(key, "fused_rows_deep")) {
        MG_REQUIRE(value >= 8 && value <= 1024, "mg_set_tuning: %s must be 8..1024", key);
        (strcmp(key, "fused_rows_deep") ? g_fused_rows_shallow : g_fused_rows_deep) = value;
    } else if (!strcmp(key, "fused_lockstep")) {
        MG_REQUIRE(value >= 0 && value <= 1024, "mg_set_tuning: fused_lockstep must be 0..1024");
        g_fused_lockstep = value;
    } else if (!strcmp(key, "xcd_stagger")) {
        MG_REQUIRE(value >= -1 && value <= (1 << 20), "mg_set_tuning: xcd_stagger must be -1 (an eighth of the width) or >= 0");
        g_xcd_stagger = value;
    } else if (!strcmp(key, "stream_min_mib")) {
        MG_REQUIRE(value >= 0, "mg_set_tuning: stream_min_mib must be >= 0");
        g_stream_min_bytes = (i64)value << 20;
    } else if (!strcmp(key, "sor_transpose")) {
        MG_REQUIRE(value == 0 || value == 1, "mg_set_tuning: sor_transpose must be 0 or 1");
        g_sor_transpose = value;
    } else if (!strcmp(key, "sor_single_wave")) {
        MG_REQUIRE(value >= 0 && value <= 2, "mg_set_tuning: sor_single_wave must be 0, 1 or 2");
        g_sor_single_wave = value;
    } else {
        mg_set_error("mg_set_tuning: unknown key '%s'", key);
        return 1;
    }
    return 0;
}

static int rows_per_block(i64 nrows, int ncb, i64 nbatch)
{
    if (g_rows > 0) return g_rows;
    // Short tiles: workgroups are dispatched in address order and live for a few rows, so what the chip
    // touches at any moment is one dense band of the arrays, which HBM serves faster than many rows far
    // apart (measured at 32769 x 32769: 4 rows 4.78 ms per sweep, 128 rows 5.08 ms, 1 row 5.65 ms; a plain
    // a + b kernel: one access per thread and no loop 6.0 TB/s, grid-stride loop 4.9-5.5, tools/copy_probe).
    // The two halo rows a tile re-reads come from L2 / the Infinity Cache.
    int rpb = 4;
    while (rpb > 4 && (i64)ncb * ((nrows + rpb - 1) / rpb) * nbatch < 4096) rpb >>= 1;
    return rpb;
}

template <int OP, int V, int U>
static int launch_stencil(const double *b, const double *x, double *out, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1, double s,
                          double om, i64 nbatch, i64 bstride, u64 *slots, int flags, hipStream_t st, const char *name)
{
    const int bw = 256 * V * U;                        // columns per workgroup
    const int ncb = (int)((nx - 1 + bw - 1) / bw);     // columns 0 .. nx-2 need a lane
    const int rpb = rows_per_block(j1 - j0, ncb, nbatch);
    const int nrc = (int)((j1 - j0 + rpb - 1) / rpb);
    dim3 grid((unsigned)(ncb * nrc), (unsigned)nbatch);
    // streaming stores for outputs that dwarf the caches (which also gives those launches a kernel name of their own)
    (void)flags;
    if (mg_streams(ny * ld * nbatch * (i64)sizeof(double)))
        hipLaunchKernelGGL((mg_stencil5_kernel<OP, V, U, true>), grid, dim3(256), 0, st, b, x, out, nx, ld, j0, j1,
                           rpb, ncb, nrc, g_linear_map, s, om, bstride, slots);
    else
        hipLaunchKernelGGL((mg_stencil5_kernel<OP, V, U, false>), grid, dim3(256), 0, st, b, x, out, nx, ld, j0, j1,
                           rpb, ncb, nrc, g_linear_map, s, om, bstride, slots);
    MG_LAUNCH_CHECK(name);
    return 0;
}

static int check_grid(const char *fn, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1, i64 nbatch, i64 bstride)
{
    MG_REQUIRE(ny >= 3 && nx >= 3, "%s: grid %lld x %lld is smaller than 3 x 3", fn, (long long)ny, (long long)nx);
    MG_REQUIRE(ld >= nx, "%s: ld %lld < nx %lld", fn, (long long)ld, (long long)nx);
    MG_REQUIRE(j0 >= 1 && j1 <= ny - 1 && j0 <= j1, "%s: row range [%lld, %lld) outside the interior of %lld rows", fn,
               (long long)j0, (long long)j1, (long long)ny);
    MG_REQUIRE(nbatch >= 1 && nbatch <= 65535, "%s: nbatch %lld outside 1..65535", fn, (long long)nbatch);
    MG_REQUIRE(nbatch == 1 || bstride >= ny * ld, "%s: bstride %lld overlaps consecutive grids", fn, (long long)bstride);
    MG_REQUIRE(nx < ((i64)1 << 30) && ny < ((i64)1 << 30), "%s: grid side too large", fn);
    return 0;
}

// 16-byte accesses need an even leading dimension and 16-byte aligned bases
static int vec_width(i64 ld, i64 bstride, i64 nbatch, const void *p0, const void *p1, const void *p2)
{
    bool ok = g_vec_cap >= 2 && ld % 2 == 0 && (nbatch == 1 || bstride % 2 == 0);
    ok = ok && (!p0 || mg_aligned(p0, 16)) && (!p1 || mg_aligned(p1, 16)) && (!p2 || mg_aligned(p2, 16));
    return ok ? 2 : 1;
}

// U accesses per lane only pay when the grid is wide enough to fill the wider tiles
static int pick_unroll(i64 nx, int v)
{
    int u = g_unroll;
    while (u > 1 && nx < (i64)256 * v * u) u >>= 1;
    return u;
}

#define DISPATCH_VU(OP, ...)                                                    \
    do {                                                                        \
        const int u_ = pick_unroll(nx, v);                                      \
        if (v == 2) {                                                           \
            if (u_ == 4) return launch_stencil<OP, 2, 4>(__VA_ARGS__);          \
            if (u_ == 2) return launch_stencil<OP, 2, 2>(__VA_ARGS__);          \
            return launch_stencil<OP, 2, 1>(__VA_ARGS__);                       \
        }                                                                       \
        if (u_ >= 2) return launch_stencil<OP, 1, 2>(__VA_ARGS__);              \
        return launch_stencil<OP, 1, 1>(__VA_ARGS__);                           \
    } while (0)

extern "C" int mg_jacobi(const double *b, const double *x, double *x_out, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1,
                         double h, double omega, i64 nbatch, i64 bstride, int flags, void *stream)
{
    MG_REQUIRE(omega > 0 && omega <= 1, "mg_jacobi: omega %g outside (0, 1]", omega);
    if (check_grid("mg_jacobi", ny, nx, ld, j0, j1, nbatch, bstride)) return 1;
    MG_REQUIRE(b && x && x_out, "mg_jacobi: null array");
    MG_REQUIRE(x != x_out, "mg_jacobi: x_out aliases x");
    if (j0 == j1) return 0;
    const int v = vec_width(ld, bstride, nbatch, b, x, x_out);
    DISPATCH_VU(OP_JACOBI, b, x, x_out, ny, nx, ld, j0, j1, h * h, omega, nbatch, bstride, nullptr, flags, (hipStream_t)stream,
                "mg_jacobi");
}

extern "C" int mg_jacobi_from_zero(const double *b, double *x_out, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1, double h,
                                   double omega, i64 nbatch, i64 bstride, int flags, void *stream)
{
    MG_REQUIRE(omega > 0 && omega <= 1, "mg_jacobi_from_zero: omega %g outside (0, 1]", omega);
    if (check_grid("mg_jacobi_from_zero", ny, nx, ld, j0, j1, nbatch, bstride)) return 1;
    MG_REQUIRE(b && x_out, "mg_jacobi_from_zero: null array");
    if (j0 == j1) return 0;
    const int v = vec_width(ld, bstride, nbatch, b, x_out, nullptr);
    DISPATCH_VU(OP_JACOBI_ZERO, b, nullptr, x_out, ny, nx, ld, j0, j1, h * h, omega, nbatch, bstride, nullptr, flags,
                (hipStream_t)stream, "mg_jacobi_from_zero");
}

extern "C" int mg_residual(double *r, const double *x, const double *b, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1,
                           double h, uint64_t *norm_slots, i64 nbatch, i64 bstride, int flags, void *stream)
{
    if (check_grid("mg_residual", ny, nx, ld, j0, j1, nbatch, bstride)) return 1;
    MG_REQUIRE(r && x && b, "mg_residual: null array");
    MG_REQUIRE(r != x, "mg_residual: r aliases x");
    if (j0 == j1) return 0;
    const int v = vec_width(ld, bstride, nbatch, b, x, r);
    const double hh = 1 / (h * h);
    if (norm_slots)
        DISPATCH_VU(OP_RESID_NORM, b, x, r, ny, nx, ld, j0, j1, hh, 1.0, nbatch, bstride, (u64 *)norm_slots, flags,
                    (hipStream_t)stream, "mg_residual");
    DISPATCH_VU(OP_RESID, b, x, r, ny, nx, ld, j0, j1, hh, 1.0, nbatch, bstride, nullptr, flags, (hipStream_t)stream,
                "mg_residual");
}

extern "C" int mg_laplacian(const double *x, double *lplc, i64 ny, i64 nx, i64 ld, double h, void *stream)
{
    if (check_grid("mg_laplacian", ny, nx, ld, 1, ny - 1, 1, 0)) return 1;
    MG_REQUIRE(x && lplc && x != lplc, "mg_laplacian: null or aliased array");
    const int v = vec_width(ld, 0, 1, x, lplc, nullptr);
    DISPATCH_VU(OP_LAPLACE, nullptr, x, lplc, ny, nx, ld, 1, ny - 1, 1 / (h * h), 1.0, 1, 0, nullptr, 0, (hipStream_t)stream,
                "mg_laplacian");
}

extern "C" int mg_norm_reset(uint64_t *norm_slots, void *stream)
{
    MG_REQUIRE(norm_slots, "mg_norm_reset: null");
    MG_HIP(hipMemsetAsync(norm_slots, 0, MG_NORM_SLOTS * sizeof(uint64_t), (hipStream_t)stream));
    return 0;
}

extern "C" int mg_norm_read(const uint64_t *norm_slots, double *out, void *stream)
{
    MG_REQUIRE(norm_slots && out, "mg_norm_read: null");
    uint64_t h[MG_NORM_SLOTS];
    MG_HIP(hipMemcpyAsync(h, norm_slots, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    MG_HIP(hipStreamSynchronize((hipStream_t)stream));
    uint64_t m = 0;
    for (int k = 0; k < MG_NORM_SLOTS; ++k) m = h[k] > m ? h[k] : m;
    union { uint64_t u; double d; } cv;
    cv.u = m;
    *out = cv.d;
    return 0;
}

extern "C" int mg_maxabs(const double *a, i64 ny, i64 nx, i64 ld, uint64_t *norm_slots, void *stream)
{
    MG_REQUIRE(a && norm_slots, "mg_maxabs: null");
    MG_REQUIRE(ny >= 1 && nx >= 1 && ld >= nx, "mg_maxabs: bad shape %lld x %lld (ld %lld)", (long long)ny,
               (long long)nx, (long long)ld);
    unsigned gx = (unsigned)((nx + 255) / 256);
    if (gx > 64) gx = 64;
    unsigned gy = (unsigned)(ny < 1024 ? ny : 1024);
    hipLaunchKernelGGL(mg_maxabs_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, a, ny, nx, ld,
                       (u64 *)norm_slots);
    MG_LAUNCH_CHECK("mg_maxabs");
    return 0;
}

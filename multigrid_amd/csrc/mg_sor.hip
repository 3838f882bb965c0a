// Coarse-grid smoother of the V-cycle on gfx950: SOR (omega = 1.5), forward [+ backward] sweeps,
// then the residual.
//
// MG_SOR_LEXICOGRAPHIC reproduces the reference's row-major Gauss-Seidel order exactly, in
// parallel: when cell (j,i) is updated in a forward row-major sweep it sees the NEW values of
// (j,i-1) and (j-1,i) and the OLD values of (j,i+1) and (j+1,i); cells on one anti-diagonal
// d = j+i touch only diagonals d-1 and d+1, so a sweep is a wavefront over d = 2 .. ny+nx-4
// with all cells of a diagonal updated together (backward sweep: d descending).  One workgroup
// per grid, thread t owns row t+1 and walks along it one column per step, with one barrier per
// diagonal.  The grid lives in LDS (row pitch rounded to an even number of doubles so that the
// diagonal accesses of consecutive rows fall in different banks); the west neighbour and the
// cell's own old value stay in registers, and b is prefetched 8 diagonals ahead so its
// latency is not on the critical path.  The critical path is ny+nx dependent steps per sweep,
// which no ordering-preserving algorithm can shorten; at ~0.15 us per step a 129 x 129 grid
// takes ~40 us per sweep.
// Grids too large for LDS use the same wavefront on global memory (correct but slow).
//
// MG_SOR_REDBLACK is the usual two-colour Gauss-Seidel, one launch per colour.
#include "mg_common.h"

#define SOR_PF 8 // prefetch distance for b, in diagonals

template <bool LDS>
__global__ __launch_bounds__(1024) void mg_sor_wavefront_kernel(const double *__restrict__ b, double *x,
                                                                double *__restrict__ r, i64 ny, i64 nx, i64 ld,
                                                                int ldl, double h2, int iterations, int backward,
                                                                i64 bstride)
{
    extern __shared__ double sx[];
    b += (i64)blockIdx.x * bstride;
    x += (i64)blockIdx.x * bstride;
    if (r) r += (i64)blockIdx.x * bstride;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double a = 1.5, alpha = 1 - a, beta = a / 4;

    if constexpr (LDS) {
        for (i64 idx = tid; idx < ny * nx; idx += nt) {
            const i64 j = idx / nx, i = idx - j * nx;
            sx[j * ldl + i] = x[j * ld + i];
        }
        __syncthreads();
    }
#define X(j, i) (LDS ? sx[(j) * (i64)ldl + (i)] : x[(j) * ld + (i)])
#define XW(j, i, v)                         \
    do {                                    \
        if (LDS) sx[(j) * (i64)ldl + (i)] = (v); \
        else x[(j) * ld + (i)] = (v);       \
    } while (0)

    const i64 dlo = 2, dhi = (ny - 2) + (nx - 2);
    const bool one_row = (ny - 2) <= nt; // every thread owns at most one row: register pipeline
    const i64 jrow = 1 + tid;
    const bool have_row = jrow <= ny - 2;

    for (int it = 0; it < iterations; ++it) {
        for (int dir = 0; dir < (backward ? 2 : 1); ++dir) {
            const i64 step = dir == 0 ? 1 : -1;
            const i64 dstart = dir == 0 ? dlo : dhi;
            const i64 nsteps = dhi - dlo + 1;
            if (one_row) {
                // Row-walking pipeline.  At step k the thread is at column i = dstart + step*k - jrow.
                double breg[SOR_PF];
                double self = 0, lat = 0; // own old value; value of the cell just left behind (new)
                const double *brow = b + (have_row ? jrow : 1) * ld;
#pragma unroll
                for (int u = 0; u < SOR_PF; ++u) {
                    const i64 i = dstart + step * u - jrow;
                    breg[u] = (have_row && i >= 1 && i <= nx - 2) ? brow[i] : 0.0;
                }
                if (have_row) {
                    const i64 ifirst = dir == 0 ? 1 : nx - 2;
                    self = X(jrow, ifirst);
                    lat = X(jrow, ifirst - step); // boundary cell next to the first column
                }
                for (i64 k0 = 0; k0 < nsteps; k0 += SOR_PF) {
#pragma unroll
                    for (int u = 0; u < SOR_PF; ++u) {
                        const i64 k = k0 + u;
                        const i64 i = dstart + step * k - jrow;
                        const bool act = have_row && k < nsteps && i >= 1 && i <= nx - 2;
                        const double bv = breg[u];
                        const i64 ip = i + step * SOR_PF;
                        breg[u] = (have_row && ip >= 1 && ip <= nx - 2) ? brow[ip] : 0.0;
                        if (act) {
                            const double ahead = X(jrow, i + step);   // (j, i+1) forward / (j, i-1) backward: old
                            const double below = X(jrow + 1, i);
                            const double above = X(jrow - 1, i);
                            const double E = dir == 0 ? ahead : lat;
                            const double W = dir == 0 ? lat : ahead;
                            const double v = alpha * self + beta * (E + W + below + above - h2 * bv);
                            XW(jrow, i, v);
                            lat = v;
                            self = ahead;
                        }
                        if (k < nsteps) {
                            if constexpr (LDS) {
                                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                            } else {
                                __syncthreads();
                            }
                        }
                    }
                }
            } else {
                for (i64 k = 0; k < nsteps; ++k) {
                    const i64 d = dstart + step * k;
                    for (i64 j = 1 + tid; j <= ny - 2; j += nt) {
                        const i64 i = d - j;
                        if (i >= 1 && i <= nx - 2) {
                            const double v = alpha * X(j, i) +
                                             beta * (X(j, i + 1) + X(j, i - 1) + X(j + 1, i) + X(j - 1, i) -
                                                     h2 * b[j * ld + i]);
                            XW(j, i, v);
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }

    if constexpr (LDS) {
        __syncthreads();
        const double hh = 1 / h2;
        for (i64 idx = tid; idx < ny * nx; idx += nt) {
            const i64 j = idx / nx, i = idx - j * nx;
            const double c = sx[j * ldl + i];
            x[j * ld + i] = c;
            if (r && j >= 1 && j <= ny - 2 && i >= 1 && i <= nx - 2) {
                const double left = sx[j * ldl + i - 1] * hh, cent = c * hh;
                r[j * ld + i] = b[j * ld + i] - left -
                                (sx[j * ldl + i + 1] + sx[(j - 1) * ldl + i] + sx[(j + 1) * ldl + i]) * hh + cent * 4;
            }
        }
    }
#undef X
#undef XW
}

// ---- single-wave wavefront -------------------------------------------------------------------
// For grids of at most 64*RPL+1 rows one wave walks the whole wavefront without any barrier:
// lane l owns rows RPL*l+1 .. RPL*l+RPL and every row advances one column per step, row j being at
// column d-j at step d (d descending in the backward sweep).  What a cell needs from the rows
// above and below sits in registers of the same lane or of the neighbouring lane:
//   forward   above (new) = the value the row above produced one step ago, below (old) = the value
//             the row below has just loaded as its own next cell;
//   backward  the same with the roles of above and below exchanged.
// The values cross the lane seam by DPP wave shifts, so the dependent chain of a step is two lane
// shifts and a handful of FP64 operations.  LDS keeps the grid between sweeps; a lane reads and
// writes only its own rows there (plus the constant top boundary row), SORW_PF steps ahead of use,
// like b from global memory.  Rows beyond the last interior row (the bottom boundary row included)
// stream their values without being updated, which supplies the bottom boundary to the row above.
// The remaining waves of the workgroup only help to load the grid and to write x and r back.
#define SORW_PF 8

template <int RPL, int DIR>
__device__ __forceinline__ void sor_wave_sweep(const double *__restrict__ b, double *sx, int ny, int nx, i64 ld, int ldl,
                                               double h2, int lane)
{
    const double a = 1.5, alpha = 1 - a, beta = a / 4;
    int j[RPL];
    bool strm[RPL], inter[RPL];
    double *srow[RPL];
    const double *brow[RPL];
    double self[RPL], lat[RPL];
    constexpr int PF = RPL >= 4 ? 4 : SORW_PF;
    double aq[RPL][PF], bq[RPL][PF], tq[PF]; // the chunk of PF steps being computed
    double an[RPL][PF], bn[RPL][PF], tn[PF]; // the next chunk, in flight
    const int d0 = DIR > 0 ? 1 : ny + nx - 2;                 // first step: row 1 at column 0 / row ny-1 at column nx-1
    const int nsteps = DIR > 0 ? ny + nx - 4 : ny + nx - 3;   // last step: row ny-2 at column nx-2 / row 1 at column 1
    const int clo = DIR > 0 ? 0 : 1, chi = DIR > 0 ? nx - 2 : nx - 1; // columns at which a row shifts its window
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        j[k] = RPL * lane + 1 + k;
        strm[k] = j[k] <= ny - 1;
        inter[k] = j[k] <= ny - 2;
        srow[k] = sx + (j[k] < ny - 1 ? j[k] : ny - 1) * ldl;
        brow[k] = b + (i64)(j[k] < ny - 1 ? j[k] : ny - 1) * ld;
        self[k] = srow[k][DIR > 0 ? 0 : nx - 1];
        lat[k] = 0;
    }
    // all loads are unconditional with clamped (always valid) addresses; a value loaded for a column
    // outside the row's range is never used
    auto fetch = [&](double (&qa)[RPL][PF], double (&qb)[RPL][PF], double (&qt)[PF], int dc) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int d = dc + DIR * u;
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const int c = d - j[k];
                int ca = c + DIR, cb = c;
                ca = ca < 0 ? 0 : (ca > nx - 1 ? nx - 1 : ca);
                cb = cb < 0 ? 0 : (cb > nx - 1 ? nx - 1 : cb);
                qa[k][u] = srow[k][ca];
                qb[k][u] = brow[k][cb];
            }
            int c0 = d - 1;
            c0 = c0 < 0 ? 0 : (c0 > nx - 1 ? nx - 1 : c0);
            qt[u] = sx[c0];
        }
    };
    fetch(aq, bq, tq, d0);
    for (int t0 = 0; t0 < nsteps; t0 += PF) {
        fetch(an, bn, tn, d0 + DIR * (t0 + PF)); // used PF .. 2*PF-1 steps from now
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int d = d0 + DIR * (t0 + u); // steps past the last one update nothing: no row is in range
            double ahead[RPL], bv[RPL], above[RPL], below[RPL];
#pragma unroll
            for (int k = 0; k < RPL; ++k) ahead[k] = aq[k][u], bv[k] = bq[k][u];
            const double top = tq[u];
            if (DIR > 0) {
                const double p = wave_from_prev(lat[RPL - 1]);
                above[0] = lane == 0 ? top : p;
                below[RPL - 1] = wave_from_next(ahead[0]);
#pragma unroll
                for (int k = 1; k < RPL; ++k) above[k] = lat[k - 1], below[k - 1] = ahead[k];
            } else {
                const double p = wave_from_prev(ahead[RPL - 1]);
                above[0] = lane == 0 ? top : p;
                below[RPL - 1] = wave_from_next(lat[0]);
#pragma unroll
                for (int k = 1; k < RPL; ++k) above[k] = ahead[k - 1], below[k - 1] = lat[k];
            }
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const int c = d - j[k];
                const bool inr = strm[k] && c >= clo && c <= chi;
                const bool act = inter[k] && c >= 1 && c <= nx - 2;
                const double E = DIR > 0 ? ahead[k] : lat[k];
                const double W = DIR > 0 ? lat[k] : ahead[k];
                double v = alpha * self[k] + beta * (E + W + below[k] + above[k] - h2 * bv[k]);
                v = act ? v : self[k];
                if (act) srow[k][c] = v;
                lat[k] = inr ? v : lat[k];
                self[k] = inr ? ahead[k] : self[k];
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            tq[u] = tn[u];
#pragma unroll
            for (int k = 0; k < RPL; ++k) aq[k][u] = an[k][u], bq[k][u] = bn[k][u];
        }
    }
}

template <int RPL>
__global__ __launch_bounds__(256) void mg_sor_wave_kernel(const double *__restrict__ b, double *x,
                                                          double *__restrict__ r, int ny, int nx, i64 ld, int ldl,
                                                          double h2, int iterations, int backward, i64 bstride)
{
    extern __shared__ double sx[];
    b += (i64)blockIdx.x * bstride;
    x += (i64)blockIdx.x * bstride;
    if (r) r += (i64)blockIdx.x * bstride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int j = wv; j < ny; j += 4)
        for (int i = lane; i < nx; i += 64) sx[j * ldl + i] = x[j * ld + i];
    __syncthreads();
    if (wv == 0) {
        for (int it = 0; it < iterations; ++it) {
            sor_wave_sweep<RPL, 1>(b, sx, ny, nx, ld, ldl, h2, lane);
            if (backward) sor_wave_sweep<RPL, -1>(b, sx, ny, nx, ld, ldl, h2, lane);
        }
    }
    __syncthreads();
    const double hh = 1 / h2;
    for (int j = wv; j < ny; j += 4)
        for (int i = lane; i < nx; i += 64) {
            const double c = sx[j * ldl + i];
            x[j * ld + i] = c;
            if (r && j >= 1 && j <= ny - 2 && i >= 1 && i <= nx - 2) {
                const double left = sx[j * ldl + i - 1] * hh, cent = c * hh;
                r[j * ld + i] = b[j * ld + i] - left -
                                (sx[j * ldl + i + 1] + sx[(j - 1) * ldl + i] + sx[(j + 1) * ldl + i]) * hh + cent * 4;
            }
        }
}

__global__ __launch_bounds__(256) void mg_sor_rb_kernel(const double *__restrict__ b, double *x, i64 ny, i64 nx, i64 ld,
                                                        double h2, int colour, i64 bstride)
{
    b += (i64)blockIdx.z * bstride;
    x += (i64)blockIdx.z * bstride;
    const double a = 1.5, alpha = 1 - a, beta = a / 4;
    for (i64 j = 1 + blockIdx.y; j <= ny - 2; j += gridDim.y) {
        const i64 i0 = 1 + ((colour + j + 1) & 1);
        for (i64 i = i0 + 2 * ((i64)blockIdx.x * 256 + threadIdx.x); i <= nx - 2; i += 2 * (i64)gridDim.x * 256) {
            double *p = x + j * ld + i;
            *p = alpha * *p + beta * (p[1] + p[-1] + p[ld] + p[-ld] - h2 * b[j * ld + i]);
        }
    }
}

int g_sor_single_wave = 1; // tuning knob "sor_single_wave": 0 = never, 1 = where it is faster, 2 = wherever it fits

extern "C" int mg_sor(const double *b, double *x, double *r, i64 ny, i64 nx, i64 ld, double h, i64 iterations,
                      int backward, int ordering, i64 nbatch, i64 bstride, void *stream)
{
    MG_REQUIRE(b && x, "mg_sor: null array");
    MG_REQUIRE(ny >= 3 && nx >= 3 && ld >= nx, "mg_sor: bad shape %lld x %lld (ld %lld)", (long long)ny, (long long)nx,
               (long long)ld);
    MG_REQUIRE(iterations >= 0 && iterations < (1 << 20), "mg_sor: bad iteration count");
    MG_REQUIRE(nbatch >= 1 && nbatch <= 65535, "mg_sor: nbatch %lld outside 1..65535", (long long)nbatch);
    MG_REQUIRE(nbatch == 1 || bstride >= ny * ld, "mg_sor: bstride overlaps consecutive grids");
    MG_REQUIRE(ordering == MG_SOR_LEXICOGRAPHIC || ordering == MG_SOR_REDBLACK, "mg_sor: unknown ordering %d", ordering);
    MG_REQUIRE(nx < ((i64)1 << 30) && ny < ((i64)1 << 30), "mg_sor: grid side too large");
    hipStream_t st = (hipStream_t)stream;
    const double h2 = h * h;
    if (ordering == MG_SOR_LEXICOGRAPHIC) {
        const int ldl = (int)((nx + 1) & ~(i64)1);
        const bool lds = ny * (i64)ldl <= MG_SOR_LDS_DOUBLES;
        int nt = (int)(((ny - 2) + 63) / 64 * 64);
        if (nt > 1024) nt = 1024;
        // measured: ahead of the one-thread-per-row kernel only with one row per lane and few grids in flight
        if (lds && ny - 1 <= 64 * 4 && (g_sor_single_wave == 2 || (g_sor_single_wave == 1 && ny - 1 <= 64 && nbatch <= 256))) {
            const size_t bytes = (size_t)ny * ldl * sizeof(double);
            static bool wattr_set = false;
            if (!wattr_set) {
                const int cap = MG_SOR_LDS_DOUBLES * (int)sizeof(double);
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                wattr_set = true;
            }
            const dim3 grid((unsigned)nbatch), block(256);
            if (ny - 1 <= 64)
                hipLaunchKernelGGL(mg_sor_wave_kernel<1>, grid, block, bytes, st, b, x, r, (int)ny, (int)nx, ld, ldl, h2,
                                   (int)iterations, backward, bstride);
            else if (ny - 1 <= 128)
                hipLaunchKernelGGL(mg_sor_wave_kernel<2>, grid, block, bytes, st, b, x, r, (int)ny, (int)nx, ld, ldl, h2,
                                   (int)iterations, backward, bstride);
            else
                hipLaunchKernelGGL(mg_sor_wave_kernel<4>, grid, block, bytes, st, b, x, r, (int)ny, (int)nx, ld, ldl, h2,
                                   (int)iterations, backward, bstride);
            MG_LAUNCH_CHECK("mg_sor (single-wave wavefront)");
            return 0;
        }
        if (lds) {
            const size_t bytes = (size_t)ny * ldl * sizeof(double);
            static bool attr_set = false;
            if (!attr_set) {
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wavefront_kernel<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           MG_SOR_LDS_DOUBLES * (int)sizeof(double)));
                attr_set = true;
            }
            hipLaunchKernelGGL(mg_sor_wavefront_kernel<true>, dim3((unsigned)nbatch), dim3(nt), bytes, st, b, x, r, ny,
                               nx, ld, ldl, h2, (int)iterations, backward, bstride);
            MG_LAUNCH_CHECK("mg_sor (LDS wavefront)");
            return 0;
        }
        hipLaunchKernelGGL(mg_sor_wavefront_kernel<false>, dim3((unsigned)nbatch), dim3(nt), 0, st, b, x, nullptr, ny,
                           nx, ld, ldl, h2, (int)iterations, backward, bstride);
        MG_LAUNCH_CHECK("mg_sor (global wavefront)");
    } else {
        unsigned gx = (unsigned)(((nx - 2 + 1) / 2 + 255) / 256);
        if (gx > 64) gx = 64;
        unsigned gy = (unsigned)(ny - 2 < 2048 ? ny - 2 : 2048);
        dim3 grid(gx, gy, (unsigned)nbatch);
        for (i64 it = 0; it < iterations; ++it)
            for (int p = 0; p < (backward ? 4 : 2); ++p) {
                const int colour = (p == 0 || p == 3) ? 0 : 1;
                hipLaunchKernelGGL(mg_sor_rb_kernel, grid, dim3(256), 0, st, b, x, ny, nx, ld, h2, colour, bstride);
                MG_LAUNCH_CHECK("mg_sor (red-black)");
            }
    }
    if (r) return mg_residual(r, x, b, ny, nx, ld, 1, ny - 1, h, nullptr, nbatch, bstride, 0, stream);
    return 0;
}

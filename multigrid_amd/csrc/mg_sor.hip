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
#include <atomic>

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
// The values cross the lane seam by DPP wave shifts.  A single wave issues one vector instruction
// per ~4 cycles, so the step time IS the instruction count and the loop is kept to the arithmetic:
//  * every row streams unconditionally (window shift without range tests): LDS has guard zones in
//    front of and behind the grid, so reads beyond a row's ends are harmless and never used;
//    b comes through a bounds-checked buffer resource (out-of-range reads return 0);
//  * addresses advance by immediate offsets inside a chunk of SORW_PF steps; the loads of the next
//    chunk are issued before the current one is computed (two chunks per loop iteration, no copies);
//  * an inactive cell (boundary column, boundary row) keeps its value by one select and writes to a
//    dummy slot instead of branching;
//  * lane 0 takes the top boundary row as the fallback operand of the lane shift.
// LDS keeps the grid between sweeps; a lane reads and writes only its own rows there (plus the
// constant top boundary row).  Rows beyond the last interior row (the bottom boundary row included)
// stream without being updated, which supplies the bottom boundary to the row above.
// The remaining waves of the workgroup only help to load the grid and to write x and r back.
#define SORW_PF 8
#define SORW_GUARD 32      // doubles in front of the grid in LDS: >= 3 * SORW_PF + 8 (see sor_wave_sweep)
#define SORW_GUARD_BACK 96 // behind it: the same, plus one dummy slot per lane for the writes of inactive cells

// TR: the wave walks the TRANSPOSED grid (ny, nx here are the transposed grid's rows and columns; sx holds x
// transposed, sb the right-hand side transposed, with guard zones like sx's).  A grid with many rows and few columns
// then needs fewer rows per lane.  The lexicographic sweep visits a cell after its upper and left neighbours and
// before its lower and right ones whichever of the two directions is the inner one, so the transposed wavefront
// applies the same updates; the sum keeps the reference's operand order, (E + W) + S + N of the original grid.
template <int RPL, int DIR, bool TR>
__device__ __forceinline__ void sor_wave_sweep(__amdgpu_buffer_rsrc_t brs, double *sx, const double *sb, double *dummy, int ny,
                                               int nx, int ld, int ldl, double h2, int lane)
{
    constexpr int PF = SORW_PF;
    const double a = 1.5, alpha = 1 - a, beta = a / 4;
    int j[RPL]; // row of slot k, clamped to the bottom boundary row: lanes beyond it stream that row again (unused)
    bool inter[RPL];
    double *srow[RPL];
    const double *sbrow[RPL];
    int boff[RPL];
    double self[RPL], lat[RPL];
    const int d0 = DIR > 0 ? 0 : ny + nx - 1;               // first step: row 1 at column -1 / row ny-1 at column nx
    const int nsteps = DIR > 0 ? ny + nx - 3 : ny + nx - 2; // last step: row ny-2 at column nx-2 / row 1 at column 1
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int jr = RPL * lane + 1 + k;
        inter[k] = jr <= ny - 2;
        // Every address below is derived from the clamped row, its column offsets included: a lane past the grid
        // reads exactly what the bottom boundary row's lane reads.  So the lowest LDS offset any lane forms is that
        // of row 1 at column -1 - 3*PF (backward sweep, three chunks of look-ahead), i.e. >= -SORW_GUARD for every
        // grid of at least 3 columns, and the highest is row ny-1 at column nx + 3*PF, inside the back guard.
        j[k] = jr < ny - 1 ? jr : ny - 1;
        srow[k] = sx + j[k] * ldl;
        sbrow[k] = TR ? sb + j[k] * ldl : nullptr;
        boff[k] = j[k] * ld;
        self[k] = 0, lat[k] = 0;
    }
    struct Chunk {
        double ah[RPL][PF], bv[RPL][PF], top[PF];
    };
    // issue the loads of the PF steps starting at step t (column of row j: cb = d0 + DIR*t - j, then cb + DIR*u)
    auto fetch = [&](Chunk &q, int t) {
        const int dd = d0 + DIR * t;
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const int cb = dd - j[k];
            // forward: ahead = X(j, c+1), c = cb+u;  backward: ahead = X(j, c-1), c = cb-u  (offsets kept >= 0)
            const double *pa = DIR > 0 ? srow[k] + cb + 1 : srow[k] + cb - 1 - (PF - 1);
            const int ob = (boff[k] + (DIR > 0 ? cb : cb - (PF - 1))) * 8;
            const double *pbv = TR ? (DIR > 0 ? sbrow[k] + cb : sbrow[k] + cb - (PF - 1)) : nullptr;
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int w = DIR > 0 ? u : PF - 1 - u;
                q.ah[k][u] = pa[w];
                if constexpr (TR)
                    q.bv[k][u] = pbv[w];
                else
                    q.bv[k][u] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(brs, ob + w * 8, 0, 0));
            }
        }
        const int c0 = dd - 1; // column of row 1: the top boundary row is read there (used by lane 0 only)
        const double *pt = DIR > 0 ? sx + c0 : sx + c0 - (PF - 1);
#pragma unroll
        for (int u = 0; u < PF; ++u) q.top[u] = pt[DIR > 0 ? u : PF - 1 - u];
    };
    auto compute = [&](const Chunk &q, int t) {
        const int dd = d0 + DIR * t;
        double *pw[RPL];
        unsigned cm[RPL];
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const int cb = dd - j[k];
            pw[k] = DIR > 0 ? srow[k] + cb : srow[k] + cb - (PF - 1);
            cm[k] = (unsigned)(cb - 1);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            double above[RPL], below[RPL];
            const int w = DIR > 0 ? u : PF - 1 - u;
            // the row above the lane's first row lives in the previous lane; lane 0 has no previous
            // lane and keeps the fallback operand: the top boundary row
            {
                const double src = DIR > 0 ? lat[RPL - 1] : q.ah[RPL - 1][u];
                const int lo = __builtin_amdgcn_update_dpp(__double2loint(q.top[u]), __double2loint(src), 0x138, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(__double2hiint(q.top[u]), __double2hiint(src), 0x138, 0xf, 0xf, false);
                above[0] = __hiloint2double(hi, lo);
            }
            {
                const double src = DIR > 0 ? q.ah[0][u] : lat[0];
                const int lo = __builtin_amdgcn_mov_dpp(__double2loint(src), 0x130, 0xf, 0xf, true);
                const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(src), 0x130, 0xf, 0xf, true);
                below[RPL - 1] = __hiloint2double(hi, lo); // lane 63's last row is never interior
            }
#pragma unroll
            for (int k = 1; k < RPL; ++k) {
                above[k] = DIR > 0 ? lat[k - 1] : q.ah[k - 1][u];
                below[k - 1] = DIR > 0 ? q.ah[k][u] : lat[k];
            }
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const unsigned cc = DIR > 0 ? cm[k] + (unsigned)u : cm[k] - (unsigned)u; // column - 1
                const bool act = inter[k] && cc < (unsigned)(nx - 2);
                const double ahead = q.ah[k][u];
                const double E = DIR > 0 ? ahead : lat[k];
                const double W = DIR > 0 ? lat[k] : ahead;
                // transposed: this grid's lower/upper neighbours are the original's east/west, its east/west the
                // original's lower/upper
                double v = TR ? alpha * self[k] + beta * (below[k] + above[k] + E + W - h2 * q.bv[k][u])
                              : alpha * self[k] + beta * (E + W + below[k] + above[k] - h2 * q.bv[k][u]);
                v = act ? v : self[k];
                double *dst = act ? pw[k] : dummy;
                dst[w] = v;
                lat[k] = v;
                self[k] = ahead;
            }
        }
    };
    Chunk A, B;
    fetch(A, 0);
    for (int t = 0; t < nsteps; t += 2 * PF) {
        fetch(B, t + PF);
        compute(A, t);
        fetch(A, t + 2 * PF);
        compute(B, t + PF);
    }
}

template <int RPL, bool TR = false>
__global__ __launch_bounds__(256) void mg_sor_wave_kernel(const double *__restrict__ b, double *x,
                                                          double *__restrict__ r, int ny, int nx, int ld, int ldl,
                                                          double h2, int iterations, int backward, i64 bstride)
{
    extern __shared__ double smem[];
    // the grid the wave walks: the array's, or its transpose (ldl is that grid's row pitch in LDS)
    const int wy = TR ? nx : ny, wx = TR ? ny : nx;
    double *const sx = smem + SORW_GUARD;           // the grid; guard zones on both sides
    double *const sb = TR ? sx + wy * ldl + SORW_GUARD_BACK + SORW_GUARD : nullptr; // transposed: b as well, laid out alike
    double *const dummy = sx + wy * ldl + (threadIdx.x & 63); // inactive cells write here (inside the back guard)
    b += (i64)blockIdx.x * bstride;
    x += (i64)blockIdx.x * bstride;
    if (r) r += (i64)blockIdx.x * bstride;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // element (j, i) of the array in LDS
    auto at = [&](int j, int i) -> double & { return TR ? sx[i * ldl + j] : sx[j * ldl + i]; };
    for (int j = wv; j < ny; j += 4)
        for (int i = lane; i < nx; i += 64) {
            at(j, i) = x[(i64)j * ld + i];
            if constexpr (TR) sb[i * ldl + j] = b[(i64)j * ld + i];
        }
    if (tid < SORW_GUARD) smem[tid] = 0.0;
    if (tid < SORW_GUARD_BACK) sx[wy * ldl + tid] = 0.0;
    if constexpr (TR) {
        if (tid < SORW_GUARD) sb[tid - SORW_GUARD] = 0.0;
        if (tid < SORW_GUARD_BACK) sb[wy * ldl + tid] = 0.0;
        // the pad column of every row (the pitch is even, the row length may be odd) is read as a cell's neighbour
        // in the streaming loop like any other: defined values, never used
        if (ldl > wx)
            for (int q = tid; q < wy; q += 256) sb[q * ldl + wx] = 0.0, sx[q * ldl + wx] = 0.0;
    }
    __syncthreads();
    if (wv == 0) {
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)b, 0, TR ? 0 : ny * ld * 8, 0x00020000);
        for (int it = 0; it < iterations; ++it) {
            sor_wave_sweep<RPL, 1, TR>(brs, sx, sb, dummy, wy, wx, ld, ldl, h2, lane);
            if (backward) sor_wave_sweep<RPL, -1, TR>(brs, sx, sb, dummy, wy, wx, ld, ldl, h2, lane);
        }
    }
    __syncthreads();
    const double hh = 1 / h2;
    for (int j = wv; j < ny; j += 4)
        for (int i = lane; i < nx; i += 64) {
            const double c = at(j, i);
            x[(i64)j * ld + i] = c;
            if (r && j >= 1 && j <= ny - 2 && i >= 1 && i <= nx - 2) {
                const double left = at(j, i - 1) * hh, cent = c * hh;
                r[(i64)j * ld + i] = b[(i64)j * ld + i] - left - (at(j, i + 1) + at(j - 1, i) + at(j + 1, i)) * hh + cent * 4;
            }
        }
}

// One colour of a red-black sweep over rows j0 <= j < j1.  GS: plain Gauss-Seidel, the Jacobi value in place;
// otherwise SOR with omega = 1.5 (the coarse smoother's weight).
template <bool GS>
__global__ __launch_bounds__(256) void mg_sor_rb_kernel(const double *__restrict__ b, double *x, i64 j0, i64 j1, i64 nx, i64 ld,
                                                        double h2, int colour, i64 bstride)
{
    b += (i64)blockIdx.z * bstride;
    x += (i64)blockIdx.z * bstride;
    const double a = 1.5, alpha = 1 - a, beta = a / 4;
    for (i64 j = j0 + blockIdx.y; j < j1; j += gridDim.y) {
        const i64 i0 = 1 + ((colour + j + 1) & 1);
        for (i64 i = i0 + 2 * ((i64)blockIdx.x * 256 + threadIdx.x); i <= nx - 2; i += 2 * (i64)gridDim.x * 256) {
            double *p = x + j * ld + i;
            if constexpr (GS)
                *p = 0.25 * (p[1] + p[-1] + p[ld] + p[-ld] - h2 * b[j * ld + i]);
            else
                *p = alpha * *p + beta * (p[1] + p[-1] + p[ld] + p[-ld] - h2 * b[j * ld + i]);
        }
    }
}

extern "C" int mg_gauss_seidel_rb(const double *b, double *x, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1, double h, int colour,
                                  i64 nbatch, i64 bstride, void *stream)
{
    MG_REQUIRE(b && x, "mg_gauss_seidel_rb: null array");
    MG_REQUIRE(ny >= 3 && nx >= 3 && ld >= nx, "mg_gauss_seidel_rb: bad shape %lld x %lld (ld %lld)", (long long)ny, (long long)nx,
               (long long)ld);
    MG_REQUIRE(j0 >= 1 && j1 <= ny - 1 && j0 <= j1, "mg_gauss_seidel_rb: row range [%lld, %lld) outside the interior", (long long)j0,
               (long long)j1);
    MG_REQUIRE(colour == 0 || colour == 1, "mg_gauss_seidel_rb: colour must be 0 or 1");
    MG_REQUIRE(nbatch >= 1 && nbatch <= 65535 && (nbatch == 1 || bstride >= ny * ld), "mg_gauss_seidel_rb: bad batch");
    if (j0 == j1) return 0;
    unsigned gx = (unsigned)(((nx - 2 + 1) / 2 + 255) / 256);
    if (gx > 64) gx = 64;
    const unsigned gy = (unsigned)(j1 - j0 < 4096 ? j1 - j0 : 4096);
    hipLaunchKernelGGL(mg_sor_rb_kernel<true>, dim3(gx, gy, (unsigned)nbatch), dim3(256), 0, (hipStream_t)stream, b, x, j0, j1, nx, ld,
                       h * h, colour, bstride);
    MG_LAUNCH_CHECK("mg_gauss_seidel_rb");
    return 0;
}

// The LDS limit of a kernel is an attribute of the function PER DEVICE: it is set once on every device a launch is
// made on (flags per device and kernel group; setting it twice from two host threads is harmless).
#define SOR_MAX_DEVICES 64
static std::atomic<unsigned char> g_lds_attr[SOR_MAX_DEVICES][3];
static bool lds_attr_done(int which)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= SOR_MAX_DEVICES) return false; // unknown: set it again
    return g_lds_attr[d][which].load(std::memory_order_acquire) != 0;
}
static void lds_attr_mark(int which)
{
    int d = 0;
    if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < SOR_MAX_DEVICES) g_lds_attr[d][which].store(1, std::memory_order_release);
}

int g_sor_transpose = 1;   // tuning knob "sor_transpose": walk tall narrow grids transposed (0 = never)
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
        // A grid with more rows than columns whose transpose needs fewer rows per lane is walked transposed
        // (129 x 17: one row per lane instead of two, 64 instead of 109 us for three sweeps both ways).
        const int ldt = (int)((ny + 1) & ~(i64)1);
        const i64 t_doubles = 2 * (nx * (i64)ldt + SORW_GUARD + SORW_GUARD_BACK);
        if (g_sor_single_wave >= 1 && g_sor_transpose && nx < ny && nx - 1 <= 128 && (nx - 1 + 63) / 64 < (ny - 1 + 63) / 64 &&
            t_doubles <= MG_SOR_LDS_DOUBLES + SORW_GUARD + SORW_GUARD_BACK) {
            if (!lds_attr_done(2)) {
                const int cap = (MG_SOR_LDS_DOUBLES + SORW_GUARD + SORW_GUARD_BACK) * (int)sizeof(double);
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                lds_attr_mark(2);
            }
            const dim3 grid((unsigned)nbatch), block(256);
            const size_t bytes = (size_t)t_doubles * sizeof(double);
            if (nx - 1 <= 64)
                hipLaunchKernelGGL((mg_sor_wave_kernel<1, true>), grid, block, bytes, st, b, x, r, (int)ny, (int)nx, (int)ld, ldt, h2,
                                   (int)iterations, backward, bstride);
            else
                hipLaunchKernelGGL((mg_sor_wave_kernel<2, true>), grid, block, bytes, st, b, x, r, (int)ny, (int)nx, (int)ld, ldt, h2,
                                   (int)iterations, backward, bstride);
            MG_LAUNCH_CHECK("mg_sor (single-wave wavefront, transposed)");
            return 0;
        }
        // measured 1.6-2.6x ahead of the one-thread-per-row kernel with one or two rows per lane
        if (lds && ny - 1 <= 64 * 4 && ny * ld < ((i64)1 << 27) && (g_sor_single_wave == 2 || (g_sor_single_wave == 1 && ny - 1 <= 128))) {
            const size_t bytes = ((size_t)ny * ldl + SORW_GUARD + SORW_GUARD_BACK) * sizeof(double);
            if (!lds_attr_done(0)) {
                const int cap = (MG_SOR_LDS_DOUBLES + SORW_GUARD + SORW_GUARD_BACK) * (int)sizeof(double);
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
                lds_attr_mark(0);
            }
            const dim3 grid((unsigned)nbatch), block(256);
            if (ny - 1 <= 64)
                hipLaunchKernelGGL(mg_sor_wave_kernel<1>, grid, block, bytes, st, b, x, r, (int)ny, (int)nx, (int)ld, ldl, h2,
                                   (int)iterations, backward, bstride);
            else if (ny - 1 <= 128)
                hipLaunchKernelGGL(mg_sor_wave_kernel<2>, grid, block, bytes, st, b, x, r, (int)ny, (int)nx, (int)ld, ldl, h2,
                                   (int)iterations, backward, bstride);
            else
                hipLaunchKernelGGL(mg_sor_wave_kernel<4>, grid, block, bytes, st, b, x, r, (int)ny, (int)nx, (int)ld, ldl, h2,
                                   (int)iterations, backward, bstride);
            MG_LAUNCH_CHECK("mg_sor (single-wave wavefront)");
            return 0;
        }
        if (lds) {
            const size_t bytes = (size_t)ny * ldl * sizeof(double);
            if (!lds_attr_done(1)) {
                MG_HIP(hipFuncSetAttribute((const void *)mg_sor_wavefront_kernel<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           MG_SOR_LDS_DOUBLES * (int)sizeof(double)));
                lds_attr_mark(1);
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
                hipLaunchKernelGGL(mg_sor_rb_kernel<false>, grid, dim3(256), 0, st, b, x, (i64)1, ny - 1, nx, ld, h2, colour, bstride);
                MG_LAUNCH_CHECK("mg_sor (red-black)");
            }
    }
    if (r) return mg_residual(r, x, b, ny, nx, ld, 1, ny - 1, h, nullptr, nbatch, bstride, 0, stream);
    return 0;
}

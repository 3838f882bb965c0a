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
#include "mg_common.h"

enum { OP_JACOBI = 0, OP_JACOBI_ZERO = 1, OP_RESID = 2, OP_RESID_NORM = 3, OP_LAPLACE = 4 };

template <int OP, int V, bool FINEST>
__global__ __launch_bounds__(256) void mg_stencil5_kernel(const double *__restrict__ b, const double *__restrict__ x,
                                                          double *__restrict__ out, i64 nx, i64 ld, i64 j0, i64 j1,
                                                          int rpb, int ncb, int nrc, double s, i64 bstride,
                                                          u64 *__restrict__ norm_slots)
{
    int cb, rc;
    mg_tile_of_block((int)blockIdx.x, ncb, nrc, cb, rc);
    const i64 boff = (i64)blockIdx.y * bstride;
    if (OP != OP_LAPLACE) b += boff;
    if (OP != OP_JACOBI_ZERO) x += boff;
    out += boff;

    const int lane = threadIdx.x & 63;
    const i64 c0 = ((i64)cb * 256 + threadIdx.x) * V; // first column of this lane
    const i64 cl = c0 + V <= ld ? c0 : ld - V;         // column actually loaded (kept inside the row)
    const i64 rb = j0 + (i64)rc * rpb;
    const i64 re = rb + rpb < j1 ? rb + rpb : j1;
    u64 nmax = 0;

    if (rb < re) { // uniform per workgroup
        if constexpr (OP == OP_JACOBI_ZERO) {
            for (i64 j = rb; j < re; ++j) {
                double bb[V], o[V];
                ldv<V>(bb, b + j * ld + cl);
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = 0.25 * (0.0 - s * bb[k]);
                double *op = out + j * ld + c0;
                if (c0 >= 1 && c0 + V - 1 <= nx - 2) {
                    stv<V>(op, o);
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k)
                        if (c0 + k >= 1 && c0 + k <= nx - 2) op[k] = o[k];
                }
            }
        } else {
            // edge element of the neighbouring wave: lane 0 needs column c0-1, lane 63 column c0+V
            const bool isL = lane == 0, isR = lane == 63;
            const i64 ce = isL ? c0 - 1 : c0 + V;
            const bool pe = (isL && c0 >= 1 && c0 - 1 < nx) || (isR && c0 + V < nx);
            const double *xr = x + (rb - 1) * ld; // row above the current one
            double up[V], cur[V], dn[V];
            ldv<V>(up, xr + cl);
            ldv<V>(cur, xr + ld + cl);
            double ecur = pe ? xr[ld + ce] : 0.0;
            for (i64 j = rb; j < re; ++j) {
                ldv<V>(dn, xr + 2 * ld + cl);
                double edn = pe ? xr[2 * ld + ce] : 0.0;
                double bb[V];
                if constexpr (OP != OP_LAPLACE) ldv<V>(bb, b + j * ld + cl);
                double west = wave_from_prev(cur[V - 1]);
                double east = wave_from_next(cur[0]);
                west = isL ? ecur : west;
                east = isR ? ecur : east;
                double o[V];
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const double E = k + 1 < V ? cur[k + 1 < V ? k + 1 : 0] : east;
                    const double W = k > 0 ? cur[k > 0 ? k - 1 : 0] : west;
                    if constexpr (OP == OP_JACOBI) {
                        o[k] = 0.25 * (E + W + dn[k] + up[k] - s * bb[k]);
                    } else if constexpr (OP == OP_LAPLACE) {
                        o[k] = (W + E + up[k] + dn[k] - 4 * cur[k]) * s;
                    } else {
                        const double left = W * s, cent = cur[k] * s;
                        o[k] = bb[k] - left - (E + up[k] + dn[k]) * s + cent * 4;
                    }
                }
                double *op = out + j * ld + c0;
                if (c0 >= 1 && c0 + V - 1 <= nx - 2) {
                    stv<V>(op, o);
                    if constexpr (OP == OP_RESID_NORM) {
#pragma unroll
                        for (int k = 0; k < V; ++k) {
                            u64 key = (u64)__double_as_longlong(fabs(o[k]));
                            nmax = key > nmax ? key : nmax;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k)
                        if (c0 + k >= 1 && c0 + k <= nx - 2) {
                            op[k] = o[k];
                            if constexpr (OP == OP_RESID_NORM) {
                                u64 key = (u64)__double_as_longlong(fabs(o[k]));
                                nmax = key > nmax ? key : nmax;
                            }
                        }
                }
#pragma unroll
                for (int k = 0; k < V; ++k) up[k] = cur[k], cur[k] = dn[k];
                ecur = edn;
                xr += ld;
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

static int rows_per_block(i64 nrows, int ncb, i64 nbatch)
{
    // Tall tiles amortise the two halo rows (2/rpb extra reads); small grids need short tiles to
    // produce enough workgroups for 256 CUs.
    int rpb = 128;
    while (rpb > 4 && (i64)ncb * ((nrows + rpb - 1) / rpb) * nbatch < 4096) rpb >>= 1;
    return rpb;
}

template <int OP, int V>
static int launch_stencil(const double *b, const double *x, double *out, i64 nx, i64 ld, i64 j0, i64 j1, double s,
                          i64 nbatch, i64 bstride, u64 *slots, int flags, hipStream_t st, const char *name)
{
    const int ncb = (int)((nx - 1 + 256 * V - 1) / (256 * V)); // columns 0 .. nx-2 need a lane
    const int rpb = rows_per_block(j1 - j0, ncb, nbatch);
    const int nrc = (int)((j1 - j0 + rpb - 1) / rpb);
    dim3 grid((unsigned)(ncb * nrc), (unsigned)nbatch);
    if (flags & MG_FLAG_FINEST)
        hipLaunchKernelGGL((mg_stencil5_kernel<OP, V, true>), grid, dim3(256), 0, st, b, x, out, nx, ld, j0, j1, rpb,
                           ncb, nrc, s, bstride, slots);
    else
        hipLaunchKernelGGL((mg_stencil5_kernel<OP, V, false>), grid, dim3(256), 0, st, b, x, out, nx, ld, j0, j1, rpb,
                           ncb, nrc, s, bstride, slots);
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

// the widest vector width the pointers and the leading dimension allow
static int vec_width(i64 ld, i64 bstride, i64 nbatch, const void *p0, const void *p1, const void *p2)
{
    for (int v = 4; v > 1; v >>= 1) {
        bool ok = ld % v == 0 && (nbatch == 1 || bstride % v == 0);
        ok = ok && (!p0 || mg_aligned(p0, 8 * v)) && (!p1 || mg_aligned(p1, 8 * v)) && (!p2 || mg_aligned(p2, 8 * v));
        if (ok) return v;
    }
    return 1;
}

#define DISPATCH_V(OP, ...)                                           \
    switch (v) {                                                      \
    case 4: return launch_stencil<OP, 4>(__VA_ARGS__);                \
    case 2: return launch_stencil<OP, 2>(__VA_ARGS__);                \
    default: return launch_stencil<OP, 1>(__VA_ARGS__);               \
    }

static int g_vec_cap = 4; // tuning/testing knob: upper bound on the vector width

extern "C" int mg_set_vector_width_cap(int cap)
{
    MG_REQUIRE(cap == 1 || cap == 2 || cap == 4, "mg_set_vector_width_cap: %d not in {1,2,4}", cap);
    g_vec_cap = cap;
    return 0;
}

static int capped(int v) { return v < g_vec_cap ? v : g_vec_cap; }

extern "C" int mg_jacobi(const double *b, const double *x, double *x_out, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1,
                         double h, i64 nbatch, i64 bstride, int flags, void *stream)
{
    if (check_grid("mg_jacobi", ny, nx, ld, j0, j1, nbatch, bstride)) return 1;
    MG_REQUIRE(b && x && x_out, "mg_jacobi: null array");
    MG_REQUIRE(x != x_out, "mg_jacobi: x_out aliases x");
    if (j0 == j1) return 0;
    const int v = capped(vec_width(ld, bstride, nbatch, b, x, x_out));
    DISPATCH_V(OP_JACOBI, b, x, x_out, nx, ld, j0, j1, h * h, nbatch, bstride, nullptr, flags, (hipStream_t)stream,
               "mg_jacobi")
}

extern "C" int mg_jacobi_from_zero(const double *b, double *x_out, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1, double h,
                                   i64 nbatch, i64 bstride, int flags, void *stream)
{
    if (check_grid("mg_jacobi_from_zero", ny, nx, ld, j0, j1, nbatch, bstride)) return 1;
    MG_REQUIRE(b && x_out, "mg_jacobi_from_zero: null array");
    if (j0 == j1) return 0;
    const int v = capped(vec_width(ld, bstride, nbatch, b, x_out, nullptr));
    DISPATCH_V(OP_JACOBI_ZERO, b, nullptr, x_out, nx, ld, j0, j1, h * h, nbatch, bstride, nullptr, flags,
               (hipStream_t)stream, "mg_jacobi_from_zero")
}

extern "C" int mg_residual(double *r, const double *x, const double *b, i64 ny, i64 nx, i64 ld, i64 j0, i64 j1,
                           double h, uint64_t *norm_slots, i64 nbatch, i64 bstride, int flags, void *stream)
{
    if (check_grid("mg_residual", ny, nx, ld, j0, j1, nbatch, bstride)) return 1;
    MG_REQUIRE(r && x && b, "mg_residual: null array");
    MG_REQUIRE(r != x, "mg_residual: r aliases x");
    if (j0 == j1) return 0;
    const int v = capped(vec_width(ld, bstride, nbatch, b, x, r));
    const double hh = 1 / (h * h);
    if (norm_slots) {
        DISPATCH_V(OP_RESID_NORM, b, x, r, nx, ld, j0, j1, hh, nbatch, bstride, (u64 *)norm_slots, flags,
                   (hipStream_t)stream, "mg_residual")
    }
    DISPATCH_V(OP_RESID, b, x, r, nx, ld, j0, j1, hh, nbatch, bstride, nullptr, flags, (hipStream_t)stream,
               "mg_residual")
}

extern "C" int mg_laplacian(const double *x, double *lplc, i64 ny, i64 nx, i64 ld, double h, void *stream)
{
    if (check_grid("mg_laplacian", ny, nx, ld, 1, ny - 1, 1, 0)) return 1;
    MG_REQUIRE(x && lplc && x != lplc, "mg_laplacian: null or aliased array");
    const int v = capped(vec_width(ld, 0, 1, x, lplc, nullptr));
    DISPATCH_V(OP_LAPLACE, nullptr, x, lplc, nx, ld, 1, ny - 1, 1 / (h * h), 1, 0, nullptr, 0, (hipStream_t)stream,
               "mg_laplacian")
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

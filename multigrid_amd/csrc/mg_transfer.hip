// Grid-transfer kernels of the V-cycle on gfx950: full-weighting restriction, bilinear
// prolongation + correction, 2-D block copies (split / stitch).
//
// Both transfer kernels stream the fine array exactly once (restriction: 8 B read per fine point
// + 2 B written; prolongation: 8 B read + 8 B written per fine point + 2 B of coarse data):
// a wave marches down the rows, each lane owns one coarse column (= two fine columns, one
// aligned 16-byte access), rows are kept in registers while they are still needed, and values
// owned by the neighbouring lane arrive by a whole-wave DPP shift; only lane 0 (restriction) or
// lane 63 (prolongation) loads one extra element per row.
//
// Arithmetic follows the reference's expression order (see include/mg_hip.h); the file is
// compiled with -ffp-contract=off.
#include <atomic>

#include "mg_common.h"

// ------------------------------------------------------------------------------- restriction

// PAIR: the two fine columns (2I+oi, 2I+oi+1) of a lane are loaded as one aligned double2
template <bool PAIR>
__global__ __launch_bounds__(256) void mg_restrict_kernel(const double *__restrict__ f, i64 fld, double *__restrict__ c,
                                                          i64 cnx, i64 cld, i64 J0, i64 J1, int rpb, int ncb, int nrc,
                                                          int oi, int oj, i64 fbs, i64 cbs)
{
    int cb, rc;
    mg_tile_of_block((int)blockIdx.x, ncb, nrc, cb, rc);
    f += (i64)blockIdx.y * fbs;
    c += (i64)blockIdx.y * cbs;
    const int lane = threadIdx.x & 63;
    const i64 I = (i64)cb * 256 + threadIdx.x; // coarse column of this lane
    const i64 Icl = I < 1 ? 1 : (I > cnx - 2 ? cnx - 2 : I);
    const i64 fc = 2 * Icl + oi; // fine centre column (lanes outside 1..cnx-2 load a valid column, unused)
    // lanes that cannot get column fc-1 from the lane before them load it themselves
    const bool isL = (lane == 0 || I == 1) && I >= 1 && I <= cnx - 2;
    const i64 Jb = J0 + (i64)rc * rpb;
    const i64 Je = Jb + rpb < J1 ? Jb + rpb : J1;
    if (Jb >= Je) return; // uniform per workgroup

    auto load = [&](const double *row, double &ctr, double &rgt, double &lft) {
        if constexpr (PAIR) {
            double2 t = *reinterpret_cast<const double2 *>(row + fc);
            ctr = t.x, rgt = t.y;
        } else {
            ctr = row[fc], rgt = row[fc + 1];
        }
        lft = isL ? row[fc - 1] : 0.0;
    };

    const double *row = f + (2 * Jb + oj - 1) * fld;
    double mc, mr, ml, zc, zr, zl, pc, pr, pl;
    load(row, mc, mr, ml);
    for (i64 J = Jb; J < Je; ++J) {
        load(row + fld, zc, zr, zl);
        load(row + 2 * fld, pc, pr, pl);
        const double a_right = zr / 8 + (pr + mr) / 16;
        double a_left = wave_from_prev(a_right);
        if (isL) a_left = zl / 8 + (pl + ml) / 16;
        double v = zc / 4 + (pc + mc) * 1 / 8;
        v += a_right + a_left;
        if (I >= 1 && I <= cnx - 2) c[J * cld + I] = v;
        mc = pc, mr = pr, ml = pl;
        row += 2 * fld;
    }
}

extern "C" int mg_restrict(const double *fine, i64 fny, i64 fnx, i64 fld, double *crs, i64 cny, i64 cnx, i64 cld,
                           i64 J0, i64 J1, i64 ofst_i, i64 ofst_j, i64 nbatch, i64 fbstride, i64 cbstride,
                           void *stream)
{
    MG_REQUIRE(fine && crs, "mg_restrict: null array");
    MG_REQUIRE((ofst_i == 0 || ofst_i == -1) && ofst_j <= 0, "mg_restrict: ofst_i must be 0 or -1, ofst_j <= 0");
    MG_REQUIRE(cny >= 3 && cnx >= 3 && cld >= cnx && fld >= fnx, "mg_restrict: bad shapes");
    MG_REQUIRE(J0 >= 1 && J1 <= cny - 1 && J0 <= J1, "mg_restrict: coarse row range [%lld, %lld) outside 1..%lld",
               (long long)J0, (long long)J1, (long long)(cny - 1));
    MG_REQUIRE(2 * (cnx - 2) + ofst_i + 1 <= fnx - 1, "mg_restrict: fine grid has %lld columns, %lld needed",
               (long long)fnx, (long long)(2 * (cnx - 2) + ofst_i + 2));
    MG_REQUIRE(J0 == J1 || (2 * (J1 - 1) + ofst_j + 1 <= fny - 1 && 2 * J0 + ofst_j - 1 >= 0),
               "mg_restrict: fine grid has %lld rows, rows %lld..%lld needed", (long long)fny,
               (long long)(2 * J0 + ofst_j - 1), (long long)(2 * (J1 - 1) + ofst_j + 1));
    MG_REQUIRE(nbatch >= 1 && nbatch <= 65535, "mg_restrict: nbatch %lld outside 1..65535", (long long)nbatch);
    if (J0 == J1) return 0;
    const int ncb = (int)((cnx - 1 + 255) / 256);
    int rpb = 64;
    while (rpb > 2 && (i64)ncb * ((J1 - J0 + rpb - 1) / rpb) * nbatch < 4096) rpb >>= 1;
    const int nrc = (int)((J1 - J0 + rpb - 1) / rpb);
    dim3 grid((unsigned)(ncb * nrc), (unsigned)nbatch);
    const bool pair = ofst_i == 0 && fld % 2 == 0 && mg_aligned(fine, 16) && (nbatch == 1 || fbstride % 2 == 0);
    if (pair)
        hipLaunchKernelGGL(mg_restrict_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, fine, fld, crs, cnx, cld,
                           J0, J1, rpb, ncb, nrc, (int)ofst_i, (int)ofst_j, fbstride, cbstride);
    else
        hipLaunchKernelGGL(mg_restrict_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, fine, fld, crs, cnx,
                           cld, J0, J1, rpb, ncb, nrc, (int)ofst_i, (int)ofst_j, fbstride, cbstride);
    MG_LAUNCH_CHECK("mg_restrict");
    return 0;
}

// ------------------------------------------------------------------------------- prolongation

// A lane owns fine columns (c, c+1), c even, and coarse columns q = c/2 and q+1.
// OI0 (ofst_i == 0):  column c is "even" (on coarse column q), c+1 is "odd" (between q and q+1).
// !OI0 (ofst_i == -1): column c is "odd" (between q and q+1), c+1 is "even" (on coarse column q+1).
template <bool PAIR, bool OI0>
__global__ __launch_bounds__(256) void mg_prolong_kernel(const double *__restrict__ a, i64 cny, i64 cnx, i64 cld,
                                                         double *__restrict__ fnew, i64 fld, i64 t0, i64 t1, i64 c_lo,
                                                         i64 c_hi, int rpb, int ncb, int nrc, int oj, i64 cbs, i64 fbs)
{
    int cb, rc;
    mg_tile_of_block((int)blockIdx.x, ncb, nrc, cb, rc);
    a += (i64)blockIdx.y * cbs;
    fnew += (i64)blockIdx.y * fbs;
    const int lane = threadIdx.x & 63;
    const i64 q = (i64)cb * 256 + threadIdx.x;
    const i64 c = 2 * q;
    const i64 qa = q > cnx - 1 ? cnx - 1 : q;         // clamped coarse columns (values unused when clamped)
    const i64 qb = q + 1 > cnx - 1 ? cnx - 1 : q + 1;
    const bool isR = lane == 63;
    const bool w0 = c >= c_lo && c <= c_hi, w1 = c + 1 >= c_lo && c + 1 <= c_hi;
    const i64 tb = t0 + (i64)rc * rpb;
    const i64 te = tb + rpb < t1 ? tb + rpb : t1;
    if (tb >= te) return; // uniform per workgroup

    auto load = [&](i64 J, double &v0, double &v1) {
        const double *row = a + (J > cny - 1 ? cny - 1 : J) * cld;
        v0 = row[qa];
        v1 = wave_from_next(v0);
        if (isR) v1 = row[qb];
    };

    i64 J = (tb - oj) >> 1;
    double a00, a01, a10, a11; // a[J][q], a[J][q+1], a[J+1][q], a[J+1][q+1]
    load(J, a00, a01);
    load(J + 1, a10, a11);
    for (i64 t = tb; t < te; ++t) {
        const i64 rel = t - oj;
        if ((rel >> 1) != J) { // uniform: moved on to the next coarse row
            J = rel >> 1;
            a00 = a10, a01 = a11;
            load(J + 1, a10, a11);
        }
        double add0, add1;
        if (rel & 1) { // between coarse rows J and J+1
            const double ctr = (a11 + a10 + a01 + a00) / 4;
            if constexpr (OI0) add0 = (a00 + a10) / 2, add1 = ctr;
            else add0 = ctr, add1 = (a01 + a11) / 2;
        } else { // on coarse row J
            const double mid = (a00 + a01) / 2;
            if constexpr (OI0) add0 = a00, add1 = mid;
            else add0 = mid, add1 = a01;
        }
        double *fp = fnew + t * fld + c;
        if (w0 && w1) {
            if constexpr (PAIR) {
                double2 v = *reinterpret_cast<double2 *>(fp);
                v.x += add0, v.y += add1;
                *reinterpret_cast<double2 *>(fp) = v;
            } else {
                fp[0] += add0, fp[1] += add1;
            }
        } else if (w0) {
            fp[0] += add0;
        } else if (w1) {
            fp[1] += add1;
        }
    }
}

extern "C" int mg_prolong_add(const double *crs, i64 cny, i64 cnx, i64 cld, double *fine, i64 fny, i64 fnx, i64 fld,
                              i64 t0, i64 t1, i64 c_lo, i64 c_hi, i64 ofst_i, i64 ofst_j, i64 nbatch, i64 cbstride,
                              i64 fbstride, void *stream)
{
    MG_REQUIRE(crs && fine, "mg_prolong_add: null array");
    MG_REQUIRE((ofst_i == 0 || ofst_i == -1) && ofst_j <= 0, "mg_prolong_add: ofst_i must be 0 or -1, ofst_j <= 0");
    MG_REQUIRE(cny >= 2 && cnx >= 2 && cld >= cnx && fld >= fnx && fny >= 1 && fnx >= 1, "mg_prolong_add: bad shapes");
    MG_REQUIRE(nbatch >= 1 && nbatch <= 65535, "mg_prolong_add: nbatch %lld outside 1..65535", (long long)nbatch);
    // the cells the reference touches: 1 <= t - ofst_j <= 2*cny-3 and 1 <= c - ofst_i <= 2*cnx-3
    // With the reference's offsets (0 / -1) every cell its four passes touch must exist; with a deeper
    // row offset (slabs with several halo rows) the row range is simply clipped to the fine array.
    MG_REQUIRE((ofst_j < -1 || 2 * cny - 3 + ofst_j <= fny - 1) && 2 * cnx - 3 + ofst_i <= fnx - 1,
               "mg_prolong_add: fine grid %lld x %lld too small for coarse %lld x %lld", (long long)fny, (long long)fnx,
               (long long)cny, (long long)cnx);
    if (t0 < 0) t0 = 0;
    if (t1 > fny) t1 = fny;
    if (t0 < 1 + ofst_j) t0 = 1 + ofst_j;
    if (t1 > 2 * cny - 3 + ofst_j + 1) t1 = 2 * cny - 3 + ofst_j + 1;
    if (c_lo < 1 + ofst_i) c_lo = 1 + ofst_i;
    if (c_hi > 2 * cnx - 3 + ofst_i) c_hi = 2 * cnx - 3 + ofst_i;
    if (t0 >= t1 || c_lo > c_hi) return 0;
    const int ncb = (int)((c_hi / 2 + 1 + 255) / 256);
    int rpb = 128;
    while (rpb > 4 && (i64)ncb * ((t1 - t0 + rpb - 1) / rpb) * nbatch < 4096) rpb >>= 1;
    const int nrc = (int)((t1 - t0 + rpb - 1) / rpb);
    dim3 grid((unsigned)(ncb * nrc), (unsigned)nbatch);
    const bool pair = fld % 2 == 0 && mg_aligned(fine, 16) && (nbatch == 1 || fbstride % 2 == 0);
    hipStream_t st = (hipStream_t)stream;
#define PL(P, O)                                                                                                     \
    hipLaunchKernelGGL((mg_prolong_kernel<P, O>), grid, dim3(256), 0, st, crs, cny, cnx, cld, fine, fld, t0, t1, c_lo, \
                       c_hi, rpb, ncb, nrc, (int)ofst_j, cbstride, fbstride)
    if (pair && ofst_i == 0) PL(true, true);
    else if (pair) PL(true, false);
    else if (ofst_i == 0) PL(false, true);
    else PL(false, false);
#undef PL
    MG_LAUNCH_CHECK("mg_prolong_add");
    return 0;
}

// ------------------------------------------------------------------------------- block copies

__global__ __launch_bounds__(256) void mg_copy2d_kernel(double *__restrict__ dst, i64 dld, const double *__restrict__ src,
                                                        i64 sld, i64 ny, i64 nx)
{
    for (i64 j = blockIdx.y; j < ny; j += gridDim.y)
        for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nx; i += (i64)gridDim.x * 256)
            dst[j * dld + i] = src[j * sld + i];
}

extern "C" int mg_copy2d(double *dst, i64 dld, const double *src, i64 sld, i64 ny, i64 nx, void *stream)
{
    MG_REQUIRE(dst && src, "mg_copy2d: null array");
    MG_REQUIRE(ny >= 0 && nx >= 0 && dld >= nx && sld >= nx, "mg_copy2d: bad shape");
    if (ny == 0 || nx == 0) return 0;
    unsigned gx = (unsigned)((nx + 255) / 256);
    if (gx > 256) gx = 256;
    unsigned gy = (unsigned)(ny < 4096 ? ny : 4096);
    hipLaunchKernelGGL(mg_copy2d_kernel, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, dst, dld, src, sld, ny, nx);
    MG_LAUNCH_CHECK("mg_copy2d");
    return 0;
}

// ------------------------------------------------------------------------------- 2x2 block halo exchange

// The reference's halo exchange for its four block ranks, executed for all four at once
// (Buffers.exchange_buffers, core/parallel_multigrid.py:19-97, width = 1, overlap = 1).
// Directions: 0 east, 1 west, 2 north (+rows), 3 south.  Every rank first packs the line just
// inside the shared centre line (:39-42), then unpacks into its outermost line (:44-47), east/west
// before north/south as its `neighbours` dict is ordered (:49-56) -- the order decides which
// message a halo corner cell ends up with.  One workgroup: the three phases are separated by
// workgroup barriers.
__global__ __launch_bounds__(1024) void mg_block_exchange_kernel(double *v, i64 m, i64 ld, i64 bs, double *stage)
{
    const int dirs[4][2] = {{0, 2}, {1, 2}, {0, 3}, {1, 3}};
    const int peers[4][2] = {{1, 2}, {0, 3}, {3, 0}, {2, 1}};
    for (i64 idx = threadIdx.x; idx < 8 * m; idx += blockDim.x) {
        const int rk = (int)(idx / (2 * m)), q = (int)((idx / m) & 1);
        const i64 k = idx % m;
        const double *a = v + rk * bs;
        const int d = dirs[rk][q];
        stage[idx] = d == 0 ? a[k * ld + (m - 3)] : d == 1 ? a[k * ld + 2] : d == 2 ? a[(m - 3) * ld + k] : a[2 * ld + k];
    }
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
        for (i64 idx = threadIdx.x; idx < 4 * m; idx += blockDim.x) {
            const int rk = (int)(idx / m);
            const i64 k = idx % m;
            double *a = v + rk * bs;
            const int d = dirs[rk][q];
            const double val = stage[(peers[rk][q] * 2 + q) * m + k]; // the peer's message of the same axis
            if (d == 0) a[k * ld + (m - 1)] = val;
            else if (d == 1) a[k * ld] = val;
            else if (d == 2) a[(m - 1) * ld + k] = val;
            else a[k] = val;
        }
    }
}

extern "C" int mg_block_exchange(double *v, i64 m, i64 ld, i64 bstride, double *stage, void *stream)
{
    MG_REQUIRE(v && stage, "mg_block_exchange: null array");
    MG_REQUIRE(m >= 4 && ld >= m && bstride >= m * ld, "mg_block_exchange: bad shape (side %lld, ld %lld, stride %lld)",
               (long long)m, (long long)ld, (long long)bstride);
    hipLaunchKernelGGL(mg_block_exchange_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, m, ld, bstride, stage);
    MG_LAUNCH_CHECK("mg_block_exchange");
    return 0;
}

// The same exchange for callers that cannot allocate or bring a staging buffer: the 8*m doubles of
// staging live in a file-scope device array.  Every GPU holds its own copy of the array at its own
// address; that address is looked up once per device (the array's name is a host-side stand-in, not
// a device address, and is never passed to a kernel).
#define MG_BLOCK_EXCHANGE_STATIC_DOUBLES (8 * MG_BLOCK_EXCHANGE_STATIC_MAX_SIDE)
#define MG_MAX_DEVICES 64
static __device__ __align__(256) double scratch[MG_BLOCK_EXCHANGE_STATIC_DOUBLES];
static std::atomic<double *> g_scratch_ptr[MG_MAX_DEVICES];

static int mg_static_stage(double **out)
{
    int dev = -1;
    MG_HIP(hipGetDevice(&dev));
    MG_REQUIRE(dev >= 0 && dev < MG_MAX_DEVICES, "mg_block_exchange_static: device %d not supported (at most %d devices)",
               dev, MG_MAX_DEVICES);
    double *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        // Two host threads may both get here: both obtain the same address, so the double store is harmless.
        MG_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        MG_REQUIRE(p, "mg_block_exchange_static: the staging array has no address on device %d", dev);
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return 0;
}

extern "C" int mg_block_exchange_static(double *v, i64 m, i64 ld, i64 bstride, void *stream)
{
    MG_REQUIRE(v, "mg_block_exchange_static: null array");
    MG_REQUIRE(m >= 4 && ld >= m && bstride >= m * ld, "mg_block_exchange_static: bad shape (side %lld, ld %lld, stride %lld)",
               (long long)m, (long long)ld, (long long)bstride);
    MG_REQUIRE(m <= MG_BLOCK_EXCHANGE_STATIC_MAX_SIDE,
               "mg_block_exchange_static: block side %lld exceeds the built-in staging array (at most %d); "
               "use mg_block_exchange with a caller-provided stage",
               (long long)m, MG_BLOCK_EXCHANGE_STATIC_MAX_SIDE);
    double *stage = nullptr;
    if (int rc = mg_static_stage(&stage)) return rc;
    // Nothing to clear: the kernel writes all 8*m staging values before it reads any of them.
    hipLaunchKernelGGL(mg_block_exchange_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v, m, ld, bstride, stage);
    MG_LAUNCH_CHECK("mg_block_exchange_static");
    return 0;
}

extern "C" int mg_zero(double *a, i64 n, void *stream)
{
    MG_REQUIRE(a && n >= 0, "mg_zero: bad argument");
    if (n) MG_HIP(hipMemsetAsync(a, 0, (size_t)n * sizeof(double), (hipStream_t)stream));
    return 0;
}

extern "C" int mg_split(const double *a_in, i64 in_ny, i64 in_nx, i64 in_ld, double *a_out, i64 nout, i64 out_ld,
                        i64 rank, void *stream)
{
    MG_REQUIRE(a_in && a_out, "mg_split: null array");
    MG_REQUIRE(nout >= 1 && nout <= in_ny && nout <= in_nx && in_ld >= in_nx && out_ld >= nout, "mg_split: bad shapes");
    if (rank < 0 || rank > 3) return 0; // the reference copies nothing for other ranks
    const i64 j0 = (rank == 2 || rank == 3) ? in_ny - nout : 0;
    const i64 i0 = (rank == 1 || rank == 3) ? in_nx - nout : 0;
    return mg_copy2d(a_out, out_ld, a_in + j0 * in_ld + i0, in_ld, nout, nout, stream);
}

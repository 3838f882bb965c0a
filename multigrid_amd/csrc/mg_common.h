// Shared helpers of libmg_hip.so (gfx950 only): error reporting across the C ABI, argument
// checks, whole-wave lane shifts, and the workgroup -> tile mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mg_hip.h"

typedef int64_t i64;
typedef unsigned long long u64;

void mg_set_error(const char *fmt, ...);

#define MG_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            mg_set_error(__VA_ARGS__); \
            return 1;                \
        }                            \
    } while (0)

#define MG_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) {                                                   \
            mg_set_error("%s failed: %s", #call, hipGetErrorString(e_));          \
            return 2;                                                             \
        }                                                                         \
    } while (0)

#define MG_LAUNCH_CHECK(name)                                                     \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            mg_set_error("launch of %s failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

static inline bool mg_aligned(const void *p, size_t a) { return ((uintptr_t)p % a) == 0; }

// ---- whole-wave shifts by one lane (wave = 64 lanes), DPP wave_shr:1 / wave_shl:1 ----------
// wave_from_prev(v): lane l receives v of lane l-1 (lane 0 keeps its own v).
// wave_from_next(v): lane l receives v of lane l+1 (lane 63 keeps its own v).
// All 64 lanes must be active at the call.
__device__ __forceinline__ double wave_from_prev(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_from_next(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// ---- vector loads/stores of V consecutive doubles (V = 1, 2, 4) ----------------------------
template <int V>
__device__ __forceinline__ void ldv(double (&d)[V], const double *p)
{
    if constexpr (V == 1) {
        d[0] = *p;
    } else if constexpr (V == 2) {
        double2 t = *reinterpret_cast<const double2 *>(p);
        d[0] = t.x, d[1] = t.y;
    } else {
        double2 t0 = reinterpret_cast<const double2 *>(p)[0];
        double2 t1 = reinterpret_cast<const double2 *>(p)[1];
        d[0] = t0.x, d[1] = t0.y, d[2] = t1.x, d[3] = t1.y;
    }
}
template <int V>
__device__ __forceinline__ void stv(double *p, const double (&d)[V])
{
    if constexpr (V == 1) {
        *p = d[0];
    } else if constexpr (V == 2) {
        *reinterpret_cast<double2 *>(p) = make_double2(d[0], d[1]);
    } else {
        reinterpret_cast<double2 *>(p)[0] = make_double2(d[0], d[1]);
        reinterpret_cast<double2 *>(p)[1] = make_double2(d[2], d[3]);
    }
}

// ---- streaming stores ------------------------------------------------------------------------
// An output array much larger than the caches (L2: 4 MB per XCD, Infinity Cache: 256 MB) gains nothing
// from being kept in them, and the lines it occupies there displace the input rows that neighbouring
// tiles are about to re-read.  Such arrays are written with the non-temporal hint (`nt`: the line is
// marked for early eviction).  Measured at 32769 x 32769: fused finest pass 5.7 -> 5.3 ms, Jacobi sweep
// 4.49 -> 4.32 ms.  The host chooses per launch, from the size of the array.
extern i64 g_stream_min_bytes; // arrays of at least this many bytes are written with streaming stores
static inline int mg_streams(i64 bytes) { return bytes >= g_stream_min_bytes; }

typedef double mg_dvec2 __attribute__((ext_vector_type(2)));
// ST is a compile-time choice: given a run-time one the compiler folds both kinds of store into a plain store.
template <bool ST>
__device__ __forceinline__ void store_pair(double *p, double v0, double v1)
{
    mg_dvec2 v = {v0, v1};
    if constexpr (ST)
        __builtin_nontemporal_store(v, reinterpret_cast<mg_dvec2 *>(p));
    else
        *reinterpret_cast<mg_dvec2 *>(p) = v;
}
template <bool ST>
__device__ __forceinline__ void store_one(double *p, double v)
{
    if constexpr (ST)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
template <int V, bool ST>
__device__ __forceinline__ void stv_stream(double *p, const double (&d)[V])
{
    if constexpr (V == 1) {
        store_one<ST>(p, d[0]);
    } else if constexpr (V == 2) {
        store_pair<ST>(p, d[0], d[1]);
    } else {
        store_pair<ST>(p, d[0], d[1]);
        store_pair<ST>(p + 2, d[2], d[3]);
    }
}

// ---- workgroup id -> (column block, row chunk) ----------------------------------------------
// Workgroups b and b+8 are observed to share an XCD (and its L2).  The map gives each of the 8
// groups a contiguous band of column blocks (or of row chunks), so that tiles that share edge
// cache lines or halo rows are fetched through the same L2.  It is a bijection for every
// (ncb, nrc); it affects speed only.
__device__ __forceinline__ void mg_tile_of_block(int bid, int ncb, int nrc, int &cb, int &rc, int stagger = 0)
{
    if ((ncb & 7) == 0) {
        int x = bid & 7, w = bid >> 3, cpx = ncb >> 3;
        cb = x * cpx + (w % cpx);
        rc = w / cpx;
    } else if ((nrc & 7) == 0) {
        int x = bid & 7, w = bid >> 3, rpx = nrc >> 3;
        rc = x * rpx + (w / ncb);
        cb = w % ncb;
        // stagger: the eight bands start at different columns, so that what the eight XCDs touch at one moment
        // differs in the low address bits too, not only by multiples of the band's size
        if (stagger) cb = (cb + x * stagger) % ncb;
    } else {
        cb = bid % ncb;
        rc = bid / ncb;
    }
}

// wave-level max of 64-bit unsigned keys, result valid in every lane
__device__ __forceinline__ u64 wave_max_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        u64 o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// ---- placement selection of the finest level's arrays (mg_solver.hip; used by both solver handles) ------------------
// The shapes and row geometry of the finest pass as the caller's cycle launches it, and of level 1's 9-stage pass.
struct MgPlaceShape {
    hipStream_t stream;
    i64 ny, nx, ld;                    // a finest-level array
    i64 cny, cnx, cld;                 // the array the finest pass restricts into (the correction it reads has that shape too)
    i64 j0, j1, flo, fhi, oj, T0, T1;  // rows the pass produces, Dirichlet rows, row offset of the transfers, coarse rows
    double h0;
    bool level1;                       // also choose level 1's x (it runs the 9-stage pass with recomputed pre-smoothing)
    i64 c2ny, c2nx, c2ld;              // level 2's x, which that pass prolongs
    i64 j0_1, j1_1, flo_1, fhi_1;      // its row geometry on level 1
};
// fin[] = {xa, xb, b, r} of the finest level, *cb_out = the array it restricts into, *xa1_out = level 1's x: all zeroed, or
// all null when the level is small (< 1 GiB per array), memory is short or anything fails.  *candidates_out: candidates timed.
void mg_place_finest(const MgPlaceShape &P, double *fin[4], double **cb_out, double **xa1_out, int *candidates_out);

// Does the physical layout behind the array the finest pass writes decide its rate?  The pass (32769 x 32769) is
// timed with x_out = (a) plain hipMalloc allocations, the process's first, (b) virtual ranges backed by physical
// chunks of a given size created one by one (hipMemCreate) and mapped in order, (c) the same mapped in a permuted
// order.  Inputs are plain allocations.
// Build: hipcc -O2 -o tools/vmm_probe tools/vmm_probe.cpp -Iinclude -Lmultigrid_amd -lmg_hip -Wl,-rpath,'$ORIGIN/../multigrid_amd'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "mg_hip.h"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));              \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static const int64_t N = 32769, LD = 32784, CN = 16385, CLD = 16400;
static double *b, *xin, *crs, *crs_b;
static uint64_t *nrm_slots;
static hipStream_t st;

static float time_out(double *xout)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        if (mg_smooth_fused_restrict(b, xin, xout, nullptr, crs, N, N, LD, CN, CN, CLD, 1, N - 1, 0, N - 1, 0, 6, 20.0 / 32768, 1.0,
                                     nrm_slots, 0, crs_b, CN, CN, CLD, 1, CN - 1, 1, 0, 0, 0, st)) {
            printf("launch failed: %s\n", mg_last_error());
            exit(1);
        }
        if (rep == 0) CK(hipEventRecord(e0, st));
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 2;
}

static double *vmm_array(size_t bytes, size_t chunk, int order /*0 in order, 1 reversed, 2 scrambled*/, std::vector<hipMemGenericAllocationHandle_t> &hs)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, total, gran, nullptr, 0));
    std::vector<size_t> slot(n);
    for (size_t k = 0; k < n; ++k) slot[k] = k;
    if (order == 1) std::reverse(slot.begin(), slot.end());
    if (order == 2)
        for (size_t k = 0; k < n; ++k) slot[k] = (k * 2654435761u + 12345) % n; // may collide: fix up below
    if (order == 2) { // a permutation: multiply by a stride coprime to n
        size_t s = n / 2 + 1;
        while (std::__gcd(s, n) != 1) ++s;
        for (size_t k = 0; k < n; ++k) slot[k] = (k * s) % n;
    }
    for (size_t k = 0; k < n; ++k) { // physical chunk k (created in this order) backs virtual slot slot[k]
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)va + slot[k] * chunk, chunk, 0, h, 0));
        hs.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    CK(hipMemsetAsync(va, 0, bytes, st));
    return (double *)va;
}

// `vmm_probe first`: the hand-mapped candidates come FIRST (small chunks, scrambled; then in order), the plain
// allocations after them: separates the chunking from the position in the process's sequence of allocations.
static int chunks_first()
{
    const size_t bytes = (size_t)N * LD * 8, cbytes = (size_t)CN * CLD * 8, MiB = (size_t)1 << 20;
    CK(hipStreamCreate(&st));
    CK(hipMalloc((void **)&b, bytes));
    CK(hipMalloc((void **)&xin, bytes));
    CK(hipMalloc((void **)&crs, cbytes));
    CK(hipMalloc((void **)&crs_b, cbytes));
    CK(hipMalloc((void **)&nrm_slots, 64 * 8));
    CK(hipMemsetAsync(b, 0, bytes, st));
    CK(hipMemsetAsync(xin, 0, bytes, st));
    CK(hipMemsetAsync(crs, 0, cbytes, st));
    CK(hipMemsetAsync(crs_b, 0, cbytes, st));
    CK(hipMemsetAsync(nrm_slots, 0, 64 * 8, st));
    const struct { size_t chunk; int order; } seq[] = {{2 * MiB, 2}, {2 * MiB, 0}, {64 * MiB, 2}, {0, 0}, {0, 0}, {2 * MiB, 2}, {0, 0}, {0, 0},
                                                       {0, 0}, {0, 0}, {0, 0}, {0, 0}, {2 * MiB, 2}, {0, 0}};
    for (auto q : seq) {
        if (q.chunk) {
            std::vector<hipMemGenericAllocationHandle_t> hs;
            double *p = vmm_array(bytes, q.chunk, q.order, hs);
            printf("hipMemCreate chunks of %6zu MiB, %s as x_out: %.3f ms\n", q.chunk / MiB,
                   q.order == 0 ? "mapped in order" : "mapped scrambled", time_out(p));
        } else {
            double *p;
            CK(hipMalloc((void **)&p, bytes));
            CK(hipMemsetAsync(p, 0, bytes, st));
            printf("hipMalloc as x_out: %.3f ms\n", time_out(p));
        }
        fflush(stdout);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'f') return chunks_first();
    const size_t bytes = (size_t)N * LD * 8, cbytes = (size_t)CN * CLD * 8;
    CK(hipStreamCreate(&st));
    double *plain[3];
    CK(hipMalloc((void **)&b, bytes));
    CK(hipMalloc((void **)&xin, bytes));
    for (auto &p : plain) CK(hipMalloc((void **)&p, bytes));
    CK(hipMalloc((void **)&crs, cbytes));
    CK(hipMalloc((void **)&crs_b, cbytes));
    CK(hipMalloc((void **)&nrm_slots, 64 * 8));
    CK(hipMemsetAsync(b, 0, bytes, st));
    CK(hipMemsetAsync(xin, 0, bytes, st));
    for (auto p : plain) CK(hipMemsetAsync(p, 0, bytes, st));
    CK(hipMemsetAsync(crs, 0, cbytes, st));
    CK(hipMemsetAsync(crs_b, 0, cbytes, st));
    CK(hipMemsetAsync(nrm_slots, 0, 64 * 8, st));
    for (int k = 0; k < 3; ++k) printf("hipMalloc #%d as x_out: %.3f ms\n", k + 2, time_out(plain[k]));
    fflush(stdout);
    const size_t MiB = (size_t)1 << 20;
    const size_t chunks[] = {bytes, 1024 * MiB, 64 * MiB, 2 * MiB};
    for (size_t c : chunks)
        for (int order = 0; order < (c == bytes ? 1 : 3); ++order) {
            std::vector<hipMemGenericAllocationHandle_t> hs;
            hipEvent_t t0, t1;
            CK(hipEventCreate(&t0));
            CK(hipEventCreate(&t1));
            double *p = vmm_array(bytes, c, order, hs);
            printf("hipMemCreate chunks of %6zu MiB, %s (%zu chunks) as x_out: %.3f ms\n", c / MiB,
                   order == 0 ? "mapped in order" : order == 1 ? "mapped in reverse" : "mapped scrambled", hs.size(), time_out(p));
            fflush(stdout);
            // left mapped: later candidates get other physical memory
        }
    // plain allocations made now, after ~100 GiB of others
    for (int k = 0; k < 3; ++k) {
        double *p;
        CK(hipMalloc((void **)&p, bytes));
        CK(hipMemsetAsync(p, 0, bytes, st));
        printf("late hipMalloc #%d as x_out: %.3f ms\n", k, time_out(p));
    }
    return 0;
}

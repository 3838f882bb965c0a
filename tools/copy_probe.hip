// Streaming-bandwidth probe for MI355X: what do plain copy / 2-read-1-write kernels reach with
// (a) linear grid-stride access and (b) the column-strip row-marching access of the stencil kernels?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/copy_probe tools/copy_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef long long i64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void copy_gs(const double2* __restrict__ a, double2* __restrict__ o, i64 n2) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n2; i += (i64)gridDim.x * 256) {
    if (NT) { double2 v; v.x = __builtin_nontemporal_load(&a[i].x); v.y = __builtin_nontemporal_load(&a[i].y);
              __builtin_nontemporal_store(v.x, &o[i].x); __builtin_nontemporal_store(v.y, &o[i].y); }
    else o[i] = a[i];
  }
}
// each block copies a contiguous chunk (block-linear), 4 loads in flight per thread
__global__ __launch_bounds__(256) void copy_chunk(const double2* __restrict__ a, double2* __restrict__ o, i64 n2, i64 per_block) {
  i64 b0 = (i64)blockIdx.x * per_block, b1 = b0 + per_block < n2 ? b0 + per_block : n2;
  for (i64 i = b0 + threadIdx.x; i < b1; i += 1024) {
    double2 v0 = a[i], v1 = i + 256 < b1 ? a[i + 256] : v0, v2 = i + 512 < b1 ? a[i + 512] : v0, v3 = i + 768 < b1 ? a[i + 768] : v0;
    o[i] = v0; if (i + 256 < b1) o[i + 256] = v1; if (i + 512 < b1) o[i + 512] = v2; if (i + 768 < b1) o[i + 768] = v3;
  }
}
__global__ __launch_bounds__(256) void add_gs(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ o, i64 n2) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n2; i += (i64)gridDim.x * 256) {
    double2 x = a[i], y = b[i]; o[i] = make_double2(x.x + y.x, x.y + y.y);
  }
}
// no loop: one workgroup per 256, 512 (two adjacent 16-B accesses per thread) or 1024 (four accesses, 4 KB apart) pairs
__global__ __launch_bounds__(256) void add_flat1(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ o, i64 n2) {
  i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) { double2 x = a[i], y = b[i]; o[i] = make_double2(x.x + y.x, x.y + y.y); }
}
__global__ __launch_bounds__(256) void add_flat2(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ o, i64 n2) {
  i64 i = ((i64)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < n2) {
    double2 x0 = a[i], x1 = a[i + 1], y0 = b[i], y1 = b[i + 1];
    o[i] = make_double2(x0.x + y0.x, x0.y + y0.y); o[i + 1] = make_double2(x1.x + y1.x, x1.y + y1.y);
  }
}
__global__ __launch_bounds__(256) void add_flat4(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ o, i64 n2) {
  i64 i = (i64)blockIdx.x * 1024 + threadIdx.x;
  if (i + 768 < n2) {
    double2 x0 = a[i], x1 = a[i + 256], x2 = a[i + 512], x3 = a[i + 768], y0 = b[i], y1 = b[i + 256], y2 = b[i + 512], y3 = b[i + 768];
    o[i] = make_double2(x0.x + y0.x, x0.y + y0.y); o[i + 256] = make_double2(x1.x + y1.x, x1.y + y1.y);
    o[i + 512] = make_double2(x2.x + y2.x, x2.y + y2.y); o[i + 768] = make_double2(x3.x + y3.x, x3.y + y3.y);
  }
}
// read-only: what the read side alone can stream (one double per block is written)
__global__ __launch_bounds__(256) void read_gs(const double2* __restrict__ a, double* __restrict__ o, i64 n2) {
  double acc = 0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n2; i += (i64)gridDim.x * 256) { double2 x = a[i]; acc += x.x + x.y; }
  if (acc == 12345.678) o[blockIdx.x] = acc;  // never true for the fill pattern; keeps the loads alive
}
// write-only
__global__ __launch_bounds__(256) void fill_gs(double2* __restrict__ o, i64 n2) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n2; i += (i64)gridDim.x * 256) o[i] = make_double2(1.0, 2.0);
}
// column strips: block = 512 columns x rpb rows, marching down rows (like the stencil kernels)
__global__ __launch_bounds__(256) void add_tile(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ o,
                                                i64 ld, int rpb, int ncb) {
  int cb = blockIdx.x % ncb, rc = blockIdx.x / ncb;
  i64 c = ((i64)cb * 256 + threadIdx.x) * 2;
  for (i64 j = (i64)rc * rpb; j < (i64)(rc + 1) * rpb; ++j) {
    double2 x = *(const double2*)(a + j * ld + c), y = *(const double2*)(b + j * ld + c);
    *(double2*)(o + j * ld + c) = make_double2(x.x + y.x, x.y + y.y);
  }
}
// The byte mix of the fused finest pass without its arithmetic and without any overlap between tiles: per fine
// point it reads x and b (16 B) and a quarter of a coarse value (2 B), writes x' (8 B) and a quarter of a coarse
// value (2 B) = 28 B.  Same shape of access: a wave owns a strip of 128 columns (one 16-byte pair per lane) and
// marches down the rows of its tile, four rows in flight; workgroups b and b+8 (one XCD) work on one band of rows.
template <bool NT>
__global__ __launch_bounds__(256) void mix_tile(const double* __restrict__ x, const double* __restrict__ b, const double* __restrict__ c,
                                                double* __restrict__ xo, double* __restrict__ cb, i64 ld, i64 cld, int rpb, int ncb, int nrc) {
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, rpx = nrc >> 3;
  const int rc = xcd * rpx + w / ncb, cbk = w % ncb;
  const i64 col = ((i64)cbk * 256 + threadIdx.x) * 2;
  typedef double dv2 __attribute__((ext_vector_type(2)));
  for (i64 j = (i64)rc * rpb; j < (i64)(rc + 1) * rpb; j += 4) {
    double2 xv[4], bv[4]; double cv[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) { xv[q] = *(const double2*)(x + (j + q) * ld + col); bv[q] = *(const double2*)(b + (j + q) * ld + col); }
#pragma unroll
    for (int q = 0; q < 2; ++q) cv[q] = c[(j / 2 + q) * cld + col / 2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dv2 o = {xv[q].x + bv[q].x + cv[q / 2], xv[q].y + bv[q].y};
      if (NT) __builtin_nontemporal_store(o, (dv2*)(xo + (j + q) * ld + col)); else *(dv2*)(xo + (j + q) * ld + col) = o;
      if ((q & 1) == 0) { if (NT) __builtin_nontemporal_store(o.x + o.y, cb + (j / 2 + q / 2) * cld + col / 2); else cb[(j / 2 + q / 2) * cld + col / 2] = o.x + o.y; }
    }
  }
}
// The same bytes, one workgroup per 2 rows x 512 columns, in row-major order of the workgroups: no loop, nobody ever
// waits for a store (loads and stores of a wave complete in order on this architecture, so a loop that stores and
// then loads again waits for its own stores).
template <bool NT>
__global__ __launch_bounds__(256) void mix_flat(const double* __restrict__ x, const double* __restrict__ b, const double* __restrict__ c,
                                                double* __restrict__ xo, double* __restrict__ cb, i64 ld, i64 cld, int ncb) {
  const i64 j = (i64)(blockIdx.x / ncb) * 2, col = ((i64)(blockIdx.x % ncb) * 256 + threadIdx.x) * 2;
  typedef double dv2 __attribute__((ext_vector_type(2)));
  const double2 x0 = *(const double2*)(x + j * ld + col), x1 = *(const double2*)(x + (j + 1) * ld + col);
  const double2 b0 = *(const double2*)(b + j * ld + col), b1 = *(const double2*)(b + (j + 1) * ld + col);
  const double cv = c[(j / 2) * cld + col / 2];
  dv2 o0 = {x0.x + b0.x + cv, x0.y + b0.y}, o1 = {x1.x + b1.x + cv, x1.y + b1.y};
  if (NT) { __builtin_nontemporal_store(o0, (dv2*)(xo + j * ld + col)); __builtin_nontemporal_store(o1, (dv2*)(xo + (j + 1) * ld + col));
            __builtin_nontemporal_store(o0.x + o1.y, cb + (j / 2) * cld + col / 2); }
  else { *(dv2*)(xo + j * ld + col) = o0; *(dv2*)(xo + (j + 1) * ld + col) = o1; cb[(j / 2) * cld + col / 2] = o0.x + o1.y; }
}
// The tile shape again, but with the loads running AHEAD rows in front of their use in a register ring (as the fused
// pass does), so that a wave never waits for the stores it has just issued.
template <bool NT, int AHEAD>
__global__ __launch_bounds__(256) void mix_tile_ahead(const double* __restrict__ x, const double* __restrict__ b, const double* __restrict__ c,
                                                      double* __restrict__ xo, double* __restrict__ cb, i64 ld, i64 cld, int rpb, int ncb, int nrc) {
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, rpx = nrc >> 3;
  const int rc = xcd * rpx + w / ncb, cbk = w % ncb;
  const i64 col = ((i64)cbk * 256 + threadIdx.x) * 2;
  typedef double dv2 __attribute__((ext_vector_type(2)));
  const i64 j0 = (i64)rc * rpb;
  double2 xv[AHEAD], bv[AHEAD]; double cv[AHEAD / 2];
#pragma unroll
  for (int q = 0; q < AHEAD; ++q) { xv[q] = *(const double2*)(x + (j0 + q) * ld + col); bv[q] = *(const double2*)(b + (j0 + q) * ld + col); }
#pragma unroll
  for (int q = 0; q < AHEAD / 2; ++q) cv[q] = c[(j0 / 2 + q) * cld + col / 2];
  for (i64 j = j0; j < j0 + rpb; j += AHEAD) {
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) {
      dv2 o = {xv[q].x + bv[q].x + cv[q / 2], xv[q].y + bv[q].y};
      const i64 jn = j + q + AHEAD < j0 + rpb ? j + q + AHEAD : j + q; // the tile's last rows are loaded twice
      xv[q] = *(const double2*)(x + jn * ld + col); bv[q] = *(const double2*)(b + jn * ld + col);
      if (NT) __builtin_nontemporal_store(o, (dv2*)(xo + (j + q) * ld + col)); else *(dv2*)(xo + (j + q) * ld + col) = o;
      if ((q & 1) == 1) {
        if (NT) __builtin_nontemporal_store(o.x + o.y, cb + ((j + q) / 2) * cld + col / 2); else cb[((j + q) / 2) * cld + col / 2] = o.x + o.y;
        cv[q / 2] = c[(jn / 2) * cld + col / 2];
      }
    }
  }
}
// pseudo-random doubles in (-1, 1): does the rate depend on the data?
__global__ __launch_bounds__(256) void fill_random(double* __restrict__ o, i64 n, unsigned long long seed) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    o[i] = (double)(long long)z * (1.0 / 9223372036854775808.0);
  }
}
int main(int argc, char** argv) {
  i64 ny = 32768, ld = 32768 + 16;
  i64 n = ny * ld, n2 = n / 2;
  double *a, *b, *o;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&o, n * 8));
  CK(hipMemset(a, 1, n * 8)); CK(hipMemset(b, 2, n * 8)); CK(hipMemset(o, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* name, double bytes, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9, tot = 0; int reps = 5;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms; }
    printf("%-58s avg %7.3f ms  %7.0f GB/s   best %7.0f GB/s\n", name, tot / reps, bytes / (tot / reps) / 1e6, bytes / best / 1e6); fflush(stdout);
  };
  for (int blocks : {2048, 4096, 8192, 65536}) {
    char nm[64]; snprintf(nm, 64, "copy_gs blocks=%d", blocks);
    report(nm, 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy_gs<false>, dim3(blocks), dim3(256), 0, 0, (const double2*)a, (double2*)o, n2); });
  }
  report("copy_gs_nt blocks=4096", 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy_gs<true>, dim3(4096), dim3(256), 0, 0, (const double2*)a, (double2*)o, n2); });
  for (int blocks : {2048, 8192, 32768}) {
    char nm[64]; snprintf(nm, 64, "copy_chunk blocks=%d", blocks);
    i64 per = (n2 + blocks - 1) / blocks; per = (per + 1023) / 1024 * 1024;
    report(nm, 2.0 * n * 8, [&] { hipLaunchKernelGGL(copy_chunk, dim3(blocks), dim3(256), 0, 0, (const double2*)a, (double2*)o, n2, per); });
  }
  report("memcpyDtoD", 2.0 * n * 8, [&] { CK(hipMemcpyAsync(o, a, n * 8, hipMemcpyDeviceToDevice, 0)); });
  for (int blocks : {2048, 8192})  {
    char nm[64]; snprintf(nm, 64, "add_gs (2R1W) blocks=%d", blocks);
    report(nm, 3.0 * n * 8, [&] { hipLaunchKernelGGL(add_gs, dim3(blocks), dim3(256), 0, 0, (const double2*)a, (const double2*)b, (double2*)o, n2); });
  }
  report("add_flat1 (2R1W, no loop)", 3.0 * n * 8, [&] { hipLaunchKernelGGL(add_flat1, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, (const double2*)a, (const double2*)b, (double2*)o, n2); });
  report("add_flat2 (2R1W, no loop)", 3.0 * n * 8, [&] { hipLaunchKernelGGL(add_flat2, dim3((unsigned)((n2 / 2 + 255) / 256)), dim3(256), 0, 0, (const double2*)a, (const double2*)b, (double2*)o, n2); });
  report("add_flat4 (2R1W, no loop)", 3.0 * n * 8, [&] { hipLaunchKernelGGL(add_flat4, dim3((unsigned)(n2 / 1024)), dim3(256), 0, 0, (const double2*)a, (const double2*)b, (double2*)o, n2); });
  for (int blocks : {65536, 262144}) {
    char nm[64]; snprintf(nm, 64, "add_gs (2R1W) blocks=%d", blocks);
    report(nm, 3.0 * n * 8, [&] { hipLaunchKernelGGL(add_gs, dim3(blocks), dim3(256), 0, 0, (const double2*)a, (const double2*)b, (double2*)o, n2); });
  }
  for (int blocks : {8192, 65536}) {
    char nm[64]; snprintf(nm, 64, "read_gs (1R) blocks=%d", blocks);
    report(nm, 1.0 * n * 8, [&] { hipLaunchKernelGGL(read_gs, dim3(blocks), dim3(256), 0, 0, (const double2*)a, o, n2); });
    snprintf(nm, 64, "fill_gs (1W) blocks=%d", blocks);
    report(nm, 1.0 * n * 8, [&] { hipLaunchKernelGGL(fill_gs, dim3(blocks), dim3(256), 0, 0, (double2*)o, n2); });
  }
  for (int rpb : {32, 128, 512}) {
    char nm[64]; snprintf(nm, 64, "add_tile (2R1W) rpb=%d", rpb);
    int ncb = 32768 / 512;
    report(nm, 3.0 * ny * 32768 * 8, [&] { hipLaunchKernelGGL(add_tile, dim3(ncb * (ny / rpb)), dim3(256), 0, 0, a, b, o, ld, rpb, ncb); });
  }
  { // the finest pass's byte mix, on arrays of its sizes (x, b, x' of 32768 x 32784, coarse arrays a quarter of that)
    double *c, *cbv; i64 cld = 16384 + 16, cn = 16384 * cld;
    CK(hipMalloc(&c, cn * 8)); CK(hipMalloc(&cbv, cn * 8)); CK(hipMemset(c, 1, cn * 8)); CK(hipMemset(cbv, 0, cn * 8));
    const double bytes = 28.0 * ny * 32768;
    for (int rpb : {64, 128, 512}) {
      char nm[64]; int ncb = 32768 / 512, nrc = (int)(ny / rpb);
      snprintf(nm, 64, "mix_tile (28 B/pt) rpb=%d", rpb);
      report(nm, bytes, [&] { hipLaunchKernelGGL(mix_tile<false>, dim3(ncb * nrc), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, rpb, ncb, nrc); });
      snprintf(nm, 64, "mix_tile nt stores rpb=%d", rpb);
      report(nm, bytes, [&] { hipLaunchKernelGGL(mix_tile<true>, dim3(ncb * nrc), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, rpb, ncb, nrc); });
    }
    report("mix_flat (2 rows x 512 cols)", bytes, [&] { hipLaunchKernelGGL(mix_flat<false>, dim3(64 * (unsigned)(ny / 2)), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, 64); });
    report("mix_flat nt stores", bytes, [&] { hipLaunchKernelGGL(mix_flat<true>, dim3(64 * (unsigned)(ny / 2)), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, 64); });
    for (int rpb : {64, 256}) {
      char nm[64]; int ncb = 32768 / 512, nrc = (int)(ny / rpb);
      snprintf(nm, 64, "mix_tile_ahead<4> nt rpb=%d", rpb);
      report(nm, bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 4>), dim3(ncb * nrc), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, rpb, ncb, nrc); });
      snprintf(nm, 64, "mix_tile_ahead<8> nt rpb=%d", rpb);
      report(nm, bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 8>), dim3(ncb * nrc), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, rpb, ncb, nrc); });
      snprintf(nm, 64, "mix_tile_ahead<8> rpb=%d", rpb);
      report(nm, bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<false, 8>), dim3(ncb * nrc), dim3(256), 0, 0, a, b, c, o, cbv, ld, cld, rpb, ncb, nrc); });
    }
    // the same with the roles of the arrays exchanged: which allocation is written decides as much as the kernel
    report("mix_tile nt rpb=64, writes a", bytes, [&] { hipLaunchKernelGGL(mix_tile<true>, dim3(64 * 512), dim3(256), 0, 0, o, b, c, a, cbv, ld, cld, 64, 64, 512); });
    report("mix_tile nt rpb=64, writes b", bytes, [&] { hipLaunchKernelGGL(mix_tile<true>, dim3(64 * 512), dim3(256), 0, 0, o, a, c, b, cbv, ld, cld, 64, 64, 512); });
  }
  { // The same movers with their arrays spread over the process's sequence of allocations, as the solver's placement
    // selection spreads the finest pass's arrays: 24 more arrays of the same size; "early" = among the first three
    // (a, b, o above), "mid" and "late" = the 10th and the 22nd of the new ones.
    double *c, *cbv; i64 cld = 16384 + 16, cn = 16384 * cld;
    CK(hipMalloc(&c, cn * 8)); CK(hipMalloc(&cbv, cn * 8)); CK(hipMemset(c, 1, cn * 8)); CK(hipMemset(cbv, 0, cn * 8));
    double* P[24]; int np = 0;
    for (; np < 24; ++np) { if (hipMalloc(&P[np], n * 8) != hipSuccess) { (void)hipGetLastError(); break; } CK(hipMemset(P[np], 1, n * 8)); }
    if (np >= 23) {
      const double bytes = 28.0 * ny * 32768;
      double *mid = P[10], *mid2 = P[11], *late = P[22];
      struct { const char* nm; double *x, *b, *o; } cases[] = {
        {"x early, b early, out early", a, b, o}, {"x early, b early, out mid", a, b, mid}, {"x mid, b mid, out mid", mid2, P[12], mid},
        {"x late, b early, out mid", late, b, mid}, {"x mid, b early, out late", mid, b, late}, {"x early, b late, out mid", a, late, mid}};
      for (int pass = 0; pass < 2; ++pass) {
      if (pass == 1) { // the same once more on pseudo-random data
        for (double* q : {a, b, o, mid, mid2, late, P[12]}) hipLaunchKernelGGL(fill_random, dim3(65536), dim3(256), 0, 0, q, n, (unsigned long long)(size_t)q);
        hipLaunchKernelGGL(fill_random, dim3(65536), dim3(256), 0, 0, c, cn, 77ull);
        CK(hipDeviceSynchronize());
        printf("--- pseudo-random data in every array\n");
      }
      for (auto& q : cases) {
        char nm[96];
        snprintf(nm, 96, "mix_flat nt: %s", q.nm);
        report(nm, bytes, [&] { hipLaunchKernelGGL(mix_flat<true>, dim3(64 * (unsigned)(ny / 2)), dim3(256), 0, 0, q.x, q.b, c, q.o, cbv, ld, cld, 64); });
        snprintf(nm, 96, "mix_tile nt 64: %s", q.nm);
        report(nm, bytes, [&] { hipLaunchKernelGGL(mix_tile<true>, dim3(64 * 512), dim3(256), 0, 0, q.x, q.b, c, q.o, cbv, ld, cld, 64, 64, 512); });
      }
      }
    }
    if (np >= 23) { // how far ahead must the loads run when only two waves per SIMD are resident (as in the fused finest pass)?
      const double bytes = 28.0 * ny * 32768;
      const size_t lds = 72 * 1024; // two workgroups of four waves per CU
      CK(hipFuncSetAttribute((const void*)mix_tile_ahead<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipFuncSetAttribute((const void*)mix_tile_ahead<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipFuncSetAttribute((const void*)mix_tile_ahead<true, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipFuncSetAttribute((const void*)mix_tile_ahead<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipFuncSetAttribute((const void*)mix_tile_ahead<true, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      double *xx = P[22], *bb = b, *oo = P[10];
      report("2 waves/SIMD, loads 2 rows ahead", bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 2>), dim3(64 * 512), dim3(256), lds, 0, xx, bb, c, oo, cbv, ld, cld, 64, 64, 512); });
      report("2 waves/SIMD, loads 4 rows ahead", bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 4>), dim3(64 * 512), dim3(256), lds, 0, xx, bb, c, oo, cbv, ld, cld, 64, 64, 512); });
      report("2 waves/SIMD, loads 6 rows ahead", bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 6>), dim3(64 * 480), dim3(256), lds, 0, xx, bb, c, oo, cbv, ld, cld, 60, 64, 480); });
      report("2 waves/SIMD, loads 8 rows ahead", bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 8>), dim3(64 * 512), dim3(256), lds, 0, xx, bb, c, oo, cbv, ld, cld, 64, 64, 512); });
      report("2 waves/SIMD, loads 12 rows ahead", bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 12>), dim3(64 * 512), dim3(256), lds, 0, xx, bb, c, oo, cbv, ld, cld, 60, 64, 512); });
      report("all waves resident, loads 4 rows ahead", bytes, [&] { hipLaunchKernelGGL((mix_tile_ahead<true, 4>), dim3(64 * 512), dim3(256), 0, 0, xx, bb, c, oo, cbv, ld, cld, 64, 64, 512); });
    }
    for (int k = 0; k < np; ++k) CK(hipFree(P[k]));
  }
  { // Does a read that follows a write of the same data closely find it in the Infinity Cache?  One array is written
    // and read back band by band (a write launch, then a read launch of the same band), for several band sizes;
    // against writing all of it and then reading all of it.
    const i64 total2 = n2; // double2 elements
    auto pass = [&](i64 band2, bool nt) {
      for (i64 off = 0; off < total2; off += band2) {
        const i64 m = off + band2 <= total2 ? band2 : total2 - off;
        const unsigned blocks = (unsigned)((m + 255) / 256 < 65536 ? (m + 255) / 256 : 65536);
        if (nt) hipLaunchKernelGGL(copy_gs<true>, dim3(blocks), dim3(256), 0, 0, (const double2*)b + off, (double2*)o + off, m);
        else hipLaunchKernelGGL(copy_gs<false>, dim3(blocks), dim3(256), 0, 0, (const double2*)b + off, (double2*)o + off, m);
        hipLaunchKernelGGL(read_gs, dim3(blocks), dim3(256), 0, 0, (const double2*)o + off, a, m);
      }
    };
    for (int mb : {8192, 1024, 256, 128, 64, 32, 16}) {
      char nm[64];
      const i64 band2 = (i64)mb * (1 << 20) / 16;
      snprintf(nm, 64, "copy+readback band %d MiB", mb);
      report(nm, 3.0 * n * 8, [&] { pass(band2, false); });
      snprintf(nm, 64, "same, nt copy, band %d MiB", mb);
      report(nm, 3.0 * n * 8, [&] { pass(band2, true); });
    }
  }
  return 0;
}

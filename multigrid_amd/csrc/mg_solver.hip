// The V-cycle solver behind a C handle: Multigrid.__init__/solve() of the reference
// (core/parallel_multigrid.py:153-348) on the undecomposed grid of ONE GPU, for callers that do not
// go through Python.  The library owns the device arrays of every level; the caller hands host (or
// device) arrays over with mg_solver_upload / reads them back with mg_solver_download.
//
// The schedule is the one of the package's slab engine on one rank: every smoothing call is one fused
// pass (mg_smooth_fused[_restrict]), the finest residual is restricted by the pass that computes it,
// large coarse levels recompute their pre-smoothing inside the post-smoothing pass, the whole-domain
// levels below the first (dead in the reference's loop bounds, :302) run on a second stream, and a
// whole cycle is one hipGraph replay.  Values are those of the reference's schedule on the global
// grid, bit for bit.
#include <string.h>

#include <chrono>
#include <new>
#include <utility>
#include <vector>

#include "mg_common.h"

namespace {

struct JLevel { // a Jacobi level: two x arrays (ping-pong), b; r on the finest only
    i64 ny, nx, ld;
    double h;
    double *xa, *xb, *b, *r;
    bool recompute; // pre-smoothing recomputed by the post-smoothing pass, x not stored on the way down
};

struct WLevel { // a whole-domain (SSOR) level
    i64 ny, nx, ld;
    double h;
    double *x, *b, *r;
};

} // namespace

struct mg_solver {
    int device;
    i64 gny, gnx;
    int n_para;
    double h0;
    bool fix_coarse_ascent, use_graph, eager_dead;
    int ordering;
    double omega;
    int max_cycles;
    i64 recompute_min_nx, recompute_min_blocks;
    std::vector<JLevel> lv;
    std::vector<WLevel> wl;
    std::vector<void *> allocs;
    u64 *norm[2];      // two max-norm accumulators, used in turn: the one the finest x's array at the start of a cycle selects
    u64 *host_norm[2]; // pinned host copies of them
    hipStream_t copy;  // reads a cycle's accumulator back while the next cycle runs
    hipEvent_t filled[2], fetched[2];
    int last_acc;      // accumulator the last cycle filled
    hipStream_t stream, side;
    hipEvent_t fork, join;
    hipGraphExec_t graphs[2][2]; // [array the finest x starts in][finest residual stored]
    int graph_end_cur[2];
    int cur0;      // which of (xa, xb) of level 0 holds x
    bool r_stale;  // the finest residual array is older than x
    int it;
    bool diverged;
    int placement_candidates; // candidate arrays timed by the placement selection, -1: placement not tuned
    std::vector<double> hist;
};

namespace {

const int N1 = 3, N2 = 6; // core/parallel_multigrid.py:255-256

i64 pitch(i64 nx) { return (nx + 15) / 16 * 16; }

int dev_alloc(mg_solver *s, double **p, i64 n)
{
    void *q = nullptr;
    MG_HIP(hipMalloc(&q, (size_t)n * sizeof(double)));
    s->allocs.push_back(q);
    MG_HIP(hipMemsetAsync(q, 0, (size_t)n * sizeof(double), s->stream));
    *p = (double *)q;
    return 0;
}

int check_device(const mg_solver *s, const char *fn)
{
    MG_REQUIRE(s, "%s: null solver", fn);
    int d = -1;
    MG_HIP(hipGetDevice(&d));
    MG_REQUIRE(d == s->device, "%s: the solver lives on device %d, the current device is %d", fn, s->device, d);
    return 0;
}

void decide_recompute(mg_solver *s)
{
    for (int l = 0; l < s->n_para; ++l) {
        JLevel &L = s->lv[l];
        // the 9-deep pass pays with wider overlaps, which a level must be wide enough and have
        // workgroups enough to absorb
        const i64 strips = (L.nx + 107) / 108, blocks = ((strips + 3) / 4) * ((L.ny - 1 + 63) / 64);
        L.recompute = l >= 1 && L.nx >= s->recompute_min_nx && blocks >= s->recompute_min_blocks;
    }
}

// where the restricted residual of level l goes, and the coarse correction of level l comes from
void coarse_of(mg_solver *s, int l, double **cb, i64 *cny, i64 *cnx, i64 *cld)
{
    if (l + 1 < s->n_para) {
        JLevel &C = s->lv[l + 1];
        *cb = C.b, *cny = C.ny, *cnx = C.nx, *cld = C.ld;
    } else {
        WLevel &C = s->wl[0];
        *cb = C.b, *cny = C.ny, *cnx = C.nx, *cld = C.ld;
    }
}

// smooth_parallel on the finest level (:100-114) with the prolongation in front (:322) and the
// restriction of its residual behind (:273 of the next cycle): one pass
int finest_pass(mg_solver *s, int sweeps, const double *crs, i64 cny, i64 cnx, i64 cld, bool want_norm, bool store_r)
{
    JLevel &L = s->lv[0];
    double *xs[2] = {L.xa, L.xb};
    double *cb;
    i64 bny, bnx, bld;
    coarse_of(s, 0, &cb, &bny, &bnx, &bld);
    u64 *const acc = s->norm[s->cur0]; // cur0 at the start of the pass: a captured graph always fills the same one
    if (want_norm) s->last_acc = s->cur0;
    if (want_norm && mg_norm_reset((uint64_t *)acc, s->stream)) return 1;
    if (mg_smooth_fused_restrict(L.b, xs[s->cur0], xs[1 - s->cur0], store_r ? L.r : nullptr, crs, L.ny, L.nx, L.ld, cny, cnx,
                                 cld, 1, L.ny - 1, 0, L.ny - 1, 0, sweeps, L.h, s->omega,
                                 want_norm ? (uint64_t *)acc : nullptr, 0, cb, bny, bnx, bld, 1, bny - 1, 1, 0, 0, 0,
                                 s->stream))
        return 1;
    s->cur0 = 1 - s->cur0;
    s->r_stale = !store_r;
    return 0;
}

// whole-domain levels below the first (:294-304)
int tail_below(mg_solver *s, hipStream_t st)
{
    const int nw = (int)s->wl.size();
    for (int i = 1; i < nw; ++i) {
        WLevel &W = s->wl[i], &F = s->wl[i - 1];
        if (mg_zero(W.x, W.ny * W.ld, st)) return 1;
        if (mg_restrict(F.r, F.ny, F.nx, F.ld, W.b, W.ny, W.nx, W.ld, 1, W.ny - 1, 0, 0, 1, 0, 0, st)) return 1;
        if (mg_sor(W.b, W.x, W.r, W.ny, W.nx, W.ld, W.h, N1, 1, s->ordering, 1, 0, st)) return 1;
    }
    // the reference's range(2, nw): the coarsest correction climbs to level 1, never into level 0
    for (int i = 2; i < nw + (s->fix_coarse_ascent ? 1 : 0); ++i) {
        WLevel &C = s->wl[nw - i + 1], &F = s->wl[nw - i];
        if (mg_prolong_add(C.x, C.ny, C.nx, C.ld, F.x, F.ny, F.nx, F.ld, 0, F.ny, 1, F.nx - 2, 0, 0, 1, 0, 0, st)) return 1;
        if (mg_sor(F.b, F.x, F.r, F.ny, F.nx, F.ld, F.h, N2, 1, s->ordering, 1, 0, st)) return 1;
    }
    return 0;
}

// the body of the loop at :270-347
int cycle(mg_solver *s, bool store_r)
{
    const int n_para = s->n_para;
    hipStream_t st = s->stream;
    // descent: x = 0, N1 sweeps, residual, restriction (:274-289); b_1 came with the finest residual
    for (int l = 1; l < n_para; ++l) {
        JLevel &L = s->lv[l];
        double *cb;
        i64 bny, bnx, bld;
        coarse_of(s, l, &cb, &bny, &bnx, &bld);
        if (mg_smooth_fused_restrict(L.b, nullptr, L.recompute ? nullptr : L.xa, nullptr, nullptr, L.ny, L.nx, L.ld, 0, 0, 0,
                                     1, L.ny - 1, 0, L.ny - 1, 0, N1, L.h, s->omega, nullptr, 0, cb, bny, bnx, bld, 1, bny - 1, 1,
                                     0, 0, 0, st))
            return 1;
    }
    // first whole-domain level (:291-293)
    WLevel &W0 = s->wl[0];
    if (mg_zero(W0.x, W0.ny * W0.ld, st)) return 1;
    if (mg_sor(W0.b, W0.x, W0.r, W0.ny, W0.nx, W0.ld, W0.h, N1, 1, s->ordering, 1, 0, st)) return 1;
    bool forked = false;
    if (s->fix_coarse_ascent) {
        if (tail_below(s, st)) return 1;
    } else if (s->eager_dead && s->wl.size() > 1) {
        // nothing in the cycle reads these levels: they run beside the ascent, on the second stream, from
        // this point of the main sequence on -- but are enqueued after it (below): captured in that order the
        // graph starts the ascent at once; captured here, the ascent waited for most of them
        MG_HIP(hipEventRecord(s->fork, st));
        forked = true;
    }
    // ascent (:306-313): prolongation + N2 sweeps; the residuals of these levels are never read
    const double *crs = W0.x;
    i64 cny = W0.ny, cnx = W0.nx, cld = W0.ld;
    for (int l = n_para - 1; l >= 1; --l) {
        JLevel &L = s->lv[l];
        if (L.recompute) {
            if (mg_smooth_fused(L.b, nullptr, L.xa, nullptr, crs, L.ny, L.nx, L.ld, cny, cnx, cld, 1, L.ny - 1, 0, L.ny - 1, 0,
                                N1 + N2, N1, L.h, s->omega, nullptr, 1, 0, 0, st))
                return 1;
            crs = L.xa;
        } else {
            if (mg_smooth_fused(L.b, L.xa, L.xb, nullptr, crs, L.ny, L.nx, L.ld, cny, cnx, cld, 1, L.ny - 1, 0, L.ny - 1, 0,
                                N2, 0, L.h, s->omega, nullptr, 1, 0, 0, st))
                return 1;
            crs = L.xb;
        }
        cny = L.ny, cnx = L.nx, cld = L.ld;
    }
    if (finest_pass(s, N2, crs, cny, cnx, cld, true, store_r)) return 1;
    if (forked) {
        MG_HIP(hipStreamWaitEvent(s->side, s->fork, 0));
        if (tail_below(s, s->side)) return 1;
        MG_HIP(hipEventRecord(s->join, s->side));
        MG_HIP(hipStreamWaitEvent(st, s->join, 0));
    }
    return 0;
}

void drop_graphs(mg_solver *s)
{
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            if (s->graphs[a][b]) {
                (void)hipGraphExecDestroy(s->graphs[a][b]);
                s->graphs[a][b] = nullptr;
            }
}

int run_cycle(mg_solver *s, bool store_r)
{
    if (!s->use_graph) return cycle(s, store_r);
    if (!s->graphs[s->cur0][store_r]) {
        // capture every variant now (capturing records, it does not execute)
        const int cur = s->cur0, acc = s->last_acc;
        const bool stale = s->r_stale;
        for (int start = 0; start < 2; ++start)
            for (int sr = 0; sr < 2; ++sr) {
                if (s->graphs[start][sr]) continue;
                s->cur0 = start;
                if (mg_graph_begin(s->stream)) return 1;
                const int rc = cycle(s, sr != 0);
                void *g = nullptr;
                if (mg_graph_end(s->stream, &g) || rc) {
                    s->cur0 = cur, s->r_stale = stale, s->last_acc = acc;
                    if (g) mg_graph_destroy(g);
                    return 1;
                }
                s->graphs[start][sr] = (hipGraphExec_t)g;
                s->graph_end_cur[start] = s->cur0;
            }
        s->cur0 = cur, s->r_stale = stale, s->last_acc = acc;
    }
    if (mg_graph_launch(s->graphs[s->cur0][store_r], s->stream)) return 1;
    s->last_acc = s->cur0; // the accumulator that graph fills
    s->cur0 = s->graph_end_cur[s->cur0];
    s->r_stale = !store_r;
    return 0;
}

int materialize_residual(mg_solver *s)
{
    if (!s->r_stale) return 0;
    JLevel &L = s->lv[0];
    if (mg_residual(L.r, s->cur0 ? L.xb : L.xa, L.b, L.ny, L.nx, L.ld, 1, L.ny - 1, L.h, nullptr, 1, 0, MG_FLAG_FINEST, s->stream))
        return 1;
    s->r_stale = false;
    return 0;
}

// max|r| of the last cycle, now (waits for the stream)
int read_err(mg_solver *s, double *err) { return mg_norm_read((const uint64_t *)s->norm[s->last_acc], err, s->stream); }

// Read-back of accumulator `a` as it stands at this point of the main stream, on the copy stream: the host
// collects it (collect_err) after it has enqueued the next cycle, so the GPU does not idle while the host waits
// for the number and launches the next graph.
int fetch_err(mg_solver *s, int a)
{
    MG_HIP(hipEventRecord(s->filled[a], s->stream));
    MG_HIP(hipStreamWaitEvent(s->copy, s->filled[a], 0));
    MG_HIP(hipMemcpyAsync(s->host_norm[a], s->norm[a], MG_NORM_SLOTS * sizeof(u64), hipMemcpyDeviceToHost, s->copy));
    MG_HIP(hipEventRecord(s->fetched[a], s->copy));
    return 0;
}

int collect_err(mg_solver *s, int a, double *err)
{
    MG_HIP(hipEventSynchronize(s->fetched[a]));
    u64 m = 0;
    for (int k = 0; k < MG_NORM_SLOTS; ++k) m = s->host_norm[a][k] > m ? s->host_norm[a][k] : m;
    union { u64 u; double d; } cv;
    cv.u = m;
    *err = cv.d;
    return 0;
}

// The same launch of the finest pass takes 5.4 or 6.1 ms at 32769 x 32769 depending on which allocations it
// WRITES: x and the coarse right-hand side; the arrays it only reads matter by 2-3 %.  What decides is the
// combination of the two written allocations, fixed when they are made and not predictable from user space (in a
// fresh process the first eleven 8-GiB allocations are slow as x beside a coarse right-hand side allocated before
// them, all later ones fast; a coarse right-hand side allocated after the fast ones makes the pass slow again).
// Unless MG_SOLVER_NO_PLACEMENT is given create therefore measures: the coarse right-hand side is allocated first AND KEPT (*cb_out:
// it becomes the next level's b), candidates of the finest array's size are allocated one by one, each is timed
// as the output of the cycle's finest pass, and as soon as both kinds have been seen and four candidates are of
// the fast kind those become fin[] = {xa, xb, b, r}, in the assignment whose pass is fastest; all zero.
// On top of that the three big arrays of the pass are spread over the sequence of allocations: measured with thirty
// candidates, the pass is fastest (by another 2-4 %) when the array it writes shares its stretch of ~96 GiB of
// that sequence with neither of the two it reads.  So b becomes a candidate of the kind that is slow as an output
// (it reads as fast as any), one x the fastest output, the other x whichever of a few candidates allocated further
// along makes the fastest pair in both directions; kept only where that beats the choice among the fast four.
// No new candidate is allocated once 1.5 s have passed (an 8-GiB allocation takes 5 to 200 ms).  fin[] and *cb_out
// stay null when the level is small (< 1 GiB per array), memory is short or anything fails: create then allocates
// as usual.
} // namespace

void mg_place_finest(const MgPlaceShape &P, double *fin[4], double **cb_out, double **xa1_out, int *candidates_out)
{
    struct { hipStream_t stream; } sh{P.stream}, *s = &sh; // (the body below names the stream s->stream)
    const int KMAX = 26, SPREAD = 14;
    const double BUDGET = 1.5;
    const auto t_begin = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    const i64 ny = P.ny, nx = P.nx, ld = P.ld, n = ny * ld;
    const size_t bytes = (size_t)n * sizeof(double);
    for (int k = 0; k < 4; ++k) fin[k] = nullptr;
    *cb_out = nullptr, *xa1_out = nullptr;
    size_t free_b = 0, total_b = 0;
    if (bytes < ((size_t)1 << 30) || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    // room for the other levels (a third of the finest's four arrays, rounded up to two arrays) and a margin
    const size_t reserve = 2 * bytes + ((size_t)8 << 30);
    int kmax = free_b > reserve ? (int)((free_b - reserve) / bytes) : 0;
    if (kmax > KMAX) kmax = KMAX;
    if (kmax < 6) return;
    const i64 cny = P.cny, cnx = P.cnx, cld = P.cld; // the next level, Jacobi or whole-domain
    const size_t cbytes = (size_t)(cny * cld) * sizeof(double);
    double *crs = nullptr, *cb = nullptr;
    u64 *norm = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc((void **)&cb, cbytes) == hipSuccess && hipMalloc((void **)&crs, cbytes) == hipSuccess &&
              hipMalloc((void **)&norm, MG_NORM_SLOTS * sizeof(u64)) == hipSuccess &&
              hipMemsetAsync(crs, 0, cbytes, s->stream) == hipSuccess && hipMemsetAsync(cb, 0, cbytes, s->stream) == hipSuccess &&
              hipMemsetAsync(norm, 0, MG_NORM_SLOTS * sizeof(u64), s->stream) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    auto time_pass = [&](double *b, double *xin, double *xout, float *ms, const double *corr = nullptr) { // the launch of finest_pass; the first one is not timed
        bool good = true;
        for (int rep = 0; rep < 3 && good; ++rep) {
            good = !mg_smooth_fused_restrict(b, xin, xout, nullptr, corr ? corr : crs, ny, nx, ld, cny, cnx, cld, P.j0, P.j1, P.flo, P.fhi, P.oj, N2,
                                             P.h0, 1.0, (uint64_t *)norm, 0, cb, cny, cnx, cld, P.T0, P.T1, 1, 0, 0, 0, s->stream);
            if (rep == 0) good = good && hipEventRecord(e0, s->stream) == hipSuccess;
        }
        good = good && hipEventRecord(e1, s->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
               hipEventElapsedTime(ms, e0, e1) == hipSuccess;
        return good;
    };
    double *a[KMAX];
    float ms[KMAX];
    int na = 0, nt = 0; // candidates allocated / timed
    float lo = 0;
    bool gap = false;
    while (ok && na < kmax) {
        if (hipMalloc((void **)&a[na], bytes) != hipSuccess) { // someone else took the memory: choose among what there is
            (void)hipGetLastError();
            break;
        }
        ok = hipMemsetAsync(a[na++], 0, bytes, s->stream) == hipSuccess;
        if (na < 3) continue;
        for (; ok && nt < na; ++nt) ok = time_pass(a[nt == 0 ? 1 : 0], a[nt <= 1 ? 2 : 1], a[nt], &ms[nt]);
        if (!ok) break;
        float hi = ms[0];
        lo = ms[0];
        for (int k = 1; k < nt; ++k) lo = ms[k] < lo ? ms[k] : lo, hi = ms[k] > hi ? ms[k] : hi;
        int fast = 0;
        for (int k = 0; k < nt; ++k) fast += ms[k] <= 1.03f * lo;
        gap = hi > 1.06f * lo; // both kinds have been seen (they are 10 % or more apart)
        if (gap && fast >= 4) break;
        if (nt >= 4 && elapsed() > BUDGET) break;
    }
    if (ok && nt >= 4) {
        int order[KMAX];
        for (int k = 0; k < nt; ++k) order[k] = k;
        for (int i = 1; i < nt; ++i) // insertion sort by time
            for (int j = i; j > 0 && ms[order[j]] < ms[order[j - 1]]; --j) std::swap(order[j], order[j - 1]);
        // the cycle's own combination: x moves between two of the four, b is a third; where the kinds are mixed two
        // candidates that are fast as outputs may still make a slow pair: try the other assignments then
        const int combos[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 2, 0}, {0, 3, 1}, {1, 3, 0}, {2, 3, 0}};
        int best = 0;
        float tbest = 0;
        for (int c = 0; c < 6 && ok; ++c) {
            double *xa = a[order[combos[c][0]]], *xb = a[order[combos[c][1]]], *bb = a[order[combos[c][2]]];
            float t0 = 0, t1 = 0;
            ok = time_pass(bb, xa, xb, &t0) && time_pass(bb, xb, xa, &t1);
            if (c == 0 || t0 + t1 < tbest) best = c, tbest = t0 + t1;
            if (t0 + t1 <= 2 * 1.02f * lo) break;
        }
        const int q3 = 6 - combos[best][0] - combos[best][1] - combos[best][2];
        int pick[4] = {order[combos[best][0]], order[combos[best][1]], order[combos[best][2]], order[q3]};
        // spread over the sequence of allocations (see above)
        int slow[2] = {-1, -1};
        for (int k = 0; k < nt; ++k)
            if (ms[k] > 1.06f * lo && slow[1] < 0) slow[slow[0] < 0 ? 0 : 1] = k;
        if (ok && gap && slow[0] >= 0 && na < kmax) {
            const int n_timed = na;
            while (ok && na < kmax && na < n_timed + SPREAD && elapsed() < BUDGET) {
                if (hipMalloc((void **)&a[na], bytes) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                ok = hipMemsetAsync(a[na++], 0, bytes, s->stream) == hipSuccess;
            }
            const int xo = pick[0], bk = slow[0], nl = na - n_timed;
            int tryx[8], ntry = 0;
            tryx[ntry++] = pick[1];
            for (int k = n_timed + nl / 2; k < na && ntry < 7; k += 3) tryx[ntry++] = k;
            if (nl > 0) tryx[ntry++] = na - 1;
            int xbest = -1;
            float tspread = 0;
            if (xo == bk) ntry = 0; // only one fast candidate was seen: nothing to spread
            for (int q = 0; q < ntry && ok; ++q) {
                const int xi = tryx[q];
                bool seen = xi == xo || xi == bk;
                for (int p = 0; p < q; ++p) seen = seen || tryx[p] == xi;
                if (seen) continue;
                float t0 = 0, t1 = 0;
                ok = time_pass(a[bk], a[xi], a[xo], &t0) && time_pass(a[bk], a[xo], a[xi], &t1);
                if (ok && (xbest < 0 || t0 + t1 < tspread)) xbest = xi, tspread = t0 + t1;
            }
            if (ok && xbest >= 0 && tspread < 0.995f * tbest) {
                int rk = slow[1];
                for (int k = 0; k < na && (rk < 0 || rk == xo || rk == xbest || rk == bk); ++k) rk = k;
                pick[0] = xo, pick[1] = xbest, pick[2] = bk, pick[3] = rk;
            }
        }
        for (int q = 0; q < 4; ++q) // never two roles in one array: give up the selection rather than alias x and b
            for (int p = 0; p < q; ++p) ok = ok && pick[p] != pick[q];
        for (int q = 0; q < 4 && ok; ++q) { // xa, xb, b, r
            fin[q] = a[pick[q]];
            a[pick[q]] = nullptr;
        }
        if (ok) *candidates_out = nt, *cb_out = cb, cb = nullptr;
    }
    // (the candidates not chosen are still held: the next level's candidates then come from further along the sequence of
    // allocations, not from the memory these give back, which is the stretch the level's right-hand side lies in)
    // The same for the one other pass that writes gigabytes: level 1's ascent pass (1.04 or 1.10 ms at 32769 x 32769
    // depending on the allocation it writes its x into), where that is the 9-stage pass with recomputed
    // pre-smoothing.  Up to eight candidates are timed as its output beside the level's right-hand side (the array
    // kept above); the fastest becomes the level's x.
    if (ok && *cb_out && P.level1) {
        const i64 c2ny = P.c2ny, c2nx = P.c2nx, c2ld = P.c2ld;
        double *crs2 = nullptr, *cand[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        float t1[8], tf[8], lo1 = 0, hi1 = 0; // ms of the level's pass into the candidate; of the finest pass reading it
        int n1 = 0;
        bool ok1 = hipMalloc((void **)&crs2, (size_t)(c2ny * c2ld) * sizeof(double)) == hipSuccess &&
                   hipMemsetAsync(crs2, 0, (size_t)(c2ny * c2ld) * sizeof(double), s->stream) == hipSuccess;
        while (ok1 && n1 < 8) {
            if (hipMalloc((void **)&cand[n1], cbytes) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            ok1 = hipMemsetAsync(cand[n1], 0, cbytes, s->stream) == hipSuccess;
            for (int rep = 0; rep < 3 && ok1; ++rep) {
                ok1 = !mg_smooth_fused(*cb_out, nullptr, cand[n1], nullptr, crs2, cny, cnx, cld, c2ny, c2nx, c2ld, P.j0_1, P.j1_1, P.flo_1,
                                       P.fhi_1, P.oj, N1 + N2, N1, 2 * P.h0, 1.0, nullptr, 1, 0, 0, s->stream);
                if (rep == 0) ok1 = ok1 && hipEventRecord(e0, s->stream) == hipSuccess;
            }
            ok1 = ok1 && hipEventRecord(e1, s->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                  hipEventElapsedTime(&t1[n1], e0, e1) == hipSuccess;
            float f0 = 0, f1 = 0; // the candidate is also the correction the finest pass reads
            ok1 = ok1 && time_pass(fin[2], fin[0], fin[1], &f0, cand[n1]) && time_pass(fin[2], fin[1], fin[0], &f1, cand[n1]);
            if (!ok1) break;
            tf[n1] = 0.5f * (f0 + f1); // like t1: the time of two launches
            ++n1;
            lo1 = hi1 = t1[0];
            int fast1 = 0;
            for (int k = 1; k < n1; ++k) lo1 = t1[k] < lo1 ? t1[k] : lo1, hi1 = t1[k] > hi1 ? t1[k] : hi1;
            for (int k = 0; k < n1; ++k) fast1 += t1[k] <= 1.01f * lo1;
            if (n1 >= 6 && hi1 > 1.05f * lo1 && fast1 >= 2) break; // both kinds seen (6-8 % apart), and the fast one twice
            if (elapsed() > BUDGET + 0.5) break;
        }
        if (ok1 && n1 > 0) {
            int best = 0;
            for (int k = 1; k < n1; ++k)
                if (t1[k] + tf[k] < t1[best] + tf[best]) best = k;
            if (hipMemsetAsync(cand[best], 0, cbytes, s->stream) == hipSuccess) *xa1_out = cand[best], cand[best] = nullptr;
        }
        if (!ok1) (void)hipGetLastError();
        (void)hipStreamSynchronize(s->stream);
        for (double *q : cand)
            if (q) (void)hipFree(q);
        if (crs2) (void)hipFree(crs2);
    }
    // the timed launches left values in the chosen arrays: back to zero
    for (int q = 0; q < 4 && ok && fin[q]; ++q) ok = hipMemsetAsync(fin[q], 0, bytes, s->stream) == hipSuccess;
    if (ok && *cb_out) ok = hipMemsetAsync(*cb_out, 0, cbytes, s->stream) == hipSuccess;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (!ok) {
        (void)hipGetLastError();
        for (int q = 0; q < 4; ++q) {
            if (fin[q]) (void)hipFree(fin[q]);
            fin[q] = nullptr;
        }
        if (*cb_out) (void)hipFree(*cb_out);
        if (*xa1_out) (void)hipFree(*xa1_out);
        *cb_out = *xa1_out = nullptr;
        *candidates_out = -1;
    }
    for (int k = 0; k < na; ++k)
        if (a[k]) (void)hipFree(a[k]);
    if (crs) (void)hipFree(crs);
    if (cb) (void)hipFree(cb);
    if (norm) (void)hipFree(norm);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
}


namespace {

// the one-GPU handle's shapes for mg_place_finest
void place_finest(mg_solver *s, double *fin[4], double **cb_out, double **xa1_out)
{
    MgPlaceShape P;
    P.stream = s->stream;
    P.ny = s->gny, P.nx = s->gnx, P.ld = pitch(P.nx);
    P.cny = ((P.ny - 1) >> 1) + 1, P.cnx = ((P.nx - 1) >> 1) + 1, P.cld = pitch(P.cnx);
    P.j0 = 1, P.j1 = P.ny - 1, P.flo = 0, P.fhi = P.ny - 1, P.oj = 0, P.T0 = 1, P.T1 = P.cny - 1;
    P.h0 = s->h0;
    const i64 strips = (P.cnx + 107) / 108, blocks = ((strips + 3) / 4) * ((P.cny - 1 + 63) / 64);
    P.level1 = s->n_para >= 3 && P.cnx >= s->recompute_min_nx && blocks >= s->recompute_min_blocks; // level 1 runs the 9-stage pass
    P.c2ny = ((P.ny - 1) >> 2) + 1, P.c2nx = ((P.nx - 1) >> 2) + 1, P.c2ld = pitch(P.c2nx);
    P.j0_1 = 1, P.j1_1 = P.cny - 1, P.flo_1 = 0, P.fhi_1 = P.cny - 1;
    mg_place_finest(P, fin, cb_out, xa1_out, &s->placement_candidates);
}

} // namespace

extern "C" int mg_solver_create(int64_t gny, int64_t gnx, int n_para, double h0, int flags, mg_solver **out)
{
    MG_REQUIRE(out, "mg_solver_create: null output");
    *out = nullptr;
    MG_REQUIRE(n_para >= 1 && n_para < 40, "mg_solver_create: n_para %d outside 1..39", n_para);
    MG_REQUIRE(gny >= 3 && gnx >= 3 && gny < ((i64)1 << 30) && gnx < ((i64)1 << 30), "mg_solver_create: bad grid %lld x %lld",
               (long long)gny, (long long)gnx);
    MG_REQUIRE((gny - 1) % ((i64)1 << n_para) == 0 && (gnx - 1) % ((i64)1 << n_para) == 0,
               "mg_solver_create: rows-1 (%lld) and columns-1 (%lld) must be divisible by 2^n_para = %lld", (long long)(gny - 1),
               (long long)(gnx - 1), (long long)((i64)1 << n_para));
    MG_REQUIRE(((gny - 1) >> n_para) >= 2 && ((gnx - 1) >> n_para) >= 2,
               "mg_solver_create: the first whole-domain level needs at least 2 cells per side");
    MG_REQUIRE(h0 > 0, "mg_solver_create: mesh width must be positive");
    MG_REQUIRE((flags & ~(MG_SOLVER_FIX_COARSE_ASCENT | MG_SOLVER_REDBLACK | MG_SOLVER_NO_GRAPH | MG_SOLVER_LAZY_DEAD_LEVELS |
                          MG_SOLVER_TUNE_PLACEMENT | MG_SOLVER_NO_PLACEMENT)) == 0,
               "mg_solver_create: unknown flag bits in %d", flags);
    mg_solver *s = new (std::nothrow) mg_solver();
    MG_REQUIRE(s, "mg_solver_create: out of host memory");
    s->gny = gny, s->gnx = gnx, s->n_para = n_para, s->h0 = h0;
    s->fix_coarse_ascent = flags & MG_SOLVER_FIX_COARSE_ASCENT;
    s->ordering = (flags & MG_SOLVER_REDBLACK) ? MG_SOR_REDBLACK : MG_SOR_LEXICOGRAPHIC;
    s->use_graph = !(flags & MG_SOLVER_NO_GRAPH);
    s->eager_dead = !(flags & MG_SOLVER_LAZY_DEAD_LEVELS);
    s->omega = 1.0, s->max_cycles = 100; // :270
    s->recompute_min_nx = 1024, s->recompute_min_blocks = 1024;
    s->norm[0] = s->norm[1] = nullptr, s->host_norm[0] = s->host_norm[1] = nullptr, s->last_acc = 0;
    s->stream = s->side = s->copy = nullptr, s->fork = s->join = nullptr;
    s->filled[0] = s->filled[1] = s->fetched[0] = s->fetched[1] = nullptr;
    s->cur0 = 0, s->r_stale = true, s->it = 0, s->diverged = false, s->placement_candidates = -1;
    for (int a = 0; a < 2; ++a) s->graphs[a][0] = s->graphs[a][1] = nullptr, s->graph_end_cur[a] = 0;
    int rc = 1;
    do {
        if (hipGetDevice(&s->device) != hipSuccess) { mg_set_error("mg_solver_create: no current device"); break; }
        if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s->filled[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->filled[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->fetched[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->fetched[1], hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void **)&s->host_norm[0], MG_NORM_SLOTS * sizeof(u64), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&s->host_norm[1], MG_NORM_SLOTS * sizeof(u64), hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->join, hipEventDisableTiming) != hipSuccess) {
            mg_set_error("mg_solver_create: cannot create streams/events");
            break;
        }
        bool ok = true;
        double *fin[4] = {nullptr, nullptr, nullptr, nullptr};
        double *cb1 = nullptr, *xa1 = nullptr; // what the finest pass restricts into and level 1's x, when the placement selection made them
        if (!(flags & MG_SOLVER_NO_PLACEMENT)) { // returns at once for grids below 1 GiB per array
            place_finest(s, fin, &cb1, &xa1);
            for (int k = 0; k < 4; ++k)
                if (fin[k]) s->allocs.push_back(fin[k]);
            if (cb1) s->allocs.push_back(cb1);
            if (xa1) s->allocs.push_back(xa1);
        }
        for (int l = 0; l < n_para && ok; ++l) {
            JLevel L;
            L.ny = ((gny - 1) >> l) + 1, L.nx = ((gnx - 1) >> l) + 1, L.ld = pitch(L.nx);
            L.h = h0 * (double)((i64)1 << l); // h_sb, :195
            L.r = nullptr, L.recompute = false;
            if (l == 0 && fin[0]) { // placed (and left all zero) by place_finest
                L.xa = fin[0], L.xb = fin[1], L.b = fin[2], L.r = fin[3];
            } else {
                if (l == 1 && xa1)
                    L.xa = xa1;
                else
                    ok = !dev_alloc(s, &L.xa, L.ny * L.ld);
                ok = ok && !dev_alloc(s, &L.xb, L.ny * L.ld);
                if (ok && l == 1 && cb1)
                    L.b = cb1;
                else
                    ok = ok && !dev_alloc(s, &L.b, L.ny * L.ld);
                if (ok && l == 0) ok = !dev_alloc(s, &L.r, L.ny * L.ld);
            }
            s->lv.push_back(L);
        }
        for (int l = n_para; ok; ++l) { // whole-domain levels down to 3 x 3 (:207-219)
            const i64 my = (gny - 1) >> l, mx = (gnx - 1) >> l;
            if (my < 2 || mx < 2 || (my << l) != gny - 1 || (mx << l) != gnx - 1) break;
            WLevel W;
            W.ny = my + 1, W.nx = mx + 1, W.ld = pitch(W.nx), W.h = h0 * (double)((i64)1 << l);
            ok = !dev_alloc(s, &W.x, W.ny * W.ld) && !dev_alloc(s, &W.r, W.ny * W.ld);
            if (ok && l == 1 && cb1) // one Jacobi level: the finest pass restricts into the first whole-domain level
                W.b = cb1;
            else
                ok = ok && !dev_alloc(s, &W.b, W.ny * W.ld);
            s->wl.push_back(W);
        }
        if (!ok) break;
        double *nrm[2] = {nullptr, nullptr};
        if (dev_alloc(s, &nrm[0], MG_NORM_SLOTS) || dev_alloc(s, &nrm[1], MG_NORM_SLOTS)) break;
        s->norm[0] = (u64 *)nrm[0], s->norm[1] = (u64 *)nrm[1];
        if (hipStreamSynchronize(s->stream) != hipSuccess) { mg_set_error("mg_solver_create: device error"); break; }
        decide_recompute(s);
        rc = 0;
    } while (0);
    if (rc) {
        mg_solver_destroy(s);
        return rc;
    }
    *out = s;
    return 0;
}

extern "C" int mg_solver_destroy(mg_solver *s)
{
    if (!s) return 0;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->copy) (void)hipStreamSynchronize(s->copy);
    drop_graphs(s);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->fork) (void)hipEventDestroy(s->fork);
    if (s->join) (void)hipEventDestroy(s->join);
    for (int a = 0; a < 2; ++a) {
        if (s->filled[a]) (void)hipEventDestroy(s->filled[a]);
        if (s->fetched[a]) (void)hipEventDestroy(s->fetched[a]);
        if (s->host_norm[a]) (void)hipHostFree(s->host_norm[a]);
    }
    if (s->copy) (void)hipStreamDestroy(s->copy);
    if (s->side) (void)hipStreamDestroy(s->side);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return 0;
}

extern "C" int mg_solver_placement(mg_solver *s, int *candidates)
{
    if (check_device(s, "mg_solver_placement")) return 1;
    MG_REQUIRE(candidates, "mg_solver_placement: null output");
    *candidates = s->placement_candidates;
    return 0;
}

extern "C" int mg_solver_set(mg_solver *s, const char *key, double value)
{
    if (check_device(s, "mg_solver_set")) return 1;
    MG_REQUIRE(key, "mg_solver_set: null key");
    if (!strcmp(key, "jacobi_omega")) {
        MG_REQUIRE(value > 0 && value <= 1, "mg_solver_set: jacobi_omega %g outside (0, 1]", value);
        s->omega = value;
    } else if (!strcmp(key, "max_cycles")) {
        MG_REQUIRE(value >= 1 && value <= 1e6, "mg_solver_set: max_cycles outside 1..1e6");
        s->max_cycles = (int)value;
        return 0; // the captured graphs do not depend on it
    } else if (!strcmp(key, "recompute_min_nx")) {
        s->recompute_min_nx = (i64)value;
    } else if (!strcmp(key, "recompute_min_blocks")) {
        s->recompute_min_blocks = (i64)value;
    } else {
        mg_set_error("mg_solver_set: unknown key '%s'", key);
        return 1;
    }
    MG_HIP(hipStreamSynchronize(s->stream));
    drop_graphs(s); // they were recorded with the old setting
    decide_recompute(s);
    return 0;
}

static int solver_array(mg_solver *s, int which, int level, double **p, i64 *ny, i64 *nx, i64 *ld)
{
    const int nj = s->n_para, nw = (int)s->wl.size();
    if (which == MG_SOLVER_X || which == MG_SOLVER_B || which == MG_SOLVER_R) {
        MG_REQUIRE(level >= 0 && level < nj, "mg_solver: Jacobi level %d outside 0..%d", level, nj - 1);
        JLevel &L = s->lv[level];
        MG_REQUIRE(which != MG_SOLVER_R || level == 0, "mg_solver: only the finest level keeps a residual array");
        // after a cycle x of a coarse level is where its post-smoothing pass left it
        *p = which == MG_SOLVER_B ? L.b : which == MG_SOLVER_R ? L.r
             : level == 0 ? (s->cur0 ? L.xb : L.xa) : (L.recompute ? L.xa : L.xb);
        *ny = L.ny, *nx = L.nx, *ld = L.ld;
        return 0;
    }
    MG_REQUIRE(which == MG_SOLVER_XW || which == MG_SOLVER_BW || which == MG_SOLVER_RW, "mg_solver: unknown array id %d", which);
    MG_REQUIRE(level >= 0 && level < nw, "mg_solver: whole-domain level %d outside 0..%d", level, nw - 1);
    WLevel &W = s->wl[level];
    *p = which == MG_SOLVER_XW ? W.x : which == MG_SOLVER_BW ? W.b : W.r;
    *ny = W.ny, *nx = W.nx, *ld = W.ld;
    return 0;
}

extern "C" int mg_solver_array(mg_solver *s, int which, int level, double **device_ptr, int64_t *ny, int64_t *nx,
                               int64_t *ld)
{
    if (check_device(s, "mg_solver_array")) return 1;
    MG_REQUIRE(device_ptr && ny && nx && ld, "mg_solver_array: null output");
    if (which == MG_SOLVER_R && level == 0) {
        if (materialize_residual(s)) return 1;
    }
    i64 a, b, c;
    if (solver_array(s, which, level, device_ptr, &a, &b, &c)) return 1;
    *ny = a, *nx = b, *ld = c;
    MG_HIP(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int mg_solver_upload(mg_solver *s, int which, const double *src, int64_t src_ld)
{
    if (check_device(s, "mg_solver_upload")) return 1;
    MG_REQUIRE(which == MG_SOLVER_X || which == MG_SOLVER_B, "mg_solver_upload: only x and b of the finest level can be set");
    MG_REQUIRE(src && src_ld >= s->gnx, "mg_solver_upload: null source or src_ld < %lld", (long long)s->gnx);
    JLevel &L = s->lv[0];
    // both x arrays take the values, so that the Dirichlet cells agree whichever a pass writes
    double *dst[2] = {which == MG_SOLVER_B ? L.b : L.xa, which == MG_SOLVER_B ? nullptr : L.xb};
    for (double *d : dst)
        if (d)
            MG_HIP(hipMemcpy2DAsync(d, (size_t)L.ld * 8, src, (size_t)src_ld * 8, (size_t)L.nx * 8, (size_t)L.ny, hipMemcpyDefault,
                                    s->stream));
    MG_HIP(hipStreamSynchronize(s->stream));
    s->r_stale = true;
    return 0;
}

extern "C" int mg_solver_download(mg_solver *s, int which, int level, double *dst, int64_t dst_ld)
{
    if (check_device(s, "mg_solver_download")) return 1;
    MG_REQUIRE(dst, "mg_solver_download: null destination");
    if (which == MG_SOLVER_R && level == 0) {
        if (materialize_residual(s)) return 1;
    }
    double *p;
    i64 ny, nx, ld;
    if (solver_array(s, which, level, &p, &ny, &nx, &ld)) return 1;
    MG_REQUIRE(dst_ld >= nx, "mg_solver_download: dst_ld %lld < %lld columns", (long long)dst_ld, (long long)nx);
    MG_HIP(hipMemcpy2DAsync(dst, (size_t)dst_ld * 8, p, (size_t)ld * 8, (size_t)nx * 8, (size_t)ny, hipMemcpyDefault, s->stream));
    MG_HIP(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int mg_solver_presmooth(mg_solver *s)
{
    if (check_device(s, "mg_solver_presmooth")) return 1;
    return finest_pass(s, 1, nullptr, 0, 0, 0, false, false); // :264-266
}

extern "C" int mg_solver_cycles(mg_solver *s, int k, double *errs)
{
    if (check_device(s, "mg_solver_cycles")) return 1;
    MG_REQUIRE(k >= 0, "mg_solver_cycles: negative cycle count");
    // Every cycle's norm is read, but one cycle late: cycle c fills the accumulator its starting array selects, its
    // read-back is queued on the copy stream, and the host waits for the number only after it has enqueued cycle
    // c + 1 (which fills the other accumulator).  Should two cycles in a row fill the same accumulator, the first is
    // collected before the second is enqueued.
    int pending = -1, pending_cycle = -1;
    for (int c = 0; c < k; ++c) {
        if (pending >= 0 && pending == s->cur0) { // the next cycle resets the accumulator still to be collected
            double e;
            if (collect_err(s, pending, &e)) return 1;
            if (errs) errs[pending_cycle] = e;
            pending = -1;
        }
        if (run_cycle(s, c == k - 1)) return 1; // the last pass leaves the residual array current
        const int a = s->last_acc;
        if (fetch_err(s, a)) return 1;
        if (pending >= 0) {
            double e;
            if (collect_err(s, pending, &e)) return 1;
            if (errs) errs[pending_cycle] = e;
        }
        pending = a, pending_cycle = c;
    }
    if (pending >= 0) {
        double e;
        if (collect_err(s, pending, &e)) return 1;
        if (errs) errs[pending_cycle] = e;
    }
    MG_HIP(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int mg_solver_solve(mg_solver *s, double eps, int *it_out, double *err_out)
{
    if (check_device(s, "mg_solver_solve")) return 1;
    s->it = 0, s->diverged = false;
    s->hist.clear();
    double err = 0.0, err_old = 100.0; // :258-259
    if (mg_solver_presmooth(s)) return 1;
    bool done = false;
    while (!done && s->it < s->max_cycles) {
        if (run_cycle(s, false)) return 1;
        if (read_err(s, &err)) return 1;
        s->hist.push_back(err);
        if (err > err_old * 1.1) s->diverged = true; // :337-340
        err_old = err;
        done = !(err > eps && !s->diverged);
        s->it += 1;
    }
    if (materialize_residual(s)) return 1;
    MG_HIP(hipStreamSynchronize(s->stream));
    if (it_out) *it_out = s->it;
    if (err_out) *err_out = err;
    return 0;
}

extern "C" int mg_solver_history(mg_solver *s, double *errs, int capacity, int *count, int *diverged)
{
    MG_REQUIRE(s, "mg_solver_history: null solver");
    const int n = (int)s->hist.size();
    for (int k = 0; k < n && k < capacity && errs; ++k) errs[k] = s->hist[k];
    if (count) *count = n;
    if (diverged) *diverged = s->diverged;
    return 0;
}

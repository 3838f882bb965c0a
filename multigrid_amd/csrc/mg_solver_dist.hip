// The V-cycle solver behind a C handle on SEVERAL GPUs, one process (rank) per GPU: Multigrid.__init__/solve() of the
// reference (core/parallel_multigrid.py:153-348) on the global grid cut into row slabs, for callers that do not go
// through Python.  What the reference's four MPI ranks exchange per sweep through Buffers.exchange_buffers (:83-97)
// and gather through four broadcasts (:117-138) is here
//   * one prepared RCCL send/receive group per fused pass of a distributed level (mg_halo_exchange: `depth` whole rows
//     to and from the rank above and the rank below),
//   * one all-gather of the first replicated level's right-hand side, after which every rank computes the remaining
//     Jacobi levels and the whole-domain levels alike, without communication (as the reference replicates its
//     whole-domain levels), and takes its row window of the result as its correction,
//   * one MAX-reduction of the norm accumulator per cycle (the stop test's allreduce, :342-345).
// The schedule is the one of the package's slab engine (multigrid_amd/engine.py) with its passes uncut: every RCCL call
// is issued on the solver's main stream, so that hipGraph capture records them like kernels (RCCL calls on a stream
// that joined a capture through an event are not recorded correctly by RCCL 2.26 / HIP 7.0) and a whole cycle of a
// rank -- passes, exchanges, all-gather -- is ONE graph launch.  Values are those of the reference's schedule on the
// undecomposed grid, bit for bit, for every number of ranks: Jacobi, residual and the transfer operators are order-free.
#include <string.h>

#include <map>
#include <new>
#include <vector>

#include "mg_common.h"

namespace {

const int N1 = 3, N2 = 6; // core/parallel_multigrid.py:255-256
const int HALO = 12;      // rows per side: as deep as the deepest pass (3 + 6 sweeps from zero with the prolongation between: 10), even

struct DLevel { // a Jacobi level (or the staging level behind the last) as this rank holds it
    bool dist;  // cut into slabs (else every rank holds the whole grid)
    i64 rows;   // owned rows (a replicated level: all but the bottom boundary row)
    i64 ny, nx, ld;
    double h;
    i64 t0, t1, flo, fhi; // rows a pass produces; Dirichlet rows (see SlabLevel in multigrid_amd/levels.py)
    double *xa, *xb, *b, *r;
    bool recompute;
};

struct WLevel {
    i64 ny, nx, ld;
    double h;
    double *x, *b, *r;
};

struct Item { // one array of a halo exchange
    double *arr;
    int level; // its (distributed) level
    int depth;
};

i64 pitch(i64 nx) { return (nx + 15) / 16 * 16; }

} // namespace

struct mg_dist_solver {
    int device, rank, world, n_para, n_dist;
    bool loopback; // rehearsal of one rank alone: every peer is this process (see mg_dist_create)
    mg_comm *comm;
    i64 gny, gnx;
    double h0, omega;
    bool fix_coarse_ascent, use_graph, eager_dead;
    int ordering, max_cycles;
    i64 recompute_min_nx, recompute_min_blocks;
    std::vector<DLevel> lv;      // 0 .. n_para (the last one: the staging level, b only)
    std::vector<DLevel> slab;    // the slab shape of levels 0 .. n_dist (what level n_dist's right-hand side is restricted into)
    std::vector<WLevel> wl;
    double *bslab, *xw0_pad, *rw0_pad;
    std::vector<void *> allocs;
    std::map<std::vector<std::pair<void *, int>>, mg_halo *> halos;
    u64 *norm[2], *host_norm[2];
    hipStream_t stream, side, copy;
    hipEvent_t fork, join, filled[2], fetched[2];
    hipGraphExec_t graphs[2][2];
    int graph_end_cur[2];
    int cur0, last_acc, it;
    bool r_stale, b0_halo_valid, diverged;
    int placement_candidates; // candidate arrays the placement selection timed, -1: none
    i64 rounds;           // exchange rounds and all-gathers carried out so far
    i64 rounds_per_graph; // ... that one replay of a captured cycle carries out
    std::vector<double> hist;
};

namespace {

int dev_alloc(mg_dist_solver *s, double **p, i64 n)
{
    void *q = nullptr;
    MG_HIP(hipMalloc(&q, (size_t)n * sizeof(double)));
    s->allocs.push_back(q);
    MG_HIP(hipMemsetAsync(q, 0, (size_t)n * sizeof(double), s->stream));
    *p = (double *)q;
    return 0;
}

int check_device(const mg_dist_solver *s, const char *fn)
{
    MG_REQUIRE(s, "%s: null solver", fn);
    int d = -1;
    MG_HIP(hipGetDevice(&d));
    MG_REQUIRE(d == s->device, "%s: the solver lives on device %d, the current device is %d", fn, s->device, d);
    return 0;
}

// One Jacobi level of a (gny x gnx) grid cut into `world` slabs, on rank `rank` (world = 1: held whole), with HALO rows
// above and below the owned ones: local row t = global row rank * rows - HALO + t.
DLevel make_level(i64 gny, i64 gnx, int l, int world, int rank, double h0, bool dist)
{
    DLevel L;
    memset(&L, 0, sizeof(L));
    L.dist = dist;
    const i64 R0 = (gny - 1) / world;
    L.rows = R0 >> l;
    L.nx = ((gnx - 1) >> l) + 1;
    L.ld = pitch(L.nx);
    L.ny = L.rows + 2 * HALO;
    L.h = h0 * (double)((i64)1 << l);
    const bool last = rank == world - 1;
    L.t0 = rank == 0 ? HALO + 1 : HALO;
    L.t1 = HALO + L.rows;
    L.flo = rank == 0 ? HALO : -1;
    L.fhi = last ? HALO + L.rows : L.ny;
    return L;
}

void decide_recompute(mg_dist_solver *s)
{
    for (int l = 0; l < s->n_para; ++l) {
        DLevel &L = s->lv[l];
        const i64 strips = (L.nx + 107) / 108, blocks = ((strips + 3) / 4) * ((L.rows + 63) / 64);
        L.recompute = l >= 1 && L.nx >= s->recompute_min_nx && blocks >= s->recompute_min_blocks;
        // the last distributed level always: its correction is a window of the replicated level below, halo included,
        // so that nothing is exchanged in front of its ascent pass
        if (l >= 1 && s->world > 1 && l + 1 == s->n_dist) L.recompute = true;
    }
}

int b_halo_rows(const mg_dist_solver *s, int l) { return s->lv[l].recompute ? N1 + N2 : (N1 + 2 > N2 ? N1 + 2 : N2); }

// How many of the Jacobi levels are cut into slabs: the cost model of multigrid_amd/levels.py (choose_n_dist) -- two
// exchange rounds per distributed level against the all-gather of the first replicated level and the redundant work
// on the replicated ones.
int choose_n_dist(i64 gny, i64 gnx, int n_para, int world, double t_round, double gather_bw)
{
    const i64 R0 = (gny - 1) / world;
    int best = 1;
    double best_cost = -1;
    auto pts = [&](int l) { return (double)(((gny - 1) >> l) + 1) * (double)(((gnx - 1) >> l) + 1); };
    for (int k = 1; k <= n_para; ++k) {
        if ((R0 >> (k - 1)) < HALO || R0 % ((i64)1 << k)) break;
        double cost = 2 * k * t_round + 8 * pts(k) * (world - 1) / world / gather_bw;
        double rep = 0;
        for (int l = k; l < n_para; ++l) rep += pts(l);
        cost += rep * 36 / 4.5e12;
        if (best_cost < 0 || cost < best_cost) best = k, best_cost = cost;
    }
    return best;
}

// Fill `depth` halo rows of every array of `items` from the neighbouring ranks, all in one RCCL group: my upper halo <-
// the last owned rows of the rank above, my lower halo <- the first owned rows of the rank below (`up` / `down` select
// which of my halos are filled).  Whole rows, padding included: every message is one contiguous block.  The group is
// prepared on first use and kept (the arrays never move).
int halo_exchange(mg_dist_solver *s, const std::vector<Item> &items, bool up = true, bool down = true)
{
    if (s->world == 1 || items.empty()) return 0;
    std::vector<std::pair<void *, int>> key;
    for (const Item &it : items) key.push_back({(void *)it.arr, it.depth});
    key.push_back({nullptr, (up ? 1 : 0) | (down ? 2 : 0)});
    auto found = s->halos.find(key);
    mg_halo *h = found == s->halos.end() ? nullptr : found->second;
    if (!h) {
        std::vector<double *> sp, rp;
        std::vector<int64_t> sc, rc;
        std::vector<int> sr, rr;
        for (const Item &it : items) {
            const DLevel &L = s->slab[it.level];
            const i64 H = HALO, R = L.rows, ld = L.ld, n = it.depth * ld;
            MG_REQUIRE(it.depth >= 1 && it.depth <= H && it.depth <= R, "mg_dist: exchange of %d rows on a level with %lld rows per rank",
                       it.depth, (long long)R);
            const int above = s->loopback ? 0 : s->rank - 1, below = s->loopback ? 0 : s->rank + 1;
            if (s->rank > 0) {
                if (down) sp.push_back(it.arr + H * ld), sc.push_back(n), sr.push_back(above);
                if (up) rp.push_back(it.arr + (H - it.depth) * ld), rc.push_back(n), rr.push_back(above);
            }
            if (s->rank < s->world - 1) {
                if (up) sp.push_back(it.arr + (H + R - it.depth) * ld), sc.push_back(n), sr.push_back(below);
                if (down) rp.push_back(it.arr + (H + R) * ld), rc.push_back(n), rr.push_back(below);
            }
        }
        if (s->loopback) { // one rank alone: equally many messages each way
            const size_t n = sp.size() < rp.size() ? sp.size() : rp.size();
            sp.resize(n), sc.resize(n), sr.resize(n), rp.resize(n), rc.resize(n), rr.resize(n);
        }
        if (mg_halo_prepare(s->comm, (int)sp.size(), sp.data(), sc.data(), sr.data(), (int)rp.size(), rp.data(), rc.data(), rr.data(), &h))
            return 1;
        s->halos[key] = h;
    }
    s->rounds += 1;
    return mg_halo_exchange(h, s->stream);
}

// every rank's owned rows of the slab right-hand side of level lc -> the replicated b of that level
int allgather_b(mg_dist_solver *s, int lc)
{
    const DLevel &S = s->slab[lc];
    double *out = s->lv[lc].b + HALO * S.ld;
    const double *in = s->bslab + HALO * S.ld;
    const i64 n = S.rows * S.ld;
    s->rounds += 1;
    if (!s->loopback) return mg_allgather_rows(s->comm, in, out, n, s->stream);
    // one rank alone: its rows into its place through the one-rank communicator, then into the other ranks' places
    if (mg_allgather_rows(s->comm, in, out + (i64)s->rank * n, n, s->stream)) return 1;
    for (int q = 0; q < s->world; ++q)
        if (q != s->rank)
            MG_HIP(hipMemcpyAsync(out + (i64)q * n, out + (i64)s->rank * n, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    return 0;
}

// where level l restricts into: the next level's right-hand side, or (last distributed level) the slab that is all-gathered
struct CoarseB {
    double *p;
    i64 ny, nx, ld, T0, T1;
};
CoarseB coarse_b_of(mg_dist_solver *s, int l)
{
    const bool to_rep = s->world > 1 && l < s->n_dist && l + 1 == s->n_dist;
    const DLevel &C = to_rep ? s->slab[l + 1] : s->lv[l + 1];
    return CoarseB{to_rep ? s->bslab : s->lv[l + 1].b, C.ny, C.nx, C.ld, C.t0, C.t1};
}

int after_restriction(mg_dist_solver *s, int l)
{
    if (s->world == 1 || l >= s->n_dist) return 0;
    if (l + 1 == s->n_dist) return allgather_b(s, l + 1);
    return halo_exchange(s, {Item{s->lv[l + 1].b, l + 1, b_halo_rows(s, l + 1)}});
}

// this rank's slab of a replicated level-lc array: its rows and the halo rows around them (one rank: the array itself)
const double *window(const mg_dist_solver *s, const double *arr, int lc)
{
    return s->world == 1 ? arr : arr + (i64)s->rank * s->slab[lc].rows * s->slab[lc].ld;
}
i64 window_ny(const mg_dist_solver *s, int lc) { return s->world == 1 ? s->lv[lc].ny : s->slab[lc].ny; }

// descent on level l >= 1: x = 0, N1 sweeps, residual, restriction (:274-289)
int descend(mg_dist_solver *s, int l)
{
    DLevel &L = s->lv[l];
    const CoarseB cb = coarse_b_of(s, l);
    if (mg_smooth_fused_restrict(L.b, nullptr, L.recompute ? nullptr : L.xa, nullptr, nullptr, L.ny, L.nx, L.ld, 0, s->lv[l + 1].nx, 0,
                                 L.t0, L.t1, L.flo, L.fhi, -HALO, N1, L.h, s->omega, nullptr, 0, cb.p, cb.ny, cb.nx, cb.ld, cb.T0, cb.T1,
                                 1, 0, 0, 0, s->stream))
        return 1;
    return after_restriction(s, l);
}

// ascent on level l >= 1: + P(crs), N2 sweeps (:309-333); *out = the array that holds the result
int ascend(mg_dist_solver *s, int l, const double *crs, i64 cny, bool crs_has_halo, double **out)
{
    DLevel &L = s->lv[l];
    const DLevel &C = s->lv[l + 1];
    const bool dist = s->world > 1 && l < s->n_dist;
    if (L.recompute) {
        if (dist && !crs_has_halo)
            if (halo_exchange(s, {Item{(double *)crs, l + 1, (N1 + N2) / 2 + 2}})) return 1;
        if (mg_smooth_fused(L.b, nullptr, L.xa, nullptr, crs, L.ny, L.nx, L.ld, cny, C.nx, C.ld, L.t0, L.t1, L.flo, L.fhi, -HALO,
                            N1 + N2, N1, L.h, s->omega, nullptr, 1, 0, 0, s->stream))
            return 1;
        *out = L.xa;
        return 0;
    }
    if (dist) {
        std::vector<Item> items{Item{L.xa, l, N2}};
        if (!crs_has_halo) items.push_back(Item{(double *)crs, l + 1, N2 / 2 + 2});
        if (halo_exchange(s, items)) return 1;
    }
    if (mg_smooth_fused(L.b, L.xa, L.xb, nullptr, crs, L.ny, L.nx, L.ld, cny, C.nx, C.ld, L.t0, L.t1, L.flo, L.fhi, -HALO, N2, 0, L.h,
                        s->omega, nullptr, 1, 0, 0, s->stream))
        return 1;
    *out = L.xb;
    return 0;
}

// the halo rows of the finest right-hand side: exchanged once after the caller has set it (never inside a captured cycle,
// which is replayed for right-hand sides to come)
int ensure_b0_halo(mg_dist_solver *s)
{
    if (s->b0_halo_valid) return 0;
    if (halo_exchange(s, {Item{s->lv[0].b, 0, HALO - 1}})) return 1;
    s->b0_halo_valid = true;
    return 0;
}

// smooth_parallel on the finest level (:100-114) with the prolongation in front and the restriction of its residual behind
int finest_pass(mg_dist_solver *s, int sweeps, const double *crs, i64 cny, bool crs_has_halo, bool want_norm, bool store_r)
{
    DLevel &L = s->lv[0];
    const DLevel &C = s->lv[1];
    double *xs[2] = {L.xa, L.xb};
    if (ensure_b0_halo(s)) return 1;
    if (s->world > 1) {
        std::vector<Item> items{Item{xs[s->cur0], 0, sweeps + 2}};
        if (crs && !crs_has_halo) items.push_back(Item{(double *)crs, 1, (sweeps + 2) / 2 + 2});
        if (halo_exchange(s, items)) return 1;
    }
    u64 *const acc = s->norm[s->cur0];
    if (want_norm) s->last_acc = s->cur0;
    if (want_norm && mg_norm_reset((uint64_t *)acc, s->stream)) return 1;
    const CoarseB cb = coarse_b_of(s, 0);
    if (mg_smooth_fused_restrict(L.b, xs[s->cur0], xs[1 - s->cur0], store_r ? L.r : nullptr, crs, L.ny, L.nx, L.ld, crs ? cny : 0, C.nx,
                                 crs ? C.ld : 0, L.t0, L.t1, L.flo, L.fhi, -HALO, sweeps, L.h, s->omega, want_norm ? (uint64_t *)acc : nullptr,
                                 0, cb.p, cb.ny, cb.nx, cb.ld, cb.T0, cb.T1, 1, 0, 0, 0, s->stream))
        return 1;
    s->cur0 = 1 - s->cur0;
    s->r_stale = !store_r;
    return after_restriction(s, 0);
}

int tail_below(mg_dist_solver *s, hipStream_t st)
{
    const int nw = (int)s->wl.size();
    for (int i = 1; i < nw; ++i) {
        WLevel &W = s->wl[i], &F = s->wl[i - 1];
        if (mg_zero(W.x, W.ny * W.ld, st)) return 1;
        if (mg_restrict(F.r, F.ny, F.nx, F.ld, W.b, W.ny, W.nx, W.ld, 1, W.ny - 1, 0, 0, 1, 0, 0, st)) return 1;
        if (mg_sor(W.b, W.x, W.r, W.ny, W.nx, W.ld, W.h, N1, 1, s->ordering, 1, 0, st)) return 1;
    }
    for (int i = 2; i < nw + (s->fix_coarse_ascent ? 1 : 0); ++i) { // the reference's range(2, nw), :302
        WLevel &C = s->wl[nw - i + 1], &F = s->wl[nw - i];
        if (mg_prolong_add(C.x, C.ny, C.nx, C.ld, F.x, F.ny, F.nx, F.ld, 0, F.ny, 1, F.nx - 2, 0, 0, 1, 0, 0, st)) return 1;
        if (mg_sor(F.b, F.x, F.r, F.ny, F.nx, F.ld, F.h, N2, 1, s->ordering, 1, 0, st)) return 1;
    }
    return 0;
}

// the body of the loop at :270-347
int cycle(mg_dist_solver *s, bool store_r)
{
    const int n_para = s->n_para, k = s->n_dist;
    hipStream_t st = s->stream;
    for (int l = 1; l < k; ++l) // distributed levels; the last one ends with the all-gather
        if (descend(s, l)) return 1;
    for (int l = k; l < n_para; ++l) // replicated levels: every rank alike
        if (descend(s, l)) return 1;
    WLevel &W0 = s->wl[0];
    const DLevel &S = s->lv[n_para];
    if (mg_zero(s->xw0_pad, S.ny * S.ld, st)) return 1;
    if (mg_sor(W0.b, W0.x, W0.r, W0.ny, W0.nx, W0.ld, W0.h, N1, 1, s->ordering, 1, 0, st)) return 1;
    if (s->fix_coarse_ascent && tail_below(s, st)) return 1;
    double *corr = s->xw0_pad;
    i64 cny = S.ny;
    for (int l = n_para - 1; l >= k; --l) {
        if (ascend(s, l, corr, cny, true, &corr)) return 1;
        cny = s->lv[l].ny;
    }
    bool forked = false;
    if (!s->fix_coarse_ascent && s->eager_dead && s->wl.size() > 1) {
        // nothing in the cycle reads the whole-domain levels below the first: they run beside the distributed ascent, on the
        // second stream, from this point of the main sequence on, and are enqueued after it (captured in that order the graph
        // starts the ascent at once)
        MG_HIP(hipEventRecord(s->fork, st));
        forked = true;
    }
    for (int l = k - 1; l >= 1; --l) {
        const bool rep = l + 1 == k; // the correction is this rank's window of a replicated array, halo rows included
        const double *c = rep ? window(s, corr, l + 1) : corr;
        if (ascend(s, l, c, rep ? window_ny(s, l + 1) : cny, rep, &corr)) return 1;
        cny = s->lv[l].ny;
    }
    const bool rep0 = k == 1;
    const double *c0 = rep0 ? window(s, corr, 1) : corr;
    if (finest_pass(s, N2, c0, rep0 ? window_ny(s, 1) : cny, rep0, true, store_r)) return 1;
    if (forked) {
        MG_HIP(hipStreamWaitEvent(s->side, s->fork, 0));
        if (tail_below(s, s->side)) return 1;
        MG_HIP(hipEventRecord(s->join, s->side));
        MG_HIP(hipStreamWaitEvent(st, s->join, 0));
    }
    return 0;
}

void drop_graphs(mg_dist_solver *s)
{
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            if (s->graphs[a][b]) {
                (void)hipGraphExecDestroy(s->graphs[a][b]);
                s->graphs[a][b] = nullptr;
            }
}

int run_cycle(mg_dist_solver *s, bool store_r)
{
    if (ensure_b0_halo(s)) return 1;
    if (!s->use_graph) return cycle(s, store_r);
    if (!s->graphs[s->cur0][store_r]) {
        const int cur = s->cur0, acc = s->last_acc;
        const bool stale = s->r_stale;
        const i64 rounds = s->rounds;
        s->rounds_per_graph = -1;
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
                if (s->rounds_per_graph < 0) s->rounds_per_graph = s->rounds - rounds; // every variant records as many
            }
        s->cur0 = cur, s->r_stale = stale, s->last_acc = acc, s->rounds = rounds;
    }
    if (mg_graph_launch(s->graphs[s->cur0][store_r], s->stream)) return 1;
    s->rounds += s->rounds_per_graph;
    s->last_acc = s->cur0;
    s->cur0 = s->graph_end_cur[s->cur0];
    s->r_stale = !store_r;
    return 0;
}

// the MAX over the ranks of the accumulator the last cycle filled (bit patterns of |r|: an integer MAX, NaN on top)
int reduce_err(mg_dist_solver *s)
{
    if (s->world == 1) return 0;
    return mg_allreduce_max(s->comm, (uint64_t *)s->norm[s->last_acc], MG_NORM_SLOTS, s->stream);
}

int materialize_residual(mg_dist_solver *s)
{
    if (!s->r_stale) return 0;
    DLevel &L = s->lv[0];
    double *x = s->cur0 ? L.xb : L.xa;
    if (halo_exchange(s, {Item{x, 0, 1}})) return 1;
    if (mg_residual(L.r, x, L.b, L.ny, L.nx, L.ld, L.t0, L.t1, L.h, nullptr, 1, 0, MG_FLAG_FINEST, s->stream)) return 1;
    s->r_stale = false;
    return 0;
}

int fetch_err(mg_dist_solver *s, int a)
{
    MG_HIP(hipEventRecord(s->filled[a], s->stream));
    MG_HIP(hipStreamWaitEvent(s->copy, s->filled[a], 0));
    MG_HIP(hipMemcpyAsync(s->host_norm[a], s->norm[a], MG_NORM_SLOTS * sizeof(u64), hipMemcpyDeviceToHost, s->copy));
    MG_HIP(hipEventRecord(s->fetched[a], s->copy));
    return 0;
}

int collect_err(mg_dist_solver *s, int a, double *err)
{
    MG_HIP(hipEventSynchronize(s->fetched[a]));
    u64 m = 0;
    for (int k = 0; k < MG_NORM_SLOTS; ++k) m = s->host_norm[a][k] > m ? s->host_norm[a][k] : m;
    union { u64 u; double d; } cv;
    cv.u = m;
    *err = cv.d;
    return 0;
}

} // namespace

extern "C" int mg_dist_destroy(mg_dist_solver *s)
{
    if (!s) return 0;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->copy) (void)hipStreamSynchronize(s->copy);
    drop_graphs(s);
    for (auto &kv : s->halos) (void)mg_halo_destroy(kv.second);
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

extern "C" int mg_dist_create(int64_t gny, int64_t gnx, int n_para, double h0, int flags, mg_comm *comm, int rank, int world,
                              int dist_levels, mg_dist_solver **out)
{
    MG_REQUIRE(out, "mg_dist_create: null output");
    *out = nullptr;
    MG_REQUIRE(n_para >= 1 && n_para < 40, "mg_dist_create: n_para %d outside 1..39", n_para);
    MG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mg_dist_create: rank %d of %d", rank, world);
    MG_REQUIRE(gny >= 3 && gnx >= 3 && gny < ((i64)1 << 30) && gnx < ((i64)1 << 30), "mg_dist_create: bad grid %lld x %lld", (long long)gny,
               (long long)gnx);
    MG_REQUIRE((gny - 1) % world == 0, "mg_dist_create: %lld rows of cells cannot be split evenly over %d ranks", (long long)(gny - 1), world);
    MG_REQUIRE((gny - 1) % ((i64)1 << n_para) == 0 && (gnx - 1) % ((i64)1 << n_para) == 0,
               "mg_dist_create: rows-1 (%lld) and columns-1 (%lld) must be divisible by 2^n_para", (long long)(gny - 1), (long long)(gnx - 1));
    MG_REQUIRE(((gny - 1) >> n_para) >= 2 && ((gnx - 1) >> n_para) >= 2, "mg_dist_create: the first whole-domain level needs at least 2 cells per side");
    MG_REQUIRE(h0 > 0, "mg_dist_create: mesh width must be positive");
    MG_REQUIRE((flags & ~(MG_SOLVER_FIX_COARSE_ASCENT | MG_SOLVER_REDBLACK | MG_SOLVER_NO_GRAPH | MG_SOLVER_LAZY_DEAD_LEVELS | MG_SOLVER_NO_PLACEMENT |
                          MG_DIST_LOOPBACK)) == 0,
               "mg_dist_create: unknown flag bits in %d", flags);
    const bool loopback = flags & MG_DIST_LOOPBACK;
    int crank = 0, cworld = 1;
    if (world > 1) {
        MG_REQUIRE(comm, "mg_dist_create: %d ranks need a communicator", world);
        if (mg_comm_info(comm, &crank, &cworld, nullptr)) return 1;
        MG_REQUIRE(loopback ? cworld == 1 : (cworld == world && crank == rank),
                   "mg_dist_create: the communicator says rank %d of %d, the caller rank %d of %d%s", crank, cworld, rank, world,
                   loopback ? " (a rehearsal of one rank alone takes a one-rank communicator)" : "");
    }
    const i64 R0 = (gny - 1) / world;
    int n_dist = 1; // one rank: the finest level counts as the (only) level in front of the replicated ones
    if (world > 1) {
        n_dist = dist_levels > 0 ? (dist_levels < n_para ? dist_levels : n_para) : choose_n_dist(gny, gnx, n_para, world, 110e-6, 100e9);
        MG_REQUIRE(R0 % ((i64)1 << n_dist) == 0, "mg_dist_create: rows per rank (%lld) must be divisible by 2^%d (slab boundaries on even rows of "
                   "every distributed level and of the first replicated one)", (long long)R0, n_dist);
        MG_REQUIRE((R0 >> (n_dist - 1)) >= HALO, "mg_dist_create: the coarsest distributed level has %lld rows per rank, fewer than the halo depth %d",
                   (long long)(R0 >> (n_dist - 1)), HALO);
    }
    mg_dist_solver *s = new (std::nothrow) mg_dist_solver();
    MG_REQUIRE(s, "mg_dist_create: out of host memory");
    s->rank = rank, s->world = world, s->n_para = n_para, s->n_dist = n_dist, s->loopback = loopback, s->comm = comm;
    s->gny = gny, s->gnx = gnx, s->h0 = h0, s->omega = 1.0, s->max_cycles = 100;
    s->fix_coarse_ascent = flags & MG_SOLVER_FIX_COARSE_ASCENT;
    s->ordering = (flags & MG_SOLVER_REDBLACK) ? MG_SOR_REDBLACK : MG_SOR_LEXICOGRAPHIC;
    s->use_graph = !(flags & MG_SOLVER_NO_GRAPH);
    s->eager_dead = !(flags & MG_SOLVER_LAZY_DEAD_LEVELS);
    s->recompute_min_nx = 1024, s->recompute_min_blocks = 1024;
    s->bslab = s->xw0_pad = s->rw0_pad = nullptr;
    s->norm[0] = s->norm[1] = nullptr, s->host_norm[0] = s->host_norm[1] = nullptr;
    s->stream = s->side = s->copy = nullptr, s->fork = s->join = nullptr;
    s->filled[0] = s->filled[1] = s->fetched[0] = s->fetched[1] = nullptr;
    for (int a = 0; a < 2; ++a) s->graphs[a][0] = s->graphs[a][1] = nullptr, s->graph_end_cur[a] = 0;
    s->cur0 = 0, s->last_acc = 0, s->it = 0, s->r_stale = true, s->b0_halo_valid = false, s->diverged = false, s->rounds = 0, s->rounds_per_graph = 0, s->placement_candidates = -1;
    int rc = 1;
    do {
        if (hipGetDevice(&s->device) != hipSuccess) { mg_set_error("mg_dist_create: no current device"); break; }
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
            mg_set_error("mg_dist_create: cannot create streams/events");
            break;
        }
        bool ok = true;
        for (int l = 0; l <= (world > 1 ? n_dist : -1); ++l) s->slab.push_back(make_level(gny, gnx, l, world, rank, h0, true));
        // Where the finest level's arrays lie decides how fast the finest pass runs (see mg_place_finest): on grids of at least
        // 1 GiB per array create chooses them among candidate allocations, timed as the outputs of that pass, and keeps the
        // array the pass restricts into (level 1's right-hand side, or the slab that is all-gathered) and level 1's x with them.
        double *fin[4] = {nullptr, nullptr, nullptr, nullptr}, *cb1 = nullptr, *xa1 = nullptr;
        const bool slab1 = world > 1 && n_dist == 1; // the finest pass restricts into the slab that is all-gathered
        if (!(flags & MG_SOLVER_NO_PLACEMENT)) {
            const DLevel F = make_level(gny, gnx, 0, world, rank, h0, true), C = make_level(gny, gnx, 1, world, rank, h0, true),
                         C2 = make_level(gny, gnx, 2, world, rank, h0, true);
            MgPlaceShape P;
            P.stream = s->stream;
            P.ny = F.ny, P.nx = F.nx, P.ld = F.ld, P.cny = C.ny, P.cnx = C.nx, P.cld = C.ld;
            P.j0 = F.t0, P.j1 = F.t1, P.flo = F.flo, P.fhi = F.fhi, P.oj = -HALO, P.T0 = C.t0, P.T1 = C.t1;
            P.h0 = h0;
            const i64 strips = (C.nx + 107) / 108, blocks = ((strips + 3) / 4) * ((C.rows + 63) / 64);
            // level 1's x is chosen with them where level 1 is a slab like the finest level and runs the 9-stage pass
            P.level1 = n_para >= 3 && (world == 1 || n_dist >= 2) &&
                       ((C.nx >= s->recompute_min_nx && blocks >= s->recompute_min_blocks) || (world > 1 && n_dist == 2));
            P.c2ny = C2.ny, P.c2nx = C2.nx, P.c2ld = C2.ld, P.j0_1 = C.t0, P.j1_1 = C.t1, P.flo_1 = C.flo, P.fhi_1 = C.fhi;
            mg_place_finest(P, fin, &cb1, &xa1, &s->placement_candidates);
            for (int q = 0; q < 4; ++q)
                if (fin[q]) s->allocs.push_back(fin[q]);
            if (cb1) s->allocs.push_back(cb1);
            if (xa1) s->allocs.push_back(xa1);
        }
        for (int l = 0; l <= n_para && ok; ++l) {
            const bool dist = world > 1 && l < n_dist;
            DLevel L = dist ? make_level(gny, gnx, l, world, rank, h0, true) : make_level(gny, gnx, l, 1, 0, h0, false);
            if (l == 0 && fin[0]) { // placed (and left all zero) by mg_place_finest
                L.xa = fin[0], L.xb = fin[1], L.b = fin[2], L.r = fin[3];
            } else {
                if (l < n_para) {
                    if (l == 1 && xa1 && !slab1)
                        L.xa = xa1;
                    else
                        ok = !dev_alloc(s, &L.xa, L.ny * L.ld);
                    ok = ok && !dev_alloc(s, &L.xb, L.ny * L.ld);
                }
                if (ok && l == 1 && cb1 && !slab1)
                    L.b = cb1;
                else
                    ok = ok && !dev_alloc(s, &L.b, L.ny * L.ld);
                if (ok && l == 0) ok = !dev_alloc(s, &L.r, L.ny * L.ld);
            }
            s->lv.push_back(L);
        }
        if (ok && world > 1) {
            if (slab1 && cb1)
                s->bslab = cb1;
            else
                ok = !dev_alloc(s, &s->bslab, s->slab[n_dist].ny * s->slab[n_dist].ld);
        }
        if (!ok) break;
        // whole-domain levels; the first shares the rows of the staging level's padded arrays (local row t = global row t - HALO)
        const DLevel &S = s->lv[n_para];
        if (dev_alloc(s, &s->xw0_pad, S.ny * S.ld) || dev_alloc(s, &s->rw0_pad, S.ny * S.ld)) break;
        for (int l = n_para; ok; ++l) {
            const i64 my = (gny - 1) >> l, mx = (gnx - 1) >> l;
            if (my < 2 || mx < 2 || (my << l) != gny - 1 || (mx << l) != gnx - 1) break;
            WLevel W;
            W.ny = my + 1, W.nx = mx + 1, W.ld = pitch(W.nx), W.h = h0 * (double)((i64)1 << l);
            if (l == n_para) {
                W.x = s->xw0_pad + HALO * S.ld, W.b = S.b + HALO * S.ld, W.r = s->rw0_pad + HALO * S.ld;
            } else {
                ok = !dev_alloc(s, &W.x, W.ny * W.ld) && !dev_alloc(s, &W.b, W.ny * W.ld) && !dev_alloc(s, &W.r, W.ny * W.ld);
            }
            s->wl.push_back(W);
        }
        if (!ok) break;
        double *nrm[2] = {nullptr, nullptr};
        if (dev_alloc(s, &nrm[0], MG_NORM_SLOTS) || dev_alloc(s, &nrm[1], MG_NORM_SLOTS)) break;
        s->norm[0] = (u64 *)nrm[0], s->norm[1] = (u64 *)nrm[1];
        if (hipStreamSynchronize(s->stream) != hipSuccess) { mg_set_error("mg_dist_create: device error"); break; }
        decide_recompute(s);
        rc = 0;
    } while (0);
    if (rc) {
        mg_dist_destroy(s);
        return rc;
    }
    *out = s;
    return 0;
}

extern "C" int mg_dist_set(mg_dist_solver *s, const char *key, double value)
{
    if (check_device(s, "mg_dist_set")) return 1;
    MG_REQUIRE(key, "mg_dist_set: null key");
    if (!strcmp(key, "jacobi_omega")) {
        MG_REQUIRE(value > 0 && value <= 1, "mg_dist_set: jacobi_omega %g outside (0, 1]", value);
        s->omega = value;
    } else if (!strcmp(key, "max_cycles")) {
        MG_REQUIRE(value >= 1 && value <= 1e6, "mg_dist_set: max_cycles outside 1..1e6");
        s->max_cycles = (int)value;
        return 0;
    } else if (!strcmp(key, "recompute_min_nx")) {
        s->recompute_min_nx = (i64)value;
    } else if (!strcmp(key, "recompute_min_blocks")) {
        s->recompute_min_blocks = (i64)value;
    } else {
        mg_set_error("mg_dist_set: unknown key '%s'", key);
        return 1;
    }
    MG_HIP(hipStreamSynchronize(s->stream));
    drop_graphs(s);
    decide_recompute(s);
    return 0;
}

extern "C" int mg_dist_placement(mg_dist_solver *s, int *candidates)
{
    MG_REQUIRE(s && candidates, "mg_dist_placement: null argument");
    *candidates = s->placement_candidates;
    return 0;
}

extern "C" int mg_dist_info(mg_dist_solver *s, int *n_dist, int64_t *first_global_row, int64_t *owned_rows, int64_t *rounds)
{
    MG_REQUIRE(s, "mg_dist_info: null solver");
    const DLevel &L = s->lv[0];
    if (n_dist) *n_dist = s->world > 1 ? s->n_dist : 0;
    if (first_global_row) *first_global_row = (i64)s->rank * L.rows;
    if (owned_rows) *owned_rows = L.rows + (s->rank == s->world - 1 ? 1 : 0);
    if (rounds) *rounds = s->rounds;
    return 0;
}

// the rows this rank owns of a finest-level array, Dirichlet rows included: local rows HALO .. HALO + owned
extern "C" int mg_dist_upload(mg_dist_solver *s, int which, const double *src, int64_t src_ld)
{
    if (check_device(s, "mg_dist_upload")) return 1;
    MG_REQUIRE(which == MG_SOLVER_X || which == MG_SOLVER_B, "mg_dist_upload: only x and b of the finest level can be set");
    MG_REQUIRE(src && src_ld >= s->gnx, "mg_dist_upload: null source or src_ld < %lld", (long long)s->gnx);
    DLevel &L = s->lv[0];
    const i64 own = L.rows + (s->rank == s->world - 1 ? 1 : 0);
    double *dst[2] = {which == MG_SOLVER_B ? L.b : L.xa, which == MG_SOLVER_B ? nullptr : L.xb};
    for (double *d : dst)
        if (d)
            MG_HIP(hipMemcpy2DAsync(d + HALO * L.ld, (size_t)L.ld * 8, src, (size_t)src_ld * 8, (size_t)L.nx * 8, (size_t)own, hipMemcpyDefault,
                                    s->stream));
    MG_HIP(hipStreamSynchronize(s->stream));
    s->r_stale = true;
    if (which == MG_SOLVER_B) s->b0_halo_valid = false;
    return 0;
}

extern "C" int mg_dist_download(mg_dist_solver *s, int which, double *dst, int64_t dst_ld)
{
    if (check_device(s, "mg_dist_download")) return 1;
    MG_REQUIRE(dst && dst_ld >= s->gnx, "mg_dist_download: null destination or dst_ld < %lld", (long long)s->gnx);
    MG_REQUIRE(which == MG_SOLVER_X || which == MG_SOLVER_B || which == MG_SOLVER_R, "mg_dist_download: x, b or r of the finest level");
    if (which == MG_SOLVER_R && materialize_residual(s)) return 1; // collective: one halo row of x is exchanged
    DLevel &L = s->lv[0];
    const i64 own = L.rows + (s->rank == s->world - 1 ? 1 : 0);
    const double *p = which == MG_SOLVER_B ? L.b : which == MG_SOLVER_R ? L.r : (s->cur0 ? L.xb : L.xa);
    MG_HIP(hipMemcpy2DAsync(dst, (size_t)dst_ld * 8, p + HALO * L.ld, (size_t)L.ld * 8, (size_t)L.nx * 8, (size_t)own, hipMemcpyDefault, s->stream));
    MG_HIP(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int mg_dist_presmooth(mg_dist_solver *s)
{
    if (check_device(s, "mg_dist_presmooth")) return 1;
    return finest_pass(s, 1, nullptr, 0, true, false, false); // :264-266
}

extern "C" int mg_dist_cycles(mg_dist_solver *s, int k, double *errs)
{
    if (check_device(s, "mg_dist_cycles")) return 1;
    MG_REQUIRE(k >= 0, "mg_dist_cycles: negative cycle count");
    // every cycle's norm is read one cycle late: the host waits for cycle c's number (reduced over the ranks on the main
    // stream, copied on a stream of its own) after it has enqueued cycle c + 1, which fills the other accumulator
    int pending = -1, pending_cycle = -1;
    for (int c = 0; c < k; ++c) {
        if (pending >= 0 && pending == s->cur0) {
            double e;
            if (collect_err(s, pending, &e)) return 1;
            if (errs) errs[pending_cycle] = e;
            pending = -1;
        }
        if (run_cycle(s, c == k - 1)) return 1;
        if (reduce_err(s)) return 1;
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

extern "C" int mg_dist_solve(mg_dist_solver *s, double eps, int *it_out, double *err_out)
{
    if (check_device(s, "mg_dist_solve")) return 1;
    s->it = 0, s->diverged = false;
    s->hist.clear();
    double err = 0.0, err_old = 100.0; // :258-259
    if (mg_dist_presmooth(s)) return 1;
    bool done = false;
    while (!done && s->it < s->max_cycles) {
        if (run_cycle(s, false)) return 1;
        if (reduce_err(s)) return 1;
        if (mg_norm_read((const uint64_t *)s->norm[s->last_acc], &err, s->stream)) return 1;
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

extern "C" int mg_dist_history(mg_dist_solver *s, double *errs, int capacity, int *count, int *diverged)
{
    MG_REQUIRE(s, "mg_dist_history: null solver");
    const int n = (int)s->hist.size();
    for (int k = 0; k < n && k < capacity && errs; ++k) errs[k] = s->hist[k];
    if (count) *count = n;
    if (diverged) *diverged = s->diverged;
    return 0;
}

/*
 * mg_oracle.c -- CPU restatement of the reference multigrid V-cycle.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path: it is linked or
 * loaded only by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of
 * bench.py.  Nothing in the multigrid_amd package may call it.
 *
 * Parity status: PINNED.  Every kernel below and the 4-rank block solver are checked
 * bit-for-bit against golden vectors produced by running the reference's own Python in the
 * build container (oracle/gen_golden.py -> tests/golden/ *.npz, tests/test_oracle_golden.py).
 *
 * Each function follows the loop order AND the floating-point expression order of the
 * reference function it cites (paths are relative to the reference checkout), and the file
 * must be compiled with -ffp-contract=off so that no multiply-add is fused.
 *
 * Three groups:
 *   orc_<kernel>      the eight exports of core/numba_modules.py
 *   orc_blocks_*      Multigrid.__init__/solve() of core/parallel_multigrid.py for the four
 *                     2x2 block ranks, run in lockstep inside one process (the halo exchange
 *                     copies what MPI would move, in the same order)
 *   orc_global_*      the same V-cycle schedule on the undecomposed grid: what the block
 *                     solver computes when every halo (corners included) is exact.  This is
 *                     the layout the GPU path works in; it also has the red-black ordering
 *                     of the coarse smoother that the GPU uses by default.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef int64_t i64;

/* grids with fewer points run their loops on the calling thread: a parallel region costs more than it saves there */
#define ORC_PAR_MIN 32768

/* limit the row-parallel loops below to n host threads (timing the algorithm on as many workers
 * as the reference runs MPI ranks: 4) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* number of host threads the row-parallel loops below use */
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ kernels */

/* core/numba_modules.py:10-20 -- the corner block of A_in that belongs to `rank` */
void orc_split(const double *in, i64 in_ny, i64 in_nx, double *out, i64 nout, i64 rank)
{
    i64 j0 = (rank == 2 || rank == 3) ? in_ny - nout : 0;
    i64 i0 = (rank == 1 || rank == 3) ? in_nx - nout : 0;
    if (rank < 0 || rank > 3)
        return; /* the reference falls through every branch and copies nothing */
    for (i64 j = 0; j < nout; j++)
        for (i64 i = 0; i < nout; i++)
            out[j * nout + i] = in[(j0 + j) * in_nx + (i0 + i)];
}

/* core/numba_modules.py:23-32 */
void orc_laplacian(const double *a, double *lplc, i64 ny, i64 nx, double h)
{
    double hh = 1 / (h * h);
    for (i64 j = 1; j < ny - 1; j++)
        for (i64 i = 1; i < nx - 1; i++)
            lplc[j * nx + i] = (a[j * nx + i - 1] + a[j * nx + i + 1] + a[(j - 1) * nx + i] +
                                a[(j + 1) * nx + i] - 4 * a[j * nx + i]) * hh;
}

/* core/numba_modules.py:35-50 -- r = b - lap(a); `left`/`cent` are pre-scaled by 1/h^2 */
void orc_residual(double *r, const double *a, const double *b, i64 ny, i64 nx, double h)
{
    double hh = 1 / (h * h);
#pragma omp parallel for schedule(static) if (ny * nx >= ORC_PAR_MIN)
    for (i64 j = 1; j < ny - 1; j++) {
        double left = a[j * nx + 0] * hh;
        for (i64 i = 1; i < nx - 1; i++) {
            double cent = a[j * nx + i] * hh;
            r[j * nx + i] = b[j * nx + i] - left -
                            (a[j * nx + i + 1] + a[(j - 1) * nx + i] + a[(j + 1) * nx + i]) * hh +
                            cent * 4;
            left = cent;
        }
    }
}

/* core/numba_modules.py:53-66 -- stencil into a_temp (interior), then copy ALL of a_temp to a */
void orc_smooth_sweep_jacobi(const double *b, double *a, double *a_temp, i64 ny, i64 nx, double h)
{
    double h2 = h * h;
#pragma omp parallel for schedule(static) if (ny * nx >= ORC_PAR_MIN)
    for (i64 j = 1; j < ny - 1; j++)
        for (i64 i = 1; i < nx - 1; i++)
            a_temp[j * nx + i] = 0.25 * (a[j * nx + i + 1] + a[j * nx + i - 1] + a[(j + 1) * nx + i] +
                                         a[(j - 1) * nx + i] - h2 * b[j * nx + i]);
#pragma omp parallel for schedule(static) if (ny * nx >= ORC_PAR_MIN)
    for (i64 j = 0; j < ny; j++)
        for (i64 i = 0; i < nx; i++)
            a[j * nx + i] = a_temp[j * nx + i];
}

/*
 * The sweep above with damping: a_temp = a + omega * (t - a), t the undamped value.  No counterpart in
 * the reference (its omega is 1, which leaves the checkerboard error mode undamped); omega = 1 is NOT
 * routed here, so the reference's arithmetic stays untouched.
 */
void orc_smooth_sweep_jacobi_damped(const double *b, double *a, double *a_temp, i64 ny, i64 nx, double h, double omega)
{
    double h2 = h * h;
#pragma omp parallel for schedule(static) if (ny * nx >= ORC_PAR_MIN)
    for (i64 j = 1; j < ny - 1; j++)
        for (i64 i = 1; i < nx - 1; i++) {
            double t = 0.25 * (a[j * nx + i + 1] + a[j * nx + i - 1] + a[(j + 1) * nx + i] + a[(j - 1) * nx + i] -
                               h2 * b[j * nx + i]);
            a_temp[j * nx + i] = a[j * nx + i] + omega * (t - a[j * nx + i]);
        }
#pragma omp parallel for schedule(static) if (ny * nx >= ORC_PAR_MIN)
    for (i64 j = 0; j < ny; j++)
        for (i64 i = 0; i < nx; i++)
            a[j * nx + i] = a_temp[j * nx + i];
}

/*
 * One red-black Gauss-Seidel sweep (an option of this build for the fine levels; the reference smooths them with
 * Jacobi): the cells with even j + i take the Jacobi value of smooth_sweep_jacobi (numba_modules.py:61-63, same
 * expression) computed from the current array, in place, then the cells with odd j + i do.  Cells of one colour
 * do not depend on each other, so the order inside a colour does not matter.
 */
void orc_smooth_sweep_rbgs(const double *b, double *a, i64 ny, i64 nx, double h)
{
    double h2 = h * h;
    for (int colour = 0; colour < 2; colour++) {
#pragma omp parallel for schedule(static) if (ny * nx >= ORC_PAR_MIN)
        for (i64 j = 1; j < ny - 1; j++)
            for (i64 i = 1 + ((colour + j + 1) & 1); i < nx - 1; i += 2)
                a[j * nx + i] = 0.25 * (a[j * nx + i + 1] + a[j * nx + i - 1] + a[(j + 1) * nx + i] + a[(j - 1) * nx + i] -
                                        h2 * b[j * nx + i]);
    }
}

static inline void sor_point(const double *b, double *x, i64 nx, i64 j, i64 i, double alpha, double beta,
                             double h2)
{
    x[j * nx + i] = alpha * x[j * nx + i] +
                    beta * (x[j * nx + i + 1] + x[j * nx + i - 1] + x[(j + 1) * nx + i] + x[(j - 1) * nx + i] -
                            h2 * b[j * nx + i]);
}

static void residual_tail(const double *b, const double *x, double *r, i64 ny, i64 nx, double hh)
{
    for (i64 j = 1; j < ny - 1; j++) {
        double left = x[j * nx + 0] * hh;
        for (i64 i = 1; i < nx - 1; i++) {
            double cent = x[j * nx + i] * hh;
            r[j * nx + i] = b[j * nx + i] - left -
                            (x[j * nx + i + 1] + x[(j - 1) * nx + i] + x[(j + 1) * nx + i]) * hh +
                            cent * 4;
            left = cent;
        }
    }
}

/* core/numba_modules.py:69-95 -- forward lexicographic SOR (omega = 1.5), then the residual */
void orc_smooth(const double *b, double *x, double h, double *r, i64 iterations, i64 ny, i64 nx)
{
    double a = 1.5, alpha = 1 - a, beta = a / 4, h2 = h * h;
    for (i64 k = 0; k < iterations; k++)
        for (i64 j = 1; j < ny - 1; j++)
            for (i64 i = 1; i < nx - 1; i++)
                sor_point(b, x, nx, j, i, alpha, beta, h2);
    residual_tail(b, x, r, ny, nx, 1 / h2);
}

/* core/numba_modules.py:98-129 -- forward then backward lexicographic SOR, then the residual */
void orc_smooth_altern(const double *b, double *x, double h, double *r, i64 iterations, i64 ny, i64 nx)
{
    double a = 1.5, alpha = 1 - a, beta = a / 4, h2 = h * h;
    for (i64 k = 0; k < iterations; k++) {
        for (i64 j = 1; j < ny - 1; j++)
            for (i64 i = 1; i < nx - 1; i++)
                sor_point(b, x, nx, j, i, alpha, beta, h2);
        for (i64 j = ny - 2; j > 0; j--)
            for (i64 i = nx - 2; i > 0; i--)
                sor_point(b, x, nx, j, i, alpha, beta, h2);
    }
    residual_tail(b, x, r, ny, nx, 1 / h2);
}

/*
 * Red-black variant of smooth_altern (NOT in the reference: it restates the ordering the GPU
 * coarse smoother uses).  A point is "red" when (j + i) is even.  The forward sweep updates
 * red then black, the backward sweep black then red, so one iteration is R,B,B,R; the point
 * update and the trailing residual are those of numba_modules.py:108-110 and :117-129.
 */
void orc_smooth_altern_rb(const double *b, double *x, double h, double *r, i64 iterations, i64 ny, i64 nx)
{
    double a = 1.5, alpha = 1 - a, beta = a / 4, h2 = h * h;
    static const int order[4] = {0, 1, 1, 0};
    for (i64 k = 0; k < iterations; k++)
        for (int p = 0; p < 4; p++)
            for (i64 j = 1; j < ny - 1; j++)
                for (i64 i = 1; i < nx - 1; i++)
                    if (((j + i) & 1) == order[p])
                        sor_point(b, x, nx, j, i, alpha, beta, h2);
    residual_tail(b, x, r, ny, nx, 1 / h2);
}

/* core/numba_modules.py:132-160 -- full weighting; column sums carried left to right */
void orc_coarse(const double *a, i64 fny, i64 fnx, double *a_crs, i64 cny, i64 cnx, i64 ofst_i, i64 ofst_j)
{
    (void)fny;
#pragma omp parallel for schedule(static) if (cny * cnx >= ORC_PAR_MIN)
    for (i64 j = 1; j < cny - 1; j++) {
        const double *am = a + (2 * j + ofst_j - 1) * fnx, *a0 = am + fnx, *ap = a0 + fnx;
        double a_left = a0[1 + ofst_i] / 8 + (ap[1 + ofst_i] + am[1 + ofst_i]) / 16;
        for (i64 i = 1; i < cnx - 1; i++) {
            i64 c = 2 * i + ofst_i;
            double a_right = a0[c + 1] / 8 + (ap[c + 1] + am[c + 1]) / 16;
            double v = a0[c] / 4 + (ap[c] + am[c]) * 1 / 8;
            v += a_right + a_left;
            a_crs[j * cnx + i] = v;
            a_left = a_right;
        }
    }
}

/* core/numba_modules.py:163-180 -- bilinear prolongation ADDED into a_new, four passes */
void orc_interpolate_add_to(const double *a, i64 cny, i64 cnx, double *a_new, i64 fny, i64 fnx, i64 ofst_i,
                            i64 ofst_j)
{
    (void)fny;
#pragma omp parallel if (cny * cnx >= ORC_PAR_MIN)
    {
#pragma omp for schedule(static)
        for (i64 j = 1; j < cny - 1; j++)
            for (i64 i = 1; i < cnx - 1; i++)
                a_new[(2 * j + ofst_j) * fnx + 2 * i + ofst_i] += a[j * cnx + i];
#pragma omp for schedule(static)
        for (i64 j = 0; j < cny - 1; j++)
            for (i64 i = 0; i < cnx - 1; i++)
                a_new[(2 * j + 1 + ofst_j) * fnx + 2 * i + 1 + ofst_i] +=
                    (a[(j + 1) * cnx + i + 1] + a[(j + 1) * cnx + i] + a[j * cnx + i + 1] + a[j * cnx + i]) / 4;
#pragma omp for schedule(static)
        for (i64 j = 1; j < cny - 1; j++)
            for (i64 i = 0; i < cnx - 1; i++)
                a_new[(2 * j + ofst_j) * fnx + 2 * i + 1 + ofst_i] += (a[j * cnx + i] + a[j * cnx + i + 1]) / 2;
#pragma omp for schedule(static)
        for (i64 j = 0; j < cny - 1; j++)
            for (i64 i = 1; i < cnx - 1; i++)
                a_new[(2 * j + 1 + ofst_j) * fnx + 2 * i + ofst_i] += (a[j * cnx + i] + a[(j + 1) * cnx + i]) / 2;
    }
}

/* np.amax(np.abs(a)) over a whole array; NaN propagates as in numpy */
double orc_maxabs(const double *a, i64 n)
{
    double m = 0;
    int nan = 0;
#pragma omp parallel for schedule(static) reduction(max : m) reduction(| : nan) if (n >= ORC_PAR_MIN)
    for (i64 k = 0; k < n; k++) {
        double v = fabs(a[k]);
        if (v != v)
            nan = 1;
        if (v > m)
            m = v;
    }
    return nan ? NAN : m;
}

/* ------------------------------------------------------------------ 4-rank block solver */

#define MAXLEV 40
#define NRANK 4
#define MAXCYC 100

typedef struct {
    int n, n_para, nwl;
    double h0, eps;
    i64 sb_n[MAXLEV + 1];         /* side of the square sub-domain arrays, level 0..n_para (staging) */
    i64 wl_n[MAXLEV];             /* side of the whole-domain arrays */
    double h_sb[MAXLEV], h_wl[MAXLEV];
    double *x_sb[NRANK][MAXLEV], *b_sb[NRANK][MAXLEV + 1], *r_sb[NRANK][MAXLEV];
    double *tmp[NRANK];           /* the `a` scratch of smooth_parallel */
    double *x_wl[MAXLEV], *b_wl[MAXLEV], *r_wl[MAXLEV];
    int it;
    int verbose;
    int max_cycles;    /* cap on the cycles of one solve (MAXCYC = the reference's 100, :270) */
    int rank_parallel; /* 1: the four ranks' kernels run on four threads, one per rank, as the reference's four processes do */
} blocks_t;

static double *zalloc(i64 n) { return (double *)calloc((size_t)n, sizeof(double)); }

/* Multigrid.__init__, core/parallel_multigrid.py:153-230 (level shapes, mesh widths) */
void *orc_blocks_create(int n, int n_para, double h0, double eps)
{
    if (n_para >= n || n_para < 1 || n >= MAXLEV)
        return NULL; /* the reference asserts n_para < n (:159) */
    blocks_t *s = (blocks_t *)calloc(1, sizeof(blocks_t));
    s->n = n, s->n_para = n_para, s->h0 = h0, s->eps = eps;
    s->max_cycles = MAXCYC;
    s->nwl = 0;
    for (int i = n_para - 1; i < n; i++) { /* :182-192 */
        i64 m = ((i64)1 << (n - i)) + 1;
        int k = s->nwl++;
        s->wl_n[k] = m;
        s->x_wl[k] = zalloc(m * m), s->b_wl[k] = zalloc(m * m), s->r_wl[k] = zalloc(m * m);
        s->h_wl[k] = h0 * (double)((i64)1 << (i + 1));
    }
    for (int i = 0; i <= n_para; i++) { /* :194-219, the last one is the extra b_sb */
        i64 m = ((i64)1 << (n - i)) + 2;
        s->sb_n[i] = m;
        for (int rk = 0; rk < NRANK; rk++) {
            s->b_sb[rk][i] = zalloc(m * m);
            if (i < n_para) {
                s->x_sb[rk][i] = zalloc(m * m);
                s->r_sb[rk][i] = zalloc(m * m);
            }
        }
        if (i < n_para)
            s->h_sb[i] = h0 * (double)((i64)1 << i);
    }
    for (int rk = 0; rk < NRANK; rk++)
        s->tmp[rk] = zalloc(s->sb_n[0] * s->sb_n[0]);
    return s;
}

void orc_blocks_destroy(void *hd)
{
    blocks_t *s = (blocks_t *)hd;
    if (!s)
        return;
    for (int k = 0; k < s->nwl; k++)
        free(s->x_wl[k]), free(s->b_wl[k]), free(s->r_wl[k]);
    for (int rk = 0; rk < NRANK; rk++) {
        for (int i = 0; i <= s->n_para; i++) {
            free(s->b_sb[rk][i]);
            if (i < s->n_para)
                free(s->x_sb[rk][i]), free(s->r_sb[rk][i]);
        }
        free(s->tmp[rk]);
    }
    free(s);
}

/* which: 0 = x_sb, 1 = b_sb, 2 = r_sb, 3 = x_wl, 4 = b_wl, 5 = r_wl (rank ignored for 3..5) */
double *orc_blocks_array(void *hd, int which, int rank, int level, i64 *side)
{
    blocks_t *s = (blocks_t *)hd;
    if (which <= 2) {
        if (level < 0 || level > s->n_para || (which != 1 && level == s->n_para))
            return NULL;
        *side = s->sb_n[level];
        return which == 0 ? s->x_sb[rank][level] : which == 1 ? s->b_sb[rank][level] : s->r_sb[rank][level];
    }
    if (level < 0 || level >= s->nwl)
        return NULL;
    *side = s->wl_n[level];
    return which == 3 ? s->x_wl[level] : which == 4 ? s->b_wl[level] : s->r_wl[level];
}

void orc_blocks_set_verbose(void *hd, int v) { ((blocks_t *)hd)->verbose = v; }
void orc_blocks_set_rank_parallel(void *hd, int v) { ((blocks_t *)hd)->rank_parallel = v; }
void orc_blocks_set_max_cycles(void *hd, int m) { ((blocks_t *)hd)->max_cycles = m < 1 ? 1 : (m > MAXCYC ? MAXCYC : m); }

/* Loop over the four ranks: serial (every kernel then uses all OpenMP threads over its rows), or one thread per
 * rank with the kernels inside running single-threaded (nested regions are inactive), which is how the reference's
 * four MPI processes share the work (core/Makefile:7).  The ranks are independent inside each such loop, so the
 * values do not change. */
#define FOR_RANKS(s) _Pragma("omp parallel for schedule(static, 1) num_threads(NRANK) if ((s)->rank_parallel)") for (int rk = 0; rk < NRANK; rk++)

/*
 * Buffers.exchange_buffers for one variable on all four ranks, core/parallel_multigrid.py:19-97
 * with width = 1, overlap = 1: every rank packs the line just inside the shared centre line
 * (:39-42), then every rank unpacks into its outermost line (:44-47) in the insertion order of
 * its `neighbours` dict (:49-56).  Whole lines are moved, corners included.
 */
static void exchange4(double *v[NRANK], i64 m)
{
    /* directions: 0 east, 1 west, 2 north, 3 south */
    static const int nb[NRANK][2][2] = {
        /* {direction, peer} in dict order */
        {{0, 1}, {2, 2}}, {{1, 0}, {2, 3}}, {{0, 3}, {3, 0}}, {{1, 2}, {3, 1}}};
    double *msg[NRANK][4] = {{0}};
    for (int rk = 0; rk < NRANK; rk++)
        for (int q = 0; q < 2; q++) {
            int d = nb[rk][q][0];
            double *buf = (double *)malloc((size_t)m * sizeof(double));
            for (i64 k = 0; k < m; k++)
                buf[k] = d == 0 ? v[rk][k * m + (m - 3)] : d == 1 ? v[rk][k * m + 2]
                         : d == 2 ? v[rk][(m - 3) * m + k] : v[rk][2 * m + k];
            msg[rk][d] = buf;
        }
    static const int opposite[4] = {1, 0, 3, 2};
    for (int rk = 0; rk < NRANK; rk++)
        for (int q = 0; q < 2; q++) {
            int d = nb[rk][q][0], peer = nb[rk][q][1];
            const double *buf = msg[peer][opposite[d]]; /* what the peer sent towards me */
            for (i64 k = 0; k < m; k++) {
                if (d == 0) v[rk][k * m + (m - 1)] = buf[k];
                else if (d == 1) v[rk][k * m + 0] = buf[k];
                else if (d == 2) v[rk][(m - 1) * m + k] = buf[k];
                else v[rk][0 * m + k] = buf[k];
            }
        }
    for (int rk = 0; rk < NRANK; rk++)
        for (int d = 0; d < 4; d++)
            free(msg[rk][d]);
}

/* smooth_parallel on all four ranks in lockstep, core/parallel_multigrid.py:100-114 */
static void smooth_parallel4(blocks_t *s, double *b[NRANK], double *x[NRANK], double h, double *r[NRANK], i64 m,
                             int iterations)
{
    FOR_RANKS(s)
        memcpy(s->tmp[rk], x[rk], (size_t)(m * m) * sizeof(double)); /* :105-106, before the exchange */
    exchange4(x, m);
    for (int k = 0; k < iterations; k++) {
        FOR_RANKS(s)
            orc_smooth_sweep_jacobi(b[rk], x[rk], s->tmp[rk], m, m, h);
        exchange4(x, m);
    }
    FOR_RANKS(s)
        orc_residual(r[rk], x[rk], b[rk], m, m, h);
}

/* gather_blocks, core/parallel_multigrid.py:117-138 */
static void gather_blocks4(double *blk[NRANK], i64 nx, double *full, i64 nx0)
{
    i64 nx2 = nx - 1;
    for (i64 j = 0; j < nx0; j++)
        for (i64 i = 0; i < nx0; i++) {
            int top = j >= nx2, right = i >= nx2;
            i64 jj = top ? j - nx2 + 2 : j, ii = right ? i - nx2 + 2 : i;
            full[j * nx0 + i] = blk[2 * top + right][jj * nx + ii];
        }
}

static const int OFS_I[NRANK] = {0, -1, 0, -1}; /* core/parallel_multigrid.py:222-228 */
static const int OFS_J[NRANK] = {0, 0, -1, -1};

/*
 * Multigrid.solve(), core/parallel_multigrid.py:232-348.
 * err_out[4]: the value each rank returns.  err_hist[4*100]: np.amax(|r_sb[0]|) of each rank at
 * each convergence vote (:336).  r_hist[4*100]: max|r_sb[0]| of each rank right after the halo
 * exchange of r at the top of each cycle (:272).  int_hist[100]: max over the four ranks of
 * max|r_sb[0]| over the block INTERIORS only (no halo line), i.e. the max-norm of the residual
 * on the whole grid, which is what a solver without stale halo copies measures.
 * Any of the three history pointers may be NULL.  Returns self.it.
 */
static double interior_maxabs(const double *a, i64 m)
{
    double v = 0;
    for (i64 j = 1; j < m - 1; j++)
        for (i64 i = 1; i < m - 1; i++)
            if (fabs(a[j * m + i]) > v)
                v = fabs(a[j * m + i]);
    return v;
}

int orc_blocks_solve(void *hd, double *err_out, double *err_hist, double *r_hist, double *int_hist)
{
    blocks_t *s = (blocks_t *)hd;
    const int n = s->n, n_para = s->n_para, n1 = 3, n2 = 6, L = s->nwl;
    double err[NRANK] = {0}, err_old[NRANK] = {100, 100, 100, 100};
    int fail[NRANK] = {0}, status = 0;
    double *x[NRANK], *b[NRANK], *r[NRANK];
#define LV(arr, lev) for (int rk = 0; rk < NRANK; rk++) arr[rk] = s->arr##_sb[rk][lev]
    s->it = 0;
    LV(x, 0); LV(b, 0); LV(r, 0);
    smooth_parallel4(s, b, x, s->h_sb[0], r, s->sb_n[0], 1); /* :264-266 */

    while (status < 4 && s->it < s->max_cycles) {
        /* descent, sub-domains */
        LV(r, 0);
        exchange4(r, s->sb_n[0]); /* :272 */
        FOR_RANKS(s) {
            if (r_hist)
                r_hist[rk * MAXCYC + s->it] = orc_maxabs(r[rk], s->sb_n[0] * s->sb_n[0]);
            orc_coarse(r[rk], s->sb_n[0], s->sb_n[0], s->b_sb[rk][1], s->sb_n[1], s->sb_n[1], OFS_I[rk], OFS_J[rk]);
        }
        for (int i = 1; i < n_para; i++) { /* :275-288 */
            i64 m = s->sb_n[i];
            LV(x, i); LV(b, i); LV(r, i);
            FOR_RANKS(s)
                memset(x[rk], 0, (size_t)(m * m) * sizeof(double));
            smooth_parallel4(s, b, x, s->h_sb[i], r, m, n1);
            exchange4(r, m);
            FOR_RANKS(s)
                orc_coarse(r[rk], m, m, s->b_sb[rk][i + 1], s->sb_n[i + 1], s->sb_n[i + 1], OFS_I[rk], OFS_J[rk]);
        }
        /* whole domain: identical on every rank, so computed once (:291-297) */
        memset(s->x_wl[0], 0, (size_t)(s->wl_n[0] * s->wl_n[0]) * sizeof(double));
        LV(b, n_para);
        gather_blocks4(b, s->sb_n[n_para], s->b_wl[0], s->wl_n[0]);
        orc_smooth_altern(s->b_wl[0], s->x_wl[0], s->h_wl[0], s->r_wl[0], n1, s->wl_n[0], s->wl_n[0]);
        for (int i = 1; i < n - n_para + 1; i++) {
            i64 m = s->wl_n[i];
            memset(s->x_wl[i], 0, (size_t)(m * m) * sizeof(double));
            orc_coarse(s->r_wl[i - 1], s->wl_n[i - 1], s->wl_n[i - 1], s->b_wl[i], m, m, 0, 0);
            orc_smooth_altern(s->b_wl[i], s->x_wl[i], s->h_wl[i], s->r_wl[i], n1, m, m);
        }
        /* ascent, whole domain: range(2, n - n_para + 1) never reaches x_wl[0] (:302-304) */
        for (int i = 2; i < n - n_para + 1; i++) {
            i64 mc = s->wl_n[L - i + 1], mf = s->wl_n[L - i];
            orc_interpolate_add_to(s->x_wl[L - i + 1], mc, mc, s->x_wl[L - i], mf, mf, 0, 0);
            orc_smooth_altern(s->b_wl[L - i], s->x_wl[L - i], s->h_wl[L - i], s->r_wl[L - i], n2, mf, mf);
        }
        /* ascent, sub-domains (:307-333) */
        FOR_RANKS(s) {
            orc_split(s->x_wl[0], s->wl_n[0], s->wl_n[0], s->b_sb[rk][n_para], s->sb_n[n_para], rk);
            orc_interpolate_add_to(s->b_sb[rk][n_para], s->sb_n[n_para], s->sb_n[n_para], s->x_sb[rk][n_para - 1],
                                   s->sb_n[n_para - 1], s->sb_n[n_para - 1], OFS_I[rk], OFS_J[rk]);
        }
        LV(b, n_para - 1); LV(x, n_para - 1); LV(r, n_para - 1);
        smooth_parallel4(s, b, x, s->h_sb[n_para - 1], r, s->sb_n[n_para - 1], n2);
        for (int i = 2; i < n_para + 1; i++) {
            int lc = n_para - i + 1, lf = n_para - i;
            FOR_RANKS(s)
                orc_interpolate_add_to(s->x_sb[rk][lc], s->sb_n[lc], s->sb_n[lc], s->x_sb[rk][lf], s->sb_n[lf],
                                       s->sb_n[lf], OFS_I[rk], OFS_J[rk]);
            LV(b, lf); LV(x, lf); LV(r, lf);
            smooth_parallel4(s, b, x, s->h_sb[lf], r, s->sb_n[lf], n2);
        }
        /* convergence vote (:336-345) */
        status = 0;
        if (int_hist) {
            int_hist[s->it] = 0;
            for (int rk = 0; rk < NRANK; rk++) {
                double v = interior_maxabs(s->r_sb[rk][0], s->sb_n[0]);
                if (v > int_hist[s->it])
                    int_hist[s->it] = v;
            }
        }
        for (int rk = 0; rk < NRANK; rk++) {
            err[rk] = orc_maxabs(s->r_sb[rk][0], s->sb_n[0] * s->sb_n[0]);
            if (err_hist)
                err_hist[rk * MAXCYC + s->it] = err[rk];
            if (err[rk] > err_old[rk] * 1.1) {
                if (s->verbose)
                    printf("Convergence compromised : r>r+1, set n1 higher\n");
                fail[rk] = 1;
            }
            err_old[rk] = err[rk];
            status += (err[rk] > s->eps && !fail[rk]) ? 0 : 1;
        }
        s->it += 1;
    }
#undef LV
    for (int rk = 0; rk < NRANK; rk++)
        err_out[rk] = err[rk];
    return s->it;
}

/* ------------------------------------------------------------------ undecomposed solver */

typedef struct {
    int n_para, nwl;
    i64 ny[MAXLEV + 1], nx[MAXLEV + 1]; /* Jacobi levels 0..n_para-1 */
    i64 wny[MAXLEV], wnx[MAXLEV];       /* whole-domain (coarse smoother) levels */
    double h[MAXLEV], hw[MAXLEV];
    double *x[MAXLEV], *b[MAXLEV], *r[MAXLEV], *tmp;
    double *xw[MAXLEV], *bw[MAXLEV], *rw[MAXLEV];
    double eps;
    int ordering;          /* 0 lexicographic (reference), 1 red-black */
    double jacobi_omega;   /* 1 = the reference's sweep */
    int fine_smoother;     /* 0 Jacobi (reference), 1 red-black Gauss-Seidel */
    int fix_coarse_ascent; /* 0: reference loop bounds (:302); 1: the ascent also reaches xw[0] */
    int max_cycles;
    int it;
} global_t;

/*
 * A (ny0 x nx0) grid including its Dirichlet boundary, ny0-1 and nx0-1 divisible by 2^n_para.
 * Levels 0..n_para-1 use Jacobi as the sub-domain levels of the reference do; level n_para and
 * every coarser grid down to the smallest with both sides >= 3 use the coarse smoother, as the
 * whole-domain levels do (parallel_multigrid.py:182-192: for a square 2^(n+1)+1 grid that is
 * n - n_para + 1 levels ending at 3x3).
 */
void *orc_global_create(i64 ny0, i64 nx0, int n_para, double h0, double eps, int ordering, int fix_coarse_ascent)
{
    if (n_para < 1 || n_para >= MAXLEV)
        return NULL;
    if (((ny0 - 1) % ((i64)1 << n_para)) || ((nx0 - 1) % ((i64)1 << n_para)))
        return NULL;
    if (((ny0 - 1) >> n_para) < 2 || ((nx0 - 1) >> n_para) < 2)
        return NULL;
    global_t *s = (global_t *)calloc(1, sizeof(global_t));
    s->n_para = n_para, s->eps = eps, s->ordering = ordering, s->fix_coarse_ascent = fix_coarse_ascent;
    s->max_cycles = MAXCYC;
    s->jacobi_omega = 1.0;
    for (int l = 0; l < n_para; l++) {
        s->ny[l] = ((ny0 - 1) >> l) + 1, s->nx[l] = ((nx0 - 1) >> l) + 1;
        s->h[l] = h0 * (double)((i64)1 << l);
        i64 sz = s->ny[l] * s->nx[l];
        s->x[l] = zalloc(sz), s->b[l] = zalloc(sz), s->r[l] = zalloc(sz);
    }
    s->tmp = zalloc(s->ny[0] * s->nx[0]);
    for (int l = n_para;; l++) {
        i64 my = (ny0 - 1) >> l, mx = (nx0 - 1) >> l;
        if (my < 2 || mx < 2 || (my << l) != ny0 - 1 || (mx << l) != nx0 - 1)
            break;
        int k = s->nwl++;
        s->wny[k] = my + 1, s->wnx[k] = mx + 1;
        s->hw[k] = h0 * (double)((i64)1 << l);
        i64 sz = s->wny[k] * s->wnx[k];
        s->xw[k] = zalloc(sz), s->bw[k] = zalloc(sz), s->rw[k] = zalloc(sz);
    }
    return s;
}

void orc_global_destroy(void *hd)
{
    global_t *s = (global_t *)hd;
    if (!s)
        return;
    for (int l = 0; l < s->n_para; l++)
        free(s->x[l]), free(s->b[l]), free(s->r[l]);
    for (int k = 0; k < s->nwl; k++)
        free(s->xw[k]), free(s->bw[k]), free(s->rw[k]);
    free(s->tmp);
    free(s);
}

void orc_global_set_max_cycles(void *hd, int m) { ((global_t *)hd)->max_cycles = m; }
void orc_global_set_jacobi_omega(void *hd, double omega) { ((global_t *)hd)->jacobi_omega = omega; }
void orc_global_set_fine_smoother(void *hd, int kind) { ((global_t *)hd)->fine_smoother = kind; }
int orc_global_num_whole_levels(void *hd) { return ((global_t *)hd)->nwl; }

/* which: 0 x, 1 b, 2 r (Jacobi levels); 3 xw, 4 bw, 5 rw (coarse-smoother levels) */
double *orc_global_array(void *hd, int which, int level, i64 *ny, i64 *nx)
{
    global_t *s = (global_t *)hd;
    if (which <= 2) {
        if (level < 0 || level >= s->n_para)
            return NULL;
        *ny = s->ny[level], *nx = s->nx[level];
        return which == 0 ? s->x[level] : which == 1 ? s->b[level] : s->r[level];
    }
    if (level < 0 || level >= s->nwl)
        return NULL;
    *ny = s->wny[level], *nx = s->wnx[level];
    return which == 3 ? s->xw[level] : which == 4 ? s->bw[level] : s->rw[level];
}

static void g_smooth(global_t *s, const double *b, double *x, double h, double *r, i64 ny, i64 nx, int iterations,
                     int want_r)
{
    memcpy(s->tmp, x, (size_t)(ny * nx) * sizeof(double));
    for (int k = 0; k < iterations; k++) {
        if (s->fine_smoother == 1)
            orc_smooth_sweep_rbgs(b, x, ny, nx, h);
        else if (s->jacobi_omega == 1.0)
            orc_smooth_sweep_jacobi(b, x, s->tmp, ny, nx, h);
        else
            orc_smooth_sweep_jacobi_damped(b, x, s->tmp, ny, nx, h, s->jacobi_omega);
    }
    if (want_r)
        orc_residual(r, x, b, ny, nx, h);
}

static void g_sor(global_t *s, const double *b, double *x, double h, double *r, i64 ny, i64 nx, int iterations)
{
    if (s->ordering == 0)
        orc_smooth_altern(b, x, h, r, iterations, ny, nx);
    else
        orc_smooth_altern_rb(b, x, h, r, iterations, ny, nx);
}

/*
 * The schedule of Multigrid.solve() (core/parallel_multigrid.py:232-348) with the four blocks
 * replaced by the one undecomposed grid (offsets 0, no exchange, no gather/split: level n_para's
 * right-hand side IS bw[0] and xw[0] IS the correction).  `ncycles` > 0 runs exactly that many
 * cycles and ignores eps and the divergence latch; 0 applies the reference stop rule.
 * err_hist[k] = max|r[0]| after cycle k.  Returns the number of cycles run.
 */
int orc_global_solve(void *hd, int ncycles, double *err_out, double *err_hist)
{
    global_t *s = (global_t *)hd;
    const int n_para = s->n_para, n1 = 3, n2 = 6, L = s->nwl;
    double err = 0, err_old = 100;
    int fail = 0, done = 0;
    s->it = 0;
    g_smooth(s, s->b[0], s->x[0], s->h[0], s->r[0], s->ny[0], s->nx[0], 1, 1);
    int cap = ncycles > 0 ? ncycles : s->max_cycles;
    while (!done && s->it < cap) {
        for (int l = 0; l < n_para; l++) { /* descent */
            if (l > 0) {
                memset(s->x[l], 0, (size_t)(s->ny[l] * s->nx[l]) * sizeof(double));
                g_smooth(s, s->b[l], s->x[l], s->h[l], s->r[l], s->ny[l], s->nx[l], n1, 1);
            }
            if (l + 1 < n_para)
                orc_coarse(s->r[l], s->ny[l], s->nx[l], s->b[l + 1], s->ny[l + 1], s->nx[l + 1], 0, 0);
            else
                orc_coarse(s->r[l], s->ny[l], s->nx[l], s->bw[0], s->wny[0], s->wnx[0], 0, 0);
        }
        memset(s->xw[0], 0, (size_t)(s->wny[0] * s->wnx[0]) * sizeof(double));
        g_sor(s, s->bw[0], s->xw[0], s->hw[0], s->rw[0], s->wny[0], s->wnx[0], n1);
        for (int i = 1; i < L; i++) {
            memset(s->xw[i], 0, (size_t)(s->wny[i] * s->wnx[i]) * sizeof(double));
            orc_coarse(s->rw[i - 1], s->wny[i - 1], s->wnx[i - 1], s->bw[i], s->wny[i], s->wnx[i], 0, 0);
            g_sor(s, s->bw[i], s->xw[i], s->hw[i], s->rw[i], s->wny[i], s->wnx[i], n1);
        }
        /* reference: for i in range(2, L) -> targets xw[L-2] .. xw[1]; fixed: .. xw[0] */
        for (int i = 2; i < L + (s->fix_coarse_ascent ? 1 : 0); i++) {
            int c = L - i + 1, f = L - i;
            orc_interpolate_add_to(s->xw[c], s->wny[c], s->wnx[c], s->xw[f], s->wny[f], s->wnx[f], 0, 0);
            g_sor(s, s->bw[f], s->xw[f], s->hw[f], s->rw[f], s->wny[f], s->wnx[f], n2);
        }
        for (int l = n_para - 1; l >= 0; l--) { /* ascent */
            const double *c = l == n_para - 1 ? s->xw[0] : s->x[l + 1];
            i64 cny = l == n_para - 1 ? s->wny[0] : s->ny[l + 1], cnx = l == n_para - 1 ? s->wnx[0] : s->nx[l + 1];
            orc_interpolate_add_to(c, cny, cnx, s->x[l], s->ny[l], s->nx[l], 0, 0);
            g_smooth(s, s->b[l], s->x[l], s->h[l], s->r[l], s->ny[l], s->nx[l], n2, 1);
        }
        err = orc_maxabs(s->r[0], s->ny[0] * s->nx[0]);
        if (err_hist)
            err_hist[s->it] = err;
        if (ncycles <= 0) {
            if (err > err_old * 1.1)
                fail = 1;
            err_old = err;
            done = !(err > s->eps && !fail);
        }
        s->it += 1;
    }
    *err_out = err;
    return s->it;
}

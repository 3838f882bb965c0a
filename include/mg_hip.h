/*
 * mg_hip.h -- C ABI of libmg_hip.so, the MI355X (gfx950) implementation of the multigrid
 * V-cycle kernels.
 *
 * Every entry point takes plain pointers and sizes.  Array pointers are DEVICE pointers to
 * IEEE float64, row-major with a leading dimension `ld` (elements between consecutive rows,
 * ld >= nx).  `stream` is a hipStream_t passed as void* (NULL = the null stream).  Unless a
 * comment says otherwise a call only enqueues work on `stream` and returns.
 * Return value: 0 on success, non-zero on error; mg_last_error() then describes it (the text
 * is per host thread).  Nothing is thrown across this boundary.  Shapes are validated on the
 * host before any launch, so a bad call is rejected instead of faulting the GPU.
 *
 * `nbatch`/`bstride`: the same operation is applied to nbatch independent grids whose arrays
 * start bstride elements apart (every array argument of the call uses the same stride).
 *
 * File references below are into the reference checkout (Tugdual-G/multigrid): each entry
 * point names the reference function it replaces.  Arithmetic follows the expression order of
 * the reference and the library is built with -ffp-contract=off, so results are bit-identical
 * to the reference's for the same inputs (mg_sor with MG_SOR_REDBLACK excepted, see there).
 */
#ifndef MG_HIP_H
#define MG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 8

/* flags for the stencil entry points */
#define MG_FLAG_FINEST 1 /* marks the launches of the finest level; ignored (large outputs get an instantiation of their own, see "stream_min_mib") */

/* number of 8-byte slots a max-norm accumulator needs (see mg_residual) */
#define MG_NORM_SLOTS 64

int mg_abi_version(void);
const char *mg_last_error(void);

/* Tuning/testing knobs of the stencil kernels (results never depend on them):
 *   "vector"      1 | 2      doubles per memory access (2 needs 16-byte aligned rows)
 *   "unroll"      1 | 2 | 4  accesses per lane and row
 *   "rows"        0..4096    rows per workgroup tile, 0 = automatic
 *   "linear_map"  0 | 1      1 disables the XCD-aware workgroup -> tile mapping
 *   "fused_waves" 1 | 2 | 4  column strips (waves) per workgroup of the fused passes
 *   "fused_min_rows" 4..128  shortest tile (rows) the fused passes choose on levels too small to fill the chip
 *   "fused_min_blocks" n     ... tiles are halved while a launch has fewer workgroups than this
 *   "fused_lockstep" n       the waves of a workgroup of the fused passes (neighbouring strips of the same rows) meet at
 *                            a barrier every n groups of six rows, which keeps them on the same DRAM pages (default 1;
 *                            0 = never)
 *   "fused_rows"     n       rows per tile of the fused passes (0 = automatic), as mg_set_fused_tuning's second argument
 *   "fused_rows_shallow" n   the automatic tile height of passes of depth <= 5 (default 32) ...
 *   "fused_rows_deep"    n   ... and of depth >= 9 (default 128); passes in between use 64 rows
 *   "xcd_stagger"    n       column blocks by which the bands of consecutive XCDs are shifted (-1: an eighth of the
 *                            width, 0 = off, the default: no effect was measured)
 *   "stream_min_mib"  n      arrays of at least n MiB are written with streaming (non-temporal) stores, which keeps
 *                            them from displacing in L2 / the Infinity Cache the rows other tiles re-read (default 256;
 *                            a value beyond every array size switches them off)
 *   "sor_single_wave" 0|1|2  mg_sor, lexicographic: barrier-free one-wave kernel never / where it is
 *                            faster (default) / wherever the grid fits
 *   "sor_transpose"   0|1    mg_sor, lexicographic: a grid with more rows than columns is walked along its columns
 *                            by the one-wave kernel when that needs fewer rows per lane (default 1; same values) */
int mg_set_tuning(const char *key, int value);

/*
 * One Jacobi sweep, rows j0 <= j < j1 (1 <= j0, j1 <= ny-1), interior columns 1..nx-2:
 *     x_out[j,i] = 0.25 * (x[j,i+1] + x[j,i-1] + x[j+1,i] + x[j-1,i] - h*h * b[j,i])
 * Replaces smooth_sweep_jacobi, core/numba_modules.py:53-66, first loop.  The reference then
 * copies the whole scratch array back (second loop); here the caller swaps x and x_out instead,
 * which is equivalent when the boundary/halo cells of both arrays hold the same values.
 * x_out must not alias x.
 * omega = 1 is the reference's sweep.  0 < omega < 1 damps it: with t the value above,
 *     x_out[j,i] = x[j,i] + omega * (t - x[j,i])
 * (no counterpart in the reference, whose undamped sweep leaves the checkerboard mode of the error
 * untouched, so that its residual stalls at that mode's share of the right-hand side).
 */
int mg_jacobi(const double *b, const double *x, double *x_out, int64_t ny, int64_t nx, int64_t ld,
              int64_t j0, int64_t j1, double h, double omega, int64_t nbatch, int64_t bstride, int flags,
              void *stream);

/*
 * The first Jacobi sweep from x = 0 without reading x:  x_out[j,i] = 0.25 * (0.0 - h*h * b[j,i]),
 * the value the sweep above produces when x is all zeros (parallel_multigrid.py:276 zeroes x
 * before smooth_parallel).  Interior cells of rows j0..j1-1 only.  omega as in mg_jacobi
 * (x_out = 0.0 + omega * (t - 0.0)).
 */
int mg_jacobi_from_zero(const double *b, double *x_out, int64_t ny, int64_t nx, int64_t ld, int64_t j0,
                        int64_t j1, double h, double omega, int64_t nbatch, int64_t bstride, int flags,
                        void *stream);

/*
 * Residual r = b - laplacian(x), rows j0 <= j < j1, interior columns, evaluated as
 *     r[j,i] = b[j,i] - x[j,i-1]*hh - (x[j,i+1] + x[j-1,i] + x[j+1,i])*hh + (x[j,i]*hh)*4,  hh = 1/(h*h)
 * Replaces residual, core/numba_modules.py:35-50.  Cells outside the range are not written.
 * If norm_slots != NULL it must point to MG_NORM_SLOTS uint64 words (zeroed with
 * mg_norm_reset before the call); the kernel max-accumulates the bit patterns of |r| of the
 * cells it wrote, so that mg_norm_read returns np.amax(np.abs(r)) over them
 * (core/parallel_multigrid.py:336), NaN included.
 */
int mg_residual(double *r, const double *x, const double *b, int64_t ny, int64_t nx, int64_t ld, int64_t j0,
                int64_t j1, double h, uint64_t *norm_slots, int64_t nbatch, int64_t bstride, int flags,
                void *stream);

/* lplc[j,i] = (x[j,i-1] + x[j,i+1] + x[j-1,i] + x[j+1,i] - 4*x[j,i]) * (1/(h*h)) on the interior.
 * Replaces laplacian, core/numba_modules.py:23-32. */
int mg_laplacian(const double *x, double *lplc, int64_t ny, int64_t nx, int64_t ld, double h, void *stream);

int mg_norm_reset(uint64_t *norm_slots, void *stream);
/* Copies the slots to the host, WAITS for `stream`, writes the max to *out. */
int mg_norm_read(const uint64_t *norm_slots, double *out, void *stream);
/* np.amax(np.abs(a)) over a whole (ny x nx) array into norm_slots (reset first). */
int mg_maxabs(const double *a, int64_t ny, int64_t nx, int64_t ld, uint64_t *norm_slots, void *stream);

/*
 * Full-weighting restriction, coarse rows J0 <= J < J1 (1 <= J0, J1 <= cny-1), coarse interior
 * columns; the fine centre of coarse (J,I) is (2J+ofst_j, 2I+ofst_i):
 *     crs[J,I] = f[c]/4 + (f[S]+f[N])*1/8 + ( (f[E]/8 + (f[SE]+f[NE])/16) + (f[W]/8 + (f[SW]+f[NW])/16) )
 * Replaces coarse, core/numba_modules.py:132-160.  The reference uses ofst_i, ofst_j in {0,-1}
 * (core/parallel_multigrid.py:222-228); ofst_j may be any value <= 0 (row slabs with an H-row halo
 * use -H).
 */
int mg_restrict(const double *fine, int64_t fny, int64_t fnx, int64_t fld, double *crs, int64_t cny,
                int64_t cnx, int64_t cld, int64_t J0, int64_t J1, int64_t ofst_i, int64_t ofst_j,
                int64_t nbatch, int64_t fbstride, int64_t cbstride, void *stream);

/*
 * Bilinear prolongation ADDED into the fine grid, fine rows t0 <= t < t1.  With (rt, rc) =
 * (t - ofst_j, c - ofst_i), J = rt>>1, I = rc>>1, a fine cell is updated iff 1 <= rt <= 2*cny-3
 * and 1 <= rc <= 2*cnx-3 (exactly the cells the reference's four passes touch):
 *   rt even, rc even:  += crs[J,I]
 *   rt odd,  rc odd :  += (crs[J+1,I+1] + crs[J+1,I] + crs[J,I+1] + crs[J,I]) / 4
 *   rt even, rc odd :  += (crs[J,I] + crs[J,I+1]) / 2
 *   rt odd,  rc even:  += (crs[J,I] + crs[J+1,I]) / 2
 * Columns are further limited to c_lo <= c <= c_hi.  Replaces interpolate_add_to,
 * core/numba_modules.py:163-180.
 */
int mg_prolong_add(const double *crs, int64_t cny, int64_t cnx, int64_t cld, double *fine, int64_t fny,
                   int64_t fnx, int64_t fld, int64_t t0, int64_t t1, int64_t c_lo, int64_t c_hi,
                   int64_t ofst_i, int64_t ofst_j, int64_t nbatch, int64_t cbstride, int64_t fbstride,
                   void *stream);

/*
 * One fused smoothing pass (temporal blocking): in a single pass over memory
 *     [x_in += prolongation(crs)]  ->  `sweeps` (1..6) Jacobi sweeps  ->  [residual of the result]
 * with  x_in == NULL : start from x = 0 (the zeroed x of core/parallel_multigrid.py:276),
 *       crs  != NULL : bilinear prolongation of crs added first (mg_prolong_add with ofst_i = 0),
 *       r    != NULL : r = b - laplacian(x_out) on the interior cells of rows j0..j1-1 (+ max-norm
 *                      into norm_slots when that is non-NULL, as mg_residual).
 * Produces rows j0 <= j < j1 of x_out (and r); reads rows j0-D .. j1+D-1 of x_in and b, D = sweeps
 * (+1 with r), so those must be valid (halo rows included).  Rows <= fix_lo and >= fix_hi and the
 * columns 0 and nx-1 are Dirichlet cells: their values pass through unchanged.  x_out's cells
 * outside rows j0..j1-1 are not written.  Bit-identical to the same sequence of mg_prolong_add,
 * mg_jacobi and mg_residual calls (smooth_parallel, core/parallel_multigrid.py:100-114, with the
 * preceding interpolate_add_to, :309/:322), but x and b are read once and x written once.
 * Needs 16-byte aligned arrays with an even ld.  (Bit-identical while no intermediate leaves the normal
 * range of a double: the pass keeps stage s's values times 4^s, exact scaling, and unscales at the end.)
  * Batches: nbatch grids, bstride elements apart in b/x_in/x_out/r (even), cstride in crs (and cbstride in
 * coarse_b below); one set of norm slots accumulates the max over all grids.
  * prolong_after = 0: the prolongation comes before the first sweep (x_in required).  prolong_after = 3
 * with sweeps = 9, x_in = NULL: a coarse level's whole smoothing in one pass -- 3 sweeps from x = 0
 * (parallel_multigrid.py:276-277), += P(crs) (:309), 6 more sweeps (:310) -- so that the descent need not
 * store x (mg_smooth_fused_restrict with x_out = NULL) and the ascent need not read it.
 * omega: every sweep of the pass is damped as in mg_jacobi; omega != 1 is built for 1, 3 and 6 sweeps
 * per pass (and 9 with prolong_after = 3), the depths of the V-cycle.
 */
int mg_smooth_fused(const double *b, const double *x_in, double *x_out, double *r, const double *crs,
                    int64_t ny, int64_t nx, int64_t ld, int64_t cny, int64_t cnx, int64_t cld, int64_t j0,
                    int64_t j1, int64_t fix_lo, int64_t fix_hi, int64_t ofst_j, int sweeps, int prolong_after,
                    double h, double omega, uint64_t *norm_slots, int64_t nbatch, int64_t bstride,
                    int64_t cstride, void *stream);
/*
 * mg_smooth_fused with the full-weighting restriction of the residual fused in as a last stage:
 * coarse rows T0 <= T < T1 of coarse_b (cbny x cbnx, interior columns) are written as mg_restrict
 * (ofst_i = 0, the same ofst_j) would compute them from the residual; each coarse row's fine centre
 * row 2T + ofst_j must be one of the produced rows j0..j1-1.  r may be NULL: the residual is then
 * not stored at all (its Dirichlet cells count as 0, which is what a never-written r array holds).
 * x_out may be NULL too: then only coarse_b (and the norm) come out of the pass.
 * Replaces the residual + exchange + coarse sequence of core/parallel_multigrid.py:114,:272-273,
 * :287-288.  Reads rows j0-D .. j1+D-1 with D = sweeps + 2.
 * norm_stride: 0 = one max-norm accumulator for all grids of a batch (as mg_smooth_fused); >= MG_NORM_SLOTS = grid g
 * accumulates into the MG_NORM_SLOTS words at norm_slots + g * norm_stride, so that every problem of a batch has
 * its own max|r| (the stop test of independent problems, core/parallel_multigrid.py:336-345 per problem).
 */
int mg_smooth_fused_restrict(const double *b, const double *x_in, double *x_out, double *r, const double *crs,
                             int64_t ny, int64_t nx, int64_t ld, int64_t cny, int64_t cnx, int64_t cld,
                             int64_t j0, int64_t j1, int64_t fix_lo, int64_t fix_hi, int64_t ofst_j, int sweeps,
                             double h, double omega, uint64_t *norm_slots, int64_t norm_stride, double *coarse_b, int64_t cbny, int64_t cbnx,
                             int64_t cbld, int64_t T0, int64_t T1, int64_t nbatch, int64_t bstride,
                             int64_t cstride, int64_t cbstride, void *stream);
/*
 * The next mg_smooth_fused / mg_smooth_fused_restrict call of the calling host thread produces the rows
 * j2 <= j < j3 as well (j1 <= j2; with a fused restriction also the coarse rows T2 <= T < T3, T1 <= T2, centred on
 * rows of that range), in the same launch and with the same values as a second call with those ranges would.
 * The setting is consumed by that one call, whether it succeeds or not.  `ny` and `stream` are those of that call:
 * a setting whose call was never made is refused (and dropped) by the next fused call of the thread when they differ,
 * instead of adding rows to an unrelated launch.  For the two ends of a row slab, which a
 * pass cut around its halo exchange produces after the interior rows (the slab layout that replaces the four
 * blocks of core/parallel_multigrid.py:39-56): one launch instead of two.
 */
int mg_fused_also_rows(int64_t j2, int64_t j3, int64_t T2, int64_t T3, int64_t ny, void *stream);
/* Tuning knobs of mg_smooth_fused: column pairs per lane (1 or 2), rows per tile (0 = automatic). */
int mg_set_fused_tuning(int unroll, int rows);

/*
 * Coarse-grid smoother: `iterations` x (forward sweep [+ backward sweep]) of SOR with omega = 1.5,
 *     x[j,i] = (1-w)*x[j,i] + (w/4)*(x[j,i+1] + x[j,i-1] + x[j+1,i] + x[j-1,i] - h*h*b[j,i]),
 * in place, followed by the residual (as mg_residual) into r when r != NULL.
 * Replaces smooth_altern (backward = 1), core/numba_modules.py:98-129, and smooth
 * (backward = 0), :69-95.
 *   MG_SOR_LEXICOGRAPHIC  the reference's row-major update order, executed as an anti-diagonal
 *                         wavefront (cells with equal j+i are independent): bit-identical to the
 *                         reference.  One workgroup per grid; the grid lives in LDS when
 *                         ny*(nx rounded up to even) <= MG_SOR_LDS_DOUBLES, else in global memory.
 *   MG_SOR_REDBLACK       colour = (j+i)&1; forward = colour 0 then 1, backward = 1 then 0.
 *                         Not the reference's ordering: a different (weaker for few sweeps)
 *                         smoother, offered for large coarse grids.
 */
#define MG_SOR_LEXICOGRAPHIC 0
#define MG_SOR_REDBLACK 1
#define MG_SOR_LDS_DOUBLES 20000
int mg_sor(const double *b, double *x, double *r, int64_t ny, int64_t nx, int64_t ld, double h,
           int64_t iterations, int backward, int ordering, int64_t nbatch, int64_t bstride, void *stream);

/*
 * One colour of a red-black Gauss-Seidel sweep, in place, rows j0 <= j < j1 (1 <= j0, j1 <= ny-1): the interior cells
 * with (j + i) % 2 == colour take the value of mg_jacobi's expression computed from the current array.  A sweep is
 * colour 0 followed by colour 1.  No counterpart in the reference (it smooths the fine levels with Jacobi,
 * core/numba_modules.py:53-66): a smoother option of this build, which unlike undamped Jacobi damps the checkerboard
 * mode of the error.
 */
int mg_gauss_seidel_rb(const double *b, double *x, int64_t ny, int64_t nx, int64_t ld, int64_t j0, int64_t j1, double h,
                       int colour, int64_t nbatch, int64_t bstride, void *stream);

/*
 * dst[0:ny, 0:nx] = src[0:ny, 0:nx] with independent leading dimensions.  The data movement of
 * split (core/numba_modules.py:10-20) and of the stitch in gather_blocks
 * (core/parallel_multigrid.py:135-138): the caller offsets the pointers to select the block.
 */
int mg_copy2d(double *dst, int64_t dld, const double *src, int64_t sld, int64_t ny, int64_t nx, void *stream);

/*
 * Halo exchange of the reference's four 2x2 block ranks, for all four at once: v holds the four
 * (m x m) blocks of one variable bstride elements apart (rank order 0..3), stage is scratch of
 * 8*m doubles.  Replaces Buffers.exchange_buffers, core/parallel_multigrid.py:83-97 (pack all,
 * then unpack east/west before north/south; corner cells included, as the reference moves whole
 * lines).  Used by the block-layout drop-in Multigrid; the slab engine exchanges rows instead.
 */
int mg_block_exchange(double *v, int64_t m, int64_t ld, int64_t bstride, double *stage, void *stream);

/*
 * mg_block_exchange without a staging argument and without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory or carry a workspace -- code under
 * hipGraph capture, tight serving loops, harnesses that forbid hipMalloc.  The 8*m doubles of staging
 * live in a device array built into the library (MG_BLOCK_EXCHANGE_STATIC_MAX_SIDE * 64 bytes = 2 MiB,
 * held by every GPU the library runs on until the process ends); blocks with m above
 * MG_BLOCK_EXCHANGE_STATIC_MAX_SIDE are rejected with an error.  The call makes no hipMalloc/hipFree;
 * the first call on a device looks up that device's copy of the array (make it outside stream capture).
 * All calls on one GPU share the array: two calls that overlap in time on one GPU (different
 * streams or host threads) race; calls on one stream are ordered and safe.  Results equal
 * mg_block_exchange bit for bit.
 */
#define MG_BLOCK_EXCHANGE_STATIC_MAX_SIDE 32768
int mg_block_exchange_static(double *v, int64_t m, int64_t ld, int64_t bstride, void *stream);

/* a[0:n] = 0.0 (the `x[:] = 0` of core/parallel_multigrid.py:276,291,295), as a memset node. */
int mg_zero(double *a, int64_t n, void *stream);

/* split(A_in, A_out, rank), core/numba_modules.py:10-20: A_out (nout x nout) = corner `rank` of A_in. */
int mg_split(const double *a_in, int64_t in_ny, int64_t in_nx, int64_t in_ld, double *a_out, int64_t nout,
             int64_t out_ld, int64_t rank, void *stream);

/*
 * hipGraph capture of whatever is enqueued on `stream` between begin and end (the coarse tail
 * of the V-cycle, or the whole cycle on one GPU).  mg_graph_end returns an executable graph.
 */
int mg_graph_begin(void *stream);
/*
 * The same on a stream the library makes for this one capture and returns: enqueue on *stream_out what is to be
 * recorded, then pass it to mg_graph_end, which releases it.  For processes that also hold an RCCL communicator
 * through torch.distributed: its watchdog thread polls the end events of outstanding collectives, and on this HIP
 * runtime that poll fails (hipErrorCapturedEvent; torch then ends the process) while a stream that has waited for
 * such an event is capturing.  A stream that has never carried anything is safe.  (No counterpart in the reference,
 * whose ranks launch nothing asynchronously.)
 */
int mg_graph_begin_own(void **stream_out);
int mg_graph_end(void *stream, void **graph_exec_out);
int mg_graph_launch(void *graph_exec, void *stream);
int mg_graph_destroy(void *graph_exec);

/*
 * Rank-to-rank data movement of the row-slab layout over RCCL (one process per GPU; the rows travel over xGMI), as raw
 * RCCL calls enqueued on the caller's stream, in order with its kernels.  The library does not link RCCL: the entry
 * points are looked up in the RCCL the process has loaded (`rccl_path`: that library's file, or NULL to search the
 * process), so a communicator made by another component of the process -- torch.distributed's, for one -- can be used.
 *
 *   mg_comm_attach     wraps an existing ncclComm_t (borrowed: mg_comm_destroy leaves it alone).  Rank, size and device
 *                      are the communicator's.
 *   mg_comm_unique_id / mg_comm_init   a communicator of the library's own: one rank obtains the 128-byte id, hands it to
 *                      the others by whatever means the caller has, every rank calls mg_comm_init (ncclCommInitRank on
 *                      the current device); mg_comm_destroy destroys it.
 *   mg_halo_prepare    fixes the messages of one halo exchange -- blocks of whole rows this rank sends to / receives
 *                      from its neighbours -- once, as the reference prepares persistent requests per array and level
 *                      (Buffers.__init__: Send_init / Recv_init, core/parallel_multigrid.py:58-81).  Messages between
 *                      one pair of ranks are matched in the order given, so every rank lists the arrays in the same
 *                      order.  A received block must not overlap any other block of the exchange.
 *   mg_halo_exchange   carries the prepared exchange out on `stream`: one RCCL group of ncclSend / ncclRecv
 *                      (Buffers.exchange_buffers, :83-97: Startall(recv), pack, Startall(send), Waitall -- no packing here,
 *                      a row block is contiguous).  Returns when the group is enqueued.
 *   mg_allgather_rows  out[q * count .. (q+1) * count) <- `in` of rank q, for every rank q: the coarse right-hand side
 *                      of the first replicated level reaches every rank (gather_blocks, :117-138: four Bcasts and a
 *                      stitch; row slabs need no stitch).  `in` may be this rank's place inside `out`.
 *   mg_allreduce_max   element-wise MAX over the ranks of `count` uint64 words, in place: the max-norm accumulator of
 *                      mg_residual / the fused passes, whose words are bit patterns of |r| (the stop test's allreduce,
 *                      :342-345, on the norm itself instead of a vote).
 * All calls of one communicator must come from one host thread at a time with its device current, in the same order
 * on every rank.  RCCL orders the operations of one communicator among themselves even when they are enqueued on
 * different streams.
 */
typedef struct mg_comm mg_comm;
typedef struct mg_halo mg_halo;
int mg_comm_attach(void *nccl_comm, const char *rccl_path, mg_comm **out);
int mg_comm_unique_id(const char *rccl_path, void *id128);
int mg_comm_init(const void *id128, int rank, int world, const char *rccl_path, mg_comm **out);
int mg_comm_info(mg_comm *c, int *rank, int *world, int *rccl_version);
int mg_comm_destroy(mg_comm *c);
int mg_halo_prepare(mg_comm *c, int nsend, double *const *send_ptr, const int64_t *send_count, const int *send_peer,
                    int nrecv, double *const *recv_ptr, const int64_t *recv_count, const int *recv_peer, mg_halo **out);
int mg_halo_exchange(mg_halo *h, void *stream);
int mg_halo_destroy(mg_halo *h);
int mg_allgather_rows(mg_comm *c, const double *in, double *out, int64_t count, void *stream);
int mg_allreduce_max(mg_comm *c, uint64_t *words, int64_t count, void *stream);

/*
 * The whole solver behind a handle: Multigrid.__init__ / Multigrid.solve of
 * core/parallel_multigrid.py:153-348 on the undecomposed (gny x gnx) grid, boundary included, on the
 * current device, for callers that do not go through the Python engine.  The library owns the
 * device arrays of every level (hipMalloc at create, hipFree at destroy) and its own streams; all
 * calls of one handle must come from one thread at a time, with the device it was created on
 * current.  Results equal the reference's schedule on the global grid bit for bit (the Python
 * engine multigrid_amd.engine.SlabMultigrid on one rank issues the same passes).
 *
 *   mg_solver_create   level layout of __init__ (:182-219): n_para Jacobi levels of
 *                      ((gny-1)>>l)+1 x ((gnx-1)>>l)+1 points with mesh width h0*2^l, then
 *                      whole-domain SSOR levels down to 3 points per side; gny-1 and gnx-1 must be
 *                      divisible by 2^n_para with at least 2 cells left.  All arrays start at zero.
 *   flags              MG_SOLVER_FIX_COARSE_ASCENT  the ascent through the whole-domain levels
 *                                                   reaches the first one (the reference's
 *                                                   range(2, ..) at :302 stops one short)
 *                      MG_SOLVER_REDBLACK           red-black instead of lexicographic SSOR
 *                      MG_SOLVER_NO_GRAPH           launch kernel by kernel instead of replaying a
 *                                                   hipGraph per cycle
 *                      MG_SOLVER_LAZY_DEAD_LEVELS   skip the whole-domain levels below the first,
 *                                                   which nothing reads under the reference's loop
 *                                                   bounds (ignored with FIX_COARSE_ASCENT)
 *                      MG_SOLVER_NO_PLACEMENT       on grids of >= 1 GiB per array create picks the finest
 *                                                   level's arrays among up to 26 candidate allocations by
 *                                                   how fast the finest pass runs when it writes into each
 *                                                   (about half a second; results unchanged; the same launch
 *                                                   is 10 % slower on some allocations than on others); this
 *                                                   flag takes the arrays as the allocator hands them out
 *                      MG_SOLVER_TUNE_PLACEMENT     accepted and ignored (the selection used to be opt-in)
 *   mg_solver_set      "jacobi_omega" (0 < w <= 1, 1 = the reference's sweep), "max_cycles"
 *                      (default 100, :270), "recompute_min_nx" / "recompute_min_blocks" (size from
 *                      which a coarse level recomputes its pre-smoothing instead of storing it)
 *   mg_solver_placement  number of candidate arrays the placement selection of create timed
 *                      (-1: none -- a small grid, MG_SOLVER_NO_PLACEMENT, or memory was short)
 *   mg_solver_upload   finest-level b (MG_SOLVER_B) or x (MG_SOLVER_X: initial guess and Dirichlet
 *                      boundary values) from a host or device array of gny rows, src_ld apart
 *   mg_solver_solve    solve() (:232-348): one sweep + residual, then V-cycles until max|r| <= eps,
 *                      the 1.1x divergence latch trips (:337-340) or max_cycles ran; returns the
 *                      cycle count (the reference's .it) and the last max|r|
 *   mg_solver_presmooth / mg_solver_cycles   the opening sweep (:264-266) and exactly k cycles
 *                      without stop test, max|r| after each into errs[k] (may be NULL)
 *   mg_solver_history  max|r| after each cycle of the last mg_solver_solve, and whether the latch
 *                      tripped; *count receives the number of cycles
 *   mg_solver_download / mg_solver_array   any level's array to host memory / as a device pointer
 *                      (valid until destroy; x of the finest level moves between two arrays, ask
 *                      again after every solve).  MG_SOLVER_R exists on the finest level only.
 */
typedef struct mg_solver mg_solver;
#define MG_SOLVER_FIX_COARSE_ASCENT 1
#define MG_SOLVER_REDBLACK 2
#define MG_SOLVER_NO_GRAPH 4
#define MG_SOLVER_LAZY_DEAD_LEVELS 8
#define MG_SOLVER_TUNE_PLACEMENT 16
#define MG_SOLVER_NO_PLACEMENT 32
enum { MG_SOLVER_X = 0, MG_SOLVER_B = 1, MG_SOLVER_R = 2, MG_SOLVER_XW = 3, MG_SOLVER_BW = 4, MG_SOLVER_RW = 5 };
int mg_solver_create(int64_t gny, int64_t gnx, int n_para, double h0, int flags, mg_solver **out);
int mg_solver_destroy(mg_solver *s);
int mg_solver_set(mg_solver *s, const char *key, double value);
int mg_solver_placement(mg_solver *s, int *candidates);
int mg_solver_upload(mg_solver *s, int which, const double *src, int64_t src_ld);
int mg_solver_download(mg_solver *s, int which, int level, double *dst, int64_t dst_ld);
int mg_solver_array(mg_solver *s, int which, int level, double **device_ptr, int64_t *ny, int64_t *nx, int64_t *ld);
int mg_solver_presmooth(mg_solver *s);
int mg_solver_cycles(mg_solver *s, int k, double *errs);
int mg_solver_solve(mg_solver *s, double eps, int *it_out, double *err_out);
int mg_solver_history(mg_solver *s, double *errs, int capacity, int *count, int *diverged);

/*
 * The same solver on SEVERAL GPUs, one process (rank) per GPU: Multigrid.__init__ / Multigrid.solve of
 * core/parallel_multigrid.py:153-348 with the reference's four block ranks (:49-56) replaced by `world` row slabs of
 * the global (gny x gnx) grid.  Rank r owns global rows [r R, (r+1) R), R = (gny-1)/world (the last rank also the bottom
 * boundary row) and keeps 12 halo rows above and below them.  The finest `n_dist` Jacobi levels are cut into slabs
 * (dist_levels, or 0 = a cost model's choice; rows per rank must be divisible by 2^n_dist and the coarsest slab at
 * least 12 rows tall); per cycle a rank issues one prepared RCCL send/receive group per fused pass of a distributed
 * level (mg_halo_exchange; the reference exchanges after every sweep, :83-97,:108-111), one all-gather of the first
 * replicated level's right-hand side (gather_blocks, :117-138) -- from there on every rank computes alike, as the
 * reference replicates its whole-domain levels -- and one MAX-reduction of the norm (:342-345).  Every call of a
 * cycle is issued on one stream, so a rank's whole cycle, communication included, is one hipGraph launch.  Results equal
 * mg_solver_* on one GPU, and the reference's schedule on the undecomposed grid, bit for bit, for every `world`.
 *
 *   mg_dist_create     collective: every rank calls it with the same grid, flags and dist_levels, its own rank, and a
 *                      communicator of `world` ranks in which it has that rank (mg_comm_attach / mg_comm_init; NULL
 *                      for world = 1).  flags: MG_SOLVER_FIX_COARSE_ASCENT, _REDBLACK, _NO_GRAPH, _LAZY_DEAD_LEVELS as
 *                      for mg_solver_create (MG_SOLVER_NO_PLACEMENT too: by default a rank whose finest arrays take
 *                      >= 1 GiB each chooses their placement as mg_solver_create does; mg_dist_placement reports the
 *                      number of candidates timed, -1: none), and MG_DIST_LOOPBACK: a rehearsal of ONE rank of `world` alone, on a
 *                      one-rank communicator with the rank as its own neighbours (values are meaningless then; the
 *                      calls, their sizes and their cost are the real job's).
 *   mg_dist_upload / mg_dist_download   the rank's OWNED rows of the finest-level b or x (download: also r), a host or
 *                      device array of owned rows x gnx (mg_dist_info tells how many rows and where they start);
 *                      downloading r is collective (one halo row of x is exchanged to bring r up to date).
 *   mg_dist_presmooth / mg_dist_cycles / mg_dist_solve / mg_dist_history / mg_dist_set   as their mg_solver_*
 *                      namesakes; collective; errs and the returned error are max|r| over ALL ranks.
 *   mg_dist_info       the number of distributed levels (0 on one rank), the global row of the first owned row, the
 *                      number of owned rows, and the communication rounds (exchanges and all-gathers) issued so far.
 * All calls of one handle come from one thread, with the device it was created on current, in the same order on
 * every rank.
 */
typedef struct mg_dist_solver mg_dist_solver;
#define MG_DIST_LOOPBACK 64
int mg_dist_create(int64_t gny, int64_t gnx, int n_para, double h0, int flags, mg_comm *comm, int rank, int world,
                   int dist_levels, mg_dist_solver **out);
int mg_dist_destroy(mg_dist_solver *s);
int mg_dist_set(mg_dist_solver *s, const char *key, double value);
int mg_dist_info(mg_dist_solver *s, int *n_dist, int64_t *first_global_row, int64_t *owned_rows, int64_t *rounds);
int mg_dist_placement(mg_dist_solver *s, int *candidates);
int mg_dist_upload(mg_dist_solver *s, int which, const double *src, int64_t src_ld);
int mg_dist_download(mg_dist_solver *s, int which, double *dst, int64_t dst_ld);
int mg_dist_presmooth(mg_dist_solver *s);
int mg_dist_cycles(mg_dist_solver *s, int k, double *errs);
int mg_dist_solve(mg_dist_solver *s, double eps, int *it_out, double *err_out);
int mg_dist_history(mg_dist_solver *s, double *errs, int capacity, int *count, int *diverged);

#ifdef __cplusplus
}
#endif
#endif /* MG_HIP_H */

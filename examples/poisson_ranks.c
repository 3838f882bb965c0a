/*
 * The script part of the reference's solver module (core/parallel_multigrid.py:351-415) from plain C on SEVERAL
 * GPUs, one process per GPU, through mg_comm_* and mg_dist_* of libmg_hip.so: what `mpiexec -n 4 python
 * parallel_multigrid.py` is to the reference (core/Makefile:7), with row slabs in the place of its four blocks.
 * A single Gaussian source on the (2^(n+1)+1)^2 grid over [-10, 10]^2, cut into `world` row slabs; this process is
 * rank `rank` and owns one of them.  Solved twice (the second solve warm-starts), the reference's lines printed
 * per solve by rank 0, then k V-cycles without stop test timed.
 *
 *   poisson_ranks n n_para world rank [k]
 *
 * How the ranks find each other: the communicator is the library's own (mg_comm_unique_id / mg_comm_init; no MPI,
 * no torch).  With MG_RANKS_ID_FILE=/some/shared/path in the environment rank 0 writes the 128-byte id there and the
 * other ranks read it: start one process per GPU, each with its own device visible (HIP_VISIBLE_DEVICES=<rank>), e.g.
 *     for r in 0 1 2 3; do HIP_VISIBLE_DEVICES=$r MG_RANKS_ID_FILE=/tmp/mg.id ./poisson_ranks 13 9 4 $r 20 & done; wait
 * Without it the process rehearses its rank ALONE on a one-rank communicator with itself as every neighbour
 * (MG_DIST_LOOPBACK): the calls, their sizes and their cost are the real job's, the values are not.
 * MG_RCCL_LIB names the RCCL library (default /opt/rocm/lib/librccl.so); world = 1 needs none.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "mg_hip.h"

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

#define CHECK(call)                                                        \
    do {                                                                   \
        if (call) {                                                        \
            fprintf(stderr, "%s failed: %s\n", #call, mg_last_error());    \
            return 1;                                                      \
        }                                                                  \
    } while (0)

/* rank 0 -> everybody: 128 bytes through a file that appears under its final name only when complete */
static int share_id(const char *path, int rank, unsigned char id[128])
{
    if (rank == 0) {
        char tmp[4096];
        snprintf(tmp, sizeof(tmp), "%s.part", path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, 128, f) != 128 || fclose(f) != 0 || rename(tmp, path) != 0) return 1;
        return 0;
    }
    for (int tries = 0; tries < 600; tries++) {
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(id, 1, 128, f);
            fclose(f);
            if (got == 128) return 0;
        }
        usleep(100000);
    }
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: poisson_ranks n n_para world rank [k]\n");
        return 2;
    }
    const int n = atoi(argv[1]), n_para = atoi(argv[2]), world = atoi(argv[3]), rank = atoi(argv[4]);
    const int k = argc > 5 ? atoi(argv[5]) : 0;
    const double b_max = 50, epsilon = b_max * 0.0005;
    const int64_t nx1 = ((int64_t)1 << n) + 1, nx0 = ((int64_t)1 << (n + 1)) + 1;
    const double h = 20.0 / (double)(nx0 - 1);
    const char *rccl = getenv("MG_RCCL_LIB") ? getenv("MG_RCCL_LIB") : "/opt/rocm/lib/librccl.so";
    const char *id_file = getenv("MG_RANKS_ID_FILE");
    mg_comm *comm = NULL;
    int flags = 0;
    if (world > 1) {
        unsigned char id[128];
        if (id_file) { /* a real job: every rank a process with its own GPU */
            if (rank == 0) CHECK(mg_comm_unique_id(rccl, id));
            if (share_id(id_file, rank, id)) {
                fprintf(stderr, "rank %d: cannot %s the communicator id at %s\n", rank, rank ? "read" : "write", id_file);
                return 1;
            }
            CHECK(mg_comm_init(id, rank, world, rccl, &comm));
        } else { /* this rank alone, its own neighbour */
            CHECK(mg_comm_unique_id(rccl, id));
            CHECK(mg_comm_init(id, 0, 1, rccl, &comm));
            flags |= MG_DIST_LOOPBACK;
        }
    }
    mg_dist_solver *s;
    CHECK(mg_dist_create(nx0, nx0, n_para, h, flags, comm, rank, world, 0, &s));
    int n_dist;
    int64_t first, rows, rounds0, rounds1;
    CHECK(mg_dist_info(s, &n_dist, &first, &rows, &rounds0));
    double *b = malloc(sizeof(double) * (size_t)(rows * nx0)), *x = malloc(sizeof(double) * (size_t)(rows * nx0));
    if (!b || !x) return 1;
    for (int64_t j = 0; j < rows; j++)
        for (int64_t i = 0; i < nx0; i++) {
            const double X = -10 + 20.0 * (double)i / (double)(nx0 - 1), Y = -10 + 20.0 * (double)(first + j) / (double)(nx0 - 1);
            b[j * nx0 + i] = b_max * exp(-(X * X + Y * Y) * 4);
        }
    CHECK(mg_dist_upload(s, MG_SOLVER_B, b, nx0));
    for (int rep = 0; rep < 2; rep++) {
        int it;
        double err;
        const double t0 = now();
        CHECK(mg_dist_solve(s, epsilon, &it, &err));
        const double t = now() - t0;
        if (rank == 0 || (flags & MG_DIST_LOOPBACK))
            printf("nx = %lld\nn_para = %d\nm_err/b_max=%.1e\ntime %.3e s/point\ntime %.3e s\nv cycles %d\n", (long long)nx1, n_para,
                   err / b_max, t / (double)(nx0 - 2), t, it);
    }
    if (k > 0) {
        double *errs = malloc(sizeof(double) * (size_t)k);
        if (!errs) return 1;
        CHECK(mg_dist_cycles(s, 2, errs)); /* warm-up */
        CHECK(mg_dist_info(s, NULL, NULL, NULL, &rounds0));
        const double t0 = now();
        CHECK(mg_dist_cycles(s, k, errs));
        const double t = now() - t0;
        CHECK(mg_dist_info(s, NULL, NULL, NULL, &rounds1));
        printf("rank %d of %d (rows %lld..%lld, %d distributed levels): %d cycles without stop test: %.4f ms per cycle, "
               "%.1f communication rounds per cycle, last max|r| %.6e\n", rank, world, (long long)first, (long long)(first + rows - 1),
               n_dist, k, 1e3 * t / k, (double)(rounds1 - rounds0) / k, errs[k - 1]);
        free(errs);
    }
    CHECK(mg_dist_download(s, MG_SOLVER_X, x, nx0));
    double sum = 0;
    for (int64_t q = 0; q < rows * nx0; q++) sum += x[q];
    printf("rank %d: sum of the owned rows of x: %.10e\n", rank, sum);
    CHECK(mg_dist_destroy(s));
    CHECK(mg_comm_destroy(comm));
    free(b), free(x);
    if (id_file && rank == 0 && world > 1) remove(id_file);
    return 0;
}

/*
 * The script part of the reference's solver module (core/parallel_multigrid.py:351-415) from plain C,
 * through the handle API of libmg_hip.so: a single Gaussian source on the (2^(n+1)+1)^2 grid over
 * [-10, 10]^2, solved twice (the second solve warm-starts from the first), the reference's six lines
 * printed per solve, then max|b - A x| of the downloaded solution evaluated on the host.
 *
 *   poisson_main [n] [n_para] [k]    (defaults 8 and 4, the reference's; k > 0: after the two solves, k more
 *                                     V-cycles without stop test are timed, mg_solver_cycles)
 *
 * Large grids (from 1 GiB per array) let create pick the placement of the finest arrays.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

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

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8, n_para = argc > 2 ? atoi(argv[2]) : 4;
    const double b_max = 50, epsilon = b_max * 0.0005;
    const int64_t nx1 = ((int64_t)1 << n) + 1, nx0 = ((int64_t)1 << (n + 1)) + 1;
    const double h = 20.0 / (double)(nx0 - 1);
    double *b = malloc(sizeof(double) * nx0 * nx0), *x = malloc(sizeof(double) * nx0 * nx0);
    if (!b || !x) return 1;
    for (int64_t j = 0; j < nx0; j++)
        for (int64_t i = 0; i < nx0; i++) {
            const double X = -10 + 20.0 * (double)i / (double)(nx0 - 1), Y = -10 + 20.0 * (double)j / (double)(nx0 - 1);
            b[j * nx0 + i] = b_max * exp(-(X * X + Y * Y) * 4);
        }
    mg_solver *s;
    CHECK(mg_solver_create(nx0, nx0, n_para, h, 0, &s)); /* no flag: the reference's schedule */
    CHECK(mg_solver_upload(s, MG_SOLVER_B, b, nx0));
    for (int rep = 0; rep < 2; rep++) {
        int it;
        double err;
        const double t0 = now();
        CHECK(mg_solver_solve(s, epsilon, &it, &err));
        const double t = now() - t0;
        printf("nx = %lld\nn_para = %d\nm_err/b_max=%.1e\ntime %.3e s/point\ntime %.3e s\nv cycles %d\n", (long long)nx1, n_para,
               err / b_max, t / (double)(nx0 - 2), t, it);
    }
    const int k = argc > 3 ? atoi(argv[3]) : 0;
    if (k > 0) {
        double *errs = malloc(sizeof(double) * (size_t)k);
        if (!errs) return 1;
        CHECK(mg_solver_cycles(s, 2, errs)); /* warm-up */
        const double t0 = now();
        CHECK(mg_solver_cycles(s, k, errs));
        const double t = now() - t0;
        printf("%d cycles without stop test: %.4f ms per cycle, %.4g unknowns*V-cycles/s, last max|r| %.6e\n", k, 1e3 * t / k,
               (double)(nx0 - 2) * (double)(nx0 - 2) * k / t, errs[k - 1]);
        free(errs);
    }
    CHECK(mg_solver_download(s, MG_SOLVER_X, 0, x, nx0));
    double rmax = 0;
    for (int64_t j = 1; j < nx0 - 1; j++)
        for (int64_t i = 1; i < nx0 - 1; i++) {
            const double lap = (x[j * nx0 + i - 1] + x[j * nx0 + i + 1] + x[(j - 1) * nx0 + i] + x[(j + 1) * nx0 + i] -
                                4 * x[j * nx0 + i]) / (h * h);
            const double r = fabs(b[j * nx0 + i] - lap);
            rmax = r > rmax ? r : rmax;
        }
    printf("residual of the downloaded solution / max(B): %.1e\n", rmax / b_max);
    CHECK(mg_solver_destroy(s));
    free(b), free(x);
    return 0;
}

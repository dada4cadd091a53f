/* bench_common.h -- bits shared by inverse_bench.c and gauss_bench.c */
#ifndef MATINV_BENCH_COMMON_H
#define MATINV_BENCH_COMMON_H

#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/types.h"
#include "../../include/helper_cpu.h"
#include "../../include/timer.h"

/* sum_i |x_i - y_i| : the reference's error metric (saxpy + sasum, src/inverse_bench.c:33-39,49-51) */
static inline double abs_diff_sum(const DataType *x, const DataType *y, size_t count)
{
    double s = 0;
    for (size_t i = 0; i < count; ++i) s += fabs((double)x[i] - (double)y[i]);
    return s;
}

/* one result line, same columns as BENCH_REPORT (src/inverse_bench.c:53-70 / src/gauss_bench.cu:503-528) */
static inline void report_line(bool csv, const char *name, int numMatrices, int n, int numReps, double total_ms,
                               double mean_ms, double var_ms, double err)
{
    if (csv) {
        if (numReps > 1) printf("%d %d %d %s %e %e %e %e\n", numMatrices, n, numReps, name, total_ms, mean_ms, var_ms, err);
        else printf("%d %d %d %s %e %e\n", numMatrices, n, numReps, name, total_ms, err);
    } else if (numReps > 1) {
        printf("%s - %d %dx%d matrices, replicated %d times, runtime %.4f ms (%.4f ms average, %.4f ms variance), "
               "average error %.4e\n", name, numMatrices, n, n, numReps, total_ms, mean_ms, var_ms, err);
    } else {
        printf("%s - %d %dx%d matrices, replicated %d times, runtime %.4f ms, average error %.4e\n", name, numMatrices,
               n, n, numReps, total_ms, err);
    }
}

static inline bool detailed_logging(void)
{
    const char *s = getenv("MATINV_DETAILED_LOGGING");
    return s && *s && *s != '0';
}

#endif

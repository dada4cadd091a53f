/*
 * timer.h -- TIMER_* macros of the bench CLIs. Same macro names, units and log line as
 * /root/reference/include/timer.h (CLOCK_MONOTONIC :99-104, Welford accumulator :106-111, log line :8-9), built
 * on one small struct instead of five loose variables per timer.
 *
 *   TIMER_INIT(name)      declare a stopwatch               TIMER_START/STOP(name)   take the two timestamps
 *   TIMER_ACC_INIT(name)  declare its accumulator           TIMER_ACC(name)          fold the last interval in
 *   TIMER_ELAPSED(name) ms of the last interval, TIMER_ELAPSED_NS(name) the same in ns,
 *   TIMER_TOTAL / TIMER_MEAN / TIMER_VARIANCE(name) over the accumulated intervals (ms, ms, ms^2; sample variance)
 *   TIMER_LOG(name, numMatrices, n)   prints  name,numMatrices,n,ms,ns\r\n   -- the format results/generate_plots.m reads
 */
#ifndef HEADER_TIMER_INCLUDED
#define HEADER_TIMER_INCLUDED

#include <stddef.h>
#include <stdio.h>
#include <time.h>

#define TIMER_BILLION 1000000000

typedef struct {
    struct timespec t0, t1;
} matinv_stopwatch;

typedef struct {
    size_t count;
    double total_ms, mean_ms, m2;
} matinv_timer_acc;

static inline unsigned long matinv_stopwatch_ns(const matinv_stopwatch *w)
{
    return (unsigned long)((long long)(w->t1.tv_sec - w->t0.tv_sec) * TIMER_BILLION + (w->t1.tv_nsec - w->t0.tv_nsec));
}

static inline void matinv_timer_fold(matinv_timer_acc *a, double ms)
{
    /* Welford's online mean / M2 */
    double delta = ms - a->mean_ms;
    a->count += 1;
    a->total_ms += ms;
    a->mean_ms += delta / (double)a->count;
    a->m2 += delta * (ms - a->mean_ms);
}

#define TIMER_INIT(name) matinv_stopwatch timer_##name = {{0, 0}, {0, 0}};
#define TIMER_ACC_INIT(name) matinv_timer_acc timer_acc_##name = {0, 0.0, 0.0, 0.0};
#define TIMER_START(name) clock_gettime(CLOCK_MONOTONIC, &timer_##name.t0);
#define TIMER_STOP(name) clock_gettime(CLOCK_MONOTONIC, &timer_##name.t1);
#define TIMER_ELAPSED_NS(name) matinv_stopwatch_ns(&timer_##name)
#define TIMER_ELAPSED(name) ((double)TIMER_ELAPSED_NS(name) / 1e6)
#define TIMER_ACC(name) matinv_timer_fold(&timer_acc_##name, TIMER_ELAPSED(name));
#define TIMER_TOTAL(name) (timer_acc_##name.total_ms)
#define TIMER_MEAN(name) (timer_acc_##name.mean_ms)
#define TIMER_VARIANCE(name) (timer_acc_##name.count > 1 ? timer_acc_##name.m2 / (double)(timer_acc_##name.count - 1) : 0.0)
#define TIMER_ACC_RESET(name) timer_acc_##name = (matinv_timer_acc){0, 0.0, 0.0, 0.0};
#define TIMER_LOG(name, numMatrices, n) \
    printf(#name ",%d,%d,%.4f,%lu\r\n", numMatrices, n, TIMER_ELAPSED(name), TIMER_ELAPSED_NS(name));

#endif

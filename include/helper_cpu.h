/*
 * helper_cpu.h -- host helpers of the bench CLIs: fatal-error macros and the `.mats` reader.
 *
 * Same names and contracts as /root/reference/include/helper_cpu.h:4-38 and src/helper.cu:15-99:
 *   ensure(cond, fmt, ...)  print "ENSURE FAILED file:line", the message, perror() when errno is set, exit(EXIT_FAILURE)
 *   fail(fmt, ...)          print "file:line<TAB>message", exit(EXIT_FAILURE)
 *   div_ceil(x, y)          ceiling division for positive x
 *   readMatricesFile        text file "K m n" + K*m rows of n numbers -> one malloc'd block, each matrix COLUMN-major
 *   replicateMatrices       the whole list repeated `numReplications` times (frees and replaces *matrices)
 *   printMatrix[List]       row-wise dump of column-major data
 * MAX_MATRIX_BYTE_READ keeps the reference's 64 MiB cap per file (helper_cpu.h:4 there).
 */
#ifndef HEADER_HELPER_CPU_INCLUDED
#define HEADER_HELPER_CPU_INCLUDED

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>

#include "types.h"

#define MAX_MATRIX_BYTE_READ 67108864

#ifdef __cplusplus
extern "C" {
#endif

/* shared tail of ensure()/fail(): never returns */
void matinv_host_die(int is_ensure, const char *file, int line, const char *fmt, ...);

#define fail(...) matinv_host_die(0, __FILE__, __LINE__, __VA_ARGS__)
#define ensure(condition, ...)                                         \
    do {                                                               \
        if (!(condition)) matinv_host_die(1, __FILE__, __LINE__, __VA_ARGS__); \
    } while (0)

#define div_ceil(x, y) (1 + (((x)-1) / (y)))

void printMatrix(Array a, int M, int N);
void printMatrixList(Array a, int N, int batchSize);
void readMatricesFile(const char *path, int *numMatrices, int *m, int *n, Array *matrices);
void replicateMatrices(Array *matrices, const int M, const int N, const int numMatrices, const int numReplications);

#ifdef __cplusplus
}
#endif

#endif

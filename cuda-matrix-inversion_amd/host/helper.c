/*
 * helper.c -- `.mats` reader, batch replication and printers for the bench CLIs.
 * Host-side counterpart of /root/reference/src/helper.cu:15-99 (readMatricesFile :15-52, replicateMatrices :54-72,
 * printMatrix :74-84, printMatrixList :87-99); the pitched device allocator of :103-118 has no equivalent here
 * because the library keeps device memory behind its C ABI.
 */
#include <stdarg.h>
#include <string.h>

#include "../../include/helper_cpu.h"

void matinv_host_die(int is_ensure, const char *file, int line, const char *fmt, ...)
{
    va_list ap;
    if (is_ensure) fprintf(stderr, "ENSURE FAILED %s:%d\r\n", file, line);
    else fprintf(stderr, "%s:%d\t", file, line);
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\r\n");
    if (is_ensure && errno) perror("possible reason for failure from ERRNO");
    exit(EXIT_FAILURE);
}

/* File: "numMatrices m n", then numMatrices*m text rows of n numbers (row by row). Memory: one block, matrix k at
 * k*m*n, element (i, j) at j*m + i (column-major) -- the layout the reference produces (helper.cu:38-48).
 * The whole file is slurped and tokenised with strtol / strtod (one pass, no per-number stdio call: the synthetic
 * 128x128 fixtures are tens of megabytes of text); failures keep the reference's messages. */
static char *slurp(const char *path, size_t *len)
{
    FILE *fp = fopen(path, "rb");
    ensure(fp != NULL, "could not open matrix file %s", path);
    size_t cap = 1 << 16, used = 0;
    char *buf = (char *)malloc(cap + 1);
    ensure(buf != NULL, "could not allocate 0x%lX bytes of memory for file %s", (unsigned long)cap, path);
    for (;;) {
        used += fread(buf + used, 1, cap - used, fp);
        if (used < cap) break;
        cap *= 2;
        buf = (char *)realloc(buf, cap + 1);
        ensure(buf != NULL, "could not allocate 0x%lX bytes of memory for file %s", (unsigned long)cap, path);
    }
    fclose(fp);
    buf[used] = '\0';
    *len = used;
    return buf;
}

void readMatricesFile(const char *path, int *numMatrices, int *m, int *n, Array *matrices)
{
    size_t len = 0;
    char *text = slurp(path, &len), *cur = text, *end = NULL;
    long dims[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) {
        errno = 0;
        dims[d] = strtol(cur, &end, 10);
        ensure(end != cur, "could not read number of matrices from file %s", path);
        cur = end;
    }
    ensure(dims[0] >= 0 && dims[1] >= 0 && dims[2] >= 0, "negative dimension in file %s", path);
    const size_t K = (size_t)dims[0], rows = (size_t)dims[1], cols = (size_t)dims[2];
    const size_t bytes = sizeof(DataType) * K * rows * cols;
    ensure(bytes <= MAX_MATRIX_BYTE_READ,
           "cannot read file %s because the allocated array would be bigger than 0x%lX bytes", path,
           (unsigned long)bytes);
    Array block = (Array)malloc(bytes ? bytes : 1);
    ensure(block != NULL, "could not allocate 0x%lX bytes of memory for file %s", (unsigned long)bytes, path);

    for (size_t e = 0, total = K * rows * cols; e < total; ++e) {
        const size_t k = e / (rows * cols), i = (e / cols) % rows, j = e % cols; /* file order: matrix, row, column */
        const double v = strtod(cur, &end);
        ensure(end != cur, "could not read matrix from file %s, stuck at matrix %d element %d, %d", path, (int)k, (int)i,
               (int)j);
        cur = end;
        block[k * rows * cols + j * rows + i] = (DataType)v;
    }
    free(text);
    errno = 0;
    *numMatrices = (int)K;
    *m = (int)rows;
    *n = (int)cols;
    *matrices = block;
}

void replicateMatrices(Array *matrices, const int M, const int N, const int numMatrices, const int numReplications)
{
    const size_t list_bytes = sizeof(DataType) * (size_t)M * N * numMatrices;
    char *out = (char *)malloc(list_bytes * (size_t)numReplications + 1);
    ensure(out != NULL, "Could not allocate memory for the replicated array (%lu bytes).",
           (unsigned long)(list_bytes * numReplications));
    for (int r = 0; r < numReplications; ++r) memcpy(out + (size_t)r * list_bytes, *matrices, list_bytes);
    free(*matrices);
    *matrices = (Array)out;
}

void printMatrix(Array a, int M, int N)
{
    for (int i = 0; i < M; ++i) {
        for (int j = 0; j < N; ++j) printf("%f\t", (double)a[(size_t)j * M + i]);
        printf("\n");
    }
    printf("\n");
}

void printMatrixList(Array a, int N, int batchSize)
{
    for (int k = 0; k < batchSize; ++k) {
        printf("=============== <%d> ===============\n", k + 1);
        printMatrix(a + (size_t)k * N * N, N, N);
    }
}

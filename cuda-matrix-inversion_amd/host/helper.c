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
 * k*m*n, element (i, j) at j*m + i (column-major), exactly what the reference produces (helper.cu:38-48). */
void readMatricesFile(const char *path, int *numMatrices, int *m, int *n, Array *matrices)
{
    int count = 0, rows = 0, cols = 0;
    FILE *fp = fopen(path, "r");
    ensure(fp != NULL, "could not open matrix file %s", path);
    ensure(fscanf(fp, "%d %d %d", &count, &rows, &cols) == 3, "could not read number of matrices from file %s", path);
    ensure(count >= 0 && rows >= 0 && cols >= 0, "negative dimension in file %s", path);

    const size_t bytes = sizeof(DataType) * (size_t)count * (size_t)rows * (size_t)cols;
    ensure(bytes <= MAX_MATRIX_BYTE_READ,
           "cannot read file %s because the allocated array would be bigger than 0x%lX bytes", path,
           (unsigned long)bytes);
    Array block = (Array)malloc(bytes ? bytes : 1);
    ensure(block != NULL, "could not allocate 0x%lX bytes of memory for file %s", (unsigned long)bytes, path);

    for (int k = 0; k < count; ++k) {
        Array mat = block + (size_t)k * rows * cols;
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < cols; ++j) {
                double v;
                ensure(fscanf(fp, "%lf", &v) == 1, "could not read matrix from file %s, stuck at matrix %d element %d, %d",
                       path, k, i, j);
                mat[(size_t)j * rows + i] = (DataType)v;
            }
    }
    fclose(fp);
    *numMatrices = count;
    *m = rows;
    *n = cols;
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

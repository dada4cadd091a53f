/*
 * types.h -- scalar and batch-pointer typedefs of the drop-in boundary.
 *
 * Replaces /root/reference/include/types.h:4,6 (`#define DataType float`,
 * `typedef DataType *Array`). The reference is fp32 only; the primary artefact
 * here is fp64 (BASELINE.json configs 1-4), so DataType defaults to double.
 * Compile the consumer with -DMATINV_DATATYPE_FLOAT to get the reference's
 * fp32 ABI; inverse_gpu.h then binds the same 17 names to the *_f32 symbols.
 */
#ifndef HEADER_TYPES_INCLUDED
#define HEADER_TYPES_INCLUDED

#ifdef MATINV_DATATYPE_FLOAT
#define DataType float
#else
#define DataType double
#endif

typedef DataType *Array;

#endif

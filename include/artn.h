/*
 * artn.h -- C ABI of libartn_hip.so, the MI355X (gfx950) numerical contraction engine
 * behind artensor's executor entry points.
 *
 * The reference (Fanerst/artensor) has no FFI: its executor is Python calling
 * torch.einsum.  Each entry point below replaces one such call site; the reference
 * file:line it stands in for is cited on the declaration.  All pointers are raw device
 * pointers (hipMalloc'ed / torch `Tensor.data_ptr()`), sizes are in elements unless a
 * name says bytes, `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Every function returns 0 on success and a negative ARTN_E_* code on failure;
 * artn_last_error() then holds a message.  No function allocates or frees device
 * memory, synchronises the device or calls a CPU fallback: work is enqueued on `stream`
 * and the caller owns every buffer (graph-capture safe).
 */
#ifndef ARTN_H
#define ARTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARTN_ABI_VERSION 7
#define ARTN_MAX_LABELS 96

/* error codes */
#define ARTN_OK 0
#define ARTN_E_INVALID (-1)     /* malformed descriptor / argument            */
#define ARTN_E_UNSUPPORTED (-2) /* valid but not implemented (e.g. dtype)     */
#define ARTN_E_LAUNCH (-3)      /* HIP launch / runtime error                 */
#define ARTN_E_NODEVICE (-4)    /* no gfx950 device visible                   */

/* dtypes (arithmetic type of the contraction) */
#define ARTN_C64 0  /* interleaved (re,im) float32 pairs  -- torch.complex64  */
#define ARTN_C128 1 /* interleaved (re,im) float64 pairs  -- torch.complex128 */
#define ARTN_C64_BF16 2 /* complex64 in memory; operands of the big steps rounded to bfloat16 for the
                           matrix cores, fp32 accumulation: the reduced-precision sampling mode (the
                           reference has no such path; its results define no parity here) */

/*
 * One pairwise contraction step  C[out labels] = sum_{labels not in C} A[...] * B[...]
 * i.e. one `torch.einsum(eq, tensors[i], tensors[j])` of
 *   artensor/contraction.py:70 (dense executor) and
 *   artensor/contraction.py:147,156,163,169,179,181,190 (sparse executor).
 * The einsum string is replaced by label lists so unions of more than 50 labels
 * (the reference's alphabet, contraction.py:9-10) are representable.
 *
 * Label l (0 <= l < n_labels) has `extent[l]` and an element stride in each operand,
 * or -1 where the operand does not carry the label.  A label carried by A and B but
 * not C is contracted; by A (or B) and C only is free; by all three is a batch label
 * (the sparse path's shared label -3, contraction.py:306-308); by A or B alone is
 * summed out (einsum semantics).  C must be dense row-major over its labels in the
 * order implied by stride_c; A and B may be arbitrary non-overlapping strided views.
 */
typedef struct ArtnStepDesc {
  int32_t dtype;
  int32_t n_labels;
  int64_t extent[ARTN_MAX_LABELS];
  int64_t stride_a[ARTN_MAX_LABELS];
  int64_t stride_b[ARTN_MAX_LABELS];
  int64_t stride_c[ARTN_MAX_LABELS];
} ArtnStepDesc;

/* What the planner decided for a step (filled on the host, no GPU needed). */
#define ARTN_KERNEL_GENERIC 0 /* one thread per output element, strided loops       */
#define ARTN_KERNEL_BITS_MFMA 1 /* LDS-tiled bit-permuted complex GEMM on fp32 MFMA */
#define ARTN_KERNEL_GEMM_MFMA 2 /* two-operand LDS GEMM on fp32 MFMA, contracted bits looped in-kernel */
#define ARTN_KERNEL_PGEMM 4     /* big x big steps (complex64: 2^11+ contracted values, 160+ FLOP per byte; bf16 operands: 2^9+): both operands packed in
                                 * tile order into a workspace, then an LDS-DMA GEMM.  ARTN_C64: fp32 images, 3M arithmetic
                                 * (artn_k_pgemm3m; workspace 8 B x (2^(m+k) + 2^(n+k))); ARTN_C64_BF16: bfloat16 images
                                 * (artn_k_pgemm; half of that) */
#define ARTN_KERNEL_XGEMM 5     /* extent-based two-operand GEMM on fp32 MFMA (3M): any extents -- bond dimensions 3, 5, 6 ... --
                                 * flattened mixed-radix indices, offset tables in LDS (artn_k_xgemm) */
typedef struct ArtnStepInfo {
  int32_t kernel;       /* ARTN_KERNEL_*                                        */
  int32_t k_bits;       /* contracted bits handled inside a tile                */
  int32_t m_tile_bits;  /* free A bits inside a tile                            */
  int32_t n_tile_bits;  /* free B bits inside a tile                            */
  int32_t tile_in_bits; /* log2 elements of A staged in LDS per tile            */
  int32_t tile_out_bits;/* log2 elements of C staged in LDS per tile            */
  int32_t run_in_bits;  /* log2 contiguous elements per global read run         */
  int32_t run_out_bits; /* log2 contiguous elements per global write run        */
  int32_t lds_bytes;
  int32_t grid;
  int64_t n_tiles;
  int64_t a_rereads;    /* how many tiles read each A element (outer-N split)   */
  double flops;         /* 8 * prod(all extents) real FLOP (c64 MAC = 8)        */
  double bytes;         /* compulsory: 8|16 * (numel A + numel B + numel C)     */
  int32_t k2_bits;      /* fused pair: contracted bits of the second step (else 0) */
  int32_t n2_tile_bits; /* fused pair: free B2 bits inside a tile                  */
  int32_t tile_mid_bits;/* log2 elements of the tile between the two stages        */
  int32_t arith;        /* 0 fp32 MFMA 4M, 1 fp32 MFMA with 3M stages, 2 bf16 operands, 3 f64 MFMA, -1 no MFMA */
  double mfma_flops;    /* real FLOP the matrix pipe executes: `flops` with 6 instead of 8 per complex
                           multiply-add in every 3M stage (0 for the strided kernel)     */
  int64_t workspace_bytes; /* scratch the step wants from artn_contract_ws (0: none; artn_contract never needs any) */
  int32_t k3_bits;      /* fused triple (development builds only: artn_contract3): contracted bits of the third step (else 0) */
  int32_t stage1_reruns;/* fused pair: how often the first stage of a tile runs (the product of the outer result labels that only the
                           second step brings: each value repeats stage 1 and its reads); 0 or 1 otherwise */
} ArtnStepInfo;

int artn_abi_version(void);
const char *artn_last_error(void);

/* Number of visible gfx950 devices (0 on a CPU-only box; never fails). */
int artn_device_count(void);

/* Host-only: run the planner for `d` and report its decision. */
int artn_contract_query(const ArtnStepDesc *d, ArtnStepInfo *info);
/* Why the step of the last artn_contract_query on this thread was given to the strided
 * kernel ("" if it was not). */
const char *artn_last_plan_note(void);

/* Enqueue one pairwise contraction (replaces torch.einsum at contraction.py:70 etc.). */
int artn_contract(const ArtnStepDesc *d, const void *A, const void *B, void *C, void *stream);

/* The same with `ws_bytes` bytes of device scratch (16-byte aligned; ArtnStepInfo::workspace_bytes of
 * artn_contract_query says how much the step can use): steps that pack their operands first (ARTN_KERNEL_PGEMM:
 * the big x big contractions of plain complex64 AND of the reduced-precision mode) run that way when the scratch
 * is big enough, and exactly like artn_contract otherwise.  The library never allocates. */
int artn_contract_ws(const ArtnStepDesc *d, const void *A, const void *B, void *C, void *ws, int64_t ws_bytes,
                     void *stream);

/*
 * artn_contract with a fused row gather: along `label` (an output label; the batch label of the
 * sparse executor) operand A is read at row rows_a[r] and operand B at row rows_b[r] for output row
 * r (device pointers to int64; NULL = r itself); extent[label] is the number of output rows and
 * stride_a/b[label] the row strides of the SOURCE tensors, which have src_rows_a/b rows.
 * Replaces `tensors[i][batch_i[k]]`, `tensors[j][batch_j[k]]` followed by the batched einsum of
 * artensor/contraction.py:149-156 and :177-179 without materialising the gathered operands.
 * Out-of-range indices read row 0 and set *err_flag (if non-NULL).  Returns ARTN_E_UNSUPPORTED
 * when the step does not fit a tiled kernel; callers then gather with artn_gather_rows.
 * (complex64 only.  Steps with 7+ contracted bits run on the two-operand GEMM kernel, the others on the
 * state-streaming kernel; both resolve the rows inside their tile-offset computation.)
 */
int artn_contract_gather(const ArtnStepDesc *d, const void *A, const void *B, void *C, int label,
                         const int64_t *rows_a, int64_t src_rows_a, const int64_t *rows_b, int64_t src_rows_b,
                         int32_t *err_flag, void *stream);

/*
 * Two consecutive steps on the same big operand in ONE pass over HBM:
 *     C1 = contract(d1; A, B1);  C = contract(d2; C1, B2)
 * i.e. two successive iterations of the loop at artensor/contraction.py:66-70 whose first
 * operand is the same `tensors[i]` (the growing state tensor is always operand 0,
 * contraction.py:41-46).  d2's stride_a must describe d1's dense result C1, which is
 * never written to memory: the second contraction runs on the tile while it is in LDS.
 * artn_contract2_query / artn_contract2 return ARTN_E_UNSUPPORTED when the pair does not
 * fit one LDS tile (artn_last_error() says why); callers then issue two artn_contract.
 * complex64 (fp32 MFMA; bf16 operands under ARTN_C64_BF16) and complex128 (f64 MFMA, tiles of half as
 * many elements); both descriptors must name the same dtype.
 */
int artn_contract2_query(const ArtnStepDesc *d1, const ArtnStepDesc *d2, ArtnStepInfo *info);
int artn_contract2(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const void *A, const void *B1,
                   const void *B2, void *C, void *stream);

/*
 * C += contract(...) in the store phase of the last launch of a slice: the reference's slice loop ends every slice with
 * `collect_tensor += tensor_contraction(...)` (artensor/simulation.py:114); with a dense output the separate add reads the
 * slice's result and the accumulator and writes the accumulator again (3 x the output), the fused form reads and writes the
 * accumulator once and the result never exists in memory.  Every element of C belongs to exactly one lane of one tile
 * of the launch (plain loads and stores; C must not alias A), so C's value does not depend on scheduling.  complex64 state-streaming
 * plans only (artn_k_bits, single steps and fused pairs, register-prefetch tile loop); ARTN_E_UNSUPPORTED (-2) otherwise:
 * the caller then contracts into a temporary and calls artn_axpy_c64.
 */
int artn_contract_acc(const ArtnStepDesc *d, const void *A, const void *B, void *C, void *stream);
int artn_contract2_acc(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const void *A, const void *B1,
                       const void *B2, void *C, void *stream);

/* (ABI 7: artn_contract3 / artn_contract3_query -- three steps in one pass, round 4 -- left the product library: built,
 * parity-green, and shorter on no committed workload.  Development builds (make dev, -DARTN_DEV_BITS3)
 * still export them with the ABI-6 signatures, declared here for those builds only.) */
#ifdef ARTN_DEV_BITS3
int artn_contract3_query(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const ArtnStepDesc *d3, ArtnStepInfo *info);
int artn_contract3(const ArtnStepDesc *d1, const ArtnStepDesc *d2, const ArtnStepDesc *d3, const void *A, const void *B1,
                   const void *B2, const void *B3, void *C, void *stream);
#endif

/*
 * dst[r, :] = src[idx[r], :] for r < nrows, rows of `row_bytes` bytes (multiple of 8).
 * Replaces the batch-row gathers `tensors[i][batch_i[k]]` of
 * artensor/contraction.py:149-150,158-159,165-166,171-172,177-178,187.
 * `idx` is a DEVICE pointer to int64 row indices; src_rows bounds-checks them
 * (out-of-range rows are written as zeros and flagged in *err_flag if non-NULL).
 */
int artn_gather_rows(const void *src, const int64_t *idx, void *dst, int64_t nrows,
                     int64_t row_bytes, int64_t src_rows, int32_t *err_flag, void *stream);

/*
 * Small-step programs.  A circuit scheme is a few big steps plus hundreds of tiny ones (rank <= 16
 * operands) that are pure launch latency when issued one by one -- on the CPU the reference spends
 * 35-65 us per torch.einsum call on them (artensor/contraction.py:66-70; its whole n12 run is that).
 * Here the tiny steps of a scheme are compiled ONCE into a device-resident image and executed by ONE launch:
 * workgroup g runs the steps of group g (steps of different groups must be independent).  Inside a group the
 * steps are sorted into the levels of their dependency tree; the steps of a level run side by side on the 16 waves
 * of the workgroup, one barrier per level; intermediates stay in an LDS arena laid out at build time, only the
 * results flagged in `keep` (read after the program) and what the arena cannot hold go through the caller-provided
 * workspace.  n12 (68 steps, 19 levels) is one launch of a few tens of microseconds.
 *
 *   artn_program_image_bytes(...)  host only: size of the image for these steps (upper bound)
 *   artn_program_build(...)        host only: fills `host_image` from the step descriptors; operand k of step s is
 *                                  the result written at workspace byte offset loc >= 0 by an earlier step, or
 *                                  external pointer number -(loc + 1) when loc < 0; keep[s] != 0 (or keep == NULL):
 *                                  the result of step s must be in the workspace after the launch.  Steps must
 *                                  have dense operands and ONE element type (all complex64 -- small matrix-core
 *                                  steps included -- or all complex128: 16-byte elements, vector ALU only);
 *                                  returns ARTN_E_UNSUPPORTED when a step does not fit a record (the caller then
 *                                  issues artn_contract per step).
 *   artn_program_run(...)          enqueue: `dev_image` is the device copy of the image, `ext` a HOST array of
 *                                  n_ext (<= 256) device pointers, `dtype` the element type the image was built for
 *                                  (ARTN_C64 / ARTN_C64_BF16 / ARTN_C128).
 */
#define ARTN_PROGRAM_MAX_EXT 256
int64_t artn_program_record_bytes(void);
int64_t artn_program_image_bytes(int32_t n_steps, const ArtnStepDesc *const *descs, int32_t n_groups);
int artn_program_build(int32_t n_steps, const ArtnStepDesc *const *descs, const int64_t *loc_a, const int64_t *loc_b,
                       const int64_t *loc_c, const uint8_t *keep, int32_t n_groups, const int32_t *group_start,
                       void *host_image, int64_t image_bytes);
int artn_program_run(const void *dev_image, int32_t n_groups, const void *const *ext, int32_t n_ext, void *workspace,
                     int32_t dtype, void *stream);

/* acc[i] += x[i], i < n complex64 elements: the slice accumulation
 * `collect_tensor += ...` of artensor/simulation.py:114 and :210. */
int artn_axpy_c64(void *acc, const void *x, int64_t n, void *stream);
/* the same for complex128 (the reference's slice loop takes any dtype, artensor/simulation.py:90, :101) */
int artn_axpy_c128(void *acc, const void *x, int64_t n, void *stream);

/* out[g][c] = sum_r in[g][r][c] for complex64 arrays in[n_groups][n_rows][n_cols] (8-byte aligned; 16-byte
 * lanes when n_cols is even and the buffers are 16-byte aligned): sums out the leading label(s) of a dense tensor.  Closes a contraction whose
 * contracted labels exceed one LDS tile: `torch.einsum` at artensor/contraction.py:70 contracts any
 * number of labels in one call; here the slowest ones become a batch label of artn_contract and are
 * summed afterwards.  Applied twice (n_rows = R * n_rows') it is a two-pass tree sum. */
int artn_sum_axis_c64(const void *in, void *out, int64_t n_groups, int64_t n_rows, int64_t n_cols, void *stream);
/* the same for complex128 arrays (any n_cols; 16-byte aligned) */
int artn_sum_axis_c128(const void *in, void *out, int64_t n_groups, int64_t n_rows, int64_t n_cols, void *stream);

/* out[0] = max_i |x[i]| over n complex64 elements (float32, device pointer), then
 * x[i] /= out[0]: the running renormalisation of artensor/contraction.py:197-200
 * (`norm_factor = tensors[i].abs().max(); tensors[i] /= norm_factor`). */
int artn_absmax_normalize_c64(void *x, int64_t n, float *out_absmax, void *stream);
/* the same for complex128 elements (out_absmax: float64 device pointer): the reference renormalises in whatever
 * dtype `TensorNetworkSimulation.contraction(dtype=...)` selected (artensor/simulation.py:90, contraction.py:197-200) */
int artn_absmax_normalize_c128(void *x, int64_t n, double *out_absmax, void *stream);

/* Measurement aid (bench.py): the rate the matrix pipes of this device sustain on back-to-back MFMAs with operands
 * in registers, in TFLOP/s -- kind 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16, 2: v_mfma_f64_16x16x4_f64
 * (the arithmetic of complex128 steps; the roofline of a complex128 run is priced against this measured figure).
 * `scratch4`: 4 bytes of device memory.  Synchronous (runs on the null stream and waits). */
int artn_probe_mfma_rate(int kind, void *scratch4, double *tflops);

#ifdef __cplusplus
}
#endif
#endif /* ARTN_H */

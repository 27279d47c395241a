/* C ABI of the cached-plan dispatch of the step's LIBRARY GEMMs (libtgt_torch_ops.so, tgt_amd/csrc/gemm_dispatch.cpp).
 *
 * Not a kernel: the same hipBLASLt / rocBLAS call torch makes for
 *     nn.Linear forward            torch.addmm(b, x, W.t()) / torch.mm           reference lib/tgt/layers/layers.py:37-38,:155-160
 *     its data gradient            dY @ W                                        (autograd of the same lines; lib/tgt/layers/triplet.py:210-211)
 *     its weight gradient          torch.bmm over row chunks, float32 partials   (tgt_amd/ops.py::_linear_backward)
 * with torch's own handle and workspace and the algorithm / solution index of the TunableOp table -- but from a plan created once per
 * problem instead of per call (torch: signature string, look-up, three layouts + a matmul descriptor created and destroyed, a support
 * query: 20-37 us of host time; a plan: 11 us).  Results are bit-identical to torch's; tgt_amd/gemm.py verifies that on a plan's first use.
 * Lives in libtgt_torch_ops.so because it must share torch's copy of the BLAS libraries (the tuned indices are that build's).
 *
 * Column-major BLAS convention, TunableOp's parameters.  Element types: TGT_F32 = 0, TGT_BF16 = 1, TGT_F16 = 2 (tgt_hip.h). */
#ifndef TGT_GEMM_H
#define TGT_GEMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* backend 0: hipBLASLt, index = algorithm index (hipblaslt_ext::getAlgosFromIndex), < 0 = the heuristic's first choice
 * backend 1: rocBLAS,   index = solution index of rocblas_gemm_ex, < 0 = rocblas_gemm_algo_standard (no bias epilogue)
 * batch > 1: strided batched with the given element strides.  Returns a plan id >= 0, or -1 (tgt_gemm_last_error). */
int tgt_gemm_plan(int backend, int index, char transa, char transb, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
                  int64_t ldc, int batch, int64_t stride_a, int64_t stride_b, int64_t stride_c, int in_dtype, int out_dtype,
                  int has_bias);
/* c = alpha * op(a) op(b) + beta * c [+ bias] on `stream` (which must be torch's current stream of the device: the workspace is its). */
int tgt_gemm_run(int plan, const void* a, const void* b, void* c, const void* bias, float alpha, float beta, void* stream);
const char* tgt_gemm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif

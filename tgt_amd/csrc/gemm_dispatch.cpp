// Cached-descriptor dispatch of the step's LIBRARY GEMMs -- host code, part of libtgt_torch_ops.so.
//
// The node channel's Linears, the projection's data gradient and every weight gradient stay library GEMMs (hipBLASLt / rocBLAS
// through torch, with the TunableOp table of tgt_amd/tuning/: DESIGN.md section 4).  torch pays ~20-37 us of host time per call --
// TunableOp builds the problem's signature string, looks it up, creates three matrix layouts and a matmul descriptor, asks the
// library whether the tuned algorithm supports the problem and destroys everything again (ATen/cuda/tunable/GemmHipblaslt.h,
// GemmRocblas.h) -- 611 calls = 16 ms of the ~70 ms the host needs to queue a step.  Here the SAME library call is made with the
// SAME handle, workspace, algorithm / solution index and problem description, from a plan created once per shape:
//     tgt_gemm_plan(...)  -> plan id        (descriptor, layouts, algorithm resolved and checked once)
//     tgt_gemm_run(plan, a, b, c, bias, alpha, beta, stream)
// Column-major BLAS convention, exactly TunableOp's parameters (transa, transb, m, n, k, lda, ldb, ldc [, batch, strides]).
// Same kernel, same arguments: the result is bit-identical to torch's (tgt_amd/gemm.py checks that once per plan on first use and
// falls back to torch for a plan that differs).  torch's own handles and workspace are used (at::cuda::getCurrentCUDABlasLtHandle,
// getCUDABlasLtWorkspace, getCurrentCUDABlasHandle), so stream semantics are torch's.
#include <ATen/hip/HIPContextLight.h>
#include <c10/hip/HIPStream.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>
#include <rocblas/rocblas.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

struct Plan {
    int backend = 0;                  // 0 hipBLASLt, 1 rocBLAS
    int index = -1;                   // hipBLASLt algorithm index / rocBLAS solution index (< 0: the library's own choice)
    char ta = 'n', tb = 'n';
    int64_t m = 0, n = 0, k = 0, lda = 0, ldb = 0, ldc = 0, sa = 0, sb = 0, sc = 0;
    int batch = 1, in_dt = 0, out_dt = 0, has_bias = 0;
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
};
std::vector<Plan> g_plans;
std::mutex g_mu;

// element types: include/tgt_hip.h TGT_F32 = 0, TGT_BF16 = 1, TGT_F16 = 2
hipDataType lt_type(int dt) { return dt == 1 ? HIP_R_16BF : (dt == 2 ? HIP_R_16F : HIP_R_32F); }
rocblas_datatype rb_type(int dt) { return dt == 1 ? rocblas_datatype_bf16_r : (dt == 2 ? rocblas_datatype_f16_r : rocblas_datatype_f32_r); }
hipblasOperation_t lt_op(char c) { return (c == 't' || c == 'T') ? HIPBLAS_OP_T : HIPBLAS_OP_N; }
rocblas_operation rb_op(char c) { return (c == 't' || c == 'T') ? rocblas_operation_transpose : rocblas_operation_none; }

#define LT(call)                                                                             \
    do {                                                                                     \
        hipblasStatus_t s_ = (call);                                                         \
        if (s_ != HIPBLAS_STATUS_SUCCESS) return fail("%s -> hipblas status %d", #call, (int)s_); \
    } while (0)

int build_lt(Plan& p) {
    const hipblasOperation_t opa = lt_op(p.ta), opb = lt_op(p.tb);
    const hipDataType ti = lt_type(p.in_dt), to = lt_type(p.out_dt);
    // (the layouts of ATen/cuda/tunable/GemmHipblaslt.h: rows x cols as stored, column-major)
    LT(hipblasLtMatrixLayoutCreate(&p.la, ti, opa == HIPBLAS_OP_N ? p.m : p.k, opa == HIPBLAS_OP_N ? p.k : p.m, p.lda));
    LT(hipblasLtMatrixLayoutCreate(&p.lb, ti, opb == HIPBLAS_OP_N ? p.k : p.n, opb == HIPBLAS_OP_N ? p.n : p.k, p.ldb));
    LT(hipblasLtMatrixLayoutCreate(&p.lc, to, p.m, p.n, p.ldc));
    if (p.batch > 1) {
        int b = p.batch;
        for (auto pr : {std::make_pair(p.la, p.sa), std::make_pair(p.lb, p.sb), std::make_pair(p.lc, p.sc)}) {
            LT(hipblasLtMatrixLayoutSetAttribute(pr.first, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &b, sizeof b));
            int64_t st = pr.second;
            LT(hipblasLtMatrixLayoutSetAttribute(pr.first, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &st, sizeof st));
        }
    }
    LT(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof opa));
    LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof opb));
    if (p.has_bias) {
        const hipDataType tb = ti;
        const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
        LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &tb, sizeof tb));
        LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi));
    }
    hipblasLtHandle_t h = at::cuda::getCurrentCUDABlasLtHandle();
    const size_t ws = at::cuda::getCUDABlasLtWorkspaceSize();
    const float alpha = 1.f, beta = 0.f;
    if (p.index >= 0) {
        std::vector<int> idx{p.index};
        std::vector<hipblasLtMatmulHeuristicResult_t> res;
        LT(hipblaslt_ext::getAlgosFromIndex(h, idx, res));
        if (res.empty()) return fail("hipBLASLt has no algorithm with index %d", p.index);
        p.algo = res[0].algo;
        size_t need = 0;
        const void* dummy = &alpha;            // (a non-null bias pointer for the support query, as the tuned call has one)
        if (p.has_bias) LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof dummy));
        LT(hipblaslt_ext::matmulIsAlgoSupported(h, p.desc, &alpha, p.la, p.lb, &beta, p.lc, p.lc, p.algo, need));
        if (need >= ws) return fail("algorithm %d needs %zu bytes of workspace, torch provides %zu", p.index, need, ws);
    } else {
        hipblasLtMatmulPreference_t pref = nullptr;
        LT(hipblasLtMatmulPreferenceCreate(&pref));
        LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws));
        hipblasLtMatmulHeuristicResult_t r;
        int got = 0;
        const hipblasStatus_t s = hipblasLtMatmulAlgoGetHeuristic(h, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, &r, &got);
        hipblasLtMatmulPreferenceDestroy(pref);
        if (s != HIPBLAS_STATUS_SUCCESS || got < 1) return fail("hipBLASLt heuristic found no algorithm (status %d)", (int)s);
        p.algo = r.algo;
    }
    p.workspace = ws;
    return 0;
}

}  // namespace

extern "C" {

const char* tgt_gemm_last_error(void) { return g_err.c_str(); }

// backend 0: hipBLASLt (index = algorithm index of hipblaslt_ext::getAlgosFromIndex, < 0: the heuristic's first choice)
// backend 1: rocBLAS   (index = solution index of rocblas_gemm_ex, < 0: rocblas_gemm_algo_standard)
int tgt_gemm_plan(int backend, int index, char transa, char transb, int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb,
                  int64_t ldc, int batch, int64_t stride_a, int64_t stride_b, int64_t stride_c, int in_dtype, int out_dtype,
                  int has_bias) {
    Plan p;
    p.backend = backend; p.index = index; p.ta = transa; p.tb = transb;
    p.m = m; p.n = n; p.k = k; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.batch = batch < 1 ? 1 : batch; p.sa = stride_a; p.sb = stride_b; p.sc = stride_c;
    p.in_dt = in_dtype; p.out_dt = out_dtype; p.has_bias = has_bias;
    if (m <= 0 || n <= 0 || k <= 0) return fail("tgt_gemm_plan: empty problem");
    if (backend == 0) {
        if (build_lt(p) != 0) return -1;
    } else if (backend == 1) {
        if (has_bias) return fail("tgt_gemm_plan: the rocBLAS backend has no bias epilogue");
    } else {
        return fail("tgt_gemm_plan: backend %d", backend);
    }
    std::lock_guard<std::mutex> g(g_mu);
    g_plans.push_back(p);
    return (int)g_plans.size() - 1;
}

int tgt_gemm_run(int plan, const void* a, const void* b, void* c, const void* bias, float alpha, float beta, void* stream) {
    Plan p;
    {
        std::lock_guard<std::mutex> g(g_mu);
        if (plan < 0 || plan >= (int)g_plans.size()) return fail("tgt_gemm_run: no plan %d", plan);
        p = g_plans[plan];
    }
    if (p.backend == 0) {
        if (p.has_bias) {
            if (!bias) return fail("tgt_gemm_run: the plan has a bias epilogue");
            LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias));
        }
        hipblasLtHandle_t h = at::cuda::getCurrentCUDABlasLtHandle();
        void* ws = at::cuda::getCUDABlasLtWorkspace();
        LT(hipblasLtMatmul(h, p.desc, &alpha, a, p.la, b, p.lb, &beta, c, p.lc, c, p.lc, &p.algo, ws, p.workspace, (hipStream_t)stream));
        return 0;
    }
    rocblas_handle h = (rocblas_handle)at::cuda::getCurrentCUDABlasHandle();      // (torch sets the current stream on it)
    const rocblas_datatype ti = rb_type(p.in_dt), to = rb_type(p.out_dt);
    const rocblas_gemm_algo algo = p.index >= 0 ? rocblas_gemm_algo_solution_index : rocblas_gemm_algo_standard;
    rocblas_status s;
    if (p.batch > 1)
        s = rocblas_gemm_strided_batched_ex(h, rb_op(p.ta), rb_op(p.tb), (int)p.m, (int)p.n, (int)p.k, &alpha, a, ti, (int)p.lda, p.sa, b, ti,
                                            (int)p.ldb, p.sb, &beta, c, to, (int)p.ldc, p.sc, c, to, (int)p.ldc, p.sc, p.batch,
                                            rocblas_datatype_f32_r, algo, p.index >= 0 ? p.index : 0, rocblas_gemm_flags_none);
    else
        s = rocblas_gemm_ex(h, rb_op(p.ta), rb_op(p.tb), (int)p.m, (int)p.n, (int)p.k, &alpha, a, ti, (int)p.lda, b, ti, (int)p.ldb, &beta, c, to,
                            (int)p.ldc, c, to, (int)p.ldc, rocblas_datatype_f32_r, algo, p.index >= 0 ? p.index : 0, rocblas_gemm_flags_none);
    if (s != rocblas_status_success) return fail("rocblas_gemm_ex -> status %d", (int)s);
    return 0;
}

}  // extern "C"

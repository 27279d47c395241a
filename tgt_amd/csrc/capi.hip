// extern "C" entry points of libtgt_hip.so (include/tgt_hip.h) + error plumbing
// (the optimizer kernels live in optimizer.hip).
#include <cstdarg>
#include <cstdio>
#include "common.hpp"

namespace tgt {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(TGT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TGT_OK;
}

int triplet_attention_run(const tgt_triplet_attention_args* a, bool bwd, hipStream_t st);
int triplet_aggregate_run(const tgt_triplet_aggregate_args* a, bool bwd, hipStream_t st);
int node_attention_run(const tgt_node_attention_args* a, bool bwd, hipStream_t st);
int layer_norm_parts();
int add_layer_norm_fwd_run(const void* x, int x_dtype, const void* res, int res_dtype, const float* scale,
                           int64_t rows_per_sample, void* s_out, const float* gamma, const float* beta, void* y,
                           int y_dtype, float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t st);
int add_layer_norm_bwd_run(const void* dy, int dy_dtype, const void* s, int s_dtype, const void* ds_in, int ds_dtype,
                           const float* scale, int64_t rows_per_sample, const float* gamma, const float* mean,
                           const float* rstd, void* d_res, void* d_x, int d_dtype, float* dgamma, float* dbeta,
                           float* d_x_colsum, float* partial, int64_t rows, int C, hipStream_t st);
int gelu_dropout_run(const void* x, const void* dy, void* out, int64_t n, int dtype, float p, uint64_t seed, bool bwd,
                     const float* row_scale, int64_t elems_per_sample, hipStream_t st);
int triangular_update_run(const void* e4, const void* v4, const float* mask, void* out, const void* d_out, void* d_e4,
                          void* d_v4, int B, int N, int H, int dtype, bool bwd, hipStream_t st);
int colsum_run(const void* x, int x_dtype, int64_t rows, int C, float* out, float* partial, hipStream_t st);
int sum_rows_run(const float* x, int rows, int C, float* out, hipStream_t st);
int gelu_dropout_bwd_colsum_run(const void* x, const void* dy, void* out, int64_t n, int dtype, float p, uint64_t seed,
                                const float* row_scale, int64_t elems_per_sample, int cols, float* partial, float* colsum,
                                hipStream_t st);
int gelu_colsum_parts();
int triplet_attention_proj_supported(const tgt_triplet_attention_args* a, int C);
int triplet_attention_proj_run(const tgt_triplet_attention_args* a, const void* x, int C, const void* w, const void* bias,
                               hipStream_t st);
int fuse_rows_run(const tgt_fuse_rows_args* a, bool scatter, hipStream_t st);
int permute_cols_run(const void* src, int sd, const int32_t* idx, void* dst, int dd, int rows, int cols, hipStream_t st);
int sum_planes_run(const float* x, int planes, int64_t n, float* out, hipStream_t st);
int transpose_many_run(const void* items, int n, int blocks_per_item, hipStream_t st);
int sum_many_run(const void* items_host, int n, hipStream_t st);
int xent_run(const void* x, int dtype, const int64_t* target, const float* lse_in, const float* w, int64_t rows, int C,
             float* lse, float* xent, void* dx, hipStream_t st);
int edge_linear_supported(const tgt_edge_linear_args* a);
int edge_linear_parts(int64_t M, int N);
int edge_linear_run(const tgt_edge_linear_args* a, hipStream_t st);
void edge_linear_set_grid_cap(int cap);
int layer_norm_fwd_run(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                       float* mean, float* rstd, int64_t rows, int C, float eps, hipStream_t st);
int layer_norm_bwd_run(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                       const float* rstd, void* dx, int dx_dtype, float* dgamma, float* dbeta, float* partial,
                       int64_t rows, int C, hipStream_t st);
int gaussian_parts(int64_t pairs);
int gaussian_run(const float* x, const float* mul, const float* bias, const float* mean, const float* std_p, int64_t pairs, int K,
                 int dtype, void* y, const void* g, float* dt, float* partial, hipStream_t st);
int dist_bins_run(const void* x, int dtype, int64_t B, int N, int NB, void* bins, int esz, int64_t ld_b, int S, int* state,
                  hipStream_t st);
int sample_commit_run(int* state, int S, hipStream_t st);
int softmax_accumulate_run(const void* x, int dtype, int64_t rows, int NB, float* acc, int* state, int S, hipStream_t st);
int probs_finish_run(const float* acc, int64_t B, int N, int NB, const int* state, int as_log, float eps, float* out, hipStream_t st);
int gap_commit_run(const void* gap, int dtype, int B, float* out, int S, int* state, hipStream_t st);
int pack_triu_run(const void* bins, int esz, int B, int S, int N, const int64_t* num_nodes, const int64_t* offsets, void* flat,
                  int64_t total, hipStream_t st);
int bins_to_dist_run(const void* bins, int kind, int64_t R, int N, const int64_t* num_nodes, int S, float bin_size, int shift_half,
                     int zero_diag, float* out, hipStream_t st);

}  // namespace tgt

namespace tgt {
static const uint64_t* g_seed_counter = nullptr;
const uint64_t* seed_counter() { return g_seed_counter; }
}  // namespace tgt

using namespace tgt;

extern "C" {

const char* tgt_last_error(void) { return g_err; }
int tgt_abi_version(void) { return 30; }
int tgt_set_seed_counter(const void* device_counter) {
    g_seed_counter = reinterpret_cast<const uint64_t*>(device_counter);
    return TGT_OK;
}

int tgt_triplet_attention_fwd(const tgt_triplet_attention_args* a, void* stream) {
    return triplet_attention_run(a, false, reinterpret_cast<hipStream_t>(stream));
}
int tgt_triplet_attention_bwd(const tgt_triplet_attention_args* a, void* stream) {
    return triplet_attention_run(a, true, reinterpret_cast<hipStream_t>(stream));
}
int tgt_triplet_aggregate_fwd(const tgt_triplet_aggregate_args* a, void* stream) {
    return triplet_aggregate_run(a, false, reinterpret_cast<hipStream_t>(stream));
}
int tgt_triplet_aggregate_bwd(const tgt_triplet_aggregate_args* a, void* stream) {
    return triplet_aggregate_run(a, true, reinterpret_cast<hipStream_t>(stream));
}
int tgt_node_attention_fwd(const tgt_node_attention_args* a, void* stream) {
    return node_attention_run(a, false, reinterpret_cast<hipStream_t>(stream));
}
int tgt_node_attention_bwd(const tgt_node_attention_args* a, void* stream) {
    return node_attention_run(a, true, reinterpret_cast<hipStream_t>(stream));
}

int tgt_triangular_update_fwd(const void* e4, const void* v4, const float* mask, void* out, int32_t B, int32_t N,
                              int32_t H, int32_t dtype, void* stream) {
    return triangular_update_run(e4, v4, mask, out, nullptr, nullptr, nullptr, B, N, H, dtype, false,
                                 reinterpret_cast<hipStream_t>(stream));
}
int tgt_triangular_update_bwd(const void* e4, const void* v4, const float* mask, const void* d_out, void* d_e4,
                              void* d_v4, int32_t B, int32_t N, int32_t H, int32_t dtype, void* stream) {
    return triangular_update_run(e4, v4, mask, nullptr, d_out, d_e4, d_v4, B, N, H, dtype, true,
                                 reinterpret_cast<hipStream_t>(stream));
}
int tgt_gelu_dropout_fwd(const void* x, void* y, int64_t n, int32_t dtype, float p, uint64_t seed, void* stream) {
    return gelu_dropout_run(x, nullptr, y, n, dtype, p, seed, false, nullptr, 0, reinterpret_cast<hipStream_t>(stream));
}
int tgt_gelu_dropout_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, float p, uint64_t seed,
                         void* stream) {
    return gelu_dropout_run(x, dy, dx, n, dtype, p, seed, true, nullptr, 0, reinterpret_cast<hipStream_t>(stream));
}
int tgt_gelu_colsum_parts(void) { return gelu_colsum_parts(); }
int tgt_gelu_dropout_bwd_colsum(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, float p, uint64_t seed,
                                const float* sample_scale, int64_t elems_per_sample, int32_t cols, float* partial, float* colsum,
                                void* stream) {
    return gelu_dropout_bwd_colsum_run(x, dy, dx, n, dtype, p, seed, sample_scale, elems_per_sample, cols, partial, colsum,
                                       reinterpret_cast<hipStream_t>(stream));
}
int tgt_gelu_dropout_scaled_fwd(const void* x, void* y, int64_t n, int32_t dtype, float p, uint64_t seed, const float* sample_scale,
                                int64_t elems_per_sample, void* stream) {
    return gelu_dropout_run(x, nullptr, y, n, dtype, p, seed, false, sample_scale, elems_per_sample, reinterpret_cast<hipStream_t>(stream));
}
int tgt_gelu_dropout_scaled_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, float p, uint64_t seed,
                                const float* sample_scale, int64_t elems_per_sample, void* stream) {
    return gelu_dropout_run(x, dy, dx, n, dtype, p, seed, true, sample_scale, elems_per_sample, reinterpret_cast<hipStream_t>(stream));
}
int tgt_add_layer_norm_fwd(const void* x, int32_t x_dtype, const void* res, int32_t res_dtype, const float* scale,
                           int64_t rows_per_sample, void* s_out, const float* gamma, const float* beta, void* y,
                           int32_t y_dtype, float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream) {
    return add_layer_norm_fwd_run(x, x_dtype, res, res_dtype, scale, rows_per_sample, s_out, gamma, beta, y, y_dtype, mean,
                                  rstd, rows, C, eps, reinterpret_cast<hipStream_t>(stream));
}
int tgt_add_layer_norm_bwd(const void* dy, int32_t dy_dtype, const void* s, int32_t s_dtype, const void* ds_in,
                           int32_t ds_dtype, const float* scale, int64_t rows_per_sample, const float* gamma,
                           const float* mean, const float* rstd, void* d_res, void* d_x, int32_t d_dtype, float* dgamma,
                           float* dbeta, float* d_x_colsum, float* partial, int64_t rows, int32_t C, void* stream) {
    return add_layer_norm_bwd_run(dy, dy_dtype, s, s_dtype, ds_in, ds_dtype, scale, rows_per_sample, gamma, mean, rstd,
                                  d_res, d_x, d_dtype, dgamma, dbeta, d_x_colsum, partial, rows, C,
                                  reinterpret_cast<hipStream_t>(stream));
}
int tgt_layer_norm_parts(void) { return layer_norm_parts(); }
int tgt_triplet_attention_proj_supported(const tgt_triplet_attention_args* a, int32_t C) { return triplet_attention_proj_supported(a, C); }
int tgt_triplet_attention_proj_fwd(const tgt_triplet_attention_args* a, const void* x, int32_t C, const void* w, const void* bias,
                                   void* stream) {
    return triplet_attention_proj_run(a, x, C, w, bias, reinterpret_cast<hipStream_t>(stream));
}
int tgt_fuse_rows(const tgt_fuse_rows_args* a, void* stream) { return fuse_rows_run(a, false, reinterpret_cast<hipStream_t>(stream)); }
int tgt_unfuse_rows(const tgt_fuse_rows_args* a, void* stream) { return fuse_rows_run(a, true, reinterpret_cast<hipStream_t>(stream)); }
int tgt_permute_cols(const void* src, int32_t src_dtype, const int32_t* idx, void* dst, int32_t dst_dtype, int32_t rows,
                     int32_t cols, void* stream) {
    return permute_cols_run(src, src_dtype, idx, dst, dst_dtype, rows, cols, reinterpret_cast<hipStream_t>(stream));
}
int tgt_cross_entropy_fwd(const void* logits, int32_t dtype, const int64_t* target, int64_t rows, int32_t C, float* lse,
                          float* xent, void* stream) {
    return xent_run(logits, dtype, target, nullptr, nullptr, rows, C, lse, xent, nullptr, reinterpret_cast<hipStream_t>(stream));
}
int tgt_cross_entropy_bwd(const void* logits, int32_t dtype, const int64_t* target, const float* lse, const float* row_weight,
                          int64_t rows, int32_t C, void* d_logits, void* stream) {
    if (!d_logits) return set_error(TGT_ERR_INVALID, "cross entropy bwd: null d_logits");
    return xent_run(logits, dtype, target, lse, row_weight, rows, C, nullptr, nullptr, d_logits, reinterpret_cast<hipStream_t>(stream));
}
int tgt_sum_planes(const float* x, int32_t planes, int64_t n, float* out, void* stream) {
    return sum_planes_run(x, planes, n, out, reinterpret_cast<hipStream_t>(stream));
}
int tgt_sum_many(const tgt_sum_item* items, int32_t n, void* stream) {
    return sum_many_run(items, n, reinterpret_cast<hipStream_t>(stream));
}
int tgt_transpose_many(const tgt_transpose_item* items, int32_t n, int32_t blocks_per_item, void* stream) {
    return transpose_many_run(items, n, blocks_per_item, reinterpret_cast<hipStream_t>(stream));
}
int tgt_sum_rows(const float* x, int32_t rows, int32_t C, float* out, void* stream) {
    return sum_rows_run(x, rows, C, out, reinterpret_cast<hipStream_t>(stream));
}
int tgt_colsum(const void* x, int32_t x_dtype, int64_t rows, int32_t C, float* out, float* partial, void* stream) {
    return colsum_run(x, x_dtype, rows, C, out, partial, reinterpret_cast<hipStream_t>(stream));
}
int tgt_layer_norm_fwd(const void* x, int32_t x_dtype, const float* gamma, const float* beta, void* y, int32_t y_dtype,
                       float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream) {
    return layer_norm_fwd_run(x, x_dtype, gamma, beta, y, y_dtype, mean, rstd, rows, C, eps,
                              reinterpret_cast<hipStream_t>(stream));
}
int tgt_layer_norm_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype, const float* gamma,
                       const float* mean, const float* rstd, void* dx, int32_t dx_dtype, float* dgamma, float* dbeta,
                       float* partial, int64_t rows, int32_t C, void* stream) {
    return layer_norm_bwd_run(dy, dy_dtype, x, x_dtype, gamma, mean, rstd, dx, dx_dtype, dgamma, dbeta, partial, rows, C,
                              reinterpret_cast<hipStream_t>(stream));
}

int tgt_dist_bins_argmax(const void* logits, int32_t dtype, int64_t B, int32_t N, int32_t NB, void* bins, int32_t bins_elem_size,
                         int64_t ld_b, int32_t S, int32_t* state, void* stream) {
    return dist_bins_run(logits, dtype, B, N, NB, bins, bins_elem_size, ld_b, S, state, reinterpret_cast<hipStream_t>(stream));
}
int tgt_sample_commit(int32_t* state, int32_t S, void* stream) { return sample_commit_run(state, S, reinterpret_cast<hipStream_t>(stream)); }
int tgt_softmax_accumulate(const void* logits, int32_t dtype, int64_t rows, int32_t NB, float* acc, int32_t* state, int32_t S,
                           void* stream) {
    return softmax_accumulate_run(logits, dtype, rows, NB, acc, state, S, reinterpret_cast<hipStream_t>(stream));
}
int tgt_probs_finish(const float* acc, int64_t B, int32_t N, int32_t NB, const int32_t* state, int32_t as_log, float eps, float* out,
                     void* stream) {
    return probs_finish_run(acc, B, N, NB, state, as_log, eps, out, reinterpret_cast<hipStream_t>(stream));
}
int tgt_gap_commit(const void* gap, int32_t dtype, int32_t B, float* out, int32_t S, int32_t* state, void* stream) {
    return gap_commit_run(gap, dtype, B, out, S, state, reinterpret_cast<hipStream_t>(stream));
}
int tgt_pack_triu(const void* bins, int32_t elem_size, int32_t B, int32_t S, int32_t N, const int64_t* num_nodes,
                  const int64_t* offsets, void* flat, int64_t total, void* stream) {
    return pack_triu_run(bins, elem_size, B, S, N, num_nodes, offsets, flat, total, reinterpret_cast<hipStream_t>(stream));
}
int tgt_bins_to_dist(const void* bins, int32_t kind, int64_t R, int32_t N, const int64_t* num_nodes, int32_t S, float bin_size,
                     int32_t shift_half, int32_t zero_diag, float* out, void* stream) {
    return bins_to_dist_run(bins, kind, R, N, num_nodes, S, bin_size, shift_half, zero_diag, out, reinterpret_cast<hipStream_t>(stream));
}

int tgt_gaussian_basis_parts(int64_t pairs) { return gaussian_parts(pairs); }
int tgt_gaussian_basis_fwd(const float* x, const float* mul, const float* bias, const float* mean, const float* std, int64_t pairs,
                           int32_t K, int32_t dtype, void* y, void* stream) {
    return gaussian_run(x, mul, bias, mean, std, pairs, K, dtype, y, nullptr, nullptr, nullptr, reinterpret_cast<hipStream_t>(stream));
}
int tgt_gaussian_basis_bwd(const float* x, const float* mul, const float* bias, const float* mean, const float* std, int64_t pairs,
                           int32_t K, int32_t dtype, const void* g, float* dt, float* partial, void* stream) {
    if (!g) return set_error(TGT_ERR_INVALID, "gaussian basis bwd: null gradient");
    return gaussian_run(x, mul, bias, mean, std, pairs, K, dtype, nullptr, g, dt, partial, reinterpret_cast<hipStream_t>(stream));
}

int tgt_edge_linear_supported(const tgt_edge_linear_args* a) { return edge_linear_supported(a); }
int tgt_edge_linear_parts(int64_t M, int32_t N) { return edge_linear_parts(M, N); }
int tgt_edge_linear(const tgt_edge_linear_args* a, void* stream) { return edge_linear_run(a, reinterpret_cast<hipStream_t>(stream)); }
void tgt_edge_linear_set_grid_cap(int32_t cap) { edge_linear_set_grid_cap(cap); }

}  // extern "C"

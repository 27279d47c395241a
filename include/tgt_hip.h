/*
 * tgt_hip.h -- C ABI of libtgt_hip.so: the MI355X (gfx950) kernels underneath
 * the TGT layer modules.
 *
 * The reference (shamim-hussain/tgt) has no FFI: its seam is Python module
 * substitution (lib/tgt/layers/layers.py:219-251 instantiates
 * `EGT_Attention` / `get_triplet_layer(...)`), and all arithmetic is ATen
 * calls inside those modules' forward().  Each entry point below replaces the
 * ATen call sequence of one such forward (cited per function); the Python
 * mirror in tgt_amd/tgt/ binds them with ctypes.
 *
 * Conventions
 *  - plain pointers to DEVICE memory + sizes; no framework types.
 *  - tensors are borrowed for the call; nothing is retained or allocated.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - every function returns 0 on success, non-zero on error
 *    (TGT_ERR_*); tgt_last_error() gives a thread-local message.
 *  - kernels are stateless and re-entrant; launches are asynchronous.
 *  - element type is chosen by `dtype` (TGT_F32 / TGT_BF16 / TGT_F16); masks,
 *    softmax statistics and reductions are always float32.
 *  - channel layouts: the triplet ops take HEAD-MAJOR channels (c = h*D + d);
 *    the node-attention ops take the reference's HEAD-MINOR channels
 *    (c = d*H + h).  The Python mirror permutes the projection weights, not
 *    the activations, to get head-major triplet operands.
 */
#ifndef TGT_HIP_H
#define TGT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TGT_F32 = 0, TGT_BF16 = 1, TGT_F16 = 2 };
enum { TGT_OK = 0, TGT_ERR_INVALID = 1, TGT_ERR_UNSUPPORTED = 2, TGT_ERR_LAUNCH = 3 };

/* flags for the triplet ops */
enum {
    TGT_TRI_BIASED      = 1,   /* third-arm bias E present      */
    TGT_TRI_GATED       = 2,   /* third-arm sigmoid gate present */
    TGT_TRI_MASK_OUT    = 4,   /* aggregate only: mask the outward direction (ungated variant) */
};

const char* tgt_last_error(void);
/* ABI version of this header; bump on any signature change. */
int tgt_abi_version(void);

/* ------------------------------------------------------------------------
 * Triplet attention core (TripletAttention / TripletAttentionUngated /
 * AxialAttention).  Replaces reference lib/tgt/layers/triplet.py:213-246
 * (the two einsum+softmax+gate+einsum chains) and its autograd backward.
 *
 * For dir in {0 = inward, 1 = outward}, per graph b, head h, edge (i,j):
 *   inward : S[k] = s * Q[i,j,h,:]·K[j,k,h,:] + E[i,k,h] + M[i,k]
 *            O[i,j,h,:] = sum_k softmax_k(S)[k] * sigmoid(G[i,k,h]+M[i,k]) * V[j,k,h,:]
 *   outward: S[k] = s * Q[i,j,h,:]·K[k,j,h,:] + E[k,i,h] + M[k,i]
 *            O[i,j,h,:] = sum_k softmax_k(S)[k] * sigmoid(G[k,i,h]+M[k,i]) * V[k,j,h,:]
 *
 * qkv[dir] : (B,N,N,ld_qkv[dir]) rows; Q,K,V of head h start at element
 *            q_off/k_off/v_off[dir] + h*D (head-major, D contiguous).
 * eg[dir]  : (B,N,N,ld_eg[dir]) rows; E[.,.,h] at e_off[dir]+h, G at g_off[dir]+h
 *            (ignored when the BIASED/GATED flags are clear).
 * mask     : (B,N,N) float32 additive mask (0 / finfo.min).
 * out      : (B,N,N,ld_out) rows; O of (dir,h) at o_off[dir] + h*D.
 * Nothing is saved for backward: softmax statistics are recomputed.
 * Backward adds: d_out (same layout as out), and writes d_qkv[dir]
 * (same layout as qkv[dir]: every Q,K,V element is written), d_eg[dir]
 * (E and G columns written; same layout as eg[dir]).
 * Supported: N <= 64, D in {8,16,32}, any H.
 * ---------------------------------------------------------------------- */
typedef struct tgt_triplet_attention_args {
    int32_t B, N, H, D;
    int32_t dtype, flags;
    float   scale;                    /* D^-0.5 */
    int32_t _pad0;
    const void* qkv[2];  int64_t ld_qkv[2];  int32_t q_off[2], k_off[2], v_off[2];
    const void* eg[2];   int64_t ld_eg[2];   int32_t e_off[2], g_off[2];
    const float* mask;
    void*   out;         int64_t ld_out;     int32_t o_off[2];
    /* backward only */
    const void* d_out;
    void*   d_qkv[2];
    void*   d_eg[2];
    /* backward, optional (both NULL or both set per direction): per-graph column sums of the
     * gradient rows, i.e. the bias gradients of the projections that produced qkv / eg
     * (`grad.sum(0)` of nn.Linear) before the sum over graphs.  d_qkv_colsum[dir] is (B, ld_qkv[dir])
     * float32, d_eg_colsum[dir] (B, ld_eg[dir]); only the Q/K/V/E/G columns of `dir` are written
     * (each exactly once), finish with tgt_sum_rows over B.  Saves a full pass over d_qkv. */
    float*  d_qkv_colsum[2];
    float*  d_eg_colsum[2];
    /* attention dropout on the gated weights (reference triplet.py:223-225, :242-244); 0 = off.
     * Counter-based: the backward must be called with the forward's (p, seed).  Generator:
     * csrc/triplet_common.hpp (tri_drop_bits), unit = ((b*2 + dir)*H + h)*N + j. */
    float    dropout_p;
    uint32_t _pad1;
    uint64_t dropout_seed;
    /* backward, optional: row length of the gradient buffers when it differs from the sources'
     * (0 = ld_qkv / ld_eg).  d_qkv / d_eg then use the same channel offsets inside rows of this
     * length, and d_qkv_colsum / d_eg_colsum are (B, ld_dqkv) / (B, ld_deg).  Lets the forward keep
     * Q/K/V (1536 channels: six full 256-wide GEMM tile columns) and E/G in two tensors while the
     * backward writes ONE fused gradient row for a single data/weight-gradient GEMM. */
    int64_t  ld_dqkv[2], ld_deg[2];
    /* optional (ABI 23): per-graph DropPath factor (B) float32 of the residual branch this call belongs to (reference
     * lib/tgt/layers/layers.py:169-174: Bernoulli(keep)/keep per graph, multiplied onto the branch output at the residual
     * add).  A graph whose factor is exactly 0 contributes nothing to the stream and receives an all-zero d_out, so the
     * kernels do not read or compute it: the forward writes zeros to its `out` rows, the backward zeros to its d_qkv / d_eg
     * rows and column sums -- equal (up to the sign of a zero) to the full computation followed by the multiplication.  The factor itself
     * is NOT applied here.  NULL = every graph is computed. */
    const float* graph_scale;
} tgt_triplet_attention_args;

int tgt_triplet_attention_fwd(const tgt_triplet_attention_args* a, void* stream);
int tgt_triplet_attention_bwd(const tgt_triplet_attention_args* a, void* stream);

/* ------------------------------------------------------------------------
 * Triplet aggregate core (TripletAggregate / TripletAggregateUngated).
 * Replaces reference lib/tgt/layers/triplet.py:56-70 (gated; outward
 * direction unmasked) and :107-123 (ungated; TGT_TRI_MASK_OUT).
 *   inward : A[i,k,h] = softmax_k(E_in[i,k,h]+M[i,k]) * sigmoid(G_in[i,k,h]+M[i,k])
 *            O[i,j,h,:] = sum_k A[i,k,h] V_in[j,k,h,:]
 *   outward: A[k,i,h] = softmax_k(E_out[k,i,h] (+M[k,i])) * sigmoid(G_out[k,i,h] (+M[k,i]))
 *            O[i,j,h,:] = sum_k A[k,i,h] V_out[k,j,h,:]
 * v[dir] : (B,N,N,ld_v[dir]) rows, V of head h at v_off[dir] + h*D.
 * eg / mask / out as above.  No saved statistics (weights are recomputed).
 * ---------------------------------------------------------------------- */
typedef struct tgt_triplet_aggregate_args {
    int32_t B, N, H, D;
    int32_t dtype, flags;
    const void* v[2];    int64_t ld_v[2];    int32_t v_off[2];
    const void* eg[2];   int64_t ld_eg[2];   int32_t e_off[2], g_off[2];
    const float* mask;
    void*   out;         int64_t ld_out;     int32_t o_off[2];
    /* backward only */
    const void* d_out;
    void*   d_v[2];
    void*   d_eg[2];
    /* attention dropout on the gated weights (reference triplet.py:59-60, :66-67); 0 = off;
     * unit = (b*2 + dir)*H + h, otherwise as tgt_triplet_attention_args. */
    float    dropout_p;
    uint32_t _pad1;
    uint64_t dropout_seed;
} tgt_triplet_aggregate_args;

int tgt_triplet_aggregate_fwd(const tgt_triplet_aggregate_args* a, void* stream);
int tgt_triplet_aggregate_bwd(const tgt_triplet_aggregate_args* a, void* stream);

/* ------------------------------------------------------------------------
 * TriangularUpdate core (reference lib/tgt/layers/triplet.py:156-172: the four
 * "siglin" gates and the two einsums 'bikh,bjkh->bijh' / 'bkih,bkjh->bijh').
 * e4, v4: (B,N,N,4H) = [in_gate | in_lin | out_gate | out_lin] (lin_E / lin_V outputs);
 * mask (B,N,N) float32; out (B,N,N,2H) = [O_in | O_out].  Backward writes d_e4, d_v4 fully.
 * ---------------------------------------------------------------------- */
int tgt_triangular_update_fwd(const void* e4, const void* v4, const float* mask, void* out,
                              int32_t B, int32_t N, int32_t H, int32_t dtype, void* stream);
int tgt_triangular_update_bwd(const void* e4, const void* v4, const float* mask, const void* d_out,
                              void* d_e4, void* d_v4, int32_t B, int32_t N, int32_t H, int32_t dtype,
                              void* stream);

/* ------------------------------------------------------------------------
 * Node attention with edge bias + gate (EGT_Attention) and the logits-only
 * EdgeUpdate.  Replaces reference lib/tgt/layers/layers.py:62-77 (and
 * :120-124 for EdgeUpdate) and its autograd backward.
 *   H_hat[l,m,h] = s * sum_d Q[l,d,h] K[m,d,h] + E[l,m,h]          (written, T)
 *   A[l,m,h]     = softmax_m(H_hat + M[l,m]) * sigmoid(G[l,m,h] + M[l,m])
 *   V_att[l,d,h] = (sum_m A[l,m,h] V[m,d,h]) * (scale_degree ? log(1+sum_m gate) : 1)
 * qkv  : (B,N,ld_qkv) rows, HEAD-MINOR: Q[l,d,h] at q_off + d*H + h (k_off, v_off alike)
 * eg   : (B,N,N,ld_eg) rows; E at e_off+h, G at g_off+h
 * mask : (B,N,N) float32 (already includes the source-dropout mask, if any)
 * vatt : (B,N,W=D*H) head-minor;  hhat: (B,N,N,H) (may be NULL: no edge update)
 * lse  : (B,N,H) float32;  gsum: (B,N,H) float32 (sum_m gate)   [saved for bwd]
 * logits_only != 0: EdgeUpdate -- only hhat is produced (V, G, mask, vatt unused).
 * Backward: the forward's vatt/lse/gsum, d_vatt (B,N,W), d_hhat (B,N,N,H, may be NULL) -> d_qkv (B,N,ld_qkv;
 * Q,K,V columns written), d_eg (B,N,N,ld_eg; E,G columns written).
 * Supported: any N, D <= 32, any H.  Kernel families behind the two entry points (one launch each way, chosen by shape; same results
 * up to the order of the softmax sums): 16-bit with D in {8,12,16} -- forward with H % 32 == 0: key-blocked, 64-byte pieces of the
 * E | G rows (csrc/node_attention_kb.hip); N <= 32, H % 8 == 0: one 32x32 tile per head (csrc/node_attention_mfma.hip);
 * 33 <= N <= 64, H % 8 == 0: 16-wide tiles (csrc/node_attention16.hip); everything else (fp32, other D / H, N > 64 backward):
 * one lane per head (csrc/node_attention.hip).
 * ---------------------------------------------------------------------- */
typedef struct tgt_node_attention_args {
    int32_t B, N, H, D;
    int32_t dtype, scale_degree, logits_only;
    int32_t _pad0;                    /* (ABI <= 25: head_major, a head-major channel order measured slower and removed) */
    float   scale;                    /* D^-0.5 */
    int32_t _pad1;
    const void* qkv;   int64_t ld_qkv;  int32_t q_off, k_off, v_off, _pad2;
    const void* eg;    int64_t ld_eg;   int32_t e_off, g_off;
    const float* mask;
    void*  vatt;
    void*  hhat;
    float* lse;
    float* gsum;
    /* backward only */
    const void* d_vatt;
    const void* d_hhat;
    void*  d_qkv;
    void*  d_eg;
    void*  _reserved0;                /* (ABI <= 25: w_ws, a pair-weight scratch between the two backward passes: neutral, removed) */
    const float* hhat_scale;          /* optional (B) float32: H_hat is WRITTEN as hhat_scale[b] * H_hat (the softmax still sees the
                                       * unscaled logits) and the backward reads d_hhat as the gradient of that scaled tensor.  The
                                       * DropPath factors of the edge branch lin_O_e(H_hat) feeds (reference layers.py:270-272),
                                       * folded in here: see TGT_EDGE_BIAS_SCALED */
} tgt_node_attention_args;

int tgt_node_attention_fwd(const tgt_node_attention_args* a, void* stream);
int tgt_node_attention_bwd(const tgt_node_attention_args* a, void* stream);

/* ------------------------------------------------------------------------
 * Flat-buffer Adam step (replaces apex.optimizers.FusedAdam, reference
 * lib/training/training.py:159-171; adam_w_mode with weight_decay=0 ==
 * torch.optim.Adam).  All buffers float32 of length n.
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
 *   p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)   [+ lr*wd*p decoupled]
 * grad_scale multiplies g first (1/world_size or AMP unscale); clip_value > 0
 * clamps the scaled gradient to [-clip_value, clip_value] (nn.utils.clip_grad_value_,
 * reference training.py:455-460).
 * shadow (may be NULL): n-element bf16/f16 buffer that receives the updated
 * parameters in the same pass (the GEMM-dtype copy the next step reads).
 * ctl (may be NULL): device control block written by tgt_grad_scaler_step; when given,
 * `step` / `grad_scale` are ignored: the update is skipped if ctl[TGT_CTL_FOUND_INF] != 0,
 * g is multiplied by ctl[TGT_CTL_MULT] (clamped) then by ctl[TGT_CTL_COEF], and the bias
 * corrections use t = ctl[TGT_CTL_STEPS] (optimizer steps actually applied).
 * ---------------------------------------------------------------------- */
int tgt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, float clip_value,
                  const float* ctl, void* shadow, int32_t shadow_dtype, void* stream);

/* ------------------------------------------------------------------------
 * Device-side GradScaler + gradient clipping decisions (no host sync).  Replaces the
 * host logic of reference lib/training/training.py:451-469: GradScaler.unscale_ /
 * found_inf (a `.item()` per step in torch), nn.utils.clip_grad_value_, clip_grad_norm_,
 * GradScaler.step's skip and GradScaler.update's backoff / growth.
 * ctl: 16 float32 on the device:
 *   [0] loss scale S (in/out; the loss is multiplied by it before backward)
 *   [1] growth tracker (in/out)        [2] found_inf of this step (out)
 *   [3] optimizer steps applied (in/out: +1 unless skipped)
 *   [4] gradient multiplier 1/(S*world) of THIS step (out)   [5] clip_grad_norm coefficient (out)
 *   [6] gradient norm after unscale + value clip (out)        [7] skipped steps (in/out)
 *   [8] total loss*samples  [9] total samples  [10] consecutive NaN losses (tgt_loss_accumulate)
 *   [12],[13] scratch pair (loss*samples, samples) between the two halves of tgt_loss_accumulate
 * grad: the (all-reduced, still scaled) flat gradient; partial: tgt_grad_stats_parts() floats.
 * dynamic != 0: GradScaler semantics (skip + S *= backoff_factor on a non-finite gradient,
 * S *= growth_factor after growth_interval consecutive clean steps); dynamic == 0: S is left alone
 * and nothing is skipped (bf16 / fp32 with clip_grad_norm).  clip_norm <= 0: coefficient 1.
 * ---------------------------------------------------------------------- */
enum { TGT_CTL_SCALE = 0, TGT_CTL_TRACKER = 1, TGT_CTL_FOUND_INF = 2, TGT_CTL_STEPS = 3, TGT_CTL_MULT = 4,
       TGT_CTL_COEF = 5, TGT_CTL_NORM = 6, TGT_CTL_SKIPPED = 7, TGT_CTL_LOSS = 8, TGT_CTL_SAMPLES = 9,
       TGT_CTL_NAN = 10, TGT_CTL_LOSS_LO = 11, TGT_CTL_PAIR = 12, TGT_CTL_SAMPLES_LO = 14, TGT_CTL_LR = 15, TGT_CTL_SIZE = 16 };
/* TGT_CTL_LR (ABI 29): tgt_adam_step called with ctl != NULL and lr < 0 reads the learning rate from ctl[15] -- a captured
 * (hipGraph) step cannot take a new host argument per replay; the schedule's value is written into the block before each one. */
/* TGT_CTL_LOSS / TGT_CTL_SAMPLES are (value, low-order part) float pairs with TGT_CTL_LOSS_LO / TGT_CTL_SAMPLES_LO: the running sums
 * are value + low part (read both, add in float64).  TGT_CTL_STEPS is an exact float32 count up to 2^24 optimizer steps (the
 * reference's longest schedule is 3e5).  TGT_CTL_COEF follows torch's clip_grad_norm_: NaN when the gradient norm is NaN. */
int tgt_grad_stats_parts(void);
int tgt_grad_scaler_step(const float* grad, int64_t n, float* ctl, float* partial, int32_t world,
                         float clip_value, float clip_norm, int32_t dynamic, float growth_factor,
                         float backoff_factor, int32_t growth_interval, void* stream);

/* ------------------------------------------------------------------------
 * update_losses without `.item()` (reference lib/training_schemes/pcqm/tgt_training.py:141-171):
 * mode 1: ctl[12..13] <- (loss*samples, samples)   (then the caller all-reduces those two floats)
 * mode 2: accumulate ctl[12..13] into ctl[8..9]; with mixed != 0 a NaN step loss is skipped unless
 *         it is the 11th in a row (the reference's _nan_loss_count rule)
 * mode 3: both (single rank).  loss: one device scalar, float32 or (loss_is_f64) float64.
 * ---------------------------------------------------------------------- */
int tgt_loss_accumulate(const void* loss, int32_t loss_is_f64, float samples, float* ctl, int32_t mixed,
                        int32_t mode, void* stream);

/* ------------------------------------------------------------------------
 * Edge-channel Linear with the neighbouring passes fused in (csrc/edge_gemm.hip):
 *     out[M, N] = epilogue( a[M, K] . w[N, K]^T + bias ),   K in {64, 128, 256}
 * Replaces, on the (B*N*N, C) edge rows, the nn.Linear -> GELU/Dropout -> nn.Linear -> residual add_ -> nn.LayerNorm
 * chains of reference lib/tgt/layers/layers.py:37-38,:62-80 (mha_ln_e, lin_EG, lin_O_e),
 * :155-160 (FFN), :270-290 (residual wiring) and lib/tgt/layers/triplet.py:207-211 (tri_ln_e, lin_EG projections),
 * forward and data-gradient.
 *   TGT_EPI_BIAS     out = z * out_scale[row / rows_per_sample]        (out_scale may be NULL)
 *   TGT_EPI_GELU     out2 = z (pre-activation);  out = dropout(gelu(z), dropout_p, dropout_seed)
 *                    (* row_scale[m / rows_per_sample] when given: N = 256, K in {64,128,256} only -- tgt_gelu_dropout_scaled_fwd's factor)
 *                    (generator of tgt_gelu_dropout_fwd on the (M, N) index space)
 *   TGT_EPI_RESID    out = res + row_scale[row / rows_per_sample] * z   (row_scale may be NULL)
 *   TGT_EPI_GELU_BWD out = z' * gelu'(res) * keep / (1-p), z' = z * out_scale[..]; res = the forward's pre-activation
 *                    N = 256: colsum_partial (tgt_edge_linear_parts(M, N), N) float32 when given = per-workgroup column sums of
 *                    `out` as stored: the bias gradient of the Linear in front of the activation (sum the rows: tgt_sum_rows)
 *   TGT_EPI_LN_BWD   z = dy, the gradient at the output of LayerNorm(res; gamma) with saved mean / rstd (N <= 256):
 *                    out  = ds_in + rstd * (dy*gamma - mean_n(dy*gamma) - xhat * mean_n(dy*gamma*xhat))   (ds_in may be NULL)
 *                    out2 = out * row_scale[..]  (when given: the gradient of the branch DropPath scaled)
 *                    colsum_partial (tgt_edge_linear_parts(M, N), 3N) float32 when given (N = 256: one row per persistent workgroup,
 *                    all written; N < 256: per row tile, ZERO-FILLED by the caller):
 *                    [sum dy*xhat | sum dy | sum round(out * row_scale)]: dgamma, dbeta and the bias gradient of the Linear
 *                    that produced the branch, to be summed over the rows (tgt_sum_planes).  No bias (a data-gradient GEMM).
 *   TGT_EPI_RESID with gamma / beta / y: additionally y = LayerNorm(out as stored; gamma, beta, eps), mean / rstd (M)
 *                    float32 written when given (N <= 256): the fused entry of the next pre-norm sub-block.
 * Element type 16-bit (TGT_BF16 / TGT_F16; bias in the same type); N % 8 == 0; K in {64, 128, 256}.
 * tgt_edge_linear_supported() tells; anything else is the caller's library GEMM + tgt_layer_norm_* path.
 * tgt_edge_linear_set_grid_cap(cap): TEST HOOK -- at most `cap` persistent workgroups per launch (0 = no cap), so that small
 * problems walk several row tiles per workgroup the way the BASELINE-size launches do.
 * ---------------------------------------------------------------------- */
enum { TGT_EPI_BIAS = 0, TGT_EPI_GELU = 1, TGT_EPI_RESID = 2, TGT_EPI_GELU_BWD = 3, TGT_EPI_LN_BWD = 4 };
/* flags.  TGT_EDGE_BIAS_SCALED (TGT_EPI_RESID, N = 256, K in {64,128,256}): a arrives PRE-SCALED by row_scale (its producer
 * folded the DropPath factor in: tgt_gelu_dropout_scaled_fwd, tgt_node_attention_args.hhat_scale), so
 *     out = res + a W^T + row_scale[row / rows_per_sample] * bias
 * and the backward of the block needs no scaled copy of the stream gradient (tgt_add_layer_norm_bwd with scale and d_x = NULL). */
enum { TGT_EDGE_BIAS_SCALED = 1 };
typedef struct tgt_edge_linear_args {
    int64_t M;
    int32_t K, N, dtype, epilogue;
    const void* a;      int64_t lda;
    const void* w;      int64_t ldw;
    const void* bias;
    const float* gamma; const float* beta; float eps;
    int32_t colsum_rows;   /* rows of colsum_partial the caller provides (ABI 28; 0 = tgt_edge_linear_parts(M, N) as of this call): with N = 256
                              the launch uses EXACTLY this many persistent workgroups, so every provided row is written and none beyond */
    float* mean;        float* rstd;
    void* y;            int64_t ldy;
    void* out;          int64_t ldo;
    void* out2;         int64_t ldo2;
    const void* res;    int64_t ldr;
    const void* ds_in;  int64_t ld_ds;
    const float* row_scale; const float* out_scale; int64_t rows_per_sample;
    float dropout_p;    uint32_t flags;  uint64_t dropout_seed;      /* flags: TGT_EDGE_* */
    float* colsum_partial;
    /* ABI 30.  Fused weight gradient (TGT_EPI_GELU_BWD / TGT_EPI_LN_BWD, K = N = 256): when given, the launch ALSO accumulates the weight
     * gradient of the Linear whose data gradient it computes, dW (N = columns of a, 256) = a^T X over the rows, with X recomputed from the
     * epilogue operand -- GELU_BWD: X = dropout(gelu(res)) * out_scale (the forward's TGT_EPI_GELU output, bit for bit); LN_BWD:
     * X = LayerNorm(res; gamma, beta, mean, rstd) (beta REQUIRED) -- so neither a nor X is read a second time by a separate weight-gradient
     * GEMM (autograd of lib/tgt/layers/layers.py:155-160).  dw_partial: (tgt_edge_linear_parts(M, 256), 256, 256) float32, one plane per
     * persistent workgroup, every plane written; sum the planes (tgt_sum_planes).  colsum_rows, when > 0, states the planes provided. */
    float* dw_partial;
} tgt_edge_linear_args;
int tgt_edge_linear_supported(const tgt_edge_linear_args* a);
int tgt_edge_linear_parts(int64_t M, int32_t N);
int tgt_edge_linear(const tgt_edge_linear_args* a, void* stream);
void tgt_edge_linear_set_grid_cap(int32_t cap);

/* ------------------------------------------------------------------------
 * LayerNorm over the last axis (the five per-layer norms of the TGT layer:
 * reference lib/tgt/layers/layers.py:37-38,:150, lib/tgt/layers/triplet.py:195,
 * i.e. the ATen layer_norm forward/backward behind nn.LayerNorm).
 *   x (rows, C) of x_dtype -> y (rows, C) of y_dtype; gamma/beta float32 (C);
 *   mean/rstd float32 (rows) are saved for backward.  C multiple of 8, <= 2048.
 * Backward: dy (dy_dtype) -> dx (dx_dtype), dgamma/dbeta float32 (C).
 * `partial` is caller-provided scratch of tgt_layer_norm_parts()*2*C floats
 * (fixed-order two-stage reduction: deterministic, no atomics).
 * ---------------------------------------------------------------------- */
int tgt_layer_norm_parts(void);
int tgt_layer_norm_fwd(const void* x, int32_t x_dtype, const float* gamma, const float* beta,
                       void* y, int32_t y_dtype, float* mean, float* rstd,
                       int64_t rows, int32_t C, float eps, void* stream);
/* Residual entry of a pre-norm block in one pass (reference lib/tgt/layers/layers.py:169-174 DropPath,
 * the in-place residual add_ of :270-290, and the LayerNorm that opens the next sub-block):
 *   s = res + x * scale[row / rows_per_sample]   (scale may be NULL = 1: no DropPath)  -> s_out (x_dtype)
 *   y = LayerNorm(s) -> y (y_dtype);  mean/rstd saved.
 * Backward: d_total = ds_in (gradient reaching s from the residual path; may be NULL) + LN_bwd(dy);
 *   d_res = d_total (d_dtype);  d_x = d_total * scale (written only when scale != NULL, else d_x may be NULL
 *   and d_res doubles as d_x);  dgamma/dbeta as tgt_layer_norm_bwd.
 *   d_x_colsum (optional, float32 C, must be dbeta + C: one 2C buffer): column sums of d_x, i.e. the
 *   bias gradient of the Linear that produced x, for free in the same pass; `partial` then needs
 *   tgt_layer_norm_parts()*3*C floats instead of *2*C. */
int tgt_add_layer_norm_fwd(const void* x, int32_t x_dtype, const void* res, int32_t res_dtype, const float* scale,
                           int64_t rows_per_sample, void* s_out, const float* gamma, const float* beta,
                           void* y, int32_t y_dtype, float* mean, float* rstd, int64_t rows, int32_t C,
                           float eps, void* stream);
int tgt_add_layer_norm_bwd(const void* dy, int32_t dy_dtype, const void* s, int32_t s_dtype, const void* ds_in,
                           int32_t ds_dtype, const float* scale, int64_t rows_per_sample, const float* gamma,
                           const float* mean, const float* rstd, void* d_res, void* d_x, int32_t d_dtype,
                           float* dgamma, float* dbeta, float* d_x_colsum, float* partial, int64_t rows, int32_t C,
                           void* stream);

/* Fused GELU (erf form) + dropout: the middle of the FFN block (reference
 * lib/tgt/layers/layers.py:157-158).  y = keep(i) ? gelu(x)/(1-p) : 0, where keep(i) is a
 * counter-based hash of (seed, i); the backward recomputes it from the same seed, so no
 * mask is stored.  p = 0: plain GELU.  n elements of `dtype`. */
int tgt_gelu_dropout_fwd(const void* x, void* y, int64_t n, int32_t dtype, float p, uint64_t seed, void* stream);
int tgt_gelu_dropout_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, float p,
                         uint64_t seed, void* stream);
/* The same with a per-sample factor folded in: y = sample_scale[i / elems_per_sample] * dropout(gelu(x)) and the matching
 * backward (dx = sample_scale[...] * ...).  sample_scale: the DropPath factors (0 or 1/keep) of the residual branch this
 * activation feeds (reference layers.py:169-174, :288-290): with the activation pre-scaled, the branch's closing Linear
 * computes  s = res + x' W^T + sample_scale * b  (TGT_EDGE_BIAS_SCALED) and the backward needs no scaled copy of the
 * stream gradient.  elems_per_sample: a multiple of 8. */
int tgt_gelu_dropout_scaled_fwd(const void* x, void* y, int64_t n, int32_t dtype, float p, uint64_t seed,
                                const float* sample_scale, int64_t elems_per_sample, void* stream);
int tgt_gelu_dropout_scaled_bwd(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, float p, uint64_t seed,
                                const float* sample_scale, int64_t elems_per_sample, void* stream);
/* (ABI 24) tgt_gelu_dropout_scaled_bwd that also returns colsum (cols) float32 = the column sums of dx seen as (n / cols, cols)
 * rows, i.e. the bias gradient of the nn.Linear in front of the activation (reference layers.py:156-157, `grad.sum(0)` of
 * lin_W1) without a separate pass over dx.  sample_scale may be NULL.  cols: a multiple of 16/sizeof(T) that divides
 * 256 * 16/sizeof(T) (64 .. 2048 for 16-bit) and n.  partial: scratch of tgt_gelu_colsum_parts() * cols floats.  Fixed summation order. */
int tgt_gelu_colsum_parts(void);
int tgt_gelu_dropout_bwd_colsum(const void* x, const void* dy, void* dx, int64_t n, int32_t dtype, float p, uint64_t seed,
                                const float* sample_scale, int64_t elems_per_sample, int32_t cols, float* partial, float* colsum,
                                void* stream);

/* Column sums of a (rows, C) tensor into float32 (C): the bias gradient of a Linear layer
 * (the `grad_output.sum(0)` ATen reduction behind nn.Linear, e.g. reference
 * lib/tgt/layers/triplet.py:198-203).  partial: tgt_layer_norm_parts()*C floats of scratch. */
int tgt_colsum(const void* x, int32_t x_dtype, int64_t rows, int32_t C, float* out, float* partial, void* stream);

/* Triplet attention forward with the Q/K/V projection fused in: the workgroup that walks node j
 * projects the edge rows it needs on the matrix cores (reference lib/tgt/layers/triplet.py:210-211
 * and :229-230, `lin_QKV_in/out(e_ln)`) and feeds the attention core from registers.
 *   x    : (B,N,N,C) layer-normed edge rows, dtype a->dtype, contiguous
 *   w    : (>= 6*H*D, C) row-major weight whose ROW index is the output channel of a->qkv
 *          (rows q_off/k_off/v_off[dir] + h*D + d), bias: same indexing
 *   a    : as tgt_triplet_attention_fwd, but a->qkv[dir] is an OUTPUT here (the projected rows, for
 *          the backward kernel); a->eg (E/G third arm) is still an input.
 * Supported: N <= 32, D = 16, H % 8 == 0, bf16/fp16, C in {64,128,256}
 * (tgt_triplet_attention_proj_supported() tells; otherwise project with a GEMM and call
 * tgt_triplet_attention_fwd). */
int tgt_triplet_attention_proj_supported(const tgt_triplet_attention_args* a, int32_t C);
int tgt_triplet_attention_proj_fwd(const tgt_triplet_attention_args* a, const void* x, int32_t C, const void* w,
                                   const void* bias, void* stream);

/* Kernel-order parameters of a triplet module in one launch.  The reference holds the
 * projections as separate nn.Linear with head-minor channels (lib/tgt/layers/triplet.py:198-203,
 * :23-43); tgt_triplet_*_args want one fused, head-major projection.
 *   tgt_fuse_rows   : fused[r,:] = src[row_src[r]][row_idx[r],:]  (row_src < 0: a zero row), and
 *                     the same for the bias vectors when fused_bias != NULL; values are converted
 *                     src_dtype -> fused_dtype.
 *   tgt_unfuse_rows : the inverse scatter (fused -> sources): the gradient path.  Source rows that
 *                     no fused row references are left untouched.
 * row_src/row_idx are device int32 (n_rows).  All matrices row-major contiguous, n_cols wide. */
typedef struct tgt_fuse_rows_args {
    int32_t n_rows, n_cols, n_src;
    int32_t src_dtype, fused_dtype;
    int32_t _pad0;
    const int32_t* row_src;
    const int32_t* row_idx;
    void* src[8];
    void* src_bias[8];
    void* fused;
    void* fused_bias;              /* NULL: weights only */
} tgt_fuse_rows_args;
int tgt_fuse_rows(const tgt_fuse_rows_args* a, void* stream);
int tgt_unfuse_rows(const tgt_fuse_rows_args* a, void* stream);

/* dst[r][c] = src[r][idx[c]] with dtype conversion: lin_O's input columns in the kernels'
 * [dir][h][d] output order (reference order d*2H + dir*H + h, triplet.py:248); the gradient uses
 * the inverse index.  idx: device int32 (cols). */
int tgt_permute_cols(const void* src, int32_t src_dtype, const int32_t* idx, void* dst, int32_t dst_dtype,
                     int32_t rows, int32_t cols, void* stream);

/* out[c] = sum_r x[r*C + c] for a small float32 (rows, C) matrix, fixed summation order (the last
 * stage of the two-stage reductions; finishes tgt_triplet_attention_args.d_*_colsum). */
int tgt_sum_rows(const float* x, int32_t rows, int32_t C, float* out, void* stream);

/* out[i] = sum_p x[p*n + i] over `planes` contiguous float32 planes of n elements, fixed summation
 * order: the closing sum of a weight gradient computed as per-row-chunk partial products
 * (dW = sum_c dY_c^T X_c -- what `grad_weight` of nn.Linear is in the lib/tgt/layers modules, contracted
 * over B*N*N = 262144 rows). */
int tgt_sum_planes(const float* x, int32_t planes, int64_t n, float* out, void* stream);

/* dst_i (cols_i, rows_i) = transpose of src_i (rows_i, cols_i), i < n, 16-bit elements, row-major contiguous, ONE launch.
 * `items` is a DEVICE array.  No reference counterpart: the data-gradient launches of tgt_edge_linear take W^T of the
 * nn.Linear they differentiate (autograd of lib/tgt/layers/layers.py:155-160, triplet.py:248-249); weights change once per
 * optimizer step, so the trainer refreshes every W^T of the model with this call instead of one copy kernel per Linear and
 * backward launch.  blocks_per_item: workgroups per matrix (each walks 32 x 32 tiles). */
typedef struct tgt_transpose_item { const void* src; void* dst; int32_t rows, cols; } tgt_transpose_item;
/* ABI 28: up to 64 plane sums (dst[i] (n) = sum over planes of src[i] (planes, n), float32, the fixed order of tgt_sum_planes: bit-identical
 * to it) as ONE launch: the closing sums behind the split-M weight gradients and column-sum partials of a layer (reference:
 * the reductions inside autograd's Linear / LayerNorm backward, lib/tgt/layers/layers.py:155-160,270-290).  `items` is a HOST array
 * (the descriptors travel as kernel arguments; nothing is read from it after the call returns). */
typedef struct tgt_sum_item { const float* src; float* dst; int32_t planes; int32_t _pad; int64_t n; } tgt_sum_item;
int tgt_sum_many(const tgt_sum_item* items, int32_t n, void* stream);

/* ABI 29.  A device-resident 64-bit step counter mixed into every dropout seed of tgt_gelu_dropout_* and tgt_edge_linear
 * (seed' = seed + *counter * 0x9E3779B97F4A7C15): a training step captured in a hipGraph bakes the seeds its host drew into the
 * graph; the counter -- incremented by the graph itself once per replay -- gives each replay its own drop patterns, forward and
 * backward of one step seeing the same value.  Process-wide, read at launch time; NULL (the default) = seeds as given.  The
 * attention-dropout seeds of tgt_triplet_attention_* / tgt_node_attention_* are NOT covered (the captured step refuses p > 0 there).
 * Replaces nothing in the reference: torch's own graph-safe Philox offsets play this role for its dropout under CUDA graphs. */
int tgt_set_seed_counter(const void* device_counter);
int tgt_transpose_many(const tgt_transpose_item* items, int32_t n, int32_t blocks_per_item, void* stream);

/* Row-wise cross entropy of the binned-distance head: replaces
 * `F.cross_entropy(dist_logits.view(-1, num_bins), dist_targ.view(-1), reduction='none')`
 * (reference lib/training_schemes/pcqm/commons.py:36-38) and its autograd chain, on the logits in
 * their storage dtype (fp32 math on the stored values, which is what autocast's fp32 cast computes).
 *   fwd: lse[r] = log sum_c exp(logits[r][c]);  xent[r] = lse[r] - logits[r][target[r]]
 *   bwd: d_logits[r][c] = row_weight[r] * (exp(logits[r][c] - lse[r]) - [c == target[r]])
 * row_weight is the upstream gradient of xent (the reference's mask / (mask.sum() + 1e-9), :44-46,
 * times whatever scales the loss); rows with weight 0 are written as zeros without being read.
 * logits / d_logits: (rows, C) contiguous, 16-byte aligned, C a multiple of 8, <= 2048; target int64. */
int tgt_cross_entropy_fwd(const void* logits, int32_t dtype, const int64_t* target, int64_t rows, int32_t C,
                          float* lse, float* xent, void* stream);
int tgt_cross_entropy_bwd(const void* logits, int32_t dtype, const int64_t* target, const float* lse,
                          const float* row_weight, int64_t rows, int32_t C, void* d_logits, void* stream);

int tgt_layer_norm_bwd(const void* dy, int32_t dy_dtype, const void* x, int32_t x_dtype,
                       const float* gamma, const float* mean, const float* rstd,
                       void* dx, int32_t dx_dtype, float* dgamma, float* dbeta, float* partial,
                       int64_t rows, int32_t C, void* stream);
/* ---- MC-sampled prediction steps and the bin format between the two inference stages ----------
 *
 * The reference decides on the HOST, after every stochastic forward, whether the sample had a NaN/Inf
 * (lib/training_schemes/pcqm/dist_pred/scheme.py:186-188, gap_pred/scheme.py:88-89: one device sync per sample).  Here
 * the accept/skip decision is device state: `state` = 4 int32 {valid samples so far, non-finite seen in the sample in
 * flight, tries, reserved}, zeroed by the caller before the loop; S = the number of samples wanted
 * (`nb_draw_samples`).  Same sequence of accepted samples as the reference's loop, no sync until results are read. */

/* One sample of `predict_bins` (dist_pred/scheme.py:186-196): softmax over the bins, p[i,j] + p[j,i], argmax (first
 * maximum).  logits (B,N,N,NB) contiguous, NB % 8 == 0, <= 2048.  Written into slot state[0] of bins (B,S,N,N)
 * (elements of bins_elem_size 1 / 2 / 4 bytes: uint8 / uint16 / int32; ld_b = elements between graphs = S*N*N);
 * nothing is written once state[0] >= S.  Sets state[1] when a logit is NaN/Inf; follow with tgt_sample_commit. */
int tgt_dist_bins_argmax(const void* logits, int32_t dtype, int64_t B, int32_t N, int32_t NB, void* bins,
                         int32_t bins_elem_size, int64_t ld_b, int32_t S, int32_t* state, void* stream);
/* if (!state[1] && state[0] < S) state[0]++;  state[1] = 0;  state[2]++ */
int tgt_sample_commit(int32_t* state, int32_t S, void* stream);
/* One sample of `predict_probs` (dist_pred/scheme.py:143-155): checks the logits (state[1]) and, when they are finite
 * and the loop is not complete, acc[row][:] += softmax(logits[row][:]) (float32 acc, same shape); follow with
 * tgt_sample_commit. */
int tgt_softmax_accumulate(const void* logits, int32_t dtype, int64_t rows, int32_t NB, float* acc, int32_t* state,
                           int32_t S, void* stream);
/* out[b,i,j,:] = (acc[b,i,j,:] + acc[b,j,i,:]) / (2 state[0])  (dist_pred/scheme.py:164-166); as_log: log(. + eps)
 * (:173).  out != acc. */
int tgt_probs_finish(const float* acc, int64_t B, int32_t N, int32_t NB, const int32_t* state, int32_t as_log, float eps,
                     float* out, void* stream);
/* One sample of the gap loop (gap_pred/scheme.py:88-96): gap (B) of `dtype`; when all B values are finite and
 * state[0] < S: out[b*S + state[0]] = gap[b] (float32) and state[0]++; state[2]++ either way. */
int tgt_gap_commit(const void* gap, int32_t dtype, int32_t B, float* out, int32_t S, int32_t* state, void* stream);
/* `pack_bins_multi` of every graph's real nodes (lib/data/pcqm/bin_ops.py:32-37, dist_pred/scheme.py:221-226):
 * flat[offsets[b] + s*T_b + k] = bins[b,s,i,j], (i,j) the k-th pair (row-major) of the strict upper triangle of the
 * n_b = num_nodes[b] real nodes, T_b = n_b(n_b-1)/2.  bins (B,S,N,N) of elem_size bytes; num_nodes (B) and
 * offsets (B+1, offsets[b] = S * sum_{b' < b} T_b') device int64; total = offsets[B]. */
int tgt_pack_triu(const void* bins, int32_t elem_size, int32_t B, int32_t S, int32_t N, const int64_t* num_nodes,
                  const int64_t* offsets, void* flat, int64_t total, void* stream);
/* `BinsProcessor.bins2dist` (lib/training_schemes/pcqm/commons.py:72-82), the same float32 operations in the same
 * order: dist[r,i,j] = (u(i,j) + h) * bin_size + (u(j,i) + h) * bin_size, h = 0.5 when shift_half, 0 on the diagonal
 * when zero_diag.  bins (R,N,N) of `kind`; u = bins, or -- num_nodes != NULL, R = B*S -- bins restricted to
 * i < j < num_nodes[r / S] (zero elsewhere): what packing the bins and unpacking them into the zero-padded batch
 * yields (bin_ops.py:39-46), so the two inference stages can be chained on the device. */
enum { TGT_BINS_U8 = 0, TGT_BINS_U16 = 1, TGT_BINS_I32 = 2, TGT_BINS_I64 = 3, TGT_BINS_F32 = 4 };
int tgt_bins_to_dist(const void* bins, int32_t kind, int64_t R, int32_t N, const int64_t* num_nodes, int32_t S,
                     float bin_size, int32_t shift_half, int32_t zero_diag, float* out, void* stream);
/* Gaussian basis of the 3-D distance embedding (reference lib/models/pcqm/layers.py:129-157, `gaussian()` +
 * GaussianLayer.forward): for every node pair p (pairs = B*N*N), t = mul[p]*x[p] + bias[p] (float32: x the pair distance,
 * mul / bias the summed atom-type embeddings), y[p,k] = exp(-((t - mean[k]) / s_k)^2 / 2) / ((2*3.14159)^0.5 * s_k),
 * s_k = |std[k]| + 0.01.  y (pairs, K) in `dtype` (the dtype the consuming Linear computes in).  K even, <= 512.
 * Backward: g = dL/dy (pairs, K) of `dtype`; dt[p] = dL/dt (float32; dmul = dt*x, dbias = dt); partial:
 * tgt_gaussian_basis_parts(pairs) rows of [dL/dmean (K) | dL/dstd (K)] float32 to be summed over rows (tgt_sum_rows). */
int tgt_gaussian_basis_parts(int64_t pairs);
int tgt_gaussian_basis_fwd(const float* x, const float* mul, const float* bias, const float* mean, const float* std,
                           int64_t pairs, int32_t K, int32_t dtype, void* y, void* stream);
int tgt_gaussian_basis_bwd(const float* x, const float* mul, const float* bias, const float* mean, const float* std,
                           int64_t pairs, int32_t K, int32_t dtype, const void* g, float* dt, float* partial, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TGT_HIP_H */

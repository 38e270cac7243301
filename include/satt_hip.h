/* satt_hip.h — C-ABI of libsatt_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the teacher-forced
 * training path of Self-attention Tacotron.
 *
 * The reference (nii-yamagishilab/self-attention-tacotron, TF1) has NO native/FFI boundary:
 * every entry point below replaces a group of implicit TensorFlow ops on the hot path; the reference call site
 * each one replaces is cited as file:line relative to the reference tree.  INTEGRATION.md shows the
 * reference-side binding (a ctypes stub) a maintainer would add.
 *
 * Conventions
 *  - all pointers are DEVICE pointers owned by the caller; the library never allocates, frees or synchronises
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (hipGraph-capturable)
 *  - return 0 on success, negative SATT_E_* otherwise; satt_strerror() maps codes to text
 *  - activations / gradients are row-major fp32; recurrent weights are passed as bf16 (uint16_t) copies
 *  - `seed` is a device pointer to one uint32 (dropout / zoneout seed of this step), masks follow
 *    keep(seed, stream_id, idx) of csrc/common.h (== oracle/rng.py); `thresh` = rate*2^32, `scale` = 1/(1-rate)
 *  - no global mutable state: re-entrant, thread-safe per (device, stream)
 */
#ifndef SATT_HIP_H
#define SATT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SATT_OK 0
#define SATT_E_BADARG (-1)
#define SATT_E_UNSUPPORTED (-2)
#define SATT_E_LAUNCH (-3)
#define SATT_E_ARCH (-4)

#define SATT_ACT_NONE 0
#define SATT_ACT_RELU 1
#define SATT_ACT_TANH 2
#define SATT_ACT_SIGMOID 3
#define SATT_ACT_SOFTSIGN 4

#define SATT_PREC_F32 0  /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32) */
#define SATT_PREC_BF16 1 /* operands rounded to bf16, fp32 accumulate (v_mfma_f32_16x16x32_bf16) */

int satt_version(void);
const char* satt_strerror(int code);
/* 1 if `device` is gfx950, 0 otherwise, negative on error */
int satt_arch_supported(int device);

/* ---- generic MFMA GEMM with fused epilogue ---------------------------------------------------------------
 * C[z][m][n] (+)= epilogue( alpha * sum_k A(z,m,k) * B(z,k,n) )
 * Replaces tf.layers.Dense / tf.tensordot / tf.layers.Conv1D / tf.matmul on the hot path:
 *   PreNet + Dense layers (modules/module.py:394,408,426,429), Projection (modules/module.py:626-643),
 *   MultiHeadAttention projections and QK^T / PV matmuls (modules/self_attention.py:45-65,102-128),
 *   Conv1d bank / projections of ZoneoutCBHG (modules/module.py:46-68,78-83),
 *   BahdanauAttention memory layer (modules/forward_attention.py:59-64), hoisted LSTM input GEMMs,
 *   and all their backward GEMMs (dX, dW).
 * a_mode: 0 A(m,k)=A[m*lda+k]            1 A(m,k)=A[k*lda+m]
 *         2 virtual im2col: m=(b,t), k=(tap,c): A[(b*conv_T+t')*lda+c], t'=t+conv_sgn*tap+conv_off, 0 outside [0,conv_T)
 *         3 same element with m=(tap,c), k=(b,t)   (weight-gradient form)
 * B(k,n) = B[(k/kin)*sb_tap + (k%kin)*sb_k + n*sb_n]
 * batch z = zo*nb_inner + zi ; operand offset = zo*stride?_o + zi*stride?_i
 * epilogue order: *alpha, +bias[n], act, dropout(idx=m*N+n), +residual[m*ldr+n], (+C if accumulate), store
 * splitk>1: K is split over splitk workgroups which atomically add into C (requires accumulate=1, no bias/act).
 */
typedef struct {
  int M, N, K;
  int nb_outer, nb_inner;
  const float* A; int64_t lda, strideA_o, strideA_i; int a_mode;
  int conv_T, conv_C, conv_sgn, conv_off;
  const float* B; int64_t sb_tap, sb_k, sb_n, strideB_o, strideB_i; int kin;
  float* C; int64_t ldc, strideC_o, strideC_i;
  const float* bias;
  const float* residual; int64_t ldr;
  int act;
  float alpha;
  int accumulate;
  int splitk;
  uint32_t drop_thresh; float drop_scale; uint32_t drop_stream; const uint32_t* seed;
  int precision;
  /* conv-bank mode (a_mode 2 only; bank_ng > 0): ONE launch for the convolutions of widths 1..bank_ng over the same
   * input (ZoneoutCBHG conv bank, modules/module.py:46-68).  Group g = width g+1: K = (g+1)*conv_C,
   * conv_off = -conv_sgn*(g/2) (SAME padding), B += bank_b_unit*g*(g+1)/2 (weights of all widths contiguous),
   * A += g*bank_a_col, C += g*bank_c_col.  bank_c_col == 0: every group adds into the same C (atomics; needs
   * accumulate, no epilogue) - the input gradient of the bank.
   * a_mode 3 with bank_ng > 0 (large-tile kernel only, else SATT_E_UNSUPPORTED): the WEIGHT gradients of all widths in
   * one launch over the shared input A: group g has M = (g+1)*conv_C rows (tap, c), conv_off = -conv_sgn*(g/2),
   * B += g*bank_c_col, and its rows start at conv_C*g*(g+1)/2 of C (ldc == N: the weights of all widths contiguous). */
  int bank_ng, bank_a_col, bank_c_col; int64_t bank_b_unit;
  /* bf16 SHADOW of B for the large-tile kernels (PREC_BF16, a_mode 0 / 2; NULL: none).  Same elements as B, laid out
   * k-contiguous per output column: B(k,n) = Bs[(k/kin)*sbs_tap + n*sbs_n + (k%kin)] (satt_shadow_pack produces both
   * orientations of every weight once per optimiser step).  With a shadow, 16-byte aligned operands, K % 8 == 0 and
   * (a_mode 2) conv_C % 32 == 0, kin % 32 == 0 the call runs on 128x128 / 64-wide double-buffered MFMA tiles;
   * otherwise, and always in PREC_F32, on the generic kernel reading B.  Results agree to bf16 operand rounding. */
  const uint16_t* Bs; int64_t sbs_tap, sbs_n;
  /* weight-gradient calls (a_mode 1 / 3, sb_n == 1): colsum[n] += sum_k B(k,n) - the bias gradient of the same layer
   * (tf.layers.Dense bias), fused into the GEMM instead of a separate satt_colsum launch.  NULL: off. */
  float* colsum;
  /* split reductions without atomics (large-tile kernels only): `ws` = satt_gemm_ws_floats(p) floats; every split
   * (splitk > 1, or the paired groups of a summed conv bank) stores its partial result to its own slab and a second
   * launch sums the slabs into C - deterministic, and with accumulate == 0 C needs no zeroing.  NULL: atomics. */
  float* ws;
} satt_gemm_params;
int satt_gemm(const satt_gemm_params* p, void* stream);
/* host-only (no launch): the kernel family satt_gemm runs this problem on - 0 generic 64x64 kernel, 1 large-tile
 * forward / input-gradient kernel (needs Bs), 2 large-tile weight-gradient kernel, 3 the conv bank's own kernel (forward and, with
 * a workspace, input gradient: input rows resident in LDS; 128 channels in, 128 filters per width, widths 1..bank_ng <= 16,
 * nothing fused behind the product);
 * negative SATT_E_* on bad arguments */
int satt_gemm_path(const satt_gemm_params* p);
int64_t satt_gemm_ws_floats(const satt_gemm_params* p);

/* bf16 shadows of GEMM weights, both orientations, for `nweights` tensors [taps][rows][cols] that live at
 * table[4w+0] (element offset) inside the flat fp32 parameter buffer; table[4w+1..3] = taps, rows, cols (int64, device).
 *   sn[off + e]                         = bf16(flat[off + e])                      plain cast  (dX: reduce over cols)
 *   st[off + (tap*cols + c)*rows + r]   = bf16(flat[off + (tap*rows + r)*cols + c])  per-tap transpose (forward)
 * One launch per optimiser step (models/models.py:485-498 has no counterpart: TF keeps one fp32 copy). */
int satt_shadow_pack(const float* flat, const int64_t* table, int nweights, uint16_t* st, uint16_t* sn, void* stream);

/* ---- small fused ops ------------------------------------------------------------------------------------ */
/* Embedding lookup (tacotron2 Embedding; call site models/models.py:351): out[i,:] = table[ids[i]-offset,:] */
int satt_embedding_fwd(const int64_t* ids, const float* table, float* out, int n, int dim, int offset, void* stream);
int satt_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int n, int dim, int offset, void* stream);
/* the same sum, deterministic: one workgroup per table row (nrows rows) adds the gradient rows of its tokens in ascending token
 * order - no atomics (dtable[ids[i]-offset, :] += dout[i, :]); falls back to satt_embedding_bwd for n > 8192 or nrows > 4096.
 * ids outside [offset, offset + nrows) are ignored.  Bit-stable from run to run; slower than the atomic form when one row
 * (e.g. the padding symbol) collects a large share of the tokens: the product path uses satt_embedding_bwd. */
int satt_embedding_bwd_rows(const int64_t* ids, const float* dout, float* dtable, int n, int dim, int offset, int nrows,
                            void* stream);

/* dx = dy * act'(y) (* scale where y != 0 for dropout-after-relu); y is the POST-activation(-dropout) output */
int satt_act_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dx, int64_t lddx,
                 int rows, int cols, int act, float scale, void* stream);
/* the same with the activation output given as z - res, where z = act(u) + res was written in one pass by a GEMM epilogue
 * with a residual (the transformer tail x + tanh(Dense(.)), modules/module.py:363-371): act(u) is recovered to within one
 * rounding of z */
int satt_act_bwd_res(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* res, int64_t ldres, float* dx,
                     int64_t lddx, int rows, int cols, int act, float scale, void* stream);

/* tf.layers.batch_normalization in training mode over rows (B*T incl. padding) of x[rows,C]
 * (external Conv1d; call sites modules/module.py:46-68).  stats: mean[C], rstd[C]; moving stats updated in place
 * (momentum 0.99, unbiased variance for the moving average).  ws: 2*C*nchunk floats (see satt_bn_ws_floats). */
int64_t satt_bn_ws_floats(int rows, int C);
int satt_bn_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                float* mean, float* rstd, float* moving_mean, float* moving_var, float* ws,
                int rows, int C, float eps, float momentum, int act, void* stream);
/* inference mode: normalise with moving statistics */
int satt_bn_infer(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* moving_mean,
                  const float* moving_var, float* y, int64_t ldy, int rows, int C, float eps, int act, void* stream);
/* dy is the gradient wrt the post-activation output; dgamma/dbeta are ACCUMULATED (+=) */
int satt_bn_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* beta,
                const float* mean, const float* rstd, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                float* ws, int rows, int C, int act, void* stream);

/* The same two operations in ONE launch each (small activations are launch-bound: three launches + two gaps per direction):
 * the last workgroup of a 64-channel group merges the partial statistics and releases the others, which then normalise their
 * own rows.  ws: satt_bn_fused_ws_floats(rows, C) floats (the backward call needs the forward's ws no longer); sync:
 * satt_bn_fused_sync_words(C) 32-bit words, ZERO before the first launch and zero again after every launch (keep one per call
 * site; two launches that may overlap must not share it).  SATT_E_UNSUPPORTED when the grid is too large for every waiting
 * workgroup to stay resident (then use satt_bn_fwd / satt_bn_bwd).  A barrier timeout writes NaN outputs. */
int64_t satt_bn_fused_ws_floats(int rows, int C);
int satt_bn_fused_sync_words(int C);
int satt_bn_fwd_fused(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, float* mean,
                      float* rstd, float* moving_mean, float* moving_var, float* ws, uint32_t* sync, int rows, int C, float eps,
                      float momentum, int act, void* stream);
int satt_bn_bwd_fused(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, float* dx, int64_t lddx, float* dgamma, float* dbeta, float* ws,
                      uint32_t* sync, int rows, int C, int act, void* stream);

/* tf.layers.MaxPooling1D(pool 2, stride 1, SAME) over time (modules/module.py:54,80): y[t]=max(x[t],x[t+1]) */
/* BatchNorm (training statistics) + activation + MaxPooling1D(2, 1, SAME) of the conv bank in one pass over contiguous [B*T, C]
 * rows (modules/module.py:46-68,80): mp[t] = max(y[t], y[t+1]) with y = act(bn(x)) NEVER stored; the backward pair recomputes y
 * from x with the same expression (identical tie decisions), returns d x, accumulates d gamma / d beta.  ws as satt_bn_fwd/bwd;
 * dbuf [B*T, C] scratch.  Forward: C % 4 == 0 and 16-byte aligned operands (SATT_E_UNSUPPORTED otherwise: use satt_bn_fwd +
 * satt_maxpool_fwd, and the separate backward calls with the stored y). */
int satt_bn_maxpool_fwd(const float* x, const float* gamma, const float* beta, float* mp, float* mean, float* rstd,
                        float* moving_mean, float* moving_var, float* ws, int B, int T, int C, float eps, float momentum, int act,
                        void* stream);
int satt_maxpool_bn_bwd(const float* dmp, const float* x, const float* gamma, const float* beta, const float* mean,
                        const float* rstd, float* dx, float* dgamma, float* dbeta, float* ws, float* dbuf, int B, int T, int C,
                        int act, void* stream);
int satt_maxpool_fwd(const float* x, float* y, int B, int T, int C, void* stream);
int satt_maxpool_bwd(const float* dy, const float* x, float* dx, int B, int T, int C, void* stream);

/* HighwayNet combine (external; call modules/module.py:72,91): z=[H_pre|T_pre] (bias already added)
 * y = relu(Hp)*sigmoid(Tp) + x*(1-sigmoid(Tp)) */
int satt_highway_fwd(const float* z, const float* x, float* y, int rows, int H, void* stream);
int satt_highway_bwd(const float* dy, const float* z, const float* x, float* dz, float* dx, int rows, int H,
                     void* stream);

/* The whole highway stack of the CBHG encoder (modules/module.py:87-90: num_highway HighwayNet layers back to back, each
 * row-wise) in ONE launch per direction, bf16 MFMA / fp32 accumulate - the arithmetic of satt_gemm in bf16 mode followed by
 * satt_highway_fwd / _bwd, layer by layer.  Per layer: Wt = bf16 [2H][H] (transposed shadow, forward), Wn = bf16 [H][2H] (plain
 * shadow, backward), b [2H], z [rows,2H] pre-activations (written forward, read backward), y [rows,H] layer output (written
 * forward; layer n+1's input in the backward pass), dz [rows,2H] (written backward: the weight-gradient GEMMs' operand).
 * H must be 128, nlayers <= SATT_HIGHWAY_MAX_LAYERS; SATT_E_UNSUPPORTED otherwise (callers keep the per-layer form). */
#define SATT_HIGHWAY_MAX_LAYERS 8
typedef struct satt_highway_layer {
  const void* Wt;
  const void* Wn;
  const float* b;
  float* z;
  float* y;
  float* dz;
} satt_highway_layer;
int satt_highway_stack_fwd(const float* x, const satt_highway_layer* layers, int nlayers, int rows, int H, void* stream);
/* dx [rows,H] = gradient wrt the stack input x; dy = gradient wrt the last layer's output */
int satt_highway_stack_bwd(const float* dy, const float* x, const satt_highway_layer* layers, int nlayers, int rows, int H,
                           float* dx, void* stream);

/* column sums: out[c] (+)= sum_r x[r*ldx+c]  (bias gradients) */
int satt_colsum(const float* x, int64_t ldx, float* out, int rows, int cols, int accumulate, void* stream);

/* Weight gradient of ForwardAttention's location convolution (modules/forward_attention.py:68-73: Conv1D(filters 5, kernel
 * 10, SAME) over the previous alignments), a 1-channel conv with 50 + 5 outputs reduced over B*Td*Ti rows:
 *   dF[j,0,k] += sum a1[b,t-1,t'+j-pad_left] * dfl[b,t,t',k]  (a1 of step -1 is zero),  dbF[k] += sum dfl[b,t,t',k]
 * SATT_E_UNSUPPORTED unless kernel == 10 and filters == 5 (use satt_gemm a_mode 3 + satt_colsum then). */
int satt_loc_filter_dw(const float* a1, const float* dfl, float* dF, float* dbF, int B, int Td, int Ti, int kernel,
                       int filters, void* stream);

/* y = a*x + b*y over a strided 2-D view */
int satt_axpby(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, float a, float b,
               void* stream);
/* y[b,t,:] = t < len[b] ? x[b,t,:] : 0  (BahdanauAttention._prepare_memory masking; forward_attention.py:59-64) */
/* round_bf16: y additionally rounded to the nearest bf16 value (kept as fp32).  The attention memory of the benchmark precision:
 * every consumer reads it as bf16 anyway, and the backward loop's identities (csrc/attn_cluster.hip, phase (b)) need the
 * value rows of both passes to be the SAME numbers */
int satt_seq_mask(const float* x, const int64_t* lengths, float* y, int B, int T, int C, int round_bf16, void* stream);

/* MultiSpeakerPreNet broadcast (modules/multi_speaker_modules.py:29): y[(b*T+t), :] += s[b, :]; and its adjoint
 * ds[b, :] (+)= sum_t x[(b*T+t), :] */
int satt_bcast_add(const float* s, float* y, int B, int T, int C, void* stream);
int satt_segment_colsum(const float* x, float* ds, int B, int T, int C, int accumulate, void* stream);

/* fp32 -> bf16 copy, optionally transposed: dst[c*rows+r] = src[r*ld+c] */
int satt_to_bf16(const float* src, int64_t ld, uint16_t* dst, int rows, int cols, int transpose, void* stream);

/* ---- softmax of attention scores (modules/self_attention.py:45-65) ---------------------------------------
 * s: [nbh, T, T] raw QK^T (scaled by `scale` here); causal: keys j>i masked to -inf (apply_subsequent_mask :79-86)
 * p: pre-dropout probabilities (the returned alignments, :59); pd: dropout(p) used for PV (:61)  */
int satt_softmax_fwd(const float* s, float* p, float* pd, int nbh, int T, float scale, int causal,
                     uint32_t drop_thresh, float drop_scale, uint32_t drop_stream, const uint32_t* seed,
                     void* stream);
/* tf.layers.dropout with the counter-based mask keep(seed, stream, r*cols + c): y = keep ? x*scale : 0.  The same call
 * applied to dy is the backward (PostNetV2 layers, models/models.py:92-100; thresh == 0: copy) */
int satt_dropout(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, uint32_t drop_thresh,
                 float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream);
/* p[r, 0:cols] = softmax(scale * s[r, 0:cols]) for rows of arbitrary leading dimension: the single query row of the
 * KV-cached incremental decoder self-attention (inference branch, modules/rnn_wrappers.py:87-124) */
int satt_softmax_rows(const float* s, int64_t lds, float* p, int64_t ldp, int rows, int cols, float scale, void* stream);
/* dpd: gradient wrt pd; ds: gradient wrt raw scores (includes `scale`) */
int satt_softmax_bwd(const float* dpd, const float* p, float* ds, int nbh, int T, float scale, int causal,
                     uint32_t drop_thresh, float drop_scale, uint32_t drop_stream, const uint32_t* seed,
                     void* stream);

/* ---- fused scaled-dot-product self-attention for head depth 16 (the ENCODER block: SelfAttentionCBHGEncoder's 2 heads x 16,
 * modules/self_attention.py:45-65,108-128; no mask).  kvq [B*T, ld] = K | V | Q, each D = H*16 wide, head h at columns h*16.
 * Forward: p [B*H, T, T] = softmax(scale Q K^T) (written: these are the `alignment` outputs of the block, models/models.py:397-408),
 * o [B*T, ldo] = dropout(p) V with the counter keep(seed, stream, ((b*H + h)*T + i)*T + j) of satt_softmax_fwd.
 * Backward: dkvq [B*T, ldd] = dK | dV | dQ, every element written once (no atomics); rowsum [B*H, T] is scratch.
 * fp32 FMAs against LDS-staged rows (an MFMA tile is mostly padding at depth 16): one launch forward, two backward, instead of
 * GEMM -> softmax -> GEMM and 4 GEMMs + softmax backward.  Other head depths: SATT_E_UNSUPPORTED (satt_small_attn_supported). */
int satt_small_attn_supported(int head_dim, int T);
int satt_small_attn_fwd(const float* kvq, int64_t ld, float* p, float* o, int64_t ldo, int B, int T, int D, int H, float scale,
                        uint32_t drop_thresh, float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream);
int satt_small_attn_bwd(const float* kvq, int64_t ld, const float* p, const float* dout, int64_t lddo, float* dkvq, int64_t ldd,
                        float* rowsum, int B, int T, int D, int H, float scale, uint32_t drop_thresh, float drop_scale,
                        uint32_t drop_stream, const uint32_t* seed, void* stream);

/* ---- fused scaled-dot-product self-attention (ScaledDotProductAttentionMechanism, modules/self_attention.py:45-65 with
 * apply_subsequent_mask :79-86; MultiHeadAttention head split :113-118).  head_dim must be 128 (the decoder's 2 x 128;
 * otherwise SATT_E_UNSUPPORTED: use satt_gemm + satt_softmax_*).  k, v, q: fp32 [B*T, .] rows with stride ld, head h at
 * columns h*128 .. (pass the K | V | Q column blocks of the fused projection); o [B*T, .] (stride ldo), same head layout.
 *   o = dropout(softmax(q k^T * scale [+ causal mask])) v     bf16 MFMA operands, fp32 online softmax
 * dropout on the probabilities: keep(seed, stream, ((b*H + h)*T + i)*T + j) - the same counter as satt_softmax_fwd.
 * lse [B*H, T] (out): log2-domain log-sum-exp of the scaled scores of every query row, the only tensor kept for backward;
 * the [B*H, T, T] probabilities are never written (the reference does not collect decoder alignments at training time,
 * models/models.py:409).  Backward recomputes them: dk, dv, dq [B*T, .] (stride ldd) are WRITTEN (every element once, no
 * atomics); delta [B*H, T] is scratch. */
int satt_flash_attn_fwd(const float* k, const float* v, const float* q, int64_t ld, float* o, int64_t ldo, float* lse,
                        int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                        float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream);
int satt_flash_attn_bwd(const float* k, const float* v, const float* q, int64_t ld, const float* o, const float* dout,
                        int64_t ldo, const float* lse, float* delta, float* dk, float* dv, float* dq, int64_t ldd,
                        int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                        float drop_scale, uint32_t drop_stream, const uint32_t* seed, void* stream);
/* The same backward restricted to the 64-row key / query tiles [tile_lo, tile_hi) (causal only; a proper sub-range of a
 * non-causal problem is SATT_E_BADARG).  Under the causal mask (modules/self_attention.py:79-86) key tile j takes query tiles
 * >= j and query tile i key tiles <= i, so one launch over the suffix [s, ceil(T/64)) leaves dk, dv, dq rows >= 64 s final and a
 * second launch over [0, s) the rows below: the training step's backward pipeline (which walks the decoder steps late to
 * early) starts on the suffix rows while the prefix launch is still running.  with_delta != 0: the launch first computes delta
 * (and the bf16 d o of the _b form) for the query rows of ITS OWN tiles.  A range reads the delta rows of its own and of LATER
 * tiles: run the ranges from the last to the first, each with with_delta != 0 (r4; until then the first launch computed every
 * row - 13 us in front of the suffix launch the recurrent pipeline waits for); with_delta == 0 re-uses rows computed before. */
int satt_flash_attn_bwd_tiles(const float* k, const float* v, const float* q, int64_t ld, const float* o, const float* dout,
                              int64_t ldo, const float* lse, float* delta, float* dk, float* dv, float* dq, int64_t ldd,
                              int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                              float drop_scale, uint32_t drop_stream, const uint32_t* seed, int tile_lo, int tile_hi,
                              int with_delta, void* stream);
/* bf16 copies of the attention operands (the "bf16 activations in HBM" of the decoder self-attention block): the forward
 * variant additionally WRITES kb | vb | qb - the K, V, Q rows rounded to bf16 (nearest-even, exactly what every kernel does to them
 * on its way into the matrix cores), same head layout, row stride ldb elements (a multiple of 8) - at no extra launch (the
 * workgroup of the diagonal tile stores the rows it stages).  The backward variant READS those copies instead of the fp32 rows and a
 * bf16 copy of d o (doutb, stride ldob: scratch written by the delta passes, see with_delta above): half the bytes per
 * staged tile and no conversion pass; results are bit-identical to satt_flash_attn_bwd_tiles. */
int satt_flash_attn_fwd_b(const float* k, const float* v, const float* q, int64_t ld, float* o, int64_t ldo, float* lse,
                          int B, int T, int H, int head_dim, float scale, int causal, uint32_t drop_thresh,
                          float drop_scale, uint32_t drop_stream, const uint32_t* seed, uint16_t* kb, uint16_t* vb,
                          uint16_t* qb, int64_t ldb, void* stream);
int satt_flash_attn_bwd_tiles_b(const uint16_t* kb, const uint16_t* vb, const uint16_t* qb, int64_t ldb, const float* o,
                                const float* dout, int64_t ldo, uint16_t* doutb, int64_t ldob, const float* lse,
                                float* delta, float* dk, float* dv, float* dq, int64_t ldd, int B, int T, int H,
                                int head_dim, float scale, int causal, uint32_t drop_thresh, float drop_scale,
                                uint32_t drop_stream, const uint32_t* seed, int tile_lo, int tile_hi, int with_delta,
                                void* stream);

/* ---- recurrent ZoneoutLSTM (tacotron2 ZoneoutLSTMCell over tf.nn.rnn_cell.LSTMCell; call sites
 * modules/module.py:93-108 (encoder BiLSTM, with sequence_length) and :1527-1534 (DecoderRNNV2 LSTM1/LSTM2)).
 * The input projection x*W_x+b is hoisted into xg by satt_gemm; this kernel runs only the h-recurrence.
 * xg     [ndir,B,T,4H]  gate pre-activations from the input (order i,j,f,o)
 * Wh     [ndir,H,4H]    bf16 recurrent weights
 * lengths[B] or NULL; direction 1 (if ndir==2) runs reversed over [0,len)
 * hout   [B,T,ld_hout] cell outputs h' (pre-zoneout), direction d written at column d*H; zero beyond len
 * saved for backward: gates [ndir,B,T,4H] (sigmoid(i),tanh(j),sigmoid(f+1),sigmoid(o)), cnew/cstate/hstate [ndir,B,T,H]
 */
int satt_lstm_fwd(const float* xg, const uint16_t* Wh, const int64_t* lengths, int ndir, int B, int T, int H,
                  int training, float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh,
                  const uint32_t* seed, const uint32_t* stream_c, const uint32_t* stream_h,
                  float* hout, int64_t ld_hout, float* gates, float* cnew, float* cstate, float* hstate,
                  void* stream);
/* WhT [ndir,4H,H] bf16 (transposed copy); dhout [B,T,ld] ; dxg [ndir,B,T,4H] = gradient wrt gate pre-activations */
int satt_lstm_bwd(const float* dhout, int64_t ld_dhout, const uint16_t* WhT, const int64_t* lengths,
                  int ndir, int B, int T, int H, int training, float zc, float zh,
                  uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed,
                  const uint32_t* stream_c, const uint32_t* stream_h,
                  const float* gates, const float* cnew, const float* cstate, float* dxg, void* stream);

/* Cluster form for sequences WITHOUT lengths (DecoderRNNV2's LSTM1/LSTM2, modules/module.py:1527-1534): C
 * workgroups per sample keep their [H x 4H/C] bf16 weight slice resident in registers and all-gather H floats per
 * step through 8-byte {tag,value} granules in `ws` (satt_lstm_cluster_ws_bytes).  Same arguments and results as
 * satt_lstm_fwd/bwd with ndir=1, lengths=NULL, except that Wh / WhT are the REGISTER-ORDER PACKS written by
 * satt_lstm_cluster_pack from the fp32 recurrent weights Wh [H,4H] (row stride ld): pack_fwd and pack_bwd of
 * satt_lstm_cluster_pack_elems(C) bf16 elements each, 16-byte aligned, valid for that C only.
 * Requires H % C == 0, (H/C) % 8 == 0, B*C <= 256.
 * [t0,t1) selects a time chunk (stream pipelining of the recurrent layers): the forward restarts from the saved
 * cstate/hstate of step t0-1; backward chunks run from late to early and carry (dc,dh) in bstate [B,2,H].
 * The FIRST launch of a pass (forward: t0 == 0; backward: t1 == T) zeroes `ws`; the later chunk launches of the pass
 * must use the same workspace and continue on it.
 * satt_lstm_cluster_status (host-synchronous) reports a hand-off timeout of any launch since the caller zeroed `ws`. */
int64_t satt_lstm_cluster_pack_elems(int C);
int satt_lstm_cluster_pack(const float* Wh, int64_t ld, int H, int C, uint16_t* pack_fwd, uint16_t* pack_bwd,
                           void* stream);
int64_t satt_lstm_cluster_ws_bytes(int B, int H, int C);
int satt_lstm_cluster_fwd(const float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training, float zc,
                          float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed, uint32_t stream_c,
                          uint32_t stream_h, float* hout, int64_t ld_hout, float* gates, float* cnew, float* cstate,
                          float* hstate, void* ws, int t0, int t1, void* stream);
/* Forward chunk with the INPUT PROJECTION of its steps formed by the launch itself: xg[b, t, :] = x[b, t, 0:Kin] Win + bin for t in
 * [t0, t1) (written to xg, then consumed as above).  x: fp32 rows [B*T, ldx]; Win_pack: satt_lstm_cluster_pack_in of the fp32 input
 * weights Win [Kin, 4H] (row stride ld) for this C, satt_lstm_cluster_pack_in_elems(Kin, C) bf16 elements, 16-byte aligned; bin:
 * [4H] or NULL.  Kin <= 544, Kin % 4 == 0, ldx % 4 == 0.  Operands are rounded to bf16 and accumulated exactly as satt_gemm does
 * (SATT_PREC_BF16): the results equal those of the separate product bit for bit.  Meant for the short chunks at the end of the
 * layer pipeline (the member's 4H/C x Kin weight slice streams from L2 once per 16 steps). */
int64_t satt_lstm_cluster_pack_in_elems(int K, int C);
int satt_lstm_cluster_pack_in(const float* Win, int64_t ld, int K, int H, int C, uint16_t* pack, void* stream);
int satt_lstm_cluster_fwd_x(float* xg, const uint16_t* Wh, int B, int T, int H, int C, int training, float zc,
                            float zh, uint32_t zc_thresh, uint32_t zh_thresh, const uint32_t* seed, uint32_t stream_c,
                            uint32_t stream_h, float* hout, int64_t ld_hout, float* gates, float* cnew, float* cstate,
                            float* hstate, void* ws, int t0, int t1, const float* x, int64_t ldx, int Kin,
                            const uint16_t* Win_pack, const float* bin, void* stream);
int satt_lstm_cluster_bwd(const float* dhout, int64_t ld_dhout, const uint16_t* WhT, int B, int T, int H, int C,
                          int training, float zc, float zh, uint32_t zc_thresh, uint32_t zh_thresh,
                          const uint32_t* seed, uint32_t stream_c, uint32_t stream_h, const float* gates,
                          const float* cnew, const float* cstate, float* dxg, void* ws, int t0, int t1, float* bstate,
                          void* stream);
int satt_lstm_cluster_status(const void* ws, int B, int H, int C, void* stream);
/* host-synchronous (tests): *count = workgroup-launches since the CALLER zeroed the workspace whose cluster sat on ONE XCD
 * and that therefore exchanged with plain stores (the fast path of csrc/cluster_xchg.h); *slow (optional) = those that
 * did not.  The 64-byte tail of a cluster workspace (error word, these two counters) is sticky: launches clear the
 * granules only, so allocate the workspace zero-filled. */
int satt_lstm_cluster_fastpath(const void* ws, int B, int H, int C, void* stream, int* count, int* slow);
/* SATT_OK if the cluster LSTM kernels accept (B, T, H) with C workgroups per sample (host-only check, no launch) */
int satt_lstm_cluster_check(int B, int T, int H, int C);
/* Resident footprint of a cluster LSTM launch on the CURRENT device (r5): *workgroups = B*C, *per_cu = workgroups of the kernel one CU
 * can hold (hipOccupancyMaxActiveBlocksPerMultiprocessor), *cus = CU count.  The launchers refuse workgroups > per_cu * cus (every
 * member of a cluster spins for its peers: all must be resident); callers that keep several cluster launches in flight size their
 * schedule from these numbers.  SATT_E_LAUNCH without a device. */
int satt_lstm_cluster_residency(int B, int T, int H, int C, int backward, int* workgroups, int* per_cu, int* cus);

/* ---- dual-source attention RNN loop (DualSourceAttentionRNN: AttentionWrapper over ZoneoutLSTMCell with
 * ForwardAttention + BahdanauAttention; modules/module.py:1011-1042,1516-1524, modules/forward_attention.py:88-136,
 * modules/attentions.py:53-57).  One persistent workgroup per sample walks all Td steps.                      */
typedef struct {
  int B, Td, Ti;
  int A;        /* attention LSTM units (256) */
  int U1, V1;   /* forward-attention units (224) / memory-1 depth (256) */
  int U2, V2;   /* additive-attention units (32) / memory-2 depth (32) */
  int kernel, filters; /* location conv (10, 5) */
  int training;
  int keys_lds_bf16;                   /* 1: stage keys1/keys2 once per launch in LDS as bf16 (fast path);
                                          0: read fp32 keys from global every step (exact parity mode) */
  float zc, zh; uint32_t zc_thresh, zh_thresh; const uint32_t* seed; uint32_t stream_c, stream_h;
  const int64_t* lengths;              /* [B] */
  const float* xg;                     /* [B,Td,4A] prenet(x_t) W_x + b */
  const uint16_t* Wrec;                /* bf16 [(V1+V2)+A, 4A]   rows: ctx1 | ctx2 | h */
  const uint16_t* Wq;                  /* bf16 [A, U1+U2]        columns: Wq1 | Wq2 */
  const float* keys1; const float* values1;   /* [B,Ti,U1] [B,Ti,V1] */
  const float* keys2; const float* values2;   /* [B,Ti,U2] [B,Ti,V2] */
  const float* locF; const float* locFb;      /* [kernel,filters] [filters] */
  const float* locU;                          /* [filters,U1] */
  const float* v1; const float* b1;           /* [U1] [U1] */
  const float* v2;                            /* [U2] */
  /* outputs */
  float* out;      /* [B,Td,A+V1+V2] = h' | ctx1 | ctx2 */
  float* align1;   /* [B,Td,Ti] normalised forward variable alpha (alignment_history[0]) */
  float* align2;   /* [B,Td,Ti] */
  /* saved for backward */
  float* a1;       /* [B,Td,Ti] softmax probabilities of mechanism 1 (next step's location-conv input) */
  float* pq;       /* [B,Td,U1+U2] processed queries */
  float* fl;       /* [B,Td,Ti,filters] location features conv(a_{t-1})+bias */
  float* gates; float* cnew; float* cstate; float* hstate;   /* [B,Td,4A] [B,Td,A] x3 */
  /* forced-alignment mode (use_forced_alignment_mode; modules/teacher_forcing_attention.py:13-78, models/models.py:411-
   * 428): when both are non-NULL the mechanisms return these alignments [B,Td,Ti] instead of their own (contexts, the
   * recorded alignment histories and everything downstream follow them).  Forward of the cluster kernels only. */
  const float* teach1; const float* teach2;
  /* first-source mechanism options (cluster kernels only; the single-workgroup kernels return SATT_E_UNSUPPORTED):
   * att1_mode 0 = ForwardAttention (alpha recursion, modules/forward_attention.py:104-110), 1 = location_sensitive (same
   * score :13-26, the returned alignments are the softmax probabilities; modules/attentions.py:35-42);
   * cumulative != 0: the location convolution sees the running sum of the softmax alignments (:118-119), which the
   * forward pass then saves in acum [B,Td,Ti] (its value AFTER step t = the conv input of step t+1; required). */
  int att1_mode, cumulative; float* acum;
  /* transition agent of the forward attention (use_forward_attention_transition_agent; modules/forward_attention.py:80-86,
   * 111-116; att1_mode 0, cluster kernels only): agentW != NULL -> the transition probability of step t+1 is
   * u = sigmoid([ctx1_t | processed query 1_t] . agentW + agentb[0]) instead of the constant 0.5 (u of step 0 = 0.5);
   * agentW [V1+U1], agentb [1]; ustate [B,Td] receives the u USED at step t (entries t >= 1; saved for backward). */
  const float* agentW; const float* agentb; float* ustate;
  /* saf (optional, cluster kernels, benchmark precision): fp16 [B,Td,Ti,U1+U2], s = r - 1/2 of the energy nonlinearity
   * (r = 1 / (1 + 2^(c x)): tanh x = -2 s, 1 - tanh^2 = 4 (1/4 - s^2)), written by the folded forward kernel for the rows below
   * the source length.  A backward launch and satt_attn_param_grads* that find it non-NULL read it instead of recomputing
   * keys + query + location term -> exp2 -> rcp per (step, row, unit) - the largest phase of the backward step.  fp16 rounding
   * (absolute 2.4e-4 on 1/4 - s^2) is the only difference; exact-fp32 mode leaves it NULL. */
  void* saf;
} satt_attn_rnn_params;
int satt_attn_rnn_fwd(const satt_attn_rnn_params* p, void* stream);

typedef struct {
  satt_attn_rnn_params f;              /* forward tensors (inputs, outputs and saved) */
  const uint16_t* WrecT;               /* bf16 [4A, (V1+V2)+A] */
  const uint16_t* WqT;                 /* bf16 [U1+U2, A] */
  const float* dout;                   /* [B,Td,A+V1+V2] gradient wrt `out` */
  const float* dalign1; const float* dalign2;  /* optional [B,Td,Ti] extra gradient on alignments (may be NULL) */
  float* dxg;                          /* [B,Td,4A] gate pre-activation gradients */
  float* dctx;                         /* [B,Td,V1+V2] total gradient on the context vectors (for dvalues) */
  float* dpq;                          /* [B,Td,U1+U2] */
  float* de1; float* de2;              /* [B,Td,Ti] energy gradients (consumed by satt_attn_param_grads) */
  float* dfl;                          /* [B,Td,Ti,filters] gradient wrt the location features */
  float* dz;                           /* transition agent (f.agentW != NULL): [B,Td] gradient wrt the agent's pre-activation of
                                          step t (d agentW = sum dz [ctx1 | pq1], d agentb = sum dz: the caller's GEMMs) */
} satt_attn_rnn_bwd_params;
/* BPTT through the loop.  Only the RECURRENT gradient flow runs in the serial loop; gradients that are plain sums
 * over steps are produced afterwards by satt_attn_param_grads (massively parallel) and batched GEMMs. */
int satt_attn_rnn_bwd(const satt_attn_rnn_bwd_params* p, void* stream);

/* Post-loop, massively parallel: dkeys1[b,t',d] = sum_t de1[b,t,t'] v1[d] (1-tanh^2(z)), z = keys1+pq1+b1+fl.U
 * (and dkeys2 likewise), plus the parameter gradients dv1, db1, dlocU, dv2 (ACCUMULATED atomically).
 * One workgroup per (sample, group of memory rows); U1+U2 <= 1024 threads, thread = attention unit. */
int satt_attn_param_grads(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1,
                          float* dkeys2, float* dv1, float* db1, float* dlocU, float* dv2, void* stream);
/* same for the steps [t0, t1) only; accumulate != 0 adds into dkeys1/2.  lds_pad_bytes of (unused) dynamic LDS keep the
 * workgroups off CUs that host the persistent recurrent kernels when the call overlaps them on another stream. */
int satt_attn_param_grads_range(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1,
                                float* dkeys2, float* dv1, float* db1, float* dlocU, float* dv2, int t0, int t1,
                                int accumulate, int lds_pad_bytes, void* stream);
/* The same with float64 accumulators for the parameter sums (saved-factor path only, f->saf != NULL): the sums over
 * (sample, step, memory row) cancel heavily - softmax gradients sum to zero over the rows - and fp32 atomics in arrival order left
 * run-to-run noise of 10 % on a near-zero d U.  acc: satt_attn_param_grads_acc_doubles(f) doubles = one slot [dv1 | db1 | dU | dv2] per
 * workgroup of the kernel's fixed grid (the count depends on f->B and f->Ti: size the buffer per problem).  A call with accumulate == 0
 * OVERWRITES the slots (no zeroing needed), accumulate != 0 adds to them with plain read-modify-writes (one writer per slot element,
 * calls ordered by the stream); satt_attn_param_grads_finish sums the slots in float64, in a fixed order, and adds the result to the
 * fp32 gradients (one small launch after the last piece; it leaves acc as it is). */
int64_t satt_attn_param_grads_acc_doubles(const satt_attn_rnn_params* f);
int satt_attn_param_grads_acc(const satt_attn_rnn_params* f, const float* de1, const float* de2, float* dkeys1, float* dkeys2,
                              double* acc, int t0, int t1, int accumulate, int lds_pad_bytes, void* stream);
int satt_attn_param_grads_finish(const satt_attn_rnn_params* f, double* acc, float* dv1, float* db1, float* dlocU, float* dv2,
                                 void* stream);

/* Cluster form of the attention RNN loop: C workgroups per sample (grid (B,C), B*C <= 256) split the recurrent
 * weight stream by gate columns and the energy / d-alpha passes by memory rows; results are identical to
 * satt_attn_rnn_fwd/bwd.  WrecP / WrecTP are the per-member bf16 weight slices produced by satt_attn_cluster_pack
 * from the fp32 matrix Wrec [(V1+V2)+A, 4A]; `ws` holds the hand-off granules (satt_attn_cluster_ws_bytes). */
#define SATT_MAX_BOUNDS 16
typedef struct {
  satt_attn_rnn_params f; int C; const uint16_t* WrecP; void* ws; int t0, t1;
  /* chunk signalling of a launch that spans several pipeline chunks (optional, progress == NULL: off): after the step
   * bound[k]-1 is complete and its outputs are visible device-wide, every workgroup (B*C of them) adds 1 to
   * progress[k] (one word per chunk: the samples advance independently, a total would let fast samples stand in for
   * slow ones); a consumer stream waits with hipStreamWaitValue32(stream, progress + k, B*C, GTE) instead of a kernel
   * boundary (one prologue per launch instead of one per chunk).  bound[] ascending, within (t0, t1]; the caller
   * zeroes progress[0..nbound) before the launch. */
  uint32_t* progress; int nbound; int bound[SATT_MAX_BOUNDS];
  /* FOLDED first-source context (optional, NULL: off; see satt_attn_cluster_fold): vw1 [B*Ti, 4A] = values1 x Wrec[ctx1 rows].
   * gates += ctx1 Wc1 is then evaluated as alpha (V1 Wc1) inside the recurrent product, WrecP must be the pack of the rows
   * [ctx2 | h] of Wrec only, and the kernel does not write the ctx1 columns of f.out. */
  const float* vw1;
} satt_attn_cluster_params;
typedef struct {
  satt_attn_rnn_bwd_params b; int C; const uint16_t* WrecTP; void* ws; int t0, t1; float* state;
  /* one launch over several pipeline chunks (optional, ready == NULL: off).  The chunks are processed late-to-early;
   * bound[k] = FIRST (lowest) step of the k-th processed chunk, descending, bound[nbound-1] == t0.
   * ready: *ready >= k+1 once the incoming gradients (b.dout rows) of the k-th chunk exist - written by the producer
   *        stream (hipStreamWriteValue32 after the LSTM1 backward of the chunk); the kernel waits for it in a bounded spin
   *        one step before it first touches the chunk.
   * done:  every workgroup adds 1 to done[k] after finishing chunk k (consumers: hipStreamWaitValue32(done + k, B*C, GTE)).
   * The caller zeroes *ready and done[0..nbound) before the launch. */
  const uint32_t* ready; uint32_t* done; int nbound; int bound[SATT_MAX_BOUNDS];
} satt_attn_cluster_bwd_params;
/* [t0,t1): time chunk of this launch (stream pipelining against LSTM1/LSTM2).  The forward restarts from its own
 * saved tensors of step t0-1; backward chunks run late-to-early and carry their recurrent gradients in `state`
 * (satt_attn_cluster_state_floats floats). */
/* Bytes of the exchange workspace (granules + the sticky 64-byte tail: error word, exchange-path counters).  A function of the
 * DIMENSIONS only - not of keys_lds_bf16 / the precision mode -: for the two specialised dimension sets (LJSpeech / VCTK
 * self-attention Tacotron, baseline Tacotron; C = 4) it is sized for Ti = 160 at least, the fixed granule layout of the folded
 * forward / saved-factor backward kernels.  Allocate it with THIS function, zero-filled; every launch and the status queries
 * locate the tail through the same formula. */
int64_t satt_attn_cluster_ws_bytes(const satt_attn_rnn_params* f, int C);
int64_t satt_attn_cluster_state_floats(const satt_attn_rnn_params* f, int C);
int64_t satt_attn_cluster_pack_elems(int K, int A, int C, int transposed);
int satt_attn_cluster_pack(const float* Wrec, int64_t ld, uint16_t* WrecP, uint16_t* WrecTP, int K, int A, int C,
                           void* stream);
int satt_attn_cluster_fwd(const satt_attn_cluster_params* p, void* stream);
int satt_attn_cluster_fold(const satt_attn_rnn_params* f, int C);
int satt_attn_cluster_bwd(const satt_attn_cluster_bwd_params* p, void* stream);
int satt_attn_cluster_status(const satt_attn_rnn_params* f, int C, const void* ws, void* stream);
/* Resident footprint of the launch satt_attn_cluster_fwd / satt_attn_cluster_bwd would make for these parameters (same kernel
 * selection, same LDS size; see satt_lstm_cluster_residency) */
int satt_attn_cluster_residency(const satt_attn_cluster_params* cp, int* workgroups, int* per_cu, int* cus);
int satt_attn_cluster_bwd_residency(const satt_attn_cluster_bwd_params* cb, int* workgroups, int* per_cu, int* cus);
/* host-synchronous (tests): *count = workgroup-launches on `ws` (since the caller zeroed it) that took the same-XCD
 * plain-store exchange (start-up handshake over HW_REG_XCC_ID succeeded): a multiple of B*C; *slow (optional) = the rest.
 * satt_attn_cluster_status: non-zero if ANY launch on `ws` since then had a hand-off timeout (sticky tail, see above). */
int satt_attn_cluster_fastpath(const satt_attn_rnn_params* f, int C, const void* ws, void* stream, int* count, int* slow);
/* SATT_OK if the cluster kernels accept this problem with C workgroups per sample (host-only check, no launch) */
int satt_attn_cluster_check(const satt_attn_rnn_params* f, int C);

/* ---- losses (tacotron2 spec_loss / binary_loss; call sites models/models.py:467-469) ---------------------
 * The decoder projection writes y[B*Td, r*nm+1] = [mel frames of the step | stop logit]; this kernel reads that
 * layout in place: mel element (b,tm,c) at mel[(b*Td+tm/r)*mel_ld + (tm%r)*nm + c], stop (b,td) at stop[(b*Td+td)*stop_ld].
 * target [B,Tm,nm], spec_mask [B,Tm]; done [B,Td], bin_mask [B,Td]
 * losses[0]=mel_loss, [1]=done_loss, [2]=loss; dmel/dstop (same layouts, may be NULL) = d loss/d mel, d stop; ws >= 4 floats */
int satt_loss_fwd_bwd(const float* mel, int64_t mel_ld, const float* target, const float* spec_mask,
                      const float* stop, int64_t stop_ld, const float* done, const float* bin_mask, int B, int Tm,
                      int nm, int Td, int l2, float* losses, float* dmel, int64_t dmel_ld, float* dstop,
                      int64_t dstop_ld, float* ws, void* stream);
/* The same in two parts for the training step (one launch between the forward and the backward pass instead of a memset and
 * two kernels): satt_loss_mask_sums computes the two mask sums - they depend on the batch only - into ws [8 floats] and resets
 * the accumulators, any time before; satt_loss_fwd_bwd_presummed then writes the gradients and the three loss values in one
 * launch.  ws must stay untouched between the two calls; dmel and dstop are required.  When dmel and dstop are one buffer of
 * rows [d mel (r*nm) | d stop | pad] (dstop == dmel + r*nm, equal leading dimensions > r*nm + 1) the pad columns are
 * zero-filled, so the rows can feed a GEMM that reads whole 8-column groups. */
int satt_loss_mask_sums(const float* spec_mask, const float* bin_mask, int B, int Tm, int Td, float* ws, void* stream);
int satt_loss_fwd_bwd_presummed(const float* mel, int64_t mel_ld, const float* target, const float* spec_mask,
                                const float* stop, int64_t stop_ld, const float* done, const float* bin_mask, int B, int Tm,
                                int nm, int Td, int l2, float* losses, float* dmel, int64_t dmel_ld, float* dstop,
                                int64_t dstop_ld, float* ws, void* stream);

/* ---- optimiser (tf.clip_by_global_norm + tf.train.AdamOptimizer; models/models.py:489-498) ---------------
 * flat fp32 buffers of n elements. state: device float[satt_sumsq_state_floats()] = {grad sumsq, global norm, lr_t,
 * clip * grad_scale, per-block partial sums ...}: the sum of squares is taken in a FIXED order (no atomics), so replicas
 * that hold bit-identical all-reduced gradients apply bit-identical updates;
 * step_dev: device int32 step counter (incremented here, 1-based t used for bias correction);
 * lr schedule models/models.py:594-598 evaluated on device from step: lr = lr0*4000^0.5*min(s*4000^-1.5, s^-0.5)
 * (decay!=0) ; grad_scale multiplies gradients first (1/world_size for data parallel). seed_dev += 1 per call. */
/* concurrency probe (see the single-launch attention backward): a 1-thread kernel on `stream` spins at most
 * max_spins (~1.3 us each) until *flag != 0 and writes 1 (seen) / 0 (timed out) to *out; a second 1-thread kernel on
 * `set_stream` (launched right behind it) sets *flag. out == 1 <=> kernels of the two streams run concurrently.
 * The caller zeroes *flag first and synchronises afterwards. */
int satt_stream_probe(uint32_t* flag, uint32_t* out, unsigned max_spins, void* stream, void* set_stream);
/* diagnostics: leave `pattern` in every LDS word of every CU (LDS survives kernel and process boundaries: a read-before-write of
 * LDS sees the previous workgroup's data - or power-on contents).  `linger`: units of ~3.4 us every workgroup stays on its CU.
 * tools/decode_cold.py --poison-lds runs the cold decode behind a NaN and a finite pattern; SATT_DEBUG_POISON_LDS=<hex> makes the
 * Python host (_lib.py) issue it in front of EVERY kernel launch of the library */
int satt_debug_poison_lds(uint32_t pattern, int linger, void* stream);
/* L2 regularisation term of ExtendedTacotronV1Model (modules/regularizers.py:11-18, models/models.py:109-114): for every
 * (offset, count) pair of table [nseg][2] (device, int64): g[off..] += scale * w[off..]; *reg += scale * sum(w^2) / 2 (the
 * caller zeroes *reg), *total += the same if total != NULL */
int satt_l2_reg(const float* w, float* g, const int64_t* table, int nseg, float scale, float* reg, float* total, void* stream);
int satt_sumsq_state_floats(void);
int satt_sumsq(const float* g, int64_t n, float* state, void* stream);
int satt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float* state, int32_t* step_dev,
                   uint32_t* seed_dev, float lr0, int decay, float step_factor, float b1, float b2, float eps,
                   float clip, float grad_scale, const uint32_t* err0, const uint32_t* err1, const uint32_t* err2,
                   void* stream);
/* err0..2 (optional, device): error words of the step's cluster workspaces (the first word of the 64-byte tail of a
 * satt_*_cluster_ws_bytes workspace).  If any is non-zero the update is skipped on the device: a hand-off timeout leaves
 * garbage gradients, which must never reach the parameters; the host raises at its next satt_*_cluster_status call.
 * The update is also skipped when the gradient's sum of squares (state[0]) is not finite.
 * state layout (satt_sumsq_state_floats() floats): [0] sum of squares, [1] global norm, [2] lr_t, [3] clip * grad scale,
 * [4] "this update was skipped" (rewritten every step; [4..n-2) are satt_sumsq's block partials before that), and two STICKY
 * counters the host clears: [n-2] updates skipped so far, [n-1] those skipped for a non-finite gradient alone (no error word set). */
/* Data-parallel form of that guard (reference train.py:68,74: MirroredStrategy replicas must apply the SAME update): call on
 * the first element of a gradient bucket before its all-reduce.  g[0] = NaN if any error word is set, so the summed gradient
 * is non-finite on every rank and satt_adam_step skips on all of them together. */
int satt_poison_on_error(float* g, const uint32_t* err0, const uint32_t* err1, const uint32_t* err2, void* stream);

/* ---- autoregressive decode step (inference branch: RNNTransformer else-branch modules/module.py:762-778,
 * RNNStateHistoryWrapper / TransformerWrapper / OutputAndStopTokenTransparentWrapper modules/rnn_wrappers.py:47-214,
 * StopTokenBasedInferenceHelper / ValidationHelper modules/helpers.py:58-166; BASELINE config 5).
 * The time index is read from DEVICE memory (`step`), so a captured hipGraph of one decoder step is replayed for every
 * step: no kernel argument changes between steps.  Every per-step tensor is addressed as
 * base + b * batch_stride + (*step) * step_stride (elements); step == NULL reads as 0. */
typedef struct {
  int B, N, nseg;                       /* rows, output features, input segments (1..3) concatenated along K */
  const float* x[3]; int64_t x_bs[3], x_ss[3]; int k[3];
  int64_t x_ps[3];                      /* parity stride: + (*step & 1) * x_ps (double-buffered recurrent state as input) */
  const float* W; const uint16_t* Wb; int64_t ldw;   /* [sum k, N]: fp32 master, or (Wb != NULL) its bf16 plain cast */
  const float* bias;                    /* [N] or NULL */
  int act;                              /* SATT_ACT_NONE / RELU / TANH / SOFTSIGN */
  const float* res; int64_t res_bs, res_ss;          /* optional residual, added AFTER the activation */
  float* y; int64_t y_bs, y_ss;
  const int* step;
  /* LSTM form (lstm_H = H > 0, N = 4H, gates i | j | f | o, forget bias 1; W's columns REGROUPED by the caller as
   * column (u / 8) * 32 + gate * 8 + u % 8 for gate column gate * H + u, bias in the original order): the product is the
   * gate pre-activation of a ZoneoutLSTMCell in inference mode (tacotron2 ZoneoutLSTMCell) whose cell runs in the epilogue.
   * c_state / h_state [2][B][H] are double-buffered by step parity: read [*step & 1], written [(*step & 1) ^ 1] with the
   * interpolating zoneout; y [B, H] receives the cell output BEFORE zoneout. */
  int lstm_H; float* c_state; float* h_state; float zc, zh;
  /* step bookkeeping carried by launches that exist anyway (plain form only; workgroup (0,0)):
   * step_out != NULL: *step_out = *step + step_add at the end - step_out must not be a word any workgroup of this launch
   * reads; stop != NULL: with t = *step >= 1, flag[0] = t the first time sigmoid(stop[b*stop_bs + (t-1)*stop_ss]) >
   * stop_threshold for every b while t - 1 > min_steps (the stop rule of the PREVIOUS step, modules/helpers.py:103-107) */
  int* step_out; int step_add;
  const float* stop; int64_t stop_bs, stop_ss; int* flag; float stop_threshold; int min_steps;
  /* dropout that stays on while synthesising (apply_dropout_on_inference: modules/module.py:564-577 hands the flag to the plain
   * PreNet layers; plain form only): drop_thresh = 0 disables; otherwise element (b, *step, n) is kept iff
   * hash(*drop_seed, drop_stream, (b * drop_T + *step) * N + n) >= drop_thresh (the training kernels' stateless mask over a
   * [B, drop_T, N] activation) and scaled by drop_scale = 1 / (1 - rate); applied after the activation, before `res`. */
  uint32_t drop_thresh; float drop_scale; uint32_t drop_stream; const uint32_t* drop_seed; int drop_T;
} satt_dec_linear_params;
/* y = act([x0 | x1 | x2] W + bias) + res ; sum k <= 1024 */
int satt_dec_linear(const satt_dec_linear_params* p, void* stream);
/* two plain Dense layers in one launch (one workgroup per sample): b->y = act_b((act_a(a->x[0] Wa + bias_a) + res_a) Wb + bias_b)
 * + res_b; the intermediate vector is not written (a->y is ignored).  bf16 weights (Wb) only; a->nseg = b->nseg = 1,
 * b->k[0] = a->N, both N <= 256, a->k[0] <= 256, ldw % 4 == 0 (SATT_E_UNSUPPORTED otherwise: launch the layers one by one);
 * the step bookkeeping fields of both blocks are honoured. */
int satt_dec_linear2(const satt_dec_linear_params* a, const satt_dec_linear_params* b, void* stream);
/* a chain in one launch: npre (1 or 2) short plain Dense layers pre[0] -> pre[1], computed redundantly by every workgroup,
 * in front of the main layer (plain or LSTM form) whose FIRST input segment they produce: main->x[0] is ignored,
 * main->k[0] == pre[npre-1].N, pre[1].k[0] == pre[0].N.  Replaces npre + 1 dependent launches (modules/module.py:1011-1042
 * pre-net -> attention cell; modules/self_attention.py:119-128 output transform -> modules/module.py:1449-1559 projections).
 * bf16 weights (Wb) with ldw % 4 == 0 for every layer, pre-layers: one segment, K, N <= 256 (SATT_E_UNSUPPORTED otherwise:
 * launch the layers one by one).  The intermediate vectors are also written to pre[j].y (may be NULL); the step bookkeeping
 * fields of the pre-layers and of a plain main layer are honoured - every layer reads its own `step` word. */
int satt_dec_linear_chain(const satt_dec_linear_params* pre, int npre, const satt_dec_linear_params* main_layer, void* stream);
typedef struct {
  int B, Td, Ti, U1, V1, U2, V2, kernel, filters;   /* U2 = V2 = 0: single source */
  int att1_mode, cumulative;            /* as satt_attn_rnn_params */
  int A;                                /* attention-RNN units: the query is the cell output h [B,A] (pre-zoneout) */
  const int64_t* lengths;
  const float* hq;                      /* [B,A] query of this step */
  const float* Wq; const uint16_t* Wqb; /* query layers [A, U1+U2] = [Wq1 | Wq2]: fp32, or (Wqb != NULL) bf16 plain cast */
  float* pq_out;                        /* [2][B, U1+U2]: the processed query of step t in buffer t & 1 (optional without the
                                           transition agent) */
  const float* agentW; const float* agentb;   /* transition agent (modules/forward_attention.py:111-116) or NULL: u of step t =
                                           sigmoid([ctx1 | pq1] of step t-1 . agentW[V1+U1] + agentb[0]), u of step 0 = 0.5 */
  const float *keys1, *values1, *keys2, *values2;    /* [B,Ti,U1], [B,Ti,V1], [B,Ti,U2], [B,Ti,V2] */
  const float *locF, *locFb, *locU, *v1, *b1, *v2;
  const float *teach1, *teach2;         /* forced alignments [B,Td,Ti] (modules/teacher_forcing_attention.py:31-38) or NULL */
  float *a_state, *alpha_state;         /* [2][B,Ti] location-conv input / previous forward variable, double-buffered by
                                           step parity (read [*step & 1], written [(*step & 1) ^ 1]); the caller initialises
                                           buffer 0 to 0 and onehot(0) (modules/forward_attention.py:128-136) */
  float *e1, *e2;                       /* [B,Ti] scratch: the energies travel between the two launches of the step */
  float* ctx;                           /* [2][B, V1+V2]: the contexts of step t in buffer t & 1 (the caller zeroes both) */
  float *align1, *align2;               /* [B,Td,Ti] histories, row *step written (align2 may be NULL) */
  const int* step;
} satt_dec_attention_params;
int satt_dec_attention(const satt_dec_attention_params* p, void* stream);

/* ---- persistent decode step (csrc/decode_mega2.hip): ONE launch runs up to `nsteps` whole decoder steps on 32 persistent
 * workgroups that hold every weight in registers and exchange {tag, value} granules, instead of nine dependent launches per step.
 * Same math and the same buffers as the launch-per-layer path above (the caller may switch between the two from one LAUNCH to the
 * next: recurrent state, contexts, location input and forward variable are handed over at the last step of a launch).  Supported
 * (otherwise satt_dec_mega_supported() == 0 and the caller uses satt_dec_linear / satt_dec_attention / satt_dec_self_attn): the
 * dual-source model with a plain two-layer pre-net, no transition agent, no forced alignments, bf16 weight shadows, B <= 2,
 * Ti <= 256, A = D = Ds = 256, one causal self-attention hop of 2 or 4 heads.  Replaces, per step: reference
 * modules/module.py:762-778, modules/rnn_wrappers.py:47-124,188-214, modules/forward_attention.py:88-136,
 * modules/helpers.py:58-166 (mirrors). */
typedef struct {
  int B, Td, Ti, A, D, Ds, heads;          /* Td: rows of the histories (yout has Td + 1 rows per sample) */
  int U1, V1, U2, V2, kernel, filters, att1_mode, cumulative;
  int P0, P1, feed, NO, ldout;            /* pre-net widths, fed-back values per step, output row width (mel | stop), row stride of Wout */
  float zc, zh, stop_threshold; int min_steps;
  /* bf16 weights, row-major [K][N]: pre-nets, attention LSTM / LSTM 1 / LSTM 2 with REGROUPED gate columns (satt_dec_linear's
   * LSTM form), query layer [A][U1+U2], K|V|Q [D][3 Ds], folded output transform [Ds][Ds], mel | stop projection [Ds][ldout] */
  const uint16_t *Wp0, *Wp1, *Wa, *Wq, *W1, *W2, *Wkvq, *Wot, *Wout;
  const float *bp0, *bp1, *ba, *b1l, *b2l, *bkvq, *bot, *bout;
  const float *locF, *locFb, *locU, *v1, *b1, *v2;
  const int64_t* lengths;
  const float *keys1, *values1, *keys2, *values2;
  /* recurrent state, double-buffered by step parity ([2][B][H]: read [*step & 1], written the other) */
  float *ca, *ha, *c1, *h1, *c2, *h2;
  float *a_state, *alpha_state;           /* [2][B][Ti] */
  float* ctx;                             /* [2][B][V1+V2]: written at the last step of a launch only (hand-over) */
  float *yout; const float* tin;          /* [B][Td+1][NO] (row 0 = go frame); teacher-fed inputs [B][Td][feed] or NULL */
  float *align1, *align2;                 /* [B][Td][Ti] */
  float* kvq;                             /* [B][Td][3 Ds] cache */
  float* part;                            /* exchange granules [satt_dec_mega_scratch_floats(B, heads, Ds / heads)]: the caller zeroes
                                             them whenever it resets the step counter (tags are step + 1) */
  /* context tables [B][Ti][4][4 * 256]: values1 W1c1 | values2 W1c2 | values1 Wac1 | values2 Wac2, W?c? = the rows of the
   * (regrouped, bf16-rounded) LSTM 1 / attention LSTM weight that multiply context 1 / context 2, products in fp32: contexts are
   * never formed inside a step.  With `flag` (free running) the kernel leaves its step loop at the step whose stop rule fires
   * (*flag = steps taken; no hand-over of state: the utterance is over) and a launch that finds *flag != 0 returns at once */
  const float* ctab;
  /* folded feedback (r6; all three NULL: off): in free-running steps the fed-back frame is a LINEAR function of the output
   * transform's result vc (y = vc Wout + bout, fed = the last `feed` mel columns of y), so the first pre-net layer is
   * relu(vc Wf + bf) with Wf = Wout[:, NO-1-feed : NO-1] Wp0 ([Ds][P0], products of the bf16 weights in fp32, carried as
   * bf16 hi + lo) and bf = bout[NO-1-feed : NO-1] Wp0 + bp0: the projection leaves the step's dependency chain (one split
   * product and one exchange less per step); the first step of a launch and teacher-fed steps take the unfolded form */
  const uint16_t *Wfh, *Wfl; const float* bfb;
  int* step;                              /* [2]: the step counter words of the launch-per-layer path (both advanced) */
  int* flag;                              /* stop flag (number of steps taken when the stop rule fired) or NULL */
  unsigned int* err;                      /* sticky error word (an exchange timed out): zeroed by the caller once */
  int nsteps;
} satt_dec_mega_params;
int satt_dec_mega_supported(const satt_dec_mega_params* p);
int64_t satt_dec_mega_scratch_floats(int B, int heads, int head_dim);
int satt_dec_mega(const satt_dec_mega_params* p, void* stream);
/* new query row of the causal self-attention over the K|V|Q cache kvq [B,Td,3D] (row *step must hold K|V|Q of the step):
 * out [B,D] = softmax(q K^T * scale over rows 0..*step) V, heads side by side (modules/self_attention.py:45-65) */
int satt_dec_self_attn(const float* kvq, float* out, const int* step, int B, int Td, int D, int heads, float scale,
                       void* stream);


#ifdef __cplusplus
}
#endif
#endif

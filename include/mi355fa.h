/* mi355fa.h -- C ABI of libmi355fa.so: FlashAttention forward + backward for MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of
 * Pearbiossom-M/FlashAttention-from-Scratch-with-Triton.  Each entry point replaces one
 * Triton kernel launch of the reference's Python launchers (file:line are relative to the
 * reference repository):
 *
 *   fa_fwd      <- flash_attention_forward_kernel launch, code/My_FlashAttention_optimized.py:53-59
 *                  (kernel code/_flash_attention_kernel_optimized.py:35-129)
 *   fa_bwd_dq   <- flash_attention_dQ_kernel launch,      code/My_FlashAttention_optimized.py:111-117
 *                  (kernel code/_flash_attention_kernel_optimized.py:165-258; also writes delta)
 *   fa_bwd_dkv  <- flash_attention_dKV_kernel launch,     code/My_FlashAttention_optimized.py:120-126
 *                  (kernel code/_flash_attention_kernel_optimized.py:292-386; reads delta)
 *
 * Contract (same as the reference's launchers, M:14-128):
 *   - every pointer is a DEVICE pointer to a contiguous row-major buffer, 16-byte aligned:
 *       q, o, dout, dq : [B, H, S_q, D]   16-bit (fp16 or bf16)
 *       k, v, dk, dv   : [B, H, S_k, D]   16-bit
 *       lse, delta     : [B, H, S_q]      fp32
 *   - the CALLER allocates every output; the kernels write every element of o, lse, dq,
 *     delta, dk, dv (no zero-init needed, nothing is accumulated into);
 *   - launches are enqueued on `stream` (a hipStream_t; NULL = the default stream) and
 *     never synchronise; the library allocates nothing and keeps no pointers;
 *   - fa_bwd_dkv must be enqueued after fa_bwd_dq on the same stream (it reads delta);
 *   - scale is the softmax scale (the reference always passes 1/sqrt(D));
 *   - causal != 0 applies the top-left aligned mask  key <= query  (K:102);
 *   - any S_q, S_k >= 1 is accepted (tails are masked); D must be 64 or 128.
 *
 * Reading guide: the boundary is SIX functions -- fa_fwd / fa_bwd_dq / fa_bwd_dkv (the reference's three launches as they
 * are) and their general forms fa_fwd_ex / fa_bwd_dq_ex / fa_bwd_dkv_ex at the end of this file (strides, packed
 * variable-length batches, dropout, the bf16 backward workspace, in any combination).  The *_strided, *_varlen and
 * *_dropout groups in between are one-call conveniences over the _ex forms, kept for callers that need one feature.
 *
 * Every function returns 0 on success.  A negative value is an argument error detected
 * before anything is launched, a positive value is the hipError_t of the failed launch;
 * fa_last_error() then returns a thread-local, human-readable description.
 */
#ifndef MI355FA_H_
#define MI355FA_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MI355FA_ABI_VERSION 7

/* dtype codes */
#define MI355FA_FP16 0
#define MI355FA_BF16 1

/* argument errors (negative return values) */
#define MI355FA_ERR_NULL (-1)      /* a required pointer is NULL */
#define MI355FA_ERR_SHAPE (-2)     /* B, H, S_q or S_k < 1, or a slice exceeds 2^31 bytes */
#define MI355FA_ERR_HEAD_DIM (-3)  /* D not in {64, 128} */
#define MI355FA_ERR_DTYPE (-4)     /* dtype not MI355FA_FP16 / MI355FA_BF16 */
#define MI355FA_ERR_ALIGN (-5)     /* a pointer is not 16-byte aligned */
#define MI355FA_ERR_STRIDE (-6)    /* a stride is negative or not a multiple of 8 elements, K and V differ in their
                                      sequence stride, or a strided slice exceeds 2^31 bytes */

int fa_abi_version(void);

/* Thread-local message for the last non-zero return on this thread ("" if none). */
const char* fa_last_error(void);

/* 1 if (D, dtype) has a kernel, else 0. */
int fa_supported(int D, int dtype);

/* O = softmax(Q K^T * scale [+causal]) V ; LSE = logsumexp of the scaled, masked scores. */
int fa_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
           int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
           void* stream);

/* dQ, and delta[b,h,i] = sum_d dO[b,h,i,d] * O[b,h,i,d] (fp32, from the 16-bit O). */
int fa_bwd_dq(const void* q, const void* k, const void* v, const void* o, const void* dout,
              const float* lse, void* dq, float* delta,
              int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
              void* stream);

/* dK and dV; `delta` is the buffer fa_bwd_dq filled.
 * bf16 accuracy note: this plain pair has no workspace, so the dK/dV launch rounds its own scaled operand (K) -- dK / dV
 * then carry a relative error of about 4.5e-4 * max|score * scale * log2 e| (nothing on ordinary activations, 2 % at 45).
 * The DOCUMENTED backward call is fa_bwd_dq_ex + fa_bwd_dkv_ex with mi355fa_opts.q_scaled (below):
 * same cost, dK / dV as accurate as O and dQ at any magnitude.  fp16 is exact either way. */
int fa_bwd_dkv(const void* q, const void* k, const void* v, const void* dout,
               const float* lse, const float* delta, void* dk, void* dv,
               int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
               void* stream);

/* ---- strided tensors ------------------------------------------------------------------------
 * The reference's binding makes every input contiguous first (code/My_FlashAttention_optimized.py:138-140,156 --
 * a 64 MiB copy per tensor at the headline size whenever Q/K/V are transposed views of a fused projection) and
 * allocates its outputs with empty_like.  The *_strided entry points read such views in place and write O / dQ / dK / dV
 * in whatever layout the caller allocated (e.g. the inputs' own, so a [B, S, H, D] model never transposes anything).
 * Each [B, H, S, D] operand gets an array of three ELEMENT strides {batch, head, sequence}; the head-dim stride is 1.
 * NULL = contiguous.  Example: a [B, S, H, D] buffer viewed as [B, H, S, D] has {S*H*D, D, H*D}.  Every stride must be a
 * multiple of 8 elements (16-byte rows) and the sequence stride at least D; K, V must share their sequence stride.
 * INPUTS may have a batch / head stride of 0 (one K/V head expanded over several query heads).  OUTPUTS (o, dq, dk, dv)
 * may not, and their rows must not overlap (not checked beyond the zero strides).  `o_strides` of fa_bwd_dq_strided
 * describes the O tensor fa_fwd* wrote.  lse and delta stay contiguous [B, H, S_q].  Everything else is as above.
 */
int fa_fwd_strided(const void* q, const long long* q_strides, const void* k, const long long* k_strides,
                   const void* v, const long long* v_strides, void* o, const long long* o_strides, float* lse,
                   int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale, void* stream);

int fa_bwd_dq_strided(const void* q, const long long* q_strides, const void* k, const long long* k_strides,
                      const void* v, const long long* v_strides, const void* o, const long long* o_strides,
                      const void* dout, const long long* dout_strides,
                      const float* lse, void* dq, const long long* dq_strides, float* delta,
                      int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale, void* stream);

int fa_bwd_dkv_strided(const void* q, const long long* q_strides, const void* k, const long long* k_strides,
                       const void* v, const long long* v_strides, const void* dout, const long long* dout_strides,
                       const float* lse, const float* delta, void* dk, const long long* dk_strides,
                       void* dv, const long long* dv_strides,
                       int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale, void* stream);

/* ---- Variable-length sequences (the extension the reference names as an exercise,
 * Phase_6.md:119-178: "concatenate the sequences of a batch into one long sequence and record where each one starts").
 *
 * Layout ("thd", packed): q, o, dout, dq : [total_q, H, D];  k, v, dk, dv : [total_k, H, D];  lse, delta : [H, total_q]
 * (fp32).  Sequence b owns the rows [cu_seqlens_q[b], cu_seqlens_q[b+1]) of the q-side tensors and
 * [cu_seqlens_k[b], cu_seqlens_k[b+1]) of the k-side ones; cu_seqlens_* are DEVICE int32 arrays of batch + 1 entries
 * starting at 0 (prefix sums of the lengths; a length of 0 is allowed on either side: a sequence with queries but no keys
 * gets O = 0, LSE = -inf, dQ = 0).  max_seqlen_* (host values, >= the longest sequence; any upper bound will do) size the
 * launch grid.  Rows of the packed outputs that no sequence covers (cu_seqlens[batch] < total) are left untouched.  Attention is computed inside each sequence only; causal != 0 applies each sequence's
 * own top-left aligned mask.  Everything else (ownership, stream, return codes, dQ before dK/dV) is as above. */
int fa_fwd_varlen(const void* q, const void* k, const void* v, void* o, float* lse, const int* cu_seqlens_q,
                  const int* cu_seqlens_k, int batch, int H, int total_q, int total_k, int max_seqlen_q, int max_seqlen_k,
                  int D, int dtype, int causal, float scale, void* stream);
int fa_bwd_dq_varlen(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                     float* delta, const int* cu_seqlens_q, const int* cu_seqlens_k, int batch, int H, int total_q,
                     int total_k, int max_seqlen_q, int max_seqlen_k, int D, int dtype, int causal, float scale,
                     void* stream);
int fa_bwd_dkv_varlen(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                      void* dk, void* dv, const int* cu_seqlens_q, const int* cu_seqlens_k, int batch, int H, int total_q,
                      int total_k, int max_seqlen_q, int max_seqlen_k, int D, int dtype, int causal, float scale,
                      void* stream);

/* ---- Attention dropout (the other extension the reference names as an exercise,
 * Phase_6.md:54-113: "a reproducible, indexable RNG -- Philox -- so that forward and backward use the same mask").
 *
 * Same tensors and contract as fa_fwd / fa_bwd_dq / fa_bwd_dkv (contiguous [B, H, S, D]).  Each attention weight
 * P[b, h, q, k] is kept with probability 1 - p and scaled by 1 / (1 - p), else set to 0; LSE is that of the undropped
 * softmax.  The keep decision is a pure function of (b*H + h, q, k, seed, offset): Philox4x32-10 with key =
 * {seed[31:0], seed[63:32]} and counter = {q >> 2, k >> 2, b*H + h, offset} (offset < 2^32, checked) yields the 16 bytes of the
 * 4 x 4 patch around (q, k) -- byte (k & 3) of output word (q & 3) -- and the weight is kept iff its byte >= round(256 p)
 * (p is quantised to multiples of 1/256; fa_dropout_keep_scale(p) returns the exact 1 / (1 - p) in use).  The three
 * kernels must be given the same (p_drop, seed, offset).  p_drop = 0 runs the plain kernels; 0 < p_drop < 1/512 would
 * quantise to 0 and is rejected (MI355FA_ERR_SHAPE) instead of silently running without dropout. */
float fa_dropout_keep_scale(float p_drop);
int fa_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int S_q, int S_k, int D,
                   int dtype, int causal, float scale, float p_drop, unsigned long long seed, unsigned long long offset,
                   void* stream);
int fa_bwd_dq_dropout(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                      float* delta, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale, float p_drop,
                      unsigned long long seed, unsigned long long offset, void* stream);
int fa_bwd_dkv_dropout(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                       void* dk, void* dv, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
                       float p_drop, unsigned long long seed, unsigned long long offset, void* stream);

/* ---- General form: strides, variable-length batches and dropout in any combination ----------------------------------
 * Every entry point above is this call with some fields of mi355fa_opts set.  Zero-initialise the struct, set
 * `size = sizeof(mi355fa_opts)` (checked: a library built for another layout refuses the call) and fill what applies:
 *   - *_strides: element strides {batch, head, seq} per tensor as for the *_strided functions (NULL = contiguous);
 *     o_strides describes O for both fa_fwd_ex (output) and fa_bwd_dq_ex (input).  Not allowed together with cu_seqlens.
 *   - cu_seqlens_q / cu_seqlens_k (both or neither) + total_q / total_k: packed [total, H, D] tensors as for the *_varlen
 *     functions; then B = number of sequences and S_q / S_k = max_seqlen_q / max_seqlen_k.
 *   - p_drop / seed / offset: attention dropout as for the *_dropout functions.  With cu_seqlens the Philox counter uses
 *     the position INSIDE each sequence and slice index (sequence * H + head), so a packed batch drops exactly the weights
 *     the same sequences would lose in a padded [B, H, S, D] launch with the same (seed, offset).
 *   - q_scaled (backward only, optional): a workspace of Q's size, contiguous [B, H, S_q, D] (packed [total_q, H, D] with
 *     cu_seqlens), given to BOTH fa_bwd_dq_ex and fa_bwd_dkv_ex of one backward pass.  bf16 only (ignored for fp16): the
 *     bf16 kernels fold softmax_scale * log2(e) into a 16-bit operand -- forward and dQ into Q, dK/dV into K, whose
 *     rounding is independent of the one behind LSE, so the dK / dV error grows with the score magnitude (about
 *     4.5e-4 * max|score * scale * log2 e| relative).  With the workspace the dQ launch stores the Q rows it multiplies and
 *     the dK/dV launch reads those instead of Q and leaves K alone: P is recomputed from exactly the operands LSE came
 *     from and dK / dV stay at the accuracy of O and dQ for any magnitude.  NULL = no workspace (the behaviour above).
 * opts == NULL is the plain launch (fa_fwd / fa_bwd_dq / fa_bwd_dkv). */
typedef struct mi355fa_opts {
  unsigned size;
  const long long *q_strides, *k_strides, *v_strides, *o_strides, *dout_strides, *dq_strides, *dk_strides, *dv_strides;
  const int *cu_seqlens_q, *cu_seqlens_k;
  int total_q, total_k;
  float p_drop;
  unsigned long long seed, offset;
  void* q_scaled;
} mi355fa_opts;

int fa_fwd_ex(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int S_q, int S_k, int D,
              int dtype, int causal, float scale, const mi355fa_opts* opts, void* stream);
int fa_bwd_dq_ex(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, void* dq,
                 float* delta, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
                 const mi355fa_opts* opts, void* stream);
int fa_bwd_dkv_ex(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                  void* dk, void* dv, int B, int H, int S_q, int S_k, int D, int dtype, int causal, float scale,
                  const mi355fa_opts* opts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355FA_H_ */

/* C ABI of libdaft_exprt_hip.so: the gfx950 kernels behind the Daft-Exprt acoustic-model forward/backward path.
 *
 * The reference (claussss/ubisoft-laforge-daft-exprt) is pure Python/PyTorch and has no FFI of its own; each entry point
 * below replaces the ATen work of one reference call site (cited per function, paths relative to
 * src/daft_exprt/).  The Python host in ubisoft_laforge_daft_exprt_amd/ binds this file with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch caching allocator); the library never allocates or frees
 *   - activations are fp32, channels-last: a (B, N, C) tensor is [B*N] rows of C floats with a leading dimension (ld*)
 *   - lens[b] (int32, device) = valid rows of batch row b; rows n >= lens[b] are padding
 *   - `stream` is a hipStream_t; all work is enqueued on it, nothing synchronises (graph-capture safe)
 *   - return value: 0 = ok, non-zero = error; dx_last_error() returns the message of the calling thread's last error
 *   - reduced precision: every entry point with a `bf16` / `*_bf16` argument (and the pack functions) also exists with the suffix
 *     `_f16` (e.g. dx_conv_gemm_f16): the same kernel built for IEEE fp16 operands and storage (v_mfma_f32_16x16x32_f16), where each
 *     of those arguments then means "fp16".  BASELINE.json config 2 runs the bf16 build, config 5 the fp16 build.
 *   - dropout: counter-based (seed, element index); the backward entry points regenerate the mask from the same seed.
 *     seed_offset (optional device scalar) is added to the seed(s) on the device: a captured HIP graph draws a new dropout
 *     stream on every replay by bumping that scalar, although the launch arguments are frozen
 */
#ifndef DAFT_EXPRT_HIP_H
#define DAFT_EXPRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ----------------------------------------------------------------------------------------------------- */
int dx_version(void);
const char* dx_last_error(void);
/* HIP-event bracketing of one kernel family (0 conv GEMM, 1 weight-grad GEMM, 2 attention fwd, 3 attention bwd,
 * 4 upsampler fwd, 5 LayerNorm rows); used by bench.py to time the dominant kernel inside the timed region. */
int dx_prof_enable(int kind, int capacity);
int dx_prof_collect(int kind, int* launches, double* total_ms);

/* ---- host-side integer durations: DaftExprt.get_int_durations, model.py:950-973 -> duration_to_integer, extract_features.py:69-125 ----
 * HOST pointers, no GPU work.  dur: (B, L) seconds, already thresholded (values < filter_length / sampling_rate / 2 set to 0);
 * out: (B, L) frames per symbol; totals[b] = sum of row b; status[b]: 0 ok, 1 the phones ran out before the frames did (the reference
 * raises IndexError / ValueError for that utterance), 2 count mismatch (the reference's index assignment raises).  Double-precision
 * arithmetic with int() truncation, operation for operation what the reference's Python does; bit-exact. */
int dx_int_durations(const float* dur, int B, int L, int sampling_rate, int filter_length, int hop_length, int centered,
                     long long* out, long long* totals, int* status);

/* ---- Conv1d / Linear as MFMA GEMM: model.py:57-72 (LinearNorm), :75-94 (ConvNorm1D), :165-186 (MHA in/out proj) ---- */
/* dims[4] = {CoutP_fwd, CinP_fwd, CinP_bwd, CoutP_bwd}: padded sizes of the packed weights (element counts
 * taps*dims[0]*dims[1] and taps*dims[2]*dims[3]); bf16 = 0 packs f32 (exact MFMA), 1 packs bf16 */
int dx_pack_dims(int Cout, int Cin, int bf16, int* dims);
/* W: checkpoint layout (Cout, Cin, taps) fp32 -> fwd pack, bwd pack (input-gradient conv: channels transposed, taps flipped) or NULL.
 * The packs are opaque operands of dx_conv_gemm at the same precision: f32 packs are [taps][CoutP][CinP]; bf16 packs hold the
 * same zero-padded matrices fragment-major (16-row x 32-k blocks in v_mfma_f32_16x16x32_bf16 A-operand order). */
int dx_pack_weights(const float* W, void* fwd, void* bwd, int Cout, int Cin, int taps, int bf16, void* stream);
/* every layer of a model in one launch; descs = device array of n 56-byte records
 * {const float* W; void* fwd; void* bwd; int32 Cout, Cin, taps, CoutP_f, CinP_f, CinP_b, CoutP_b, pad} */
int dx_pack_weights_batched(const void* descs, int n, int bf16, void* stream);
/* the same from a HOST array of the records: the descriptors travel as kernel arguments (chunks of 60 layers) and the grid holds one
 * wave per 16 x 32 block of the whole model -- what an optimiser step is followed by (one launch for the 57 packed layers of DaftExprt) */
int dx_pack_weights_host(const void* descs, int n, int bf16, void* stream);
/* Y[b,n,:] = out_scale * mask( relu_aux>0 ? . : 0 )( post_scale * relu?( bias + conv(X) ) + post_shift )  (+= if accumulate)
 * taps 1 (Linear) or 3 (zero 'same' padding inside each batch row).  NULL disables an epilogue stage.
 * skip_halo >= 0: 128-token tiles that start at or beyond lens[b] + skip_halo are padding nobody reads: zero-filled, not computed
 * (the reference computes the whole padded grid; rows within the halo of a k=3 stack are still computed).
 * rows_exist (optional, device int32 [B]): input rows n >= rows_exist[b] of batch row b do not exist -- they read as the conv's zero
 * 'same' padding although the buffers hold N rows per batch row.  NULL: every batch row has N rows (the reference's padded grid, whose
 * N is the longest utterance: model.py:14-24).  With it, ONE allocation / captured HIP graph of N rows serves every batch whose longest
 * utterance is <= N with unchanged results (rows_exist[b] = that batch's max length), and a batch row can be made to behave as if it
 * were run alone (rows_exist[b] = lens[b]: scripts/synthesize.py:420-448 runs the accent encoder per reference wav). */
int dx_conv_gemm(const void* X, int ldx, const void* Wp, const float* bias, void* Y, int ldy,
                 int B, int N, int Cin, int Cout, int taps, int bf16,
                 int relu, const float* post_scale, const float* post_shift,
                 const void* relu_aux, int ld_aux, int accumulate,
                 const int* lens, int mask_rows, float out_scale, int skip_halo,
                 int x_bf16, int y_bf16, int aux_bf16, const int* rows_exist, void* stream);
/* x_bf16 / y_bf16 / aux_bf16 = 1: that tensor is stored as bf16 (bf16 operand mode only; ld* count elements).  Used for the
 * 1024-wide hidden activations of the conv feed-forward and the prenet, whose HBM traffic otherwise bounds the step. */
/* G (fp32, caller-initialised, the parameter's OWN checkpoint layout (Cout, Cin, taps)) += dY^T * shifted X   (autograd of the conv
 * w.r.t. its weight, train.py:435 loss.backward()).  G may be a zeroed scratch gradient or a live `.grad` / all-reduce bucket view. */
int dx_conv_wgrad(const void* dY, int ldy, const void* X, int ldx, float* G,
                  int B, int N, int Cin, int Cout, int taps, const int* lens, int skip_halo,
                  int bf16, int dy_bf16, int x_bf16, float* dbias, void* stream);
/* dbias (optional, caller-zeroed [Cout]): the bias gradient sum_rows dY is accumulated by the same launch from the staged dY tiles */
/* Up to 32 weight gradients of one kind (same taps, same operand storage, bf16 / fp16 operand mode, Cin % 128 == 0) in ONE launch:
 * the eight k = 3 layers of a 4-block FFT stack fill the chip at a 4-way token split instead of 16-way per layer (a quarter of the fp32
 * atomics); more layers run as further rounds of workgroups of the same launch (their atomic epilogues overlap the next round).  `jobs` is a HOST array of njobs DxWgradJob (read during the call; the descriptors travel as kernel arguments); every field
 * has the meaning of the dx_conv_wgrad argument of the same name. */
typedef struct DxWgradJob {
  const void* dY; const void* X; float* G; float* dbias; const int* lens;
  int ldy, ldx, B, N, Cin, Cout, skip_halo, reserved;
} DxWgradJob;
int dx_conv_wgrad_batched(const void* jobs, int njobs, int taps, int dy_bf16, int x_bf16, void* stream);
/* grad (Cout, Cin, taps) (+)= G[taps][Cout][Cin]   (re-layout helper; dx_conv_wgrad itself now writes the parameter layout) */
int dx_unpack_wgrad(const float* G, float* grad, int Cout, int Cin, int taps, int accumulate, void* stream);
/* Fused conv feed-forward pair (model.py:206-217 PositionWiseConvFF.forward: conv k3 128->F, ReLU, conv k3 F->128) and the
 * input-gradient chain of the same pair, bf16 MFMA operands, ONE launch; the F-wide hidden tensor is consumed from LDS.
 *   Y[b,n,:] (+)= bias_b + conv3(Wb, Hm)[b,n,:],   Hm = mid(bias_a + conv3(Wa, X)),  H <- Hm (bf16, kept for the weight gradients)
 * X [B][N][128] bf16; Wa / Wb: bf16 packs of dx_pack_weights (forward: conv1.fwd, conv2.fwd; backward: conv2.bwd, conv1.bwd);
 * mid = ReLU if relu_mid; aux (bf16 [B][N][F], optional): mid zeroes every position where aux <= 0 (ReLU backward).
 * Token tiles starting at or beyond min(lens[b] + skip_halo, N) are padding nobody reads: zero-filled (H, and Y unless accumulate). */
int dx_ff_pair(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b,
               const void* aux, int ld_aux, void* H, int ldh, float* Y, int ldy,
               int B, int N, int F, int relu_mid, int accumulate, const int* lens, int skip_halo, const int* rows_exist, void* stream);
/* dx_ff_pair (forward) with the FFT block's SECOND LayerNorm folded into its epilogue (model.py:225-233 after :206-217): the output tile
 * is normalised while it is in LDS.  Z = res + dropout(pair output) (fp32 [B][N][128], what dx_ln_fwd leaves in `a`), Yln / mean / rstd /
 * film / seeds as in dx_ln_fwd with C = 128 and halo 0 (rows >= lens[b] are masked); lens is required; skip_halo as in dx_ff_pair (it
 * only decides which whole tiles are computed: the hidden rows of the halo are still written to H for the weight gradients). */
/* H may be NULL in the forward entry points (dx_ff_pair with relu_mid, dx_ff_pair_ln, dx_ff_pair_ln_qkv): the mid activation is then not
 * written (inference: nothing reads it; 2 KB per token saved).
 * hmask (optional; dx_ff_pair_ln / dx_ff_pair_ln_qkv write it, dx_ff_block_bwd reads it INSTEAD of aux for the ReLU gradient): the sign of
 * the hidden activation, one bit per element, as uint32 [B * ceil(N / 126)][F / 128][4][2][64] -- the kernel's own register layout, one
 * coalesced dword per lane (128 B per token instead of re-reading 2 KB of H with 8-byte strided loads in the backward's producer epilogue;
 * model.py:213 ReLU backward).  Forward and backward must use the same tile width (they do: both default to 126-token tiles). */
int dx_ff_pair_ln(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b, void* H, int ldh, float* Z,
                  int B, int N, int F, const int* lens, int skip_halo, const int* rows_exist,
                  const float* res, const float* ln_w, const float* ln_b, const float* film, int ld_film, float* Yln, float* mean, float* rstd,
                  uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, void* hmask, void* stream);
/* dx_ff_pair_ln followed, on the same tile, by the NEXT FFT block's attention in-projection (model.py:163-171, F.multi_head_attention_forward's
 * linear(x, in_proj_weight, in_proj_bias)): QKV (16-bit [B][N][384]) = Yln x Wq^T + bias_q, what dx_conv_gemm(Yln, Wq, bias_q, taps = 1,
 * lens, skip_halo = 0, y 16-bit) writes (rows >= lens[b] of a live tile = the bias; tiles beyond the halo = 0).  Wq: the forward pack of
 * the (384, 128) weight in the mode's 16-bit type. */
int dx_ff_pair_ln_qkv(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b, void* H, int ldh, float* Z,
                      int B, int N, int F, const int* lens, int skip_halo, const int* rows_exist,
                      const float* res, const float* ln_w, const float* ln_b, const float* film, int ld_film, float* Yln, float* mean, float* rstd,
                      uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, const void* Wq, const float* bias_q, void* QKV, void* hmask, void* stream);
/* dx_ff_pair (input-gradient pair, accumulate = 1) with the BACKWARD of the block's first LayerNorm folded into its epilogue: Y holds the
 * residual-branch gradient on entry and dz1 = LayerNorm-backward(Y + pair result) on return; DG (16-bit [B][N][128]) = dropout(dz1), the
 * operand of the out-projection's backward GEMMs; dw / db (caller-initialised [128]) accumulate the affine gradients.  z / mean / rstd /
 * ln_w / ln_b / seed_pre / p_pre: the arguments of dx_ln_bwd with C = 128, no FiLM, halo 0 (model.py:188-191 backward). */
int dx_ff_pair_lnbwd(const void* X, int ldx, const void* Wa, const void* Wb, const void* aux, int ld_aux, void* H, int ldh, float* Y,
                     int B, int N, int F, const int* lens, int skip_halo,
                     const float* z, const float* mean, const float* rstd, const float* ln_w, const float* ln_b, void* DG, float* dw, float* db,
                     uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, void* stream);
/* The conv feed-forward half of an FFT block's backward in ONE launch (model.py:225-233, :206-217, :188-191 backward): the backward of
 * the SECOND LayerNorm as the prologue that produces the pair's input tile, the input-gradient pair, the backward of the FIRST LayerNorm as
 * its epilogue.  dY2 [B][N][128] fp32 = d(loss)/d(block output); z2 / mean2 / rstd2 / ln2_* / film / seed2 / p2 and z1 / ... / seed1 / p1:
 * the arguments of the two dx_ln_bwd calls it replaces (C = 128, halo 0); Wa / Wb / aux / H / lens / skip_halo: those of dx_ff_pair
 * (input-gradient pair).  Outputs: Y = dz1 (fp32, the residual gradient for the attention half), DG1 = dropout(dz1) and DG2 = dropout(dz2)
 * as 16 bits (operands of the out-projection's and the second conv's backward GEMMs), H = the hidden gradient; dw* / db* / dfilm accumulate.
 * Wout_bwd / DATT (optional, together): the backward pack of the attention out-projection's (128, 128) weight and a 16-bit [B][N][128]
 * output: the epilogue then also computes DATT = DG1 x W_out, the input gradient of the out-projection (what dx_attention_bwd takes as dctx). */
int dx_ff_block_bwd(const float* dY2, const float* z2, const float* mean2, const float* rstd2, const float* ln2_w, const float* ln2_b,
                    const float* film, int ld_film, void* DG2, float* dw2, float* db2, float* dfilm, int ld_dfilm, uint64_t seed2, float p2,
                    const void* Wa, const void* Wb, const void* aux, int ld_aux, void* H, int ldh, float* Y,
                    int B, int N, int F, const int* lens, int skip_halo,
                    const float* z1, const float* mean1, const float* rstd1, const float* ln1_w, const float* ln1_b, void* DG1, float* dw1, float* db1,
                    uint64_t seed1, float p1, const void* Wout_bwd, void* DATT, const uint64_t* seed_offset, const void* hmask, void* stream);
/* out[c] += sum_rows X[row][c]   (bias gradients) */
int dx_colsum(const void* X, int ldx, float* out, long rows, int C, int x_bf16, void* stream);

/* ---- multi-head attention: model.py:165-186 (nn.MultiheadAttention slow path), called from :255 ------------------- */
/* order[0..B) = utterance indices sorted by length, longest first (stable): the optional `order` argument of the attention entry
 * points.  A workgroup's work grows with its utterance's length and workgroups start in blockIdx order: longest first removes the tail
 * of a launch.  Pure scheduling: results do not depend on it.  (No reference counterpart: get_mask_from_lengths, model.py:14-24, is
 * the only use the reference makes of the lengths.) */
int dx_length_order(const int* lens, int* order, int B, void* stream);
int dx_attention_fwd(const void* qkv, int ld, const int* lens, void* ctx, int ldc, float* lse,
                     int B, int N, int H, int D, uint64_t seed, const uint64_t* seed_offset, float p_drop, int bf16, int qkv_bf16, int ctx_bf16,
                     const int* order, void* stream);
/* dx_attention_fwd (16-bit q/k/v and context, 2 heads x 64) + dx_proj_ln_fwd on its result in ONE launch, bit-identical outputs: the
 * attention, out-projection, dropout, residual and first LayerNorm of an FFT block (model.py:165-191, forward).  Arguments as in the two
 * entry points (halo 0).  Every entry point with 16-bit operands also exists as <name>_f16. */
int dx_attention_proj_ln_fwd(const void* qkv, int ld, const int* lens, void* ctx, int ldc, float* lse, int B, int N, int H, int D,
                             uint64_t seed, const uint64_t* seed_offset, float p_drop,
                             const void* Wpack, const float* proj_bias, float* z, const float* res, const float* w, const float* bias,
                             const float* film, int ld_film, float* y, float* mean, float* rstd, uint64_t seed_pre, float p_pre,
                             void* y_bf16_copy, void* stream);
int dx_attention_bwd(const void* qkv, int ld, const void* ctx, const void* dctx, int ldc, const float* lse, float* delta,
                     const int* lens, void* dqkv, int ldg, int B, int N, int H, int D, uint64_t seed, const uint64_t* seed_offset, float p_drop, int bf16,
                     int qkv_bf16, int dqkv_bf16, int ctx_bf16, const int* order, void* stream);
/* bf16 = 1: QK^T / PV (and the five backward products) on v_mfma_f32_16x16x32_bf16, softmax and accumulation in fp32;
 * qkv_bf16 / dqkv_bf16 / ctx_bf16 = 1: the in-projection output / its gradient / the attention context are stored as bf16 (ld in
 * elements; ctx and dctx share ldc AND the storage type: ctx_bf16 = 1 means both are 16-bit).  The context is only ever consumed as a 16-bit GEMM operand (out-projection, its weight gradient)
 * and in delta = rowsum(dctx * ctx), so storing it in 16 bits halves four passes over it. */

/* ---- dropout + residual + LayerNorm + FiLM + mask: model.py:188-191, :225-233, :256-258, :655-669 ------------------- */
/* z = drop_pre(a) + res is written back over `a`; y = mask(film(drop_post(LN(z)))); C in {128, 1024}.
 * rows n >= lens[b] + halo are zero-filled and not computed: halo = 0 is the reference's padding mask, halo > 0 serves the
 * unmasked prenet LayerNorms whose rows inside the conv halo are still needed */
int dx_ln_fwd(void* a, const void* res, const float* w, const float* bias, const float* film, int ld_film,
              const int* lens, int halo, void* y, float* mean, float* rstd, int B, int N, int C,
              uint64_t seed_pre, float p_pre, uint64_t seed_post, float p_post, const uint64_t* seed_offset, int io_bf16, void* y_bf16_copy,
              void* stream);
/* FiLM parameters of all FFT blocks in one launch: StyleAdapter.forward, model.py:779-800 (scalar post-multiplier affine, gamma | beta
 * concatenation, per-block split).  gammas / betas: (B, nb*C) predictor outputs; pm: (2, nb) post-multipliers or null (then gamma + 1, beta);
 * film: (nb, B, 2C), block i = the (B, 2C) tensor the LayerNorm kernels of FFT block i read.  The backward takes a HOST array of nb device
 * pointers to the per-block gradients (null entries = no gradient), writes dgammas / dbetas and ACCUMULATES dpm (2, nb). */
int dx_film_affine_fwd(const float* gammas, const float* betas, const float* pm, float* film, int B, int nb, int C, void* stream);
int dx_film_affine_bwd(const void* dfilm_ptrs, const float* gammas, const float* betas, const float* pm, float* dgammas, float* dbetas, float* dpm,
                       int B, int nb, int C, void* stream);
/* Out-projection + dropout + residual + LayerNorm (+ FiLM + mask) of an FFT block in ONE launch (16-bit operand modes):
 *   z = dropout(X W^T + proj_bias) + res;  y = mask(FiLM(LayerNorm(z)))      attn.out_proj (model.py:165-186) + model.py:188-191
 * X: 16-bit [B*N][ldx], 128 columns (the attention context); Wpack: the forward pack (dx_pack_weights) of the (128, 128) weight.
 * z / y / mean / rstd / y_bf16_copy / film / lens / halo / seeds: exactly the arguments of dx_ln_fwd with C = 128 (z = its `a` afterwards).
 * A 64-row tile owns whole output rows, so the projection result feeds the row statistics from LDS and never goes to HBM. */
int dx_proj_ln_fwd(const void* X, int ldx, const void* Wpack, const float* proj_bias, float* z, const float* res, const float* w, const float* bias,
                   const float* film, int ld_film, const int* lens, int halo, float* y, float* mean, float* rstd, int B, int N,
                   uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, void* y_bf16_copy, void* stream);
int dx_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* w, const float* bias,
              const float* film, int ld_film, const int* lens, int halo, void* dz, void* da, float* dw, float* dbias,
              float* dfilm, int ld_dfilm, int B, int N, int C, int relu_mask,
              uint64_t seed_pre, float p_pre, uint64_t seed_post, float p_post, const uint64_t* seed_offset, int io_bf16, void* dg_bf16_copy,
              void* stream);
/* y_bf16_copy / dg_bf16_copy (optional, C = 128): a second, bf16 copy of y / of the gradient that feeds the GEMM backward.  The
 * next GEMM would round its fp32 operand to bf16 while staging anyway, so results are bit-identical and the operand costs half the bytes.
 * io_bf16 = 1 (C = 1024 only): a / res / y and dy / z / dz / da are stored as bf16; statistics and parameter gradients stay fp32 */

/* ---- embeddings, positions, masks, pooling: model.py:119-150, :597-604, :554-557, :687-716 --------------------------- */
int dx_add_pos(const float* x, const long* sym, const float* emb, const float* pe, const int* lens, float* out,
               int B, int N, int D, int pe_rows, void* stream);
int dx_mask_rows(const float* in, const int* lens, float* out, int B, int N, int C, void* stream);
int dx_embedding_bwd(const float* dout, const long* sym, const int* lens, float* demb, int B, int N, int D, void* stream);
int dx_accent_sum(const float* prenet, const float* energy, const float* pitch, const float* we, const float* be,
                  const float* wp, const float* bp, const float* pe, const int* lens, float* out,
                  int B, int N, int D, int pe_rows, void* stream);
int dx_scalar_conv_wgrad(const float* dout, int ldd, const float* rowscale, const float* s0, const float* s1, const int* lens,
                         float* dw0, float* db0, float* dw1, float* db1, int B, int N, int D, void* stream);
int dx_mean_pool(const float* x, const int* lens, float* out, int B, int N, int C, void* stream);
int dx_mean_pool_bwd(const float* dout, const int* lens, float* dx, int B, int N, int C, void* stream);
int dx_transpose(const float* in, float* out, int B, int R, int Cc, int accumulate, void* stream);   /* accumulate: out += in^T */
int dx_relu_bwd(const float* dy, const float* y, float* out, long n, void* stream);
int dx_channel_affine(const void* x, const float* scale, const float* shift, void* out, long rows, int C, int io_bf16, void* stream); /* layers/pitch_predictor.py:49-62 (BatchNorm1d, eval); io_bf16: x / out stored in the mode's 16-bit type */
int dx_l2_normalize(const float* x, float* y, int rows, int C, void* stream);            /* model.py:904 */
int dx_cross_entropy(const float* logits, const long* target, float* loss, float* dlogits, int B, int S, void* stream); /* loss.py:85 */

/* ---- Gaussian upsampling: model.py:417-510 ---------------------------------------------------------------------------- */
int dx_duration_scan(const long* dur_int, float* mu, long* totals, int B, int L, void* stream);
int dx_upsample_prep(const float* enc, const float* dur, const float* energy, const float* pitch,
                     const float* wd, const float* bd, const float* we, const float* be, const float* wp, const float* bp,
                     const float* wr, const float* br, const int* lens, float* xs, float* z, float* sigma,
                     int B, int L, int D, void* stream);
int dx_upsample_fwd(const float* xs, const float* mu, const float* sigma, const int* lens, float* weights, float* xup,
                    int B, int L, int T, int D, void* stream);
int dx_upsample_bwd(const float* dxup, const float* xs, const float* mu, const float* sigma, const float* weights, const int* lens,
                    float* dxs, float* dsigma, int B, int L, int T, int D, void* stream);
int dx_upsample_sym_bwd(const float* dxs_in, const float* dsigma, const float* xs, const float* z, const float* dur, const int* lens,
                        const float* wd, const float* bd, const float* wr, float* dxs_out, float* dz, float* dwr, float* dbr,
                        int B, int L, int D, void* stream);

/* ---- loss reductions and gradients: loss.py:99-146 ------------------------------------------------------------------- */
int dx_mel_stats(const float* mel_pred, const float* mel_target, float* ep, float* et, float* l1sum, float* l2sum,
                 int B, int M, int T, void* stream);
int dx_energy_diff(const float* ep, const float* et, const int* lens, float* des, float* esum, int B, int T, void* stream);
/* e_per_total = 1: c_e is divided by sum_b lens[b] on the device (loss.py:129 normalises the energy term by the batch's valid frames) */
int dx_mel_grad(const float* mel_pred, const float* mel_target, const float* ep, const float* des, const int* lens,
                float c_l1, float c_l2, float c_e, int e_per_total, float* dmel, int B, int M, int T, void* stream);
/* loss.py:85-157 assembled on the device: terms[7] = {speaker_loss, speaker_ce_raw, post_mult_loss, mel_l1, mel_l2, energy, pitch},
 * total[1] = speaker + post_mult + l1 + l2 + ecw * energy + pcw * pitch; d_spk = dlogits * w; d_pm = pmw * pm / ||pm||_2.
 * w = *spk_w_dev if given (device scalar, re-read by every replay of a captured graph) else spk_w.  NULL ce / pm / esum / psum: term off.
 * grad_scale multiplies d_spk and d_pm (not the terms): the factor d(total)/d(what the caller differentiates), e.g. loss scale / accumulation steps. */
int dx_loss_finalize(const float* ce, const float* spk_w_dev, float spk_w, const float* dlogits, float* d_spk, int n_logits,
                     const float* pm, float* d_pm, int n_pm, float pmw,
                     const float* l1sum, const float* l2sum, const int* lens, int B, int M, float msw,
                     const float* esum, float ecw, const float* psum, float pcw, float* terms, float* total, float grad_scale, void* stream);
int dx_pitch_mse(const float* pp, int ldp, const float* gt, const int* lens, float* sums, int B, int T, void* stream);   /* ldp / ldd: element stride between consecutive frames of pp / dpp (the predictor's last conv writes 4-wide rows, channel 0 is the prediction) */
int dx_pitch_grad(const float* pp, int ldp, const float* gt, const int* lens, const float* sums, float scale, float* dpp, int ldd, int B, int T, void* stream);
/* The frozen pitch predictor of the pitch-consistency term, one launch per direction (layers/pitch_predictor.py:38-74 applied in loss.py:131-140):
 * mel (B, M = 80, T) fp32 -> pp (B, T), and its input-gradient chain dpp (B, T) -> dmel (B, M, T) +=.  w0..w2: the 16-bit packs of the three
 * 256-wide k = 3 convolutions written by dx_pack_weights (fwd takes the forward packs, bwd the backward packs); b*: biases; s* / t*: eval-mode
 * BatchNorm folded to a scale / shift per channel; w3: row 0 of the last convolution's weight in checkpoint layout (256, 3) fp32, b3 its bias;
 * masks: (B, T, 3, 8) uint32 written by fwd (ReLU sign bits: the network is frozen, so the backward needs no activations), read by bwd.
 * Only tokens n < lens[b] are produced (the loss and the model mask the rest).  Every entry point with 16-bit packs also exists as <name>_f16. */
int dx_pitch_chain_fwd(const float* mel, int B, int M, int T, const int* lens, const void* w0, const void* w1, const void* w2,
                       const float* b0, const float* b1, const float* b2, const float* s0, const float* s1, const float* s2,
                       const float* t0, const float* t1, const float* t2, const float* w3, float b3, float* pp, void* masks, void* stream);
int dx_pitch_chain_bwd(const float* dpp, int B, int M, int T, const int* lens, const void* w0, const void* w1, const void* w2,
                       const float* s0, const float* s1, const float* s2, const float* w3, const void* masks, float* dmel, void* stream);

/* ---- on-device batch conditioning: dynamic_stats.py:131-195 ------------------------------------------------ */
int dx_condition_prosody(const float* in, float* out, const long* speaker_ids, const float* table, const int* valid,
                         int which, int B, int N, int S, void* stream);
int dx_gather_speaker_rows(const float* emb, const long* speaker_ids, const int* valid, float* out, int B, int E, int S, void* stream);

/* ---- fused optimiser step: train.py:278-280 (Adam), :443 (clip_grad_norm_) ------------------------------ */
int dx_sumsq(const float* x, long n, float* out, void* stream);
/* skipped (optional device int): a non-finite *normsq (an overflowed fp16-mode gradient) skips the whole update -- p, m, v untouched --
   and increments *skipped; the reference has no loss scaling and therefore no counterpart (train.py:443-450 steps on NaN).
   norm_out (optional): receives sqrt(*normsq) * grad_scale = the value clip_grad_norm_ returns (train.py:443).
   zero_after (optional, != normsq): one float this launch sets to zero (the caller's other squared-norm accumulator). */
int dx_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, const float* normsq, float max_norm, float grad_scale, int* skipped,
                 float* norm_out, float* zero_after, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DAFT_EXPRT_HIP_H */

/* C-ABI of libvisualrwkv_hip.so -- the MI355X (gfx950) drop-in for the native WKV7 operator of
 * howard-hou/VisualRWKV (v7.xx).
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream (as void*, NULL = the
 * legacy default stream) and returns 0 on success, a positive hipError_t value when the HIP
 * runtime reported an error, or a negative VRWKV_E* code for bad arguments.  No torch types, no
 * allocation, no retained state: the caller owns every buffer (as in the reference, where
 * WindBackstepping allocates y/s/sa and the six gradients, src/model.py:52-54,63).
 *
 * Reference interface each function replaces (paths relative to VisualRWKV-v7/v7.00/):
 *   vrwkv_wkv7_forward_bf16   <- cuda_forward(B,T,H,w,q,k,v,z,a,y,s,sa)        cuda/wkv7_op.cpp:5, cuda/wkv7_cuda.cu:132-134
 *   vrwkv_wkv7_backward_bf16  <- cuda_backward(B,T,H,w,q,k,v,z,a,dy,s,sa,d*)   cuda/wkv7_op.cpp:12, cuda/wkv7_cuda.cu:135-138
 * which sit under torch.ops.wind_backstepping.{forward,backward} (cuda/wkv7_op.cpp:21-29).
 *
 * Layouts: activations (B,T,H,64) contiguous bf16; `s` (B,H,T/16,64,64) f32 holding
 * S^T at the end of every 16-token chunk (s[b,h,c,j,i] = S[i][j]); `sa` (B,T,H,64) f32.
 * T must be a multiple of 16; the head size is 64.  Argument names follow the op schema:
 * z = the kernel's `a` (= -kk), a = the kernel's `b` (= kk * gate).
 */
#ifndef VISUALRWKV_HIP_H
#define VISUALRWKV_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define VRWKV_OK 0
#define VRWKV_EINVAL (-1)   /* null pointer, non-positive size                     */
#define VRWKV_ESHAPE (-2)   /* T % 16 != 0                                         */
#define VRWKV_EALIGN (-3)   /* a pointer is not 16-byte aligned                    */

#define VRWKV_HEAD_SIZE 64
#define VRWKV_CHUNK_LEN 16

/* ABI version of this header; bumped on any signature change. */
int vrwkv_abi_version(void);

/* Human-readable text for a return code of this library (static storage). */
const char* vrwkv_strerror(int code);

int vrwkv_wkv7_forward_bf16(int B, int T, int H,
                            const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a,
                            void* y, float* s, float* sa, void* stream);

int vrwkv_wkv7_backward_bf16(int B, int T, int H,
                             const void* w, const void* q, const void* k, const void* v,
                             const void* z, const void* a, const void* dy,
                             const float* s, const float* sa,
                             void* dw, void* dq, void* dk, void* dv, void* dz, void* da,
                             void* stream);

/* The same operator on the host cores -- what the `CPU` dispatch key of torch.ops.wind_backstepping runs (BASELINE config 1:
 * fp32 CPU WKV path; cuda/wkv7_op.cpp:26 registers the CUDA key only).  Host pointers; dtype 0 = bf16 activations (the
 * op's contract), 1 = float32 activations (RWKV_FLOAT_MODE=fp32); s / sa as above, always f32.  One task per (b, head)
 * on n_threads host threads (<= 0: all hardware threads).  csrc/wkv7_host.hip. */
int vrwkv_wkv7_forward_host(int B, int T, int H, int dtype, const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a, void* y, float* s, float* sa, int n_threads);
int vrwkv_wkv7_backward_host(int B, int T, int H, int dtype, const void* w, const void* q, const void* k, const void* v,
                             const void* z, const void* a, const void* dy, const float* s, const float* sa,
                             void* dw, void* dq, void* dk, void* dv, void* dz, void* da, int n_threads);

/* Sequence-parallel WKV7 backward (no counterpart in the reference): nseg workgroups per head, each
 * walks a contiguous range of 16-token chunks of the same tensors as vrwkv_wkv7_backward_bf16.  ds_in (B,H,nseg,64,64)
 * f32, [i][j]: dL/dS at the END of every range (NULL = zeros); ds_out, same shape: dL/dS at the START of the range.
 * The recurrence is linear in dS -- dS_start = dS_end M^T + C with M the forward map of the range (S_end = S_start M + ..,
 * obtained from vrwkv_wkv7_forward_state_bf16 with s0 = I, v = 0) -- so: run with ds_in = NULL to get C, scan the
 * ranges, run again with the true ds_in for the gradients (visualrwkv_amd/wkv7.py::wkv7_backward_tparallel).
 * 1 <= nseg <= T/16. */
int vrwkv_wkv7_backward_segments_bf16(int B, int T, int H, int nseg, const void* w, const void* q, const void* k,
                                      const void* v, const void* z, const void* a, const void* dy, const float* s,
                                      const float* sa, const float* ds_in, float* ds_out,
                                      void* dw, void* dq, void* dk, void* dv, void* dz, void* da, void* stream);

/* WKV6 (BASELINE config 4): replaces cuda_forward / cuda_backward of VisualRWKV-v6/v6.0/cuda/wkv6_cuda.cu:229-242 as bound
 * by cuda/wkv6_op.cpp:8-13 (forward(B,T,C,H,r,k,v,w,u,y), backward(B,T,C,H,r,k,v,w,u,gy,gr,gk,gv,gw,gu)).
 * r,k,v,y,gy,gr,gk,gv,gw: (B,T,C) bf16; ew: (B,T,C) f32 = -exp(w_raw) as WKV_6.forward computes it (src/model.py:62);
 * u: (C) bf16; gu: (B,C) bf16, summed over the batch by the caller (src/model.py:84); C == 64*H; any T >= 1.
 * s_ckpt: vrwkv_wkv6_ckpt_floats(B,T,H) floats of chunk-start states -- optional output of the forward (NULL for
 * inference), required input of the backward (the reference re-sweeps the sequence five times instead). */
long vrwkv_wkv6_ckpt_floats(int B, int T, int H);
int vrwkv_wkv6_forward_bf16(int B, int T, int C, int H, const void* r, const void* k, const void* v, const float* ew,
                            const void* u, void* y, float* s_ckpt, void* stream);
int vrwkv_wkv6_backward_bf16(int B, int T, int C, int H, const void* r, const void* k, const void* v, const float* ew,
                             const void* u, const void* gy, const float* s_ckpt, void* gr, void* gk, void* gv, void* gw,
                             void* gu, void* stream);
/* Kernel generation of the WKV6 backward (same-box A/B in benchmarks): -1 = default (2), 2 = three-role pipeline of 12 waves
 * (csrc/wkv6_bwd_v2.h), 1 = the four-wave kernel of rounds 1-3 (csrc/wkv6_chunked.h).  Anything else: VRWKV_EINVAL. */
int vrwkv_wkv6_set_backward_variant(int variant);

/* Residual add + LayerNorm in one pass (Block.forward: x = x + att(ln1(x)); x = x + ffn(ln2(x)), and ln_out --
 * VisualRWKV-v7/v7.00/src/model.py:247-254,318; the reference runs a bf16 add followed by nn.LayerNorm).
 *   fwd: xn = bf16(x + delta) (delta may be NULL: no add, xn not written), y = LayerNorm(xn; w, b, eps); mean/rstd
 *        (ntok fp32 each) are kept for the backward.
 *   bwd: dx = dres (may be NULL) + LayerNorm-backward(dy); dwb = [dw | db] (2*C fp32, overwritten);
 *        ws: vrwkv_add_ln_ws_floats(ntok, C) floats of scratch.
 * x, delta, xn, y, dy, dres, dx: (ntok, C) bf16; w, b: (C) bf16; C % 64 == 0, C <= 8192. */
long vrwkv_add_ln_ws_floats(long ntok, int C);
int vrwkv_add_ln_fwd_bf16(long ntok, int C, float eps, const void* x, const void* delta, const void* w, const void* b,
                          void* xn, void* y, float* mean, float* rstd, void* stream);
int vrwkv_add_ln_bwd_bf16(long ntok, int C, const void* dy, const void* dres, const void* xn, const float* mean,
                          const float* rstd, const void* w, void* dx, float* dwb, float* ws, void* stream);

/* Residual add + LayerNorm + token shift + M lerps in one pass (M = 6: x = x + att(ln1(x)) up to the six inputs of
 * RWKV_Tmix_x070, VisualRWKV-v7/v7.00/src/model.py:247-254,166-173;  M = 1: ln2 + RWKV_CMix_x070's x_k lerp, :222-223).
 * xn = bf16(x + delta) (delta NULL: xn is not written), y = bf16(LayerNorm(xn)) is formed in registers only,
 * out[j][n] = y[n] + (y[n-1] - y[n]) mu[j]  with y[-1] = 0 at the first token of every sample (T tokens per sample);
 * mean / rstd (ntok) fp32 are kept for the backward.  Bit-identical to vrwkv_add_ln_fwd_bf16 followed by vrwkv_mix_fwd_bf16. */
long vrwkv_ln_mix_ws_floats(long ntok, int C, int M);
int vrwkv_ln_mix_fwd_bf16(long ntok, int T, int C, float eps, int M, const void* x, const void* delta, const void* w, const void* b,
                          const void* const* mu, void* xn, void* const* out, float* mean, float* rstd, void* stream);
/* backward of the above for M = 1: dx (ntok, C) bf16 = dres + LN'(d y), dwb (2, C) fp32 = (dgamma, dbeta), dmu (M, C) fp32;
 * dout3_second must be NULL, dres: gradient of xn from the residual path (NULL: none).  M = 6 (VRWKV_ESHAPE here): the six lerps'
 * backward with y recomputed from xn and the statistics, vrwkv_mix_bwd_ln_bf16, then vrwkv_add_ln_bwd_bf16 on its dx */
int vrwkv_ln_mix_bwd_bf16(long ntok, int T, int C, int M, const void* xn, const float* mean, const float* rstd, const void* w,
                          const void* b, const void* const* mu, const void* const* dout, const void* dout3_second, const void* dres,
                          void* dx, float* dwb, float* dmu, float* ws, void* stream);
/* M = 6: as vrwkv_mix_bwd2_bf16 (dx = gradient of the lerps' input, dmu (M, C) fp32; ws: vrwkv_param_grad_ws_floats(ntok, C, M)) with the
 * input y = bf16(LayerNorm(xn)) recomputed from xn, mean, rstd, ln_w, ln_b instead of read */
int vrwkv_mix_bwd_ln_bf16(long ntok, int T, int C, int M, const void* xn, const float* mean, const float* rstd, const void* ln_w,
                          const void* ln_b, const void* const* mu, const void* const* dout, const void* dout3_second, void* dx,
                          float* dmu, float* ws, void* stream);
/* Inference form for the frozen ViT towers: xn = x + delta * dscale (dscale = LayerScale gamma (C) bf16 or NULL; delta NULL: no
 * add, xn not written), y = LayerNorm(xn); no statistics kept (timm blocks via src/vision.py:123-134, src/sam.py:231-247). */
int vrwkv_add_ln_scaled_fwd_bf16(long ntok, int C, float eps, const void* x, const void* delta, const void* dscale, const void* w,
                                 const void* b, void* xn, void* y, void* stream);

/* Cross-entropy over the vocabulary + L2Wrap gradient (training_step / L2Wrap, VisualRWKV-v7/v7.00/src/model.py:418-434,
 * 257-271), one pass over the (nrows, V) bf16 logits per direction.  labels: int64 per row, the already SHIFTED target
 * (-100 = ignored).  fwd writes per row: loss (0 for ignored rows), max, log-sum-exp, first arg-max.
 * bwd: dlogits = row_w (softmax - onehot(label)) + onehot(argmax) * max * l2_factor;  row_w carries the upstream
 * gradient and the per-sample normalisation and is 0 for ignored rows.  V % 8 == 0. */
int vrwkv_ce_fwd_bf16(long nrows, int V, const void* logits, const long* labels, float* row_loss, float* row_max,
                      float* row_lse, int* row_argmax, void* stream);
int vrwkv_ce_bwd_bf16(long nrows, int V, const void* logits, const long* labels, const float* row_w, const float* row_max,
                      const float* row_lse, const int* row_argmax, float l2_factor, void* dlogits, void* stream);

/* WKV7 forward from / to an explicit state, for inference and stateful prefill (no counterpart in the reference, whose
 * forward always starts from S = 0 and always writes its training checkpoints -- cuda/wkv7_cuda.cu:15,44-50).
 * s0: optional (B,H,64,64) f32 initial state S[i][j] (i = value row, j = key column; NULL = zeros);
 * s_final: optional output, same layout; s_ckpt / sa: the training outputs of vrwkv_wkv7_forward_bf16, optional here
 * (22 of the forward's 24 written bytes per element are these two).  T % 16 == 0. */
int vrwkv_wkv7_forward_state_bf16(int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                                  const void* z, const void* a, void* y, const float* s0, float* s_final, float* s_ckpt,
                                  float* sa, void* stream);

/* Batched GEMV for single-token decode: y_j = act_j(W_j x_j) (+ res_j) for n_jobs <= 8 independent products in one
 * launch (at T = 1 every nn.Linear / LoRA product of RWKV_Tmix_x070 / RWKV_CMix_x070.forward, src/model.py:175-194,222-225,
 * is a GEMV).  W_j: (N_j,K_j) bf16 row-major, x_j: (B,K_j), res_j: (B,N_j) or NULL, y_j: (B,N_j); B <= 4, K_j % 8 == 0;
 * act: 0 none, 1 tanh, 2 sigmoid, 3 relu^2. */
int vrwkv_gemv_multi_bf16(int n_jobs, int B, const void* const* W, const void* const* x, const void* const* res,
                          void* const* y, const int* N, const int* K, const int* act, void* stream);
/* Same, plus a side job: copy_dst[0..copy_elems) = copy_src (bf16 elements, copy_elems % 8 == 0), done by one workgroup. */
int vrwkv_gemv_multi_copy_bf16(int n_jobs, int B, const void* const* W, const void* const* x, const void* const* res,
                               void* const* y, const int* N, const int* K, const int* act, const void* copy_src,
                               void* copy_dst, long copy_elems, void* stream);
/* Batched GEMV with LayerNorm + token shift + lerp folded into the input (Block.forward's ln1 / ln2 followed by the
 * lerps of src/model.py:169-173,222-223 at T = 1):  h = LayerNorm(x; ln_w, ln_b, eps),  in_j = h + (x_prev - h) mu_j,
 * y_j = act_j(W_j in_j).  All jobs share x (B,K) and K (512 <= K <= 4096); h is also written to h_out (B,K).  x_prev is
 * only read: the caller replaces it by h_out in a LATER launch (the side jobs above / of vrwkv_decode_tmix_head_bf16). */
int vrwkv_gemv_ln_multi_bf16(int n_jobs, int B, int K, const void* const* W, const void* x, const void* ln_w,
                             const void* ln_b, float eps, const void* x_prev, const void* const* mu, void* h_out,
                             void* const* y, const int* N, const int* act, void* stream);

/* Decode step, fused glue (T = 1; replaces the launch-bound chains of src/model.py:166-194,247-254 for one token).
 * ln_mix: h = LayerNorm(x) (src/model.py:250,253: ln1 / ln2), out_j = h + (x_prev - h) * mu_j for M <= 6 lerps
 * (src/model.py:169-173,222-223 with the shift reading the carried row), then x_prev = h.  x, x_prev, out_j: (B,C) bf16;
 * ln_w, ln_b, mu_j: (C) bf16; C % 8 == 0, C <= 8192. */
int vrwkv_decode_ln_mix_bf16(int B, int C, int M, const void* x, const void* ln_w, const void* ln_b, float eps,
                             void* x_prev, const void* const* mu, void* const* out, void* stream);
/* tmix_head, one workgroup per (b, head): second LoRA stage of w / a / g / v-gate (hid_i (B,D_i) bf16 hidden vectors
 * with their activation applied, W2t_i (C,D_i) bf16 = the reference's w2, a2, g2, v2 parameters TRANSPOSED;
 * src/model.py:176,181-183), decay soft-clamp (:176), k_k normalisation, k_a, value residual (:180-187), the WKV7 step
 * of vrwkv_wkv7_step_bf16 on `state` in place, GroupNorm ln_x + r_k bonus + gate (:190-193).  r, k, v, out: (B,C) bf16
 * with C = 64 H; v_first / v0 / hid[3] / W2t[3] NULL on layer 0; D_i % 32 == 0.  Optional side job: carry_dst = carry_src
 * ((B,C) bf16 each; the carried token-shift row replaced by the LayerNorm row vrwkv_gemv_ln_multi_bf16 produced). */
int vrwkv_decode_tmix_head_bf16(int B, int H, const void* r, const void* k, const void* v, const void* v_first,
                                const void* const* hid, const void* const* W2t, const int* D,
                                const void* w0, const void* a0, const void* v0, const void* k_k, const void* k_a,
                                const void* r_k, const void* ln_w, const void* ln_b, float eps,
                                float* state, void* out, const void* carry_src, void* carry_dst, void* stream);

/* WKV7 single-token step with carried state (stateful generation; the reference re-runs the whole forward per new
 * token, VisualRWKV-v7/v7.00/src/model.py:513-529).  w..a, y: (B,H,64) bf16; state: (B,H,64,64) f32, S[i][j] with
 * i = value row, j = key column, updated in place.  (The training op's checkpoint `s` holds S^T.) */
int vrwkv_wkv7_step_bf16(int B, int H, const void* w, const void* q, const void* k, const void* v,
                         const void* z, const void* a, float* state, void* y, void* stream);

/* Kernel-generation override for tests and A/B benchmarks; -1 restores the default.
 * forward:  -1 = the default: csrc/wkv7_fwd_v4.h (chunked MFMA kernel with producer / consumer waves; rows in by LDS-DMA, results out
 *            through LDS images, 1 KB per store instruction) for B*H > 128, else the two-workgroups-per-head instantiation of
 *            csrc/wkv7_fwd_v3.h; 7 = wkv7_fwd_v4.h whatever the size; 6 = the two-workgroups-per-head form whatever the size; 1..5 = instantiations of wkv7_fwd_v3.h for A/B
 *            (4 = its default one).  Measured at T = 2624, H = 32 (round 6): 128 heads 0.283 (6) vs 0.297 ms (7); 192 heads 0.452 vs 0.299; 256 heads 0.493 vs 0.319.
 * backward: -1 = the default (9 for B x H > 256, else 8; a launch whose `sa` reaches 4 GiB runs as batch slices of the same kernel), 9 = 8 with the score pieces of a chunk formed one step ahead by the
 *            memory-role waves in what was their barrier wait (0 ... -3 % against 8 at B = 4 ... 32, -2.3 % inside the training step), 8 = three-role
 *            pipeline of 12 waves with ONE copy of dL/dS (handed from the i-split to the j-split waves as an operand image), the T chain on a
 *            memory-role wave and a memory role that moves full 128-byte rows by LDS-DMA (csrc/wkv7_bwd_v8.h, wkv7_bwd_rows.h), 5 = producer / consumer schedule of
 *            8 waves (csrc/wkv7_bwd_v5.h; also the sequence-parallel kernel, and the kernel of a single sample of 4 GiB and more).  Anything else:
 *            VRWKV_EINVAL (experiment builds of benchmarks/build_alt.sh accept more: benchmarks/experiments/wkv7_experiments.h). */
int vrwkv_wkv7_set_forward_variant(int variant);
int vrwkv_wkv7_set_backward_variant(int variant);
/* Test hook: the tensor size (bytes of the fp32 `sa`, B*T*H*64*4) from which vrwkv_wkv7_backward_bf16 cuts a launch into batch slices (the default
 * kernel forms 32-bit byte offsets inside a tensor); 0 = the default, 4 GiB.  Results do not depend on it. */
int vrwkv_wkv7_set_backward_slice_limit(unsigned long long bytes);
/* The kernel generation the LAST vrwkv_wkv7_forward_bf16 (backward == 0) / vrwkv_wkv7_backward_bf16 (backward != 0) launch of this
 * process resolved to, in the numbering above (forward: 7 = wkv7_fwd_v4.h, 6 = two workgroups per head, 4 = wkv7_fwd_v3.h; 0 = none yet);
 * vrwkv_wkv7_forward_state_bf16 records into the forward slot as well.  Lets a single-threaded parity test assert WHICH kernel the default
 * dispatch chose for its shape. */
int vrwkv_wkv7_last_variant(int backward);
/* Which kernel generation a launch of shape (B,T,H) resolves to under the current override, WITHOUT launching: kind 0 = vrwkv_wkv7_forward_bf16,
 * 1 = vrwkv_wkv7_backward_bf16, 2 = vrwkv_wkv7_forward_state_bf16 (same rule as kind 0; its A/B overrides 1..5 all mean 4).  A pure function of
 * its arguments and the override: unlike vrwkv_wkv7_last_variant it does not depend on which thread launched last (the reference calls the op
 * from the Python thread and from autograd's backward thread).  A bad shape or kind gives VRWKV_EINVAL (< 0). */
int vrwkv_wkv7_resolve_variant(int kind, int B, int T, int H);

/* ---- Fused element-wise glue of RWKV_Tmix_x070 / RWKV_CMix_x070 (VisualRWKV-v7/v7.00/src/model.py), forward and
 * backward.  Activations are (ntok, C) bf16 contiguous (ntok = B*T), parameters C bf16.  Parameter gradients
 * are written (not accumulated) as fp32; the backward entry points need a scratch buffer `ws` of
 * vrwkv_param_grad_ws_floats(ntok, C, nvec) floats (nvec = number of C-sized gradient vectors: M for mix, 1 for
 * decay, 4 for kva, 3 for post) for the per-workgroup partial sums.  C % 64 == 0, C <= 8192.
 *   mix    : token-shift + M lerps, model.py:166-173 (M = 6) and :222-224 (M = 1); `mu`, `out`, `dout` are host
 *            arrays of M device pointers; dmu is M*C floats.  Shift indexing is exact: x[t-1], zero at t = 0 of
 *            every sample (ntok % T == 0).
 *   decay  : w = -softplus(-(w0 + h)) - 0.5, model.py:176.
 *   kva    : a = sigmoid(a0+al); v2 = v + (v_first - v) sigmoid(v0+vl) (layers > 0: has_vres); kk = normalize(k k_k)
 *            per 64-channel head; k2 = k (1 + (a-1) k_a); z = -kk; b = kk a.   model.py:179-188.
 *   post   : GroupNorm(H, C, eps)(y) + (sum_head r k r_k) v, times g.   model.py:191-194.
 *   relusq : relu(h)^2, model.py:225.
 */
int vrwkv_mix_fwd_bf16(long ntok, int T, int C, int M, const void* x, const void* const* mu, void* const* out, void* stream);
/* same, for stateful inference: x_prev (ntok/T, C) bf16 = the token before the first one of every sample (NULL: zeros) */
int vrwkv_mix_fwd_prev_bf16(long ntok, int T, int C, int M, const void* x, const void* x_prev, const void* const* mu,
                            void* const* out, void* stream);
long vrwkv_param_grad_ws_floats(long ntok, int C, int nvec);
int vrwkv_mix_bwd_bf16(long ntok, int T, int C, int M, const void* x, const void* const* mu, const void* const* dout,
                       void* dx, float* dmu, float* ws, void* stream);
/* same with a second gradient for output 3 (M == 6 only; NULL = none): x_v feeds the value projection AND the v-gate
 * LoRA (src/model.py:178,182); the two gradients are summed in the kernel instead of by autograd's element-wise add */
int vrwkv_mix_bwd2_bf16(long ntok, int T, int C, int M, const void* x, const void* const* mu, const void* const* dout,
                        const void* dout3_second, void* dx, float* dmu, float* ws, void* stream);
/* RWKV-6 (BASELINE config 4) glue, same conventions:  ddmix = token shift with per-token lerp weights, 5 outputs
 * (VisualRWKV-v6/v6.0/src/model.py:150-160: out_j = x + (x[t-1] - x) (mu_j + mm_j), mm_j (ntok, C) bf16 from the 5-way LoRA);
 * gn_silu = GroupNorm(C/64 groups, eps)(y) * silu(gg) (model.py:166,176-184).  mix accepts M = 1, 2 (RWKV_CMix_x060) and 6. */
int vrwkv_ddmix_fwd_bf16(long ntok, int T, int C, const void* x, const void* const* mu, const void* const* mm, void* const* out, void* stream);
int vrwkv_ddmix_bwd_bf16(long ntok, int T, int C, const void* x, const void* const* mu, const void* const* mm, const void* const* dout,
                         void* dx, void* const* dmm, float* dmu /* 5*C */, float* ws, void* stream);
int vrwkv_gn_silu_fwd_bf16(long ntok, int C, float eps, const void* y, const void* gg, const void* ln_w, const void* ln_b, void* out, void* stream);
int vrwkv_gn_silu_bwd_bf16(long ntok, int C, float eps, const void* y, const void* gg, const void* ln_w, const void* ln_b, const void* dout,
                           void* dy, void* dgg, float* dparams /* 2*C: dln_w dln_b */, float* ws, void* stream);
int vrwkv_decay_fwd_bf16(long ntok, int C, const void* h, const void* w0, void* w, void* stream);
int vrwkv_decay_bwd_bf16(long ntok, int C, const void* h, const void* w0, const void* dw, void* dh, float* dw0, float* ws,
                         void* stream);
int vrwkv_kva_fwd_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl,
                       const void* al, const void* k_k, const void* k_a, const void* a0, const void* v0,
                       void* k2, void* v2, void* z, void* b, void* stream);
int vrwkv_kva_bwd_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl,
                       const void* al, const void* k_k, const void* k_a, const void* a0, const void* v0,
                       const void* dk2, const void* dv2, const void* dz, const void* db,
                       void* dk, void* dv, void* dvfirst, void* dvl, void* dal,
                       float* dparams /* 4*C: dk_k dk_a da0 dv0 */, float* ws, void* stream);
/* same with optional second gradients of k2 and v2 (NULL = none): both feed the WKV7 op AND the bonus term of `post`
 * (src/model.py:190,193) */
int vrwkv_kva_bwd2_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl,
                        const void* al, const void* k_k, const void* k_a, const void* a0, const void* v0,
                        const void* dk2, const void* dv2, const void* dz, const void* db, const void* dk2_second,
                        const void* dv2_second, void* dk, void* dv, void* dvfirst, void* dvl, void* dal,
                        float* dparams, float* ws, void* stream);
/* the same with dvfirst_in (ntok, C) bf16 or NULL added to dvfirst: the layers pass the gradient of v_first down a chain (each layer's
 * kva returns an alias of v_first for the next layer) instead of autograd summing one term per layer */
int vrwkv_kva_bwd3_bf16(long ntok, int C, int has_vres, const void* k, const void* v, const void* vfirst, const void* vl, const void* al,
                        const void* k_k, const void* k_a, const void* a0, const void* v0,
                        const void* dk2, const void* dv2, const void* dz, const void* db, const void* dk2_second, const void* dv2_second,
                        const void* dvfirst_in, void* dk, void* dv, void* dvfirst, void* dvl, void* dal,
                        float* dparams, float* ws, void* stream);
int vrwkv_post_fwd_bf16(long ntok, int C, float eps, const void* y, const void* r, const void* k, const void* v,
                        const void* g, const void* ln_w, const void* ln_b, const void* r_k, void* out, void* stream);
int vrwkv_post_bwd_bf16(long ntok, int C, float eps, const void* y, const void* r, const void* k, const void* v,
                        const void* g, const void* ln_w, const void* ln_b, const void* r_k, const void* dout,
                        void* dy, void* dr, void* dk, void* dv, void* dg, float* dparams /* 3*C: dln_w dln_b dr_k */,
                        float* ws, void* stream);
int vrwkv_relusq_fwd_bf16(long n, const void* h, void* y, void* stream);
int vrwkv_relusq_bwd_bf16(long n, const void* h, const void* dy, void* dh, void* stream);

/* Softmax attention forward of the frozen ViT towers: o = softmax(q k^T / sqrt(D)) v for D in {64, 72}
 * (replaces the attention inside the timm VisionTransformer blocks run by SamDinoSigLIPViTBackbone.forward,
 * VisualRWKV-v7/v7.00/src/vision.py:123-134).  q/k/v element (b,l,h,d) at b*stride_b + l*stride_l + h*stride_h + d
 * (bf16, strides in elements, multiples of 8); o is (B,L,H,D) contiguous bf16. */
int vrwkv_attention_fwd_bf16(int B, int L, int H, int D, const void* q, const void* k, const void* v,
                             long stride_b, long stride_l, long stride_h, void* o, void* stream);

/* The same with SAM's decomposed relative-position bias added to the logits inside the kernel (replaces
 * Attention.forward + add_decomposed_rel_pos, VisualRWKV-v7/v7.00/src/sam.py:289-305, 392-426, without the
 * (B, H, L, L) bias tensor): L = S*S tokens of an S x S window (S = 14: windowed blocks, S = 64: global blocks),
 * logit(q, k) = q.k / sqrt(D) + q.rel_h[qh - kh + S - 1] + q.rel_w[qw - kw + S - 1] with the UNSCALED q.
 * rel_h / rel_w: (2S-1, D) bf16 contiguous (already interpolated to 2S-1 rows, sam.py:371-381).  D = 64. */
int vrwkv_attention_relpos_fwd_bf16(int B, int S, int H, int D, const void* q, const void* k, const void* v,
                                    long stride_b, long stride_l, long stride_h, const void* rel_h, const void* rel_w,
                                    void* o, void* stream);
/* Patch embedding of the ViT towers as an implicit GEMM over the NCHW pixels, bias and position embedding fused
 * (replaces timm PatchEmbed + pos_embed add run via VisualRWKV-v7/v7.00/src/vision.py:123-134 and
 * PatchEmbed.forward + pos_embed of src/sam.py:118-121,468-483):
 *   out[b, prefix + m, n] = sum_{c,py,px} pixels[b, c, gy P + py, gx P + px] w[n, (c P + py) P + px] + bias[n] + pos[m, n]
 * pixels (B,3,Himg,Wimg) bf16; w_padded (N, KP) bf16 = the conv weight flattened to (N, 3 P P) and zero-padded to
 * KP = vrwkv_patch_embed_kp(P) columns; bias (N) / pos (M, N) bf16 or NULL; out (B, tokens_per_image, N) bf16, rows
 * < prefix of each image are left untouched (class / register tokens).  P in {14, 16}; (Himg/P)(Wimg/P) % 64 == 0;
 * N % 32 == 0. */
int vrwkv_patch_embed_bf16(int B, int Himg, int Wimg, int P, int N, const void* pixels, const void* w_padded,
                           const void* bias, const void* pos, void* out, int tokens_per_image, int prefix, void* stream);
int vrwkv_patch_embed_kp(int P);
/* Tower image transform on the device: antialiased bicubic resize of one decoded (H,W,3) uint8 image to
 * S x S, clip to [0,255], normalise ((v/255 - mean)/std), planar (3,S,S) bf16 or fp32 output (replaces, per image and tower,
 * Resize((S,S), bicubic) + ToTensor + Normalize run by the reference's DataLoader worker: src/vision.py:96-121).
 * mean3 / std3: host pointers to 3 floats. */
int vrwkv_resize_normalize_u8(int H, int W, const void* src_hwc_u8, int S, const float* mean3, const float* std3,
                              void* dst_chw, int dst_is_f32, void* stream);
/* Test hook: query tiles of 16 per wave in the attention kernels (1 or 2; 0 = default = 2). */
int vrwkv_attention_set_qtiles(int qt);

/* Fused AdamW step on a flat ZeRO-1 shard (replaces DeepSpeed's FusedAdam(adam_w_mode=True) that the
 * reference configures in VisualRWKV-v7/v7.00/src/model.py:410): fp32 master/m/v, bf16 gradient in, bf16
 * parameter out, gradient pre-scaled by grad_scale (clip coefficient / world size), bias correction for
 * `step` (1-based).  Elements whose global index (global_offset + i) is >= wd_boundary get no weight
 * decay (model.py:391-393).  n % 4 == 0. */
int vrwkv_adamw_step_bf16(long n, float* master, float* m, float* v, const void* grad, void* param,
                          float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                          float grad_scale, long global_offset, long wd_boundary, void* stream);

/* ---- image side: pooling, context gate, LayerNorm + scatter (src/model.py:328-338,442-447,485-493) ------------------ */
/* nn.AdaptiveAvgPool2d(side_out) of a token-major ViT feature map: x (B, side_in^2, D) -> y (B, side_out^2, D) bf16,
 * PyTorch's window rule (also side_out > side_in), fp32 accumulation.  D % 8 == 0. */
int vrwkv_adaptive_pool_bf16(int B, int side_in, int side_out, int D, const void* x, void* y, void* stream);
/* MLPWithContextGating's gate: out = x * sigmoid(g) over n bf16 elements (n % 8 == 0); backward dg = dout x s (1 - s)
 * and, when dx is not NULL, dx = dout s. */
int vrwkv_gate_fwd_bf16(long n, const void* x, const void* g, void* out, void* stream);
int vrwkv_gate_bwd_bf16(long n, const void* x, const void* g, const void* dout, void* dg, void* dx, void* stream);
/* nn.GELU of the frozen towers' MLPs (timm `Mlp` run by src/vision.py:123-134: DINOv2 exact, SigLIP `gelu_tanh`; src/sam.py MLPBlock exact) over n bf16
 * elements (n % 8 == 0), y may be x: tanh_approx != 0 = 0.5 x (1 + tanh(sqrt(2/pi)(x + 0.044715 x^3))); 0 = 0.5 x (1 + erf(x / sqrt 2)), erfc to 1.2e-7
 * relative (fp32 arithmetic, one rounding to bf16). */
int vrwkv_gelu_bf16(long n, const void* x, void* y, int tanh_approx, void* stream);
/* ln_v of the projector fused with the masked scatter of preparing_embedding: out[row_index[n]] = LayerNorm(x[n]) for the
 * ntok projected image tokens, written into the (rows, C) token-embedding tensor `out`; row_index: device int64, distinct; a negative entry drops that
 * feature row (fewer placeholders than features: the reference truncates, src/model.py:487-491).  mean /
 * rstd (ntok fp32 each) are saved for the backward, which reads dout[row_index[n]] and returns dx (ntok, C) and
 * dwb = (dgamma, dbeta) (2 C fp32); ws: vrwkv_add_ln_ws_floats(ntok, C) floats. */
int vrwkv_ln_scatter_fwd_bf16(long ntok, int C, float eps, const void* x, const void* w, const void* b, const long* row_index,
                              void* out, float* mean, float* rstd, void* stream);
int vrwkv_ln_gather_bwd_bf16(long ntok, int C, const void* dout, const long* row_index, const void* x, const float* mean,
                             const float* rstd, const void* w, void* dx, float* dwb, float* ws, void* stream);

/* The same step with the clip factor formed on the device: sqnorm[0] = squared L2 norm of the (unscaled, summed over
 * ranks) gradient, grad scale = inv_world * min(1, clip / (sqrt(sqnorm) * inv_world + 1e-6)) (clip <= 0: no clipping).
 * Replaces Lightning's gradient_clip_val=1.0 host-side norm (train.py:92) without a device -> host synchronisation. */
int vrwkv_adamw_step_clip_bf16(long n, float* master, float* m, float* v, const void* grad, void* param,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               const float* sqnorm, float inv_world, float clip, long global_offset, long wd_boundary,
                               void* stream);

/* out[0] += sum of squares of a bf16 buffer (n % 8 == 0); used for gradient_clip_val=1.0 (train.py:92). */
int vrwkv_sqnorm_bf16(long n, const void* x, float* out, void* stream);

/* Profiling build of the chunked WKV7 kernels: same computation, plus dbg[0..15] (device memory, zero it first)
 * += shader-clock cycles that workgroup 0 spent in each phase.  backward = 0: uses w..a, y, s, sa; 1: all. */
int vrwkv_wkv7_profile_bf16(int backward, int B, int T, int H, const void* w, const void* q, const void* k, const void* v,
                            const void* z, const void* a, const void* dy, void* y, float* s, float* sa,
                            void* dw, void* dq, void* dk, void* dv, void* dz, void* da,
                            unsigned long long* dbg, void* stream);

/* Weight gradient of a skinny projection, out = Wide^T Narrow: Wide (M,Nw) and Narrow (M,D) bf16 row-major, out bf16
 * (Nw,D), or (D,Nw) with transposed != 0.  These are the gradients of the LoRA factors of RWKV_Tmix_x070
 * (VisualRWKV-v7/v7.00/src/model.py:176,181-183; dW1 = x^T dH: Wide = x; dW2 = h^T dOut: Wide = dOut, transposed) that
 * autograd computes with torch.mm in the reference.  Nw % 128 == 0, D in {32,64,96,128,160,256};
 * ws: vrwkv_wgrad_skinny_ws_floats(M,Nw,D) floats of scratch (-1: unsupported shape). */
long vrwkv_wgrad_skinny_ws_floats(long M, int Nw, int D);
int vrwkv_wgrad_skinny_bf16(long M, int Nw, int D, const void* wide, const void* narrow, void* out, int transposed,
                            float* ws, void* stream);

/* ---- Weight gradient of a square / wide projection: out (N1 x N2, bf16) = A^T B for A (M x N1), B (M x N2) bf16 row-major, fp32
 * accumulation (dW = dy^T x of nn.Linear: VisualRWKV-v7/v7.00/src/model.py:150-153 receptance / key / value / output, :214-215
 * channel-mix key / value, :281 head; autograd's dy.t().mm(x)).  csrc/wgrad_big.h: both MFMA operands by transposing LDS reads from
 * tiles stored as they lie in memory, split over M when the output has fewer tiles than the chip has CUs.  M % 32 == 0,
 * N1 % 256 == 0, N2 % 256 == 0.  ws: vrwkv_wgrad_big_ws_floats(M,N1,N2) floats (0: none needed; -1: unsupported shape). */
long vrwkv_wgrad_big_ws_floats(long M, int N1, int N2);
int vrwkv_wgrad_big_bf16(long M, int N1, int N2, const void* A, const void* B, void* out, float* ws, void* stream);


/* Streaming copy dst = src (bytes % 16 == 0): the on-box copy ceiling the WKV roofline fraction is also reported
 * against.  Moves 2 * bytes of HBM traffic. */
int vrwkv_stream_copy(const void* src, void* dst, long bytes, void* stream);
/* The same streaming tiling with other read : write mixes (HBM ceilings for a write-heavy / read-heavy kernel): mode 1 fill
 * (0 : 1), 2 one read + two writes, 3 read only, 4 two reads + one write.  `bytes` = size of one array (multiple of 16). */
int vrwkv_stream_probe(int mode, const void* a, const void* b, void* d0, void* d1, long bytes, void* stream);

/* out[c][r] = in[r][c] for a row-major (rows, cols) bf16 matrix (rows, cols multiples of 64): the transposed weight copy
 * that lets a Linear's input gradient run in the forward GEMMs' operand layout (replaces the strided torch copy inside
 * autograd's mm; fused.linear). */
int vrwkv_transpose_bf16(long rows, long cols, const void* in, void* out, void* stream);

/* Hardware probe for the GPU tests (MFMA lane maps, cross-lane primitives); one wave.
 * which: 0 = 16x16x4 f32, 1 = 32x32x2 f32, 2 = 16x16x32 bf16, 3 = 32x32x16 bf16 (d = a*b, row-major
 * f32 operands), 4 = cross-lane primitives (a: 64 floats, d: 896 floats). */
int vrwkv_debug_probe(int which, const float* a, const float* b, float* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif

/* libmia_hip.so -- C ABI of the MI355X (gfx950) kernels behind the EfficientNet-B3 + CBAM hot path.
 *
 * The reference (MMHS-09/medical-image-analysis) has no FFI seam of its own: its hot path is
 * Python calling stock ATen/timm/OpenCV ops. Each entry point below therefore names the reference
 * call site whose arithmetic it replaces (paths relative to the reference's foundation_model/).
 *
 * Conventions
 *   - plain pointers to DEVICE memory, sizes as ints; no framework types. `stream` is a hipStream_t
 *     passed as void* (0 = default stream). Nothing here allocates or synchronises.
 *   - activations are NHWC, i.e. row-major [rows = B*H*W, C]; dtype: MIA_F32 or MIA_BF16 storage,
 *     fp32 accumulation. Parameters, statistics and weight gradients are fp32.
 *   - return 0 on success, negative on error; mia_last_error() gives the message (thread-local).
 *   - train-mode BatchNorm: producers write per-workgroup partial sums into a caller-provided
 *     fp32 slab part[part_rows][2][C] (every row is written; the caller picks 0 < part_rows <= 65535),
 *     mia_bn_finalize_* reduces it. Consumers apply the resulting per-channel affine while loading.
 */
#ifndef MIA_HIP_H
#define MIA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MIA_F32 = 0, MIA_BF16 = 1 };
enum { MIA_ACT_NONE = 0, MIA_ACT_SILU = 1, MIA_ACT_RELU = 2 };

const char* mia_last_error(void);
int mia_version(void);
/* 0 iff the current HIP device is gfx950. Also does the per-device set-up that must not happen for the first time
 * inside a HIP graph capture (address lookup of the library's built-in device arrays). */
int mia_check_device(void);
/* Kernel launches this library has made in this process so far (all entry points, all devices). No counterpart in
 * the reference: the benchmark reports launches per training step from the difference of two readings. */
long long mia_launch_count(void);

/* ---- pointwise (1x1) convolutions and linear layers: timm conv_pw / conv_pwl (src/model.py:283-288, :350),
 *      nn.Linear in the heads (src/model.py:125-130, :138, :141, :312, :315) --------------------------------
 * C[M,N] = pro(A)[M,K] . B[N,K]^T (+bias[N]) (epi_act) (+D[M,N], leading dim ldc)
 * pro(a)[m,k] = pro_act(a*pscale[k] + pshift[k]) * gate[(m / rows_per_img)*K + k]   (each part optional)
 * stats (optional): part[part_rows][2][N] <- column sums and sums of squares of the raw accumulators.
 * bf16 projections of the large maps (M >= 400 000, K <= 192, N <= 32, BatchNorm+SiLU prologue, statistics, maps of
 * 1024 pixels and more when gated) run in a streaming kernel that keeps the weights in registers and folds the gate
 * into them, (a*g).w = a.(g*w): the one bf16 rounding of the prologue then falls on g*w instead of on a*g. The
 * environment variable MIA_GEMM_NO_STREAM, read once per process, keeps the tiled kernel everywhere (A/B runs). */
int mia_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* D, int M,
                int N, int K, const float* pscale, const float* pshift, int pro_act, const float* gate,
                int rows_per_img, const float* bias, int epi_act, float* stats, int part_rows, void* stream);

/* Weight gradient (autograd of the ops above, train.py:370): O[N,K] (fp32, leading dim ldo) = G[M,N]^T . pro(X)[M,K].
 * The reduction over M is split across workgroups. With a workspace (ws != NULL, at least
 * mia_gemm_tn_ws_floats(...) floats; or ws == NULL and ws_floats < 0 when that count is 0) every split writes its
 * partial tile to ws and a second kernel adds the partials in a fixed order: bit-reproducible results, O is
 * OVERWRITTEN and need not be initialised; ws and O must be 16-byte aligned and ldo a multiple of 4. With ws == NULL and ws_floats == 0 the partials are added into O with
 * float atomics: O must then be zero-filled (or hold a gradient to accumulate into) and the rounding varies from run
 * to run. */
int mia_gemm_tn(int dtype, const void* G, int ldg, const void* X, int ldx, float* O, int ldo, int M, int N, int K,
                const float* pscale, const float* pshift, int pro_act, const float* gate, int rows_per_img,
                float* ws, long long ws_floats, void* stream);
/* Workspace size in floats for the call above (has_prologue = pscale or gate given); a count, not an error code
 * (-1: bad shape; 0: the product runs unsplit). */
long long mia_gemm_tn_ws_floats(int dtype, int M, int N, int K, int has_prologue);

/* ---- depthwise k x k convolutions: timm conv_dw (k in {3,5}, stride in {1,2}, pad (k-1)/2, no bias) ----------
 * y[B,OH,OW,C] = dw(pro(x)[B,H,W,C]); pro = SiLU(x*pscale+pshift) when pscale != NULL; w is [C][k*k] fp32.
 * bf16 with C >= 64 and OH*OW <= 1600 runs on the matrix cores (forward, stride-1 data gradient, 5x5 weight
 * gradient): there the taps are rounded to bf16 like every other operand (autocast semantics, fp32 accumulation);
 * all other cases multiply by the fp32 taps. fp32 tensors never take that path. Two environment variables, read once
 * per process, keep the fp32-tap kernels everywhere (for A/B measurements of speed and of bf16 gradient fidelity):
 * MIA_DW_NO_MFMA (forward and data gradient) and MIA_DWW_NO_MFMA (weight gradient). */
int mia_dwconv_fwd(int dtype, const void* x, int B, int H, int W, int C, const float* pscale, const float* pshift,
                   const float* w, int K, int S, void* y, float* stats, int part_rows, void* stream);
/* dx[B,H,W,C] from dy[B,OH,OW,C]. With yfwd != NULL the epilogue turns it into the gradient w.r.t. the
 * pre-BatchNorm tensor's normalised activation: dx <- dx * silu'(yfwd*escale+eshift), and stats receives
 * partials of sum(dx) and sum(dx * (yfwd-emean)*erstd). */
int mia_dwconv_bwd_data(int dtype, const void* dy, int B, int H, int W, int C, const float* w, int K, int S, void* dx,
                        const void* yfwd, const float* escale, const float* eshift, const float* emean,
                        const float* erstd, float* stats, int part_rows, void* stream);
/* Statistics rows mia_dwconv_fwd (kind 0) / mia_dwconv_bwd_data (kind 1) fill for this shape when at most `cap` rows
 * are allowed (H, W: the forward convolution's INPUT map): passing exactly this count as part_rows avoids the extra
 * zero-fill launch for rows no workgroup owns. Returns a count (-1: bad shape). */
int mia_dwconv_part_rows(int dtype, int kind, int B, int H, int W, int C, int K, int S, int cap);
/* The stride-2 data gradient of an inverted-residual block (timm InvertedResidual: conv_pw -> bn1 -> conv_dw) fused
 * with the two products of conv_pw's backward that consume it, for blocks whose bn1 backward is carried through
 * conv_pw (mia_bn_lin_prep / mia_bn_lin_fix). With g = dx * silu'(yfwd*escale+eshift) as in mia_dwconv_bwd_data, g
 * itself is never written; instead
 *     dxp[B*H*W, Cin] = g . ((gamma*erstd) o wexp)          t1[C, Cin] (fp32) = g^T x
 * wexp [C, Cin] and x [B*H*W, Cin] are bf16, 16-byte aligned like dxp. stats as in mia_dwconv_bwd_data (required).
 * ws: fp32 scratch of rows*C*Cin floats, rows = mia_dwconv_bwd_data_lin_rows(B, H, W, K, C, part_rows) (a count; -1:
 * bad shape). mia_dwconv_bwd_data_lin_supported is 1 for the shapes a kernel exists for (bf16, stride 2,
 * K = 3, C = 144, Cin in {8, 16, 24, 32}): EfficientNet-B3's block 1.0, the largest expanded tensor of the trunk. */
int mia_dwconv_bwd_data_lin_supported(int dtype, int C, int K, int S, int Cin);
int mia_dwconv_bwd_data_lin_rows(int B, int H, int W, int K, int C, int cap);
int mia_dwconv_bwd_data_lin(int dtype, const void* dy, int B, int H, int W, int C, const float* w, int K, int S,
                            const void* yfwd, const float* escale, const float* eshift, const float* emean,
                            const float* erstd, const float* gamma, const void* wexp, const void* x, int Cin,
                            void* dxp, float* t1, float* stats, int part_rows, float* ws, long long ws_floats,
                            void* stream);
/* dw[C][k*k] (fp32) = sum dy * pro(x). ws: fp32 scratch of ws_floats >= C*k*k floats (256*C*k*k lets every
 * workgroup row keep its own partial; fewer rows are used when it is smaller). */
int mia_dwconv_bwd_weight(int dtype, const void* x, int B, int H, int W, int C, const float* pscale,
                          const float* pshift, const void* dy, int K, int S, float* dw, float* ws,
                          long long ws_floats, void* stream);

/* ---- BatchNorm: timm BatchNormAct2d (eps 1e-5, momentum 0.1) and CBAM's nn.BatchNorm2d(1, momentum 0.01)
 *      (src/model.py:16) ------------------------------------------------------------------------------------
 * training: batch statistics from part (count = rows reduced), running stats updated in place (unbiased
 * variance), *num_batches_tracked += 1. eval: statistics = running stats. Outputs: scale = gamma*rstd,
 * shift = beta - mean*scale, mean, rstd. conv_bias (nullable): the normalised tensor is y + conv_bias[c] (a biased
 * convolution, DecoderBlock src/model.py:477-480) while only the bias-free y is stored: the outputs are relative to
 * y, the running mean tracks y + bias. */
int mia_bn_finalize_fwd(const float* part, int P, int C, double count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                        float eps, int training, const float* conv_bias, float* scale, float* shift, float* mean,
                        float* rstd, void* stream);
/* part holds partials of (sum g, sum g*xhat). Outputs dgamma, dbeta (nullable) and the coefficients of
 * dy = ca*g + cb*y + cd (y = raw pre-BN tensor). dconv_bias (nullable): gradient of a convolution bias added before
 * this normalisation (= column sum of dy: zero with batch statistics, ca * sum g with running statistics). */
int mia_bn_finalize_bwd(const float* part, int P, int C, double count, const float* gamma, const float* mean,
                        const float* rstd, int training, float* dgamma, float* dbeta, float* ca, float* cb, float* cd,
                        float* dconv_bias, void* stream);
/* g = (G*gate[b] + dpool[b]) * act'(Y*scale+shift) written to Out (b = row / rows_per_img; gate, dpool optional);
 * part <- partials of sum(g), sum(g*(Y-mean)*rstd). Out may be G itself (in place) when the caller owns G; with
 * neither activation nor gate nor dpool g equals G, nothing is written and Out may be null. */
int mia_bn_act_bwd_reduce(int dtype, const void* G, void* Out, const void* Y, long long R, int C, const float* scale,
                          const float* shift, const float* mean, const float* rstd, int act, const float* gate,
                          const float* dpool, int rows_per_img, float* part, int part_rows, void* stream);
/* G <- ca*[act'(Y*scale+shift) * (G*gate[b] + dpool[b])] + cb*Y + cd: activation, gate and BatchNorm backward
 * applied in one pass (gate/dpool optional, b = row / rows_per_img). */
int mia_bn_act_bwd_apply(int dtype, void* G, const void* Y, long long R, int C, const float* scale, const float* shift,
                         int act, const float* gate, const float* dpool, int rows_per_img, const float* ca,
                         const float* cb, const float* cd, void* stream);
/* Out = act(a[c]*X + b[c]*Y + d[c]) + Res   (Y/b and Res optional; Out may alias X). */
int mia_affine2(int dtype, const void* X, const void* Y, const void* Res, void* Out, long long R, int C,
                const float* a, const float* b, const float* d, int act, void* stream);

/* BatchNorm backward carried through a 1x1 convolution (autograd of timm conv_pw + bn1, train.py:370) instead of
 * being applied to the convolution's wide output: for y = x W^T and dy = ca*g + cb*y + cd,
 *   dx = g (ca o W) + x S + r      with S = W^T diag(cb) W [Cin][Cin] and r = cd^T W [Cin]        (mia_bn_lin_prep)
 *   dW = ca o (g^T x) + cb o (W G) + cd (x) cs    with G = x^T x and cs = column sums of x         (mia_bn_lin_fix)
 * W [Cm][Cin] in the compute dtype (the matrix the forward product used). prep writes WcaT [Cin][Cm] and S in the
 * compute dtype (B operands of mia_gemm_nt) and r in fp32 (its bias); fix turns dW = g^T x (fp32, from mia_gemm_tn)
 * into the weight gradient in place, given G and cs in fp32. */
int mia_bn_lin_prep(int dtype, const void* W, int Cm, int Cin, const float* ca, const float* cb, const float* cd,
                    void* WcaT, void* S, float* r, void* stream);
int mia_bn_lin_fix(int dtype, const void* W, int Cm, int Cin, const float* ca, const float* cb, const float* cd,
                   const float* G, const float* cs, float* dW, void* stream);

/* ---- squeeze-excite: timm SqueezeExcite (mean over H,W -> conv_reduce+bias -> SiLU -> conv_expand+bias -> sigmoid)
 * out[b][split][c] = sum over the split's rows of Wt[r,c] * act(Y*scale+shift)   (Wt optional) */
int mia_img_colsum(int dtype, const void* Y, const void* Wt, int B, int HW, int C, const float* scale,
                   const float* shift, int act, int splits, float* out, void* stream);
/* Wr = conv_reduce.weight [RD][C]; WeT = conv_expand.weight TRANSPOSED to [RD][C] (coalesced reads); gradients
 * dWr [RD][C] and dWe [C][RD] come back in the parameters' own layouts. */
int mia_se_gate_fwd(const float* pool_part, int B, int splits, int HW, int C, int RD, const float* Wr,
                    const float* br, const float* WeT, const float* be, float* pooled, float* hpre, float* gate,
                    void* stream);
/* mia_se_gate_bwd: dpre2 may be the same array as dgate_part when splits == 1 and the array already holds dpre2 (see
 * mia_se_bwd_sums); it is then read, not rewritten. */
int mia_se_gate_bwd(const float* dgate_part, int B, int splits, int HW, int C, int RD, const float* Wr,
                    const float* WeT, const float* gate, const float* hpre, const float* pooled, float* dpre2,
                    float* dpre1, float* dpool, float* dWr, float* dbr, float* dWe, float* dbe, void* stream);
/* The parameter gradients of the two FCs alone, from the dpre2 / dpre1 that mia_se_gate_bwd produced (pass dWr = dbr =
 * dWe = dbe = NULL there): they are needed by nobody before the optimizer, so a caller can run them on another stream. */
int mia_se_param_grad(int B, int C, int RD, const float* dpre2, const float* dpre1, const float* pooled,
                      const float* hpre, float* dWr, float* dbr, float* dWe, float* dbe, void* stream);

/* Backward of a = silu(scale*Y+shift)*gate[b] given dZ = dL/da, in one pass over (Y, dZ): per image and split the
 * five channel sums  dZ*silu(z) | silu'(z)*dZ | silu'(z)*dZ*xhat | silu'(z) | silu'(z)*xhat  (xhat = (Y-mean)*rstd)
 * as out [5][B][splits][C]. Plane 0 is the dgate_part of mia_se_gate_bwd; mia_se_bn_combine turns planes 1..4 plus
 * (gate, dpool) into the partial slab part [B][2][C] of mia_bn_finalize_bwd (timm SqueezeExcite + BatchNormAct2d
 * backward, done by autograd in the reference). */
/* gate (nullable, needs splits == 1): plane 0 comes out multiplied by gate*(1-gate), the sigmoid's derivative, i.e.
 * as the dpre2 of mia_se_gate_bwd -- pass that plane there as BOTH dgate_part and dpre2 and the entry point skips its
 * own pass over it. */
int mia_se_bwd_sums(int dtype, const void* Y, const void* dZ, int B, int HW, int C, const float* scale,
                    const float* shift, const float* mean, const float* rstd, int splits, float* out,
                    const float* gate, void* stream);
int mia_se_bn_combine(const float* sums, int B, int splits, int C, const float* gate, const float* dpool, float* part,
                      void* stream);

/* ---- CBAM (src/model.py:28-108) on an NHWC feature map x [B,H*W,C] -----------------------------------------
 * channel gate (ChannelGate, model.py:45-67): avg/max pool over H*W, shared MLP C->HID->C (ReLU) on both, sum,
 * sigmoid -> cscale [B,C]. avg, mx, amax (arg-max pixel), ha, hm (hidden activations) are kept for the backward.
 * ws: fp32 scratch of mia_cbam_ws_floats(B, H, W, C, HID, 0) floats (0 for small maps: ws may be NULL then). */
int mia_cbam_channel_fwd(int dtype, const void* x, int B, int HW, int C, int HID, const float* W1, const float* b1,
                         const float* W2, const float* b2, float* avg, float* mx, int* amax, float* ha, float* hm,
                         float* cscale, float* ws, void* stream);
/* Workspace size in floats of the CBAM forward (backward = 0) or backward (backward = 1) for this shape; returns a
 * count, not an error code (-1 if it does not fit an int). Large maps (the decoder levels of
 * EnhancedUNetDecoder, model.py:211) split their per-image reductions over row ranges and need room for partials. */
int mia_cbam_ws_floats(int B, int H, int W, int C, int HID, int backward);
/* spatial gate, first half (ChannelPool + 7x7 conv, model.py:77-91): comp [B,HW,2] = (max_c, mean_c) of x*cscale,
 * cmax = arg-max channel, pre = conv7x7(comp) (w7 [2][7][7]); part [part_rows][2][1] feeds mia_bn_finalize_fwd (C=1). */
int mia_cbam_spatial_fwd(int dtype, const void* x, const float* cscale, int B, int H, int W, int C, const float* w7,
                         float* comp, int* cmax, float* pre, float* part, int part_rows, void* stream);
/* out = x * cscale[b,c] * sigmoid(pre*bnst[0] + bnst[1])[b,p]  (model.py:66-67, :92-93); sgate [B,HW] stored. */
int mia_cbam_apply(int dtype, const void* x, const float* cscale, const float* pre, const float* bnst, int B, int HW,
                   int C, void* out, float* sgate, void* stream);
/* Backward of the whole CBAM module. bnst = {scale, shift, mean, rstd} of the BatchNorm(1); ws: fp32 scratch of
 * mia_cbam_ws_floats(B, H, W, C, HID, 1) floats. */
int mia_cbam_bwd(int dtype, const void* dx2, const void* x, int B, int H, int W, int C, int HID, const float* cscale,
                 const float* sgate, const float* pre, const float* bnst, const float* bn_gamma, int bn_training,
                 const float* comp, const int* cmax, const int* amax, const float* avg, const float* mx,
                 const float* ha, const float* hm, const float* W1, const float* W2, const float* w7, float* ws,
                 void* dx, float* dW1, float* db1, float* dW2, float* db2, float* dw7, float* dbn_gamma,
                 float* dbn_beta, void* stream);

/* ---- multi-head self-attention core (MultiHeadAttentionClassifier.forward, model.py:153-163) --------------------
 * qkv [B*HW, 3C] (query | key | value projections, head h = columns h*HD..) -> out [B*HW, C];
 * softmax(QK^T/sqrt(HD)) is written to P [B,heads,HW,HW] (fp32) for the backward; dropout(pdrop) on the weights
 * uses a counter-based mask from `seed` that the backward regenerates. HW <= 256, HD in {8,16,24,32,48,64,96,128}.
 * seed_offset (nullable, here and in every entry point that takes a seed): a counter in DEVICE memory added to the
 * seed (times an odd constant) at launch time, so that a captured HIP graph, which replays fixed arguments, draws
 * new masks on every replay when the caller rewrites the counter before it (mia_set_words). */
int mia_attention_fwd(int dtype, const void* qkv, int B, int HW, int C, int heads, float pdrop,
                      unsigned long long seed, const unsigned long long* seed_offset, float* P, void* out, void* stream);
/* out: the forward's output (the row sums of P o dP are formed as dout . out). */
int mia_attention_bwd(int dtype, const void* qkv, const float* P, const void* dout, const void* out, int B, int HW,
                      int C, int heads, float pdrop, unsigned long long seed, const unsigned long long* seed_offset,
                      void* dqkv, void* stream);

/* ---- classifier front end: AdaptiveAvgPool2d(1) -> Flatten -> [LayerNorm] -> Dropout (model.py:134-137, :309-311)
 * in_part [B][S][C]: per-image channel sums over H*W (mia_img_colsum). gamma/beta NULL = no LayerNorm. */
int mia_pool_ln_dropout_fwd(const float* in_part, int B, int S, int HW, int C, const float* gamma, const float* beta,
                            float eps, float pdrop, unsigned long long seed, const unsigned long long* seed_offset,
                            float* out, float* xhat, float* rstd,
                            void* stream);
int mia_pool_ln_dropout_bwd(const float* dout, int B, int C, const float* gamma, const float* xhat, const float* rstd,
                            float pdrop, unsigned long long seed, const unsigned long long* seed_offset, float* dpooled,
                            float* dgamma_rows,
                            float* dbeta_rows, void* stream);
/* out[b,p,c] = v[b,c]*mult (+ res[b,p,c]): gradient of the average pool spread back over H*W. */
int mia_broadcast_rows(int dtype, const float* v, int B, int HW, int C, float mult, const void* res, void* out,
                       void* stream);
/* x *= dropout mask (and zero where relu_ref <= 0): nn.Dropout after nn.ReLU, forward and backward. */
int mia_dropout_f32(float* x, const float* relu_ref, long long n, float pdrop, unsigned long long seed,
                    const unsigned long long* seed_offset, void* stream);
/* out[c] = sum_r x[r][c]: bias gradients. ws (nullable): fp32 scratch of ws_floats floats; with min(128, R/64) * C
 * floats the rows are summed by that many workgroups per 64 columns and folded in a fixed order (no float atomics);
 * with less, by fewer (one without a workspace). */
int mia_colsum(int dtype, const void* x, long long R, int C, int ld, float* out, float* ws, long long ws_floats,
               void* stream);
/* nn.CrossEntropyLoss (src/loss.py:147; train.py:352): mean loss and d loss / d logits (zero padded to ldd columns).
 * A label of -100 (the default ignore_index) drops its row from the sum and the divisor; any other label outside
 * [0, n) (a device-side assert in torch) gives a zero gradient row and a NaN loss, and is never used as an index. */
int mia_softmax_ce(const float* logits, int ld, const long long* labels, int B, int n, float* loss, float* dlogits,
                   int ldd, void* stream);

/* ---- stem: timm conv_stem (3->40, k3 s2 p1) as im2col + mia_gemm_nt. x is NCHW fp32 (the tensor the
 *      reference feeds the model, train.py:345-351); col is [B*OH*OW][32], columns 27..31 zero. */
int mia_stem_im2col(int dtype, const float* x, int B, int H, int W, void* col, void* stream);
/* The same lowering from a channels-last image in the compute type, x [B,H,W,3] (dtype) -> col [B*OH*OW,32] (dtype). */
int mia_stem_im2col_nhwc(int dtype, const void* x, int B, int H, int W, void* col, void* stream);

/* fp32 master weight [R][Cc] -> compute-dtype copy [R][ldd] and/or transpose [Cc][ldt], zero padded. */
int mia_cast_pad_transpose(int dtype, const float* src, int R, int Cc, void* dst, int ldd, void* dstT, int ldt,
                           void* stream);
/* The same for n matrices in one launch per 48 (all 1x1-conv weights of a step). The arrays are HOST arrays of
 * device pointers / sizes; dst[k] or dstT[k] may be NULL. */
int mia_cast_pad_transpose_multi(int dtype, int n, const void* const* src, const int* rows, const int* cols,
                                 void* const* dst, const int* ldd, void* const* dstT, const int* ldt, void* stream);

/* ---- preprocessing: A.Resize + A.Normalize + ToTensorV2 (src/dataset.py:247-251, inference.py:116-140) and the
 *      no-transform fallback cv2.resize + /255 (src/dataset.py:180-182). src u8 [N,h,w,ch] (ch 1 or 3) ->
 *      dst f32 [N,3,oh,ow]. mode 0: (x - 255*mean)*(1/(255*std)); mode 1: x/255. */
int mia_resize_normalize_u8(const uint8_t* src, int N, int h, int w, int ch, float* dst, int oh, int ow,
                            const float* mean3, const float* std3, int mode, void* stream);
/* The same transform with the result in the trunk's compute type and channels-last order, dst [N,oh,ow,3] in dtype
 * (MIA_F32 / MIA_BF16): values are the float results above rounded once to dtype. This is the form
 * mia_stem_im2col_nhwc reads, so the normalised image is written once at 6 bytes a pixel (bf16) and the first
 * convolution (timm conv_stem, src/model.py:283-288) takes it without a layout or type conversion. */
int mia_resize_normalize_u8_nhwc(const uint8_t* src, int N, int h, int w, int ch, void* dst, int dtype, int oh, int ow,
                                 const float* mean3, const float* std3, int mode, void* stream);

/* ---- zero-intensity slice filter. No counterpart in the reference (zi_remover.py:3-19 deletes Zone.Identifier
 *      files); specification: keep[z] = any(vol[z] != 0), idx = ascending kept indices, count[0] = their number.
 *      flags_ws: int32[n_slices] scratch, contents irrelevant on entry. elem_bytes in {1,2,4}; is_float selects the
 *      IEEE compare for 4-byte data. Slices of up to 1 MiB take ONE launch (a workgroup per slice, the last one to
 *      finish compacts); that path draws a ticket counter from a pool of 64 built into the library, so at most 64
 *      calls may be in flight on one GPU at a time. */
int mia_zero_slice_filter(const void* vol, int elem_bytes, int is_float, long long n_slices, long long slice_elems,
                          int* flags_ws, uint8_t* keep, int* idx, int* count, void* stream);
/* The same filter for callers that cannot allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc. No counterpart in the reference. The flag words live in a device array built into
 * the library (256 KiB per GPU, held until the process ends) instead of flags_ws; the entry point allocates nothing and
 * rejects n_slices > 65535. Calls on one GPU share that array: calls that overlap on the same GPU race. Results are
 * identical to mia_zero_slice_filter. The array's address is looked up once per device (hipGetSymbolAddress, under a
 * host lock): make the first call on a device -- or call mia_check_device(), which does the lookup -- OUTSIDE any HIP
 * graph capture. */
int mia_zero_slice_filter_static(const void* vol, int elem_bytes, int is_float, long long n_slices,
                                 long long slice_elems, uint8_t* keep, int* idx, int* count, void* stream);
/* out[k] = vol[idx[k]] for k < count[0] (count stays on the device: no host round trip). */
int mia_gather_slices(const void* vol, int elem_bytes, long long n_slices, long long slice_elems, const int* idx,
                      const int* count, void* out, void* stream);

/* ---- optimizer: torch.optim.AdamW as configured at train.py:193-197 / finetune.py:232-248 --------------------------
 * One multi-tensor pass over (param, grad, exp_avg, exp_avg_sq), all fp32 and contiguous. params/grads/exp_avg/
 * exp_avg_sq are HOST arrays of n_tensors DEVICE pointers, sizes a host array of element counts; step is the 1-based
 * number of this update; gradients are multiplied by grad_scale first. Arithmetic = torch's AdamW step:
 * p *= 1-lr*wd; m = lerp(m,g,1-b1); v = b2*v+(1-b2)*g*g; p -= lr/(1-b1^t) * m/(sqrt(v)/sqrt(1-b2^t)+eps).
 * dev_hyper (nullable): 3 floats in DEVICE memory {1-lr*wd, lr/(1-b1^t), 1/sqrt(1-b2^t)} read at run time in place
 * of the values derived from lr, weight_decay and step, for launches replayed from a captured HIP graph. */
int mia_adamw_step(int n_tensors, const void* const* params, const void* const* grads, const void* const* exp_avg,
                   const void* const* exp_avg_sq, const long long* sizes, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int step, double grad_scale, const float* dev_hyper,
                   void* stream);
/* Stores n (<= 64) 32-bit words, passed by value in the launch, to device memory dst. No counterpart in the
 * reference: it refreshes the per-step words a replayed graph reads (seed_offset, dev_hyper) without a host buffer
 * that must stay untouched until an asynchronous copy has run. */
int mia_set_words(void* dst, int n, const unsigned* host_words, void* stream);

/* ---- segmentation decoders (src/model.py:179-237 EnhancedUNetDecoder, :417-468 UNetDecoder, :471-494
 *      DecoderBlock) and losses (src/loss.py:27-80 CombinedSegmentationLoss/DiceLoss, :105-136 IoULoss) ----------
 * 3x3 / stride 1 / pad 1 convolution (nn.Conv2d(cin, cout, 3, padding=1), model.py:477,479) = im2col + mia_gemm_nt
 * with the weight as [cout][tap][cin]: col[pix][tap][c] = gate[b][c] * act(scale[c]*x + shift[c]) of the shifted
 * pixel, zero outside the image (scale/shift, gate optional: BatchNorm apply, ReLU and the Dropout2d mask of the
 * producing layer). C % 4 == 0. */
int mia_im2col3x3(int dtype, const void* x, int B, int H, int W, int C, const float* scale, const float* shift, int act,
                  const float* gate, void* col, void* stream);
/* data gradient of the lowering: dx[b,h,w,c] = sum_taps dcol[(b,h-dy,w-dx)][tap][c]. */
int mia_col2im3x3(int dtype, const void* dcol, int B, int H, int W, int C, void* dx, void* stream);
/* F.interpolate(x, size=(H,W), mode='bilinear', align_corners=False) followed by torch.cat([x, skip], dim=1)
 * (model.py:227-228, :235 with C2 = 0 / skip = NULL): out [B,H,W,C1+C2] from x [B,h,w,C1] and skip [B,H,W,C2]. */
int mia_upsample_concat_fwd(int dtype, const void* x, int B, int h, int w, int C1, const void* skip, int C2, int H,
                            int W, void* out, void* stream);
/* its gradient: dx [B,h,w,C1] gathered from dout [B,H,W,C1+C2], dskip [B,H,W,C2] = the trailing channels. */
int mia_upsample_concat_bwd(int dtype, const void* dout, int B, int h, int w, int C1, int C2, int H, int W, void* dx,
                            void* dskip, void* stream);
/* final_conv = nn.Conv2d(16, num_classes, 1) (model.py:214, :237): logits [B][n][HW] fp32 (NCHW) from x [B,HW,C];
 * Wt [n][C], bias [n] (nullable). C % 4 == 0, C <= 64, n <= 32. */
int mia_pixel_linear_fwd(int dtype, const void* x, int B, int HW, int C, int n, const float* Wt, const float* bias,
                         float* logits, void* stream);
/* dx [B,HW,C]; partial [partial_rows][n*C + n] = per-workgroup sums of (dWt, dbias): column-sum them. */
int mia_pixel_linear_bwd(int dtype, const void* x, const float* dlogits, int B, int HW, int C, int n, const float* Wt,
                         void* dx, float* partial, int partial_rows, void* stream);
/* nn.Dropout2d mask (model.py:481): mask[b][c] = 0 with probability p, else 1/(1-p); counter-based, from seed. */
int mia_channel_dropout_mask(float* mask, int B, int C, float p, unsigned long long seed,
                             const unsigned long long* seed_offset, void* stream);
/* loss = ce_weight * CrossEntropy(logits, target, weight=class_weights) + overlap_weight * (1 - mean score),
 * score = Dice (iou = 0) or IoU (iou = 1) per (image, class) on softmax probabilities. logits [B][n][HW] fp32,
 * target [B][HW] int64, n <= 32. partial: [B*splits][3n+2] scratch; coef: [1 + 2*B*n] kept for the backward. */
int mia_seg_loss_fwd(const float* logits, const long long* target, int B, int HW, int n, const float* class_weights,
                     int iou, float smooth, float ce_weight, float overlap_weight, int splits, float* partial,
                     float* coef, float* loss, void* stream);
/* dlogits = upstream[0] * d loss / d logits (upstream: device scalar, NULL = 1). */
int mia_seg_loss_bwd(const float* logits, const long long* target, int B, int HW, int n, const float* class_weights,
                     const float* coef, const float* upstream, float* dlogits, void* stream);

/* ---- evaluation counters (MetricsCalculator.update, src/metric.py:29-43: argmax + Python lists + scikit-learn
 *      in the reference) -----------------------------------------------------------------------------------------
 * conf[t*n + argmax_k logits] += 1 per sample/pixel (uint64 [n][n], caller zero-fills at reset), first maximum on
 * ties; logits element (image b, class k, position p) at b*stride_image + k*stride_class + p*stride_pos (fp32);
 * target [images*positions] int64, values outside [0,n) skipped. loss/loss_acc (nullable): loss_acc[0] += loss[0]
 * (the running loss of train.py:355, kept on the device: no .item() per step). n <= 64. */
int mia_confusion_update(const float* logits, long long stride_image, long long stride_class, long long stride_pos,
                         const long long* target, long long images, long long positions, int n,
                         unsigned long long* conf, const float* loss, double* loss_acc, void* stream);

/* ---- training-time augmentation (get_transforms(augment=True), src/dataset.py:229-239: A.Resize, HorizontalFlip,
 *      VerticalFlip, Rotate(limit=15), RandomBrightnessContrast(0.2, 0.2), GaussianBlur(3), Normalize, ToTensorV2;
 *      albumentations/OpenCV on DataLoader workers in the reference) ------------------------------------------------
 * 8-bit bilinear resize to HWC x 3 (gray replicated): the uint8 stage the later transforms operate on. */
int mia_resize_u8(const uint8_t* src, int N, int h, int w, int ch, uint8_t* dst, int oh, int ow, void* stream);
/* src u8 [N,H,W,3] (resized) -> dst f32 [N,3,H,W] normalised. params: 9 doubles per image on the device:
 * {hflip, vflip, rotate?, cos(angle), sin(angle), brightness/contrast?, alpha, beta*255, blur?}. Rotation about
 * (W/2-0.5, H/2-0.5), bilinear with 1/32-pixel coordinates, reflect-101 border; brightness/contrast v*alpha+beta
 * clipped and truncated to 8 bits; blur = [1 2 1]x[1 2 1]/16 with rounding, reflect-101. mean3/std3: host arrays. */
int mia_augment_normalize_u8(const uint8_t* src, int N, int H, int W, const double* params, float* dst,
                             const float* mean3, const float* std3, void* stream);
/* the geometric part for label masks (dataset.py:208-213): flips + nearest-neighbour rotation; u8 [N,H,W] -> int64. */
int mia_augment_mask_u8(const uint8_t* src, int N, int H, int W, const double* params, long long* dst, void* stream);

/* ---- the other loss variants of src/loss.py: FocalLoss (:83-102) and class-weighted nn.CrossEntropyLoss (:143-147) on
 *      class logits [B,n] or pixel logits [B,n,HW] (strides as in mia_confusion_update) ------------------------------
 * plain: l_i = w[t_i]*ce_i, mean = sum l / sum w[t]; focal: l_i = alpha*(1-pt_i)^gamma*ce_i, mean = sum l / count.
 * mean = 0 gives the sum. per_elem (nullable): l_i. partial: [partial_rows][2] scratch. out[0] = loss, out[1] = the
 * factor the backward applies (kept for it). */
int mia_ce_general_fwd(const float* logits, long long stride_image, long long stride_class, long long stride_pos,
                       const long long* target, long long images, long long positions, int n,
                       const float* class_weights, int focal, float alpha, float gamma, int mean, float* per_elem,
                       float* partial, int partial_rows, float* out, void* stream);
/* dlogits (same strides) = upstream * out[1] * d l_i/d logits; upstream: device scalar (upstream_per_elem = 0) or one
 * value per element (reduction 'none'; pass out = NULL then). */
int mia_ce_general_bwd(const float* logits, long long stride_image, long long stride_class, long long stride_pos,
                       const long long* target, long long images, long long positions, int n,
                       const float* class_weights, int focal, float alpha, float gamma, const float* out,
                       const float* upstream, int upstream_per_elem, float* dlogits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIA_HIP_H */

// Segmentation-decoder kernels (gfx950): 3x3 convolution lowering (im2col / col2im around the MFMA GEMMs of
// gemm.hip), bilinear upsample + skip concat and its gradient, the final per-pixel linear layer with NCHW fp32
// logits, channel-dropout masks, and the cross-entropy + Dice / IoU segmentation loss.
// Activations are NHWC [B, H, W, C] in the compute dtype T; all arithmetic is fp32.
#include "common.h"
#include <type_traits>

static inline unsigned grid_for(long long items, int cap = 16384) {
    long long b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------------------
// im2col for a 3x3 / stride 1 / pad 1 convolution: col[pix][tap][c] = f(x[b, h+dy, w+dx, c]) (zero outside the
// image), f = optional per-channel affine + activation + per-image channel gate (BatchNorm apply, ReLU and the
// channel-dropout mask of the producing layer folded into this pass).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col3_kernel(const T* x, int B, int H, int W, int C, const float* scale,
                                                      const float* shift, int act, const float* gate, T* col) {
    const unsigned cv = (unsigned)C >> 2;
    const long long total = (long long)B * H * W * 9 * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)((unsigned long long)i % cv) << 2;
        const long long t = i / cv;
        const int tap = (int)(t % 9);
        const long long pix = t / 9;
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        const int ih = h + tap / 3 - 1, iw = w + tap % 3 - 1;
        f4 o = {0.f, 0.f, 0.f, 0.f};
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
            o = load4<T>(x + ((size_t)(b * H + ih) * W + iw) * C + c);
            if (scale) {
                const f4 sc = load4<float>(scale + c), sh = load4<float>(shift + c);
                o = {o.x * sc.x + sh.x, o.y * sc.y + sh.y, o.z * sc.z + sh.z, o.w * sc.w + sh.w};
            }
            if (act == MIA_ACT_RELU) o = {fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f)};
            else if (act == MIA_ACT_SILU) o = {siluf_(o.x), siluf_(o.y), siluf_(o.z), siluf_(o.w)};
            if (gate) {
                const f4 g = load4<float>(gate + (size_t)b * C + c);
                o = {o.x * g.x, o.y * g.y, o.z * g.z, o.w * g.w};
            }
        }
        store4<T>(col + (size_t)i * 4, o);
    }
}

extern "C" int mia_im2col3x3(int dtype, const void* x, int B, int H, int W, int C, const float* scale,
                             const float* shift, int act, const float* gate, void* col, void* stream) {
    MIA_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "mia_im2col3x3: bad arguments");
    MIA_CHECK_ARG(!(scale == nullptr) == !(shift == nullptr), "mia_im2col3x3: scale/shift come together");
    const long long total = (long long)B * H * W * 9 * (C / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(im2col3_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)x, B, H, W, C,
                           scale, shift, act, gate, (bf16_t*)col);
    else
        MIA_LAUNCH(im2col3_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, B, H, W, C,
                           scale, shift, act, gate, (float*)col);
    MIA_CHECK_LAUNCH("mia_im2col3x3");
    return 0;
}

// transpose of im2col: dx[b,h,w,c] = sum over taps of dcol[(b, h-dy, w-dx)][tap][c]
template <typename T>
__global__ __launch_bounds__(256) void col2im3_kernel(const T* dcol, int B, int H, int W, int C, T* dx) {
    const unsigned cv = (unsigned)C >> 2;
    const long long total = (long long)B * H * W * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)((unsigned long long)i % cv) << 2;
        const long long pix = i / cv;
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const long long b = pix / ((long long)W * H);
        f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int oh = h - (tap / 3 - 1), ow = w - (tap % 3 - 1);
            if (oh < 0 || oh >= H || ow < 0 || ow >= W) continue;
            const f4 v = load4<T>(dcol + (((size_t)(b * H + oh) * W + ow) * 9 + tap) * C + c);
            a = {a.x + v.x, a.y + v.y, a.z + v.z, a.w + v.w};
        }
        store4<T>(dx + (size_t)pix * C + c, a);
    }
}

extern "C" int mia_col2im3x3(int dtype, const void* dcol, int B, int H, int W, int C, void* dx, void* stream) {
    MIA_CHECK_ARG(dcol && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "mia_col2im3x3: bad arguments");
    const long long total = (long long)B * H * W * (C / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(col2im3_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dcol, B, H, W,
                           C, (bf16_t*)dx);
    else
        MIA_LAUNCH(col2im3_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dcol, B, H, W, C,
                           (float*)dx);
    MIA_CHECK_LAUNCH("mia_col2im3x3");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Bilinear resampling with half-pixel centres (align_corners=False): src = scale*(dst+0.5)-0.5 clamped at 0,
// scale = in/out in fp32, second tap = first + 1 unless on the last row/column.
// ------------------------------------------------------------------------------------------------
struct BilTap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ BilTap bil_tap(int dst, float scale, int in) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    int i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    float l1 = src - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    return {i0, i0 + (i0 < in - 1 ? 1 : 0), 1.f - l1, l1};
}

// out[b,H,W, 0:C1] = bilinear(x[b,h,w,C1]);  out[..., C1:C1+C2] = skip[b,H,W,C2]
template <typename T>
__global__ __launch_bounds__(256) void upsample_concat_kernel(const T* x, int B, int h, int w, int C1, const T* skip,
                                                              int C2, int H, int W, T* out) {
    const int Ct = C1 + C2;
    const unsigned cv = (unsigned)Ct >> 2;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)B * H * W * cv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)((unsigned long long)i % cv) << 2;
        const long long pix = i / cv;
        f4 o;
        if (c >= C1) {
            o = load4<T>(skip + (size_t)pix * C2 + (c - C1));
        } else {
            const int X = (int)(pix % W), Y = (int)((pix / W) % H);
            const long long b = pix / ((long long)W * H);
            const BilTap ty = bil_tap(Y, sh, h), tx = bil_tap(X, sw, w);
            const T* xb = x + (size_t)b * h * w * C1 + c;
            const f4 v00 = load4<T>(xb + ((size_t)ty.i0 * w + tx.i0) * C1), v01 = load4<T>(xb + ((size_t)ty.i0 * w + tx.i1) * C1);
            const f4 v10 = load4<T>(xb + ((size_t)ty.i1 * w + tx.i0) * C1), v11 = load4<T>(xb + ((size_t)ty.i1 * w + tx.i1) * C1);
            o.x = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
            o.y = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
            o.z = ty.l0 * (tx.l0 * v00.z + tx.l1 * v01.z) + ty.l1 * (tx.l0 * v10.z + tx.l1 * v11.z);
            o.w = ty.l0 * (tx.l0 * v00.w + tx.l1 * v01.w) + ty.l1 * (tx.l0 * v10.w + tx.l1 * v11.w);
        }
        store4<T>(out + (size_t)pix * Ct + c, o);
    }
}

extern "C" int mia_upsample_concat_fwd(int dtype, const void* x, int B, int h, int w, int C1, const void* skip, int C2,
                                       int H, int W, void* out, void* stream) {
    MIA_CHECK_ARG(x && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0,
                  "mia_upsample_concat_fwd: bad arguments");
    MIA_CHECK_ARG((C2 == 0) == (skip == nullptr), "mia_upsample_concat_fwd: skip and C2 come together");
    const long long total = (long long)B * H * W * ((C1 + C2) / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(upsample_concat_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)x, B, h,
                           w, C1, (const bf16_t*)skip, C2, H, W, (bf16_t*)out);
    else
        MIA_LAUNCH(upsample_concat_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, B, h, w,
                           C1, (const float*)skip, C2, H, W, (float*)out);
    MIA_CHECK_LAUNCH("mia_upsample_concat_fwd");
    return 0;
}

// gradient: dx[b,i,j,c] gathers from every output pixel whose two taps per axis include (i,j) — deterministic, no
// atomics; dskip = the trailing C2 channels of dout, copied out densely.
template <typename T>
__global__ __launch_bounds__(256) void upsample_concat_bwd_kernel(const T* dout, int B, int h, int w, int C1, int C2,
                                                                  int H, int W, T* dx, T* dskip) {
    const int Ct = C1 + C2;
    const unsigned cv1 = (unsigned)C1 >> 2, cv2 = (unsigned)C2 >> 2;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long n1 = (long long)B * h * w * cv1, n2 = (long long)B * H * W * cv2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * 256) {
        if (i >= n1) {
            const long long k = i - n1;
            const int c = (int)((unsigned long long)k % cv2) << 2;
            const long long pix = k / cv2;
            store4<T>(dskip + (size_t)pix * C2 + c, load4<T>(dout + (size_t)pix * Ct + C1 + c));
            continue;
        }
        const int c = (int)((unsigned long long)i % cv1) << 2;
        const long long pix = i / cv1;
        const int sx = (int)(pix % w), sy = (int)((pix / w) % h);
        const long long b = pix / ((long long)w * h);
        int ylo = (int)floorf(((float)sy - 0.5f) / sh - 0.5f) - 1, yhi = (int)ceilf(((float)sy + 1.5f) / sh - 0.5f) + 1;
        int xlo = (int)floorf(((float)sx - 0.5f) / sw - 0.5f) - 1, xhi = (int)ceilf(((float)sx + 1.5f) / sw - 0.5f) + 1;
        ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
        yhi = yhi > H - 1 ? H - 1 : yhi; xhi = xhi > W - 1 ? W - 1 : xhi;
        f4 a = {0.f, 0.f, 0.f, 0.f};
        const T* gb = dout + (size_t)b * H * W * Ct + c;
        for (int Y = ylo; Y <= yhi; ++Y) {
            const BilTap ty = bil_tap(Y, sh, h);
            const float wy = (ty.i0 == sy ? ty.l0 : 0.f) + (ty.i1 == sy ? ty.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int X = xlo; X <= xhi; ++X) {
                const BilTap tx = bil_tap(X, sw, w);
                const float wx = (tx.i0 == sx ? tx.l0 : 0.f) + (tx.i1 == sx ? tx.l1 : 0.f);
                if (wx == 0.f) continue;
                const f4 g = load4<T>(gb + ((size_t)Y * W + X) * Ct);
                const float wt = wy * wx;
                a = {a.x + wt * g.x, a.y + wt * g.y, a.z + wt * g.z, a.w + wt * g.w};
            }
        }
        store4<T>(dx + (size_t)pix * C1 + c, a);
    }
}

extern "C" int mia_upsample_concat_bwd(int dtype, const void* dout, int B, int h, int w, int C1, int C2, int H, int W,
                                       void* dx, void* dskip, void* stream) {
    MIA_CHECK_ARG(dout && dx && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C1 > 0 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0,
                  "mia_upsample_concat_bwd: bad arguments");
    MIA_CHECK_ARG((C2 == 0) == (dskip == nullptr), "mia_upsample_concat_bwd: dskip and C2 come together");
    const long long total = (long long)B * h * w * (C1 / 4) + (long long)B * H * W * (C2 / 4);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(upsample_concat_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dout,
                           B, h, w, C1, C2, H, W, (bf16_t*)dx, (bf16_t*)dskip);
    else
        MIA_LAUNCH(upsample_concat_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)dout, B,
                           h, w, C1, C2, H, W, (float*)dx, (float*)dskip);
    MIA_CHECK_LAUNCH("mia_upsample_concat_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// final 1x1 convolution: logits[b][k][p] (NCHW, fp32) = bias[k] + sum_c x[b,p,c] * Wt[k][c]
// ------------------------------------------------------------------------------------------------
// classes per pixel are held in registers: kernels are built for up to 8 (the reference's heads have 1-4) and up to 32
#define SEG_MAXN 32
#define SEG_MAXC 64

template <typename T, int MAXN>
__global__ __launch_bounds__(256) void pixel_linear_fwd_kernel(const T* x, int B, int HW, int C, int n, const float* Wt,
                                                               const float* bias, float* logits) {
    __shared__ float ws[MAXN * SEG_MAXC + MAXN];
    for (int i = threadIdx.x; i < n * C; i += 256) ws[i] = Wt[i];
    if (threadIdx.x < n) ws[MAXN * SEG_MAXC + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    __syncthreads();
    const long long total = (long long)B * HW;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
        float acc[MAXN];
#pragma unroll
        for (int k = 0; k < MAXN; ++k) acc[k] = ws[MAXN * SEG_MAXC + (k < n ? k : 0)];
        for (int c = 0; c < C; c += 4) {
            const f4 v = load4<T>(x + (size_t)pix * C + c);
#pragma unroll
            for (int k = 0; k < MAXN; ++k)
                if (k < n) {
                    const float* wk = ws + k * C + c;
                    acc[k] += v.x * wk[0] + v.y * wk[1] + v.z * wk[2] + v.w * wk[3];
                }
        }
        const long long b = pix / HW, p = pix % HW;
#pragma unroll
        for (int k = 0; k < MAXN; ++k)
            if (k < n) logits[((size_t)b * n + k) * HW + p] = acc[k];
    }
}

// dx[b,p,c] = sum_k dl[b][k][p] * Wt[k][c]
template <typename T, int MAXN>
__global__ __launch_bounds__(256) void pixel_linear_bwd_data_kernel(const float* dl, int B, int HW, int C, int n,
                                                                    const float* Wt, T* dx) {
    __shared__ float ws[MAXN * SEG_MAXC];
    for (int i = threadIdx.x; i < n * C; i += 256) ws[i] = Wt[i];
    __syncthreads();
    const long long total = (long long)B * HW;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
        const long long b = pix / HW, p = pix % HW;
        float g[MAXN];
#pragma unroll
        for (int k = 0; k < MAXN; ++k) g[k] = k < n ? dl[((size_t)b * n + k) * HW + p] : 0.f;
        for (int c = 0; c < C; c += 4) {
            f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < MAXN; ++k)
                if (k < n) {
                    const float* wk = ws + k * C + c;
                    o = {o.x + g[k] * wk[0], o.y + g[k] * wk[1], o.z + g[k] * wk[2], o.w + g[k] * wk[3]};
                }
            store4<T>(dx + (size_t)pix * C + c, o);
        }
    }
}

// partial[blk][k][c] = sum over the block's pixels of dl[k] * x[c];  partial[blk][n*C + k] = sum dl[k]
// 256 threads = (256 / (C/4)) pixel lanes x (C/4) channel-vector lanes
template <typename T, int MAXN>
__global__ __launch_bounds__(256) void pixel_linear_bwd_weight_kernel(const T* x, const float* dl, int B, int HW, int C,
                                                                      int n, float* partial) {
    __shared__ f4 red[256];
    __shared__ float redb[256];
    const int cvn = C >> 2, pl = 256 / cvn;
    const int cvl = threadIdx.x % cvn, plane = threadIdx.x / cvn;
    const bool active = plane < pl;
    f4 acc[MAXN];
    float bs[MAXN];
#pragma unroll
    for (int k = 0; k < MAXN; ++k) { acc[k] = {0.f, 0.f, 0.f, 0.f}; bs[k] = 0.f; }
    const long long total = (long long)B * HW;
    if (active)
        for (long long pix = (long long)blockIdx.x * pl + plane; pix < total; pix += (long long)gridDim.x * pl) {
            const long long b = pix / HW, p = pix % HW;
            const f4 v = load4<T>(x + (size_t)pix * C + cvl * 4);
#pragma unroll
            for (int k = 0; k < MAXN; ++k)
                if (k < n) {
                    const float g = dl[((size_t)b * n + k) * HW + p];
                    acc[k] = {acc[k].x + g * v.x, acc[k].y + g * v.y, acc[k].z + g * v.z, acc[k].w + g * v.w};
                    bs[k] += g;
                }
        }
    float* out = partial + (size_t)blockIdx.x * (n * C + n);
#pragma unroll
    for (int k = 0; k < MAXN; ++k) {
        if (k >= n) break;
        __syncthreads();
        red[threadIdx.x] = acc[k];
        redb[threadIdx.x] = bs[k];
        __syncthreads();
        if (threadIdx.x < cvn) {
            f4 s = {0.f, 0.f, 0.f, 0.f};
            float sb = 0.f;
            for (int q = 0; q < pl; ++q) {
                const f4 r = red[q * cvn + threadIdx.x];
                s = {s.x + r.x, s.y + r.y, s.z + r.z, s.w + r.w};
                sb += redb[q * cvn + threadIdx.x];
            }
            store4<float>(out + k * C + threadIdx.x * 4, s);
            if (threadIdx.x == 0) out[n * C + k] = sb;
        }
    }
}

extern "C" int mia_pixel_linear_fwd(int dtype, const void* x, int B, int HW, int C, int n, const float* Wt,
                                    const float* bias, float* logits, void* stream) {
    MIA_CHECK_ARG(x && Wt && logits && B > 0 && HW > 0, "mia_pixel_linear_fwd: bad arguments");
    MIA_CHECK_ARG(C > 0 && C % 4 == 0 && C <= SEG_MAXC && n > 0 && n <= SEG_MAXN,
                  "mia_pixel_linear_fwd: needs C %% 4 == 0, C <= %d, classes <= %d (got C=%d n=%d)", SEG_MAXC, SEG_MAXN, C, n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = grid_for((long long)B * HW, 4096);
    auto go = [&](auto nc_tag) {
        constexpr int NC_ = decltype(nc_tag)::value;
        if (dtype == MIA_BF16)
            MIA_LAUNCH((pixel_linear_fwd_kernel<bf16_t, NC_>), dim3(g), dim3(256), 0, s, (const bf16_t*)x, B, HW, C, n, Wt,
                               bias, logits);
        else
            MIA_LAUNCH((pixel_linear_fwd_kernel<float, NC_>), dim3(g), dim3(256), 0, s, (const float*)x, B, HW, C, n, Wt, bias,
                               logits);
    };
    if (n <= 8) go(std::integral_constant<int, 8>{});
    else go(std::integral_constant<int, 32>{});
    MIA_CHECK_LAUNCH("mia_pixel_linear_fwd");
    return 0;
}

extern "C" int mia_pixel_linear_bwd(int dtype, const void* x, const float* dlogits, int B, int HW, int C, int n,
                                    const float* Wt, void* dx, float* partial, int partial_rows, void* stream) {
    MIA_CHECK_ARG(x && dlogits && Wt && dx && partial && B > 0 && HW > 0 && partial_rows > 0 && partial_rows <= 65535,
                  "mia_pixel_linear_bwd: bad arguments");
    MIA_CHECK_ARG(C > 0 && C % 4 == 0 && C <= SEG_MAXC && n > 0 && n <= SEG_MAXN,
                  "mia_pixel_linear_bwd: needs C %% 4 == 0, C <= %d, classes <= %d (got C=%d n=%d)", SEG_MAXC, SEG_MAXN, C, n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const unsigned g = grid_for((long long)B * HW, 4096);
    auto go = [&](auto nc_tag) {
        constexpr int NC_ = decltype(nc_tag)::value;
        if (dtype == MIA_BF16) {
            MIA_LAUNCH((pixel_linear_bwd_data_kernel<bf16_t, NC_>), dim3(g), dim3(256), 0, s, dlogits, B, HW, C, n, Wt,
                               (bf16_t*)dx);
            MIA_LAUNCH((pixel_linear_bwd_weight_kernel<bf16_t, NC_>), dim3(partial_rows), dim3(256), 0, s, (const bf16_t*)x,
                               dlogits, B, HW, C, n, partial);
        } else {
            MIA_LAUNCH((pixel_linear_bwd_data_kernel<float, NC_>), dim3(g), dim3(256), 0, s, dlogits, B, HW, C, n, Wt,
                               (float*)dx);
            MIA_LAUNCH((pixel_linear_bwd_weight_kernel<float, NC_>), dim3(partial_rows), dim3(256), 0, s, (const float*)x,
                               dlogits, B, HW, C, n, partial);
        }
    };
    if (n <= 8) go(std::integral_constant<int, 8>{});
    else go(std::integral_constant<int, 32>{});
    MIA_CHECK_LAUNCH("mia_pixel_linear_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// channel dropout (Dropout2d): one keep/scale factor per (image, channel), from a counter-based hash
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_dropout_mask_kernel(float* mask, int n, float p, unsigned long long seed,
                                                                   const unsigned long long* seed_off) {
    if (seed_off) seed += seed_off[0] * 0x9E3779B97F4A7C15ull;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = (unsigned long long)i + seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.f;
}

extern "C" int mia_channel_dropout_mask(float* mask, int B, int C, float p, unsigned long long seed,
                                        const unsigned long long* seed_offset, void* stream) {
    MIA_CHECK_ARG(mask && B > 0 && C > 0 && p >= 0.f && p < 1.f, "mia_channel_dropout_mask: bad arguments");
    MIA_LAUNCH(channel_dropout_mask_kernel, dim3(cdiv((long long)B * C, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), mask, B * C, p, seed, seed_offset);
    MIA_CHECK_LAUNCH("mia_channel_dropout_mask");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Segmentation loss on NCHW fp32 logits [B, n, HW] and int64 targets [B, HW]:
//   loss = ce_w * CE(weighted mean) + ov_w * (1 - mean_{b,c} score[b,c])
//   score = Dice (2I+s)/(P+T+s) or IoU (I+s)/(P+T-I+s), I = sum p_c*[t==c], P = sum p_c, T = sum [t==c]
// pass 1: per-(image, split) partial sums [3n+2] = I[n], P[n], T[n], sum w[t]*(-log p_t), sum w[t]
// finalize (one workgroup): loss, and per-(image, class) coefficients of d score / d p = A*[t==c] + E
// pass 2: dlogits = upstream * (ce term + softmax Jacobian applied to the overlap term)
// ------------------------------------------------------------------------------------------------
template <int MAXN>
__device__ __forceinline__ void softmax_n(const float* z, int n, float* p, float& lse) {
    float m = z[0];
#pragma unroll
    for (int k = 1; k < MAXN; ++k)
        if (k < n) m = fmaxf(m, z[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXN; ++k) {
        p[k] = k < n ? expf(z[k] - m) : 0.f;
        s += p[k];
    }
    const float inv = 1.0f / s;
#pragma unroll
    for (int k = 0; k < MAXN; ++k) p[k] *= inv;
    lse = m + logf(s);
}

template <int MAXN>
__global__ __launch_bounds__(256) void seg_loss_sums_kernel(const float* logits, const long long* target, int HW, int n,
                                                            int splits, const float* cw, float* partial) {
    __shared__ float red[4][3 * MAXN + 2];
    const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int per = (HW + splits - 1) / splits;
    const int p0 = sp * per, p1 = (p0 + per) < HW ? (p0 + per) : HW;
    float I[MAXN], P[MAXN], Tn[MAXN], cen = 0.f, ced = 0.f;
#pragma unroll
    for (int k = 0; k < MAXN; ++k) I[k] = P[k] = Tn[k] = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) {
        float z[MAXN], pr[MAXN], lse;
#pragma unroll
        for (int k = 0; k < MAXN; ++k) z[k] = k < n ? logits[((size_t)b * n + k) * HW + p] : 0.f;
        softmax_n<MAXN>(z, n, pr, lse);
        const int t = (int)target[(size_t)b * HW + p];
        float zt = 0.f, wt = 1.f;
#pragma unroll
        for (int k = 0; k < MAXN; ++k)
            if (k < n) {
                const bool hit = (k == t);
                P[k] += pr[k];
                if (hit) { I[k] += pr[k]; Tn[k] += 1.f; zt = z[k]; if (cw) wt = cw[k]; }
            }
        if (t >= 0 && t < n) { cen += wt * (lse - zt); ced += wt; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < MAXN; ++k) {
        if (k >= n) break;
        const float a = wave_sum(I[k]), c = wave_sum(P[k]), d = wave_sum(Tn[k]);
        if (lane == 0) { red[wave][k] = a; red[wave][n + k] = c; red[wave][2 * n + k] = d; }
    }
    cen = wave_sum(cen);
    ced = wave_sum(ced);
    if (lane == 0) { red[wave][3 * n] = cen; red[wave][3 * n + 1] = ced; }
    __syncthreads();
    if (threadIdx.x < 3 * n + 2)
        partial[(size_t)blockIdx.x * (3 * n + 2) + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// coef layout: [0] = ce_w / sum w,  then A[B][n], E[B][n] (already scaled by -ov_w / (B*n))
__global__ __launch_bounds__(256) void seg_loss_finalize_kernel(const float* partial, int B, int n, int splits, int iou,
                                                                float smooth, float ce_w, float ov_w, float* loss,
                                                                float* coef) {
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int F = 3 * n + 2;
    double score_sum = 0.0;
    for (int i = tid; i < B * n; i += 256) {
        const int b = i / n, c = i % n;
        double I = 0.0, P = 0.0, T = 0.0;
        for (int s = 0; s < splits; ++s) {
            const float* r = partial + ((size_t)b * splits + s) * F;
            I += r[c]; P += r[n + c]; T += r[2 * n + c];
        }
        const double U = P + T, sm = smooth;
        double score, A, E;
        if (iou) {
            const double D = U - I + sm;
            score = (I + sm) / D;
            A = (U + 2.0 * sm) / (D * D);
            E = -(I + sm) / (D * D);
        } else {
            const double D = U + sm;
            score = (2.0 * I + sm) / D;
            A = 2.0 / D;
            E = -(2.0 * I + sm) / (D * D);
        }
        score_sum += score;
        const double k = -(double)ov_w / ((double)B * n);
        coef[1 + i] = (float)(k * A);
        coef[1 + B * n + i] = (float)(k * E);
    }
    sh[tid] = score_sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        double cen = 0.0, ced = 0.0;
        for (int r = 0; r < B * splits; ++r) { cen += partial[(size_t)r * F + 3 * n]; ced += partial[(size_t)r * F + 3 * n + 1]; }
        const double ce = ced > 0.0 ? cen / ced : 0.0;
        loss[0] = (float)(ce_w * ce + ov_w * (1.0 - sh[0] / ((double)B * n)));
        coef[0] = ced > 0.0 ? (float)(ce_w / ced) : 0.f;
    }
}

template <int MAXN>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* logits, const long long* target, int B, int HW,
                                                           int n, const float* cw, const float* coef,
                                                           const float* upstream, float* dlogits) {
    const float up = upstream ? upstream[0] : 1.f;
    const float ces = coef[0];
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        float z[MAXN], pr[MAXN], lse;
#pragma unroll
        for (int k = 0; k < MAXN; ++k) z[k] = k < n ? logits[((size_t)b * n + k) * HW + p] : 0.f;
        softmax_n<MAXN>(z, n, pr, lse);
        const int t = (int)target[i];
        const float wt = (t >= 0 && t < n) ? (cw ? cw[t] : 1.f) : 0.f;
        float G[MAXN], dot = 0.f;
#pragma unroll
        for (int k = 0; k < MAXN; ++k) {
            G[k] = 0.f;
            if (k < n) {
                G[k] = coef[1 + b * n + k] * (k == t ? 1.f : 0.f) + coef[1 + B * n + b * n + k];
                dot += pr[k] * G[k];
            }
        }
#pragma unroll
        for (int k = 0; k < MAXN; ++k)
            if (k < n) {
                const float ce_g = ces * wt * (pr[k] - (k == t ? 1.f : 0.f));
                dlogits[((size_t)b * n + k) * HW + p] = up * (ce_g + pr[k] * (G[k] - dot));
            }
    }
}

extern "C" int mia_seg_loss_fwd(const float* logits, const long long* target, int B, int HW, int n,
                                const float* class_weights, int iou, float smooth, float ce_weight, float overlap_weight,
                                int splits, float* partial, float* coef, float* loss, void* stream) {
    MIA_CHECK_ARG(logits && target && partial && coef && loss && B > 0 && HW > 0 && splits > 0 && splits <= HW,
                  "mia_seg_loss_fwd: bad arguments");
    MIA_CHECK_ARG(n > 0 && n <= SEG_MAXN, "mia_seg_loss_fwd: classes must be 1..%d (got %d)", SEG_MAXN, n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto go = [&](auto nc_tag) {
        constexpr int NC_ = decltype(nc_tag)::value;
        MIA_LAUNCH(seg_loss_sums_kernel<NC_>, dim3(B * splits), dim3(256), 0, s, logits, target, HW, n, splits,
                           class_weights, partial);
    };
    if (n <= 8) go(std::integral_constant<int, 8>{});
    else go(std::integral_constant<int, 32>{});
    MIA_LAUNCH(seg_loss_finalize_kernel, dim3(1), dim3(256), 0, s, partial, B, n, splits, iou, smooth, ce_weight,
                       overlap_weight, loss, coef);
    MIA_CHECK_LAUNCH("mia_seg_loss_fwd");
    return 0;
}

extern "C" int mia_seg_loss_bwd(const float* logits, const long long* target, int B, int HW, int n,
                                const float* class_weights, const float* coef, const float* upstream, float* dlogits,
                                void* stream) {
    MIA_CHECK_ARG(logits && target && coef && dlogits && B > 0 && HW > 0, "mia_seg_loss_bwd: bad arguments");
    MIA_CHECK_ARG(n > 0 && n <= SEG_MAXN, "mia_seg_loss_bwd: classes must be 1..%d (got %d)", SEG_MAXN, n);
    auto go = [&](auto nc_tag) {
        constexpr int NC_ = decltype(nc_tag)::value;
        MIA_LAUNCH(seg_loss_bwd_kernel<NC_>, dim3(grid_for((long long)B * HW, 4096)), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), logits, target, B, HW, n, class_weights, coef, upstream,
                           dlogits);
    };
    if (n <= 8) go(std::integral_constant<int, 8>{});
    else go(std::integral_constant<int, 32>{});
    MIA_CHECK_LAUNCH("mia_seg_loss_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// General softmax cross-entropy over strided fp32 logits (class logits [B,n] or pixel logits [B,n,HW]; element
// (image b, class k, position p) at b*sb + k*sk + p*sp) with optional class weights and focal modulation:
//   plain:  l_i = w[t_i] * ce_i,                     mean = sum l_i / sum w[t_i]
//   focal:  l_i = alpha * (1 - pt_i)^gamma * ce_i,   pt_i = exp(-ce_i),   mean = sum l_i / count
// Targets outside [0, n) are skipped. Any number of classes (the class loop recomputes instead of keeping arrays).
// ------------------------------------------------------------------------------------------------
struct CeElem {
    float ce, pt, lse;
    int t;
};
__device__ __forceinline__ CeElem ce_elem(const float* z, long long sk, int n, long long t) {
    float m = z[0];
    for (int k = 1; k < n; ++k) m = fmaxf(m, z[k * sk]);
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += expf(z[k * sk] - m);
    CeElem e;
    e.lse = m + logf(s);
    e.t = (t >= 0 && t < n) ? (int)t : -1;
    e.ce = e.t >= 0 ? e.lse - z[e.t * sk] : 0.f;
    e.pt = expf(-e.ce);
    return e;
}

__global__ __launch_bounds__(256) void ce_general_fwd_kernel(const float* logits, long long sb, long long sk, long long sp,
                                                             const long long* target, long long total, long long P, int n,
                                                             const float* cw, int focal, float alpha, float gamma,
                                                             float* per_elem, float* partial) {
    __shared__ float red[2][4];
    float num = 0.f, den = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const CeElem e = ce_elem(logits + (i / P) * sb + (i % P) * sp, sk, n, target[i]);
        float l = 0.f;
        if (e.t >= 0) {
            if (focal) {
                l = alpha * powf(fmaxf(1.f - e.pt, 0.f), gamma) * e.ce;
                den += 1.f;
            } else {
                const float w = cw ? cw[e.t] : 1.f;
                l = w * e.ce;
                den += w;
            }
        }
        num += l;
        if (per_elem) per_elem[i] = l;
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = num; red[1][threadIdx.x >> 6] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// out[0] = loss, out[1] = the factor every element's gradient carries (1/denominator for the mean, 1 for the sum)
__global__ __launch_bounds__(256) void ce_general_finalize_kernel(const float* partial, int parts, int mean, float* out) {
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256) { a += partial[i * 2]; b += partial[i * 2 + 1]; }
    sh[0][threadIdx.x] = a; sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double num = sh[0][0], den = sh[1][0];
        out[0] = (float)(mean ? (den > 0.0 ? num / den : 0.0) : num);
        out[1] = (float)(mean ? (den > 0.0 ? 1.0 / den : 0.0) : 1.0);
    }
}

// dlogits (same strides as logits) = upstream * scale * d l_i / d z;  upstream: one scalar, or one value per element
__global__ __launch_bounds__(256) void ce_general_bwd_kernel(const float* logits, long long sb, long long sk, long long sp,
                                                             const long long* target, long long total, long long P, int n,
                                                             const float* cw, int focal, float alpha, float gamma,
                                                             const float* scale, const float* upstream, int upstream_per_elem,
                                                             float* dlogits) {
    const float sc = scale ? scale[1] : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long off = (i / P) * sb + (i % P) * sp;
        const CeElem e = ce_elem(logits + off, sk, n, target[i]);
        float f = 0.f;
        if (e.t >= 0) {
            if (focal) {
                const float q = fmaxf(1.f - e.pt, 1e-12f);
                f = alpha * (powf(q, gamma) + gamma * e.pt * powf(q, gamma - 1.f) * e.ce);
            } else {
                f = cw ? cw[e.t] : 1.f;
            }
        }
        const float up = upstream ? upstream[upstream_per_elem ? i : 0] : 1.f;
        f *= sc * up;
        for (int k = 0; k < n; ++k) {
            const float p = expf(logits[off + k * sk] - e.lse);
            dlogits[off + k * sk] = f * (p - (k == e.t ? 1.f : 0.f));
        }
    }
}

extern "C" int mia_ce_general_fwd(const float* logits, long long stride_image, long long stride_class,
                                  long long stride_pos, const long long* target, long long images, long long positions,
                                  int n, const float* class_weights, int focal, float alpha, float gamma, int mean,
                                  float* per_elem, float* partial, int partial_rows, float* out, void* stream) {
    MIA_CHECK_ARG(logits && target && partial && out && images > 0 && positions > 0 && n > 0 && partial_rows > 0 &&
                      partial_rows <= 4096,
                  "mia_ce_general_fwd: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MIA_LAUNCH(ce_general_fwd_kernel, dim3(partial_rows), dim3(256), 0, s, logits, stride_image, stride_class,
                       stride_pos, target, images * positions, positions, n, class_weights, focal, alpha, gamma, per_elem,
                       partial);
    MIA_LAUNCH(ce_general_finalize_kernel, dim3(1), dim3(256), 0, s, partial, partial_rows, mean, out);
    MIA_CHECK_LAUNCH("mia_ce_general_fwd");
    return 0;
}

extern "C" int mia_ce_general_bwd(const float* logits, long long stride_image, long long stride_class,
                                  long long stride_pos, const long long* target, long long images, long long positions,
                                  int n, const float* class_weights, int focal, float alpha, float gamma,
                                  const float* out, const float* upstream, int upstream_per_elem, float* dlogits,
                                  void* stream) {
    MIA_CHECK_ARG(logits && target && dlogits && images > 0 && positions > 0 && n > 0, "mia_ce_general_bwd: bad arguments");
    MIA_LAUNCH(ce_general_bwd_kernel, dim3(grid_for(images * positions, 4096)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), logits, stride_image, stride_class, stride_pos, target,
                       images * positions, positions, n, class_weights, focal, alpha, gamma, out, upstream,
                       upstream_per_elem, dlogits);
    MIA_CHECK_LAUNCH("mia_ce_general_bwd");
    return 0;
}

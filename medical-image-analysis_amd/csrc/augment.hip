// Training-time augmentation on the device (gfx950), after the 8-bit resize of preprocess.hip and fused with the
// normalise step: flips -> rotation (bilinear, reflect-101 border) -> brightness/contrast -> 3x3 Gaussian blur ->
// (x - 255*mean) / (255*std) -> CHW fp32, the transform list of the reference's get_transforms(augment=True).
// Every stage rounds to 8 bits exactly where the uint8 pipeline of the reference does, so the result is a
// deterministic function of the image and the per-image parameters and can be held bit-exact to a CPU restatement.
// Floating-point contraction is off: the rotated coordinates must round the same way as in that restatement.
#include "common.h"

#pragma clang fp contract(off)

// per-image parameters, 9 doubles: hflip, vflip, rotate?, cos, sin, brightness/contrast?, alpha, beta*255, blur?
#define AUG_NP 9

// reflect-101 index (... 2 1 | 0 1 2 .. n-1 | n-2 n-3 ...) in closed form: no loop, any int in (garbage parameters
// cannot hang a wave)
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    int m = i % period;
    if (m < 0) m += period;
    return m < n ? m : period - m;
}

struct AugImg {
    const uint8_t* img;   // resized image [H][W][3]
    int H, W;
    bool hflip, vflip, rot, bc;
    double a, b, cx, cy;
    float alpha, beta;
};

// pixel (x, y) of the flipped image, in range
__device__ __forceinline__ int flipped(const AugImg& g, int x, int y, int c) {
    const int sx = g.hflip ? g.W - 1 - x : x, sy = g.vflip ? g.H - 1 - y : y;
    return g.img[((size_t)sy * g.W + sx) * 3 + c];
}

// value after flips, rotation and brightness/contrast at integer (x, y) inside the image
__device__ __forceinline__ void stage3(const AugImg& g, int x, int y, int out[3]) {
    if (g.rot) {
        const double dx = (double)x - g.cx, dy = (double)y - g.cy;
        const double sx = g.a * dx - g.b * dy + g.cx, sy = g.b * dx + g.a * dy + g.cy;
        const long long SX = llrint(sx * 32.0), SY = llrint(sy * 32.0);
        const int ix = (int)(SX >> 5), iy = (int)(SY >> 5), fx = (int)(SX & 31), fy = (int)(SY & 31);
        const int x0 = reflect101(ix, g.W), x1 = reflect101(ix + 1, g.W);
        const int y0 = reflect101(iy, g.H), y1 = reflect101(iy + 1, g.H);
        const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out[c] = (w00 * flipped(g, x0, y0, c) + w01 * flipped(g, x1, y0, c) + w10 * flipped(g, x0, y1, c) +
                      w11 * flipped(g, x1, y1, c) + 512) >> 10;
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = flipped(g, x, y, c);
    }
    if (g.bc) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)out[c];
            v = v * g.alpha;
            v = v + g.beta;
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
            out[c] = (int)v;   // truncation, as a uint8 look-up table built with astype() does
        }
    }
}

__global__ __launch_bounds__(256) void augment_normalize_kernel(const uint8_t* src, int N, int H, int W,
                                                                const double* params, float* dst, float m0, float m1,
                                                                float m2, float r0, float r1, float r2) {
    const long long total = (long long)N * H * W;
    const float mm[3] = {m0, m1, m2}, rr[3] = {r0, r1, r2};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((long long)W * H));
        const double* p = params + (size_t)n * AUG_NP;
        AugImg g;
        g.img = src + (size_t)n * H * W * 3;
        g.H = H; g.W = W;
        g.hflip = p[0] != 0.0; g.vflip = p[1] != 0.0; g.rot = p[2] != 0.0;
        g.a = p[3]; g.b = p[4];
        g.cx = (double)W * 0.5 - 0.5; g.cy = (double)H * 0.5 - 0.5;
        g.bc = p[5] != 0.0;
        g.alpha = (float)p[6]; g.beta = (float)p[7];
        int v[3];
        if (p[8] != 0.0) {
            int acc[3] = {0, 0, 0};
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    int t[3];
                    stage3(g, reflect101(x + dx, W), reflect101(y + dy, H), t);
                    const int wgt = (2 - (dx < 0 ? -dx : dx)) * (2 - (dy < 0 ? -dy : dy));
                    acc[0] += wgt * t[0]; acc[1] += wgt * t[1]; acc[2] += wgt * t[2];
                }
            v[0] = (acc[0] + 8) >> 4; v[1] = (acc[1] + 8) >> 4; v[2] = (acc[2] + 8) >> 4;
        } else {
            stage3(g, x, y, v);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float o = (float)v[c] - mm[c];
            o = o * rr[c];
            dst[(((size_t)n * 3 + c) * H + y) * W + x] = o;
        }
    }
}

// the geometric part for label masks: flips + nearest-neighbour rotation (reflect-101), int64 out
__global__ __launch_bounds__(256) void augment_mask_kernel(const uint8_t* src, int N, int H, int W, const double* params,
                                                           long long* dst) {
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / ((long long)W * H));
        const double* p = params + (size_t)n * AUG_NP;
        const bool hflip = p[0] != 0.0, vflip = p[1] != 0.0;
        int sx = x, sy = y;
        if (p[2] != 0.0) {
            const double cx = (double)W * 0.5 - 0.5, cy = (double)H * 0.5 - 0.5;
            const double dx = (double)x - cx, dy = (double)y - cy;
            const double fx = p[3] * dx - p[4] * dy + cx, fy = p[4] * dx + p[3] * dy + cy;
            sx = reflect101((int)((llrint(fx * 32.0) + 16) >> 5), W);
            sy = reflect101((int)((llrint(fy * 32.0) + 16) >> 5), H);
        }
        if (hflip) sx = W - 1 - sx;
        if (vflip) sy = H - 1 - sy;
        dst[i] = (long long)src[((size_t)n * H + sy) * W + sx];
    }
}

extern "C" int mia_augment_normalize_u8(const uint8_t* src, int N, int H, int W, const double* params, float* dst,
                                        const float* mean3, const float* std3, void* stream) {
    MIA_CHECK_ARG(src && params && dst && mean3 && std3 && N > 0 && H > 0 && W > 0, "mia_augment_normalize_u8: bad arguments");
    float m[3], r[3];
    for (int i = 0; i < 3; ++i) {
        volatile float mv = mean3[i] * 255.0f;
        volatile float sv = std3[i] * 255.0f;
        volatile float rv = 1.0f / sv;
        m[i] = mv;
        r[i] = rv;
    }
    long long blocks = ((long long)N * H * W + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    MIA_LAUNCH(augment_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, N, H, W, params, dst, m[0], m[1], m[2], r[0], r[1], r[2]);
    MIA_CHECK_LAUNCH("mia_augment_normalize_u8");
    return 0;
}

extern "C" int mia_augment_mask_u8(const uint8_t* src, int N, int H, int W, const double* params, long long* dst,
                                   void* stream) {
    MIA_CHECK_ARG(src && params && dst && N > 0 && H > 0 && W > 0, "mia_augment_mask_u8: bad arguments");
    long long blocks = ((long long)N * H * W + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    MIA_LAUNCH(augment_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src, N, H, W, params, dst);
    MIA_CHECK_LAUNCH("mia_augment_mask_u8");
    return 0;
}

// Multi-tensor AdamW (decoupled weight decay), one pass over (p, g, m, v): 28 bytes per parameter.
// Replaces torch.optim.AdamW.step as the reference configures it (train.py:193-197: lr 1e-3, weight_decay 1e-4).
//
// The arithmetic follows torch's single-tensor AdamW step operation by operation, in fp32:
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g  (as lerp);  v = b2*v + (1-b2)*g*g
//   denom = sqrt(v)/sqrt(1-b2^t) + eps;  p -= (lr/(1-b1^t)) * m/denom
//
// Up to MIA_ADAMW_CHUNK tensors travel in one launch's kernel arguments (pointers + sizes by value: no device-side
// table to allocate or upload). A workgroup owns BLOCK_ELEMS consecutive elements of one tensor; it finds its tensor
// by scanning the chunk's block prefix.
#include "common.h"

#define MIA_ADAMW_CHUNK 48
#define MIA_ADAMW_BLOCK_ELEMS 4096

struct AdamWChunk {
    float* p[MIA_ADAMW_CHUNK];
    const float* g[MIA_ADAMW_CHUNK];
    float* m[MIA_ADAMW_CHUNK];
    float* v[MIA_ADAMW_CHUNK];
    long long n[MIA_ADAMW_CHUNK];
    int first_block[MIA_ADAMW_CHUNK + 1];
    int count;
    float lr_wd_keep;   // 1 - lr*wd
    float beta1, beta2, eps;
    float omb1, omb2;   // 1 - beta, formed in double on the host
    float step_size;    // lr / (1 - beta1^t)
    float inv_sqrt_bc2; // 1 / sqrt(1 - beta2^t)
    float grad_scale;   // gradients are multiplied by this first (1 = none)
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamWChunk c, const float* __restrict__ dev_hyper) {
    // step-dependent factors from device memory when given: a captured graph replays fixed kernel arguments.
    // (Kept in locals: assigning to fields of `c` would force a private copy of the whole 2 KB argument block in
    // every lane -- measured at 73 us instead of 13 us a launch.)
    float lr_wd_keep = c.lr_wd_keep, step_size = c.step_size, inv_sqrt_bc2 = c.inv_sqrt_bc2;
    if (dev_hyper) {
        lr_wd_keep = dev_hyper[0];
        step_size = dev_hyper[1];
        inv_sqrt_bc2 = dev_hyper[2];
    }
    // wave-uniform scan: which tensor does this block belong to
    int t = 0;
    while (t + 1 < c.count && (int)blockIdx.x >= c.first_block[t + 1]) ++t;
    const long long base = (long long)((int)blockIdx.x - c.first_block[t]) * MIA_ADAMW_BLOCK_ELEMS;
    const long long n = c.n[t];
    float* __restrict__ p = c.p[t];
    const float* __restrict__ g = c.g[t];
    float* __restrict__ m = c.m[t];
    float* __restrict__ v = c.v[t];
    const float omb1 = c.omb1, omb2 = c.omb2;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= c.grad_scale;
        pp *= lr_wd_keep;
        mm = mm + omb1 * (gg - mm);          // lerp(m, g, 1-b1)
        vv = vv * c.beta2 + omb2 * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + c.eps;
        pp -= step_size * (mm / denom);
    };
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
#pragma unroll
    for (int it = 0; it < MIA_ADAMW_BLOCK_ELEMS / (256 * 4); ++it) {
        const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            upd(pv.x, gv.x, mv.x, vv.x);
            upd(pv.y, gv.y, mv.y, vv.y);
            upd(pv.z, gv.z, mv.z, vv.z);
            upd(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (long long j = i; j < n && j < i + 4; ++j) upd(p[j], g[j], m[j], v[j]);
        }
    }
}

// params/grads/exp_avg/exp_avg_sq: HOST arrays of n_tensors DEVICE pointers (fp32, contiguous); sizes: host array of
// element counts. step is the 1-based step number of this update.
// dev_hyper (nullable): 3 floats in DEVICE memory {1 - lr*weight_decay, lr/(1 - beta1^step), 1/sqrt(1 - beta2^step)}
// read at run time in place of the values derived from lr/weight_decay/step, for launches replayed from a HIP graph
// (the caller rewrites them before each replay, e.g. with mia_set_words).
extern "C" int mia_adamw_step(int n_tensors, const void* const* params, const void* const* grads,
                              const void* const* exp_avg, const void* const* exp_avg_sq, const long long* sizes,
                              double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                              double grad_scale, const float* dev_hyper, void* stream) {
    MIA_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && sizes)),
                  "mia_adamw_step: null table");
    MIA_CHECK_ARG(step >= 1, "mia_adamw_step: step must be >= 1 (got %d)", step);
    MIA_CHECK_ARG(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && lr >= 0.,
                  "mia_adamw_step: bad hyper-parameters");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamWChunk c;
    // hyper-parameters arrive as doubles (Python floats) and are rounded to fp32 once, after the derived
    // quantities are formed: 1 - 0.999 in fp32 would be off by 1e-5 relative
    c.lr_wd_keep = (float)(1.0 - lr * weight_decay);
    c.beta1 = (float)beta1; c.beta2 = (float)beta2; c.eps = (float)eps;
    c.omb1 = (float)(1.0 - beta1); c.omb2 = (float)(1.0 - beta2);
    c.step_size = (float)(lr / bc1);
    c.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    c.grad_scale = (float)grad_scale;
    int i = 0;
    while (i < n_tensors) {
        c.count = 0;
        int blocks = 0;
        while (i < n_tensors && c.count < MIA_ADAMW_CHUNK) {
            MIA_CHECK_ARG(sizes[i] >= 0 && params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i],
                          "mia_adamw_step: tensor %d has a null pointer or a negative size", i);
            if (sizes[i] > 0) {
                const int k = c.count++;
                c.p[k] = (float*)params[i]; c.g[k] = (const float*)grads[i];
                c.m[k] = (float*)exp_avg[i]; c.v[k] = (float*)exp_avg_sq[i];
                c.n[k] = sizes[i];
                c.first_block[k] = blocks;
                blocks += cdiv(sizes[i], MIA_ADAMW_BLOCK_ELEMS);
            }
            ++i;
        }
        if (c.count == 0) continue;
        c.first_block[c.count] = blocks;
        MIA_LAUNCH(adamw_kernel, dim3(blocks), dim3(256), 0, s, c, dev_hyper);
        MIA_CHECK_LAUNCH("mia_adamw_step");
    }
    return 0;
}

// ---- small by-value upload: n 32-bit words travel in the kernel arguments and are stored to dst by one wave.
// Unlike an asynchronous copy from host memory, the values are fixed when the call returns, so the caller may
// overwrite its buffer at once; used to refresh the per-step words a replayed graph reads (seed offset, dev_hyper).
#define MIA_SET_WORDS_MAX 64
struct WordPack {
    unsigned w[MIA_SET_WORDS_MAX];
};

__global__ __launch_bounds__(64) void set_words_kernel(unsigned* __restrict__ dst, int n, WordPack v) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = v.w[i];
}

extern "C" int mia_set_words(void* dst, int n, const unsigned* host_words, void* stream) {
    MIA_CHECK_ARG(n >= 0 && n <= MIA_SET_WORDS_MAX, "mia_set_words: n must be 0..%d (got %d)", MIA_SET_WORDS_MAX, n);
    if (n == 0) return 0;
    MIA_CHECK_ARG(dst && host_words, "mia_set_words: null pointer");
    WordPack v;
    for (int i = 0; i < MIA_SET_WORDS_MAX; ++i) v.w[i] = i < n ? host_words[i] : 0u;
    MIA_LAUNCH(set_words_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (unsigned*)dst, n,
                       v);
    MIA_CHECK_LAUNCH("mia_set_words");
    return 0;
}

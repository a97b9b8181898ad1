// Bandwidth-bound BatchNorm / squeeze-excite / elementwise kernels on NHWC [rows, C] matrices (gfx950).
//
// Train-mode BatchNorm is split in two: the producing conv/GEMM kernel emits per-workgroup partial
// sums (rows of a [P][2][C] fp32 slab, fixed summation order => bit-reproducible), a one-wave-per-
// channel finalize kernel turns them into the per-channel affine (scale, shift) that the CONSUMER
// applies while loading, so a normalised tensor is never written to HBM. The backward mirrors it:
// reduce kernels emit partials of sum(g) and sum(g*xhat), finalize emits the three coefficients of
// dy = ca*g + cb*y + cd.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// thread -> (row-thread, 4-channel vector) mapping shared by the row-streaming kernels
// ------------------------------------------------------------------------------------------------
struct RowMap {
    int CB, nvec, npt, c0, cbv, cv, pt, c;
    bool valid;
};
// channel blocking: up to 1024 channels (256 vectors) per workgroup column
static inline void host_colblock(int C, int* ncb, int* CB) {
    *ncb = cdiv(C, 1024);
    *CB = cdiv(cdiv(C, 4), *ncb) * 4;
}
__device__ __forceinline__ RowMap make_rowmap(int C, int CB, int block_col, int tid) {
    RowMap m;
    m.CB = CB;
    m.nvec = CB / 4;
    m.npt = 256 / m.nvec;
    m.c0 = block_col * CB;
    m.cbv = (C - m.c0) < CB ? (C - m.c0) : CB;
    m.cv = tid % m.nvec;
    m.pt = tid / m.nvec;
    m.c = m.c0 + m.cv * 4;
    m.valid = (m.cv * 4 < m.cbv) && (m.pt < m.npt);
    return m;
}
__device__ __forceinline__ f22 f22_zero() { return f22{v2f{0.f, 0.f}, v2f{0.f, 0.f}}; }
__device__ __forceinline__ f22 f22_splat(float a) { return f22{v2f{a, a}, v2f{a, a}}; }

// sum the per-thread f4 pairs over row-threads and write channel sums to outA[c0 + ch] (and outB)
__device__ __forceinline__ void block_colsum2(float* red, const RowMap& m, int tid, f4 s, f4 q, float* outA,
                                              float* outB) {
    __syncthreads();
    red[tid * 8 + 0] = s.x; red[tid * 8 + 1] = s.y; red[tid * 8 + 2] = s.z; red[tid * 8 + 3] = s.w;
    red[tid * 8 + 4] = q.x; red[tid * 8 + 5] = q.y; red[tid * 8 + 6] = q.z; red[tid * 8 + 7] = q.w;
    __syncthreads();
    for (int ch = tid; ch < m.cbv; ch += 256) {
        const int cv = ch >> 2, e = ch & 3;
        float a = 0.f, b = 0.f;
        for (int pt = 0; pt < m.npt; ++pt) {
            a += red[(pt * m.nvec + cv) * 8 + e];
            b += red[(pt * m.nvec + cv) * 8 + 4 + e];
        }
        outA[m.c0 + ch] = a;
        if (outB) outB[m.c0 + ch] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize (forward): partials -> scale/shift/mean/rstd, running-stat update
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* part, int P, int C, double count,
                                                              const float* gamma, const float* beta, float* rmean,
                                                              float* rvar, long long* nbt, float momentum, float eps,
                                                              int training, const float* conv_bias,
                                                              float* scale, float* shift, float* mean, float* rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    // conv_bias: the normalised tensor is (y + conv_bias[c]) but only the bias-free y is stored; the returned
    // mean/shift are relative to y, the running mean tracks y + bias
    const float cb = conv_bias ? conv_bias[c] : 0.f;
    float mu, var;
    if (training) {
        double s = 0.0, q = 0.0;
        for (int r = lane; r < P; r += 64) {
            s += (double)part[((size_t)r * 2 + 0) * C + c];
            q += (double)part[((size_t)r * 2 + 1) * C + c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        const double m = s / count;
        double v = q / count - m * m;
        if (v < 0.0) v = 0.0;
        mu = (float)m;
        var = (float)v;
        if (lane == 0 && rmean) {
            const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (mu + cb);
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
            if (c == 0 && nbt) *nbt += 1;
        }
    } else {
        mu = rmean[c] - cb;
        var = rvar[c];
    }
    if (lane == 0) {
        const float rs = 1.0f / sqrtf(var + eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        scale[c] = g * rs;
        shift[c] = b - mu * g * rs;
        mean[c] = mu;
        rstd[c] = rs;
    }
}

extern "C" int mia_bn_finalize_fwd(const float* part, int P, int C, double count, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, float momentum, float eps, int training,
                                   const float* conv_bias, float* scale, float* shift, float* mean, float* rstd,
                                   void* stream) {
    MIA_CHECK_ARG(C > 0 && scale && shift && mean && rstd, "mia_bn_finalize_fwd: bad arguments");
    MIA_CHECK_ARG(!training || (part && P > 0 && count > 0), "mia_bn_finalize_fwd: training needs partials");
    MIA_CHECK_ARG(training || (running_mean && running_var), "mia_bn_finalize_fwd: eval needs running stats");
    MIA_LAUNCH(bn_finalize_fwd_kernel, dim3(cdiv(C, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       part, P, C, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                       training, conv_bias, scale, shift, mean, rstd);
    MIA_CHECK_LAUNCH("mia_bn_finalize_fwd");
    return 0;
}

// backward finalize: partials of (sum g, sum g*xhat) -> dgamma, dbeta and dy = ca*g + cb*y + cd
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* part, int P, int C, double count,
                                                              const float* gamma, const float* mean, const float* rstd,
                                                              int training, float* dgamma, float* dbeta, float* ca,
                                                              float* cb, float* cd, float* dconv_bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int r = lane; r < P; r += 64) {
        s += (double)part[((size_t)r * 2 + 0) * C + c];
        q += (double)part[((size_t)r * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) {
        const double g = gamma ? (double)gamma[c] : 1.0;
        const double rs = rstd[c], mu = mean[c];
        if (dgamma) dgamma[c] = (float)q;
        if (dbeta) dbeta[c] = (float)s;
        // gradient of a bias added before the normalisation = column sum of dy: exactly zero with batch statistics
        if (dconv_bias) dconv_bias[c] = training ? 0.f : (float)(g * rs * s);
        if (training) {
            ca[c] = (float)(g * rs);
            cb[c] = (float)(-g * rs * rs * q / count);
            cd[c] = (float)(g * rs * (-s / count + mu * rs * q / count));
        } else {
            ca[c] = (float)(g * rs);
            cb[c] = 0.f;
            cd[c] = 0.f;
        }
    }
}

extern "C" int mia_bn_finalize_bwd(const float* part, int P, int C, double count, const float* gamma,
                                   const float* mean, const float* rstd, int training, float* dgamma, float* dbeta,
                                   float* ca, float* cb, float* cd, float* dconv_bias, void* stream) {
    MIA_CHECK_ARG(part && P > 0 && C > 0 && count > 0 && mean && rstd && ca && cb && cd,
                  "mia_bn_finalize_bwd: bad arguments");
    MIA_LAUNCH(bn_finalize_bwd_kernel, dim3(cdiv(C, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       part, P, C, count, gamma, mean, rstd, training, dgamma, dbeta, ca, cb, cd, dconv_bias);
    MIA_CHECK_LAUNCH("mia_bn_finalize_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// backward reduce: g = (G * gate[b] + dpool[b]) * act'(y*scale+shift), written to Out (which may be G itself);
// partials sum(g), sum(g * (y-mean)*rstd) per channel
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const T* G, T* Out, const T* Y, long long R, int C, int CB,
                                                                const float* scale, const float* shift,
                                                                const float* mean, const float* rstd, int act,
                                                                const float* gate, const float* dpool,
                                                                int rows_per_img, float* part) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const RowMap m = make_rowmap(C, CB, blockIdx.y, tid);
    f4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    if (m.valid) {
        const f22 sc = load22<float>(scale + m.c), sh = load22<float>(shift + m.c);
        const f22 mu = load22<float>(mean + m.c), rs = load22<float>(rstd + m.c);
        const bool writes = (act != MIA_ACT_NONE) || gate || dpool;
        const long long step = (long long)gridDim.x * m.npt;
        f22 s2 = f22_zero(), q2 = f22_zero();
        auto body = [&](long long r, f22 g, const f22 y) {
            if (gate) {
                const f22 gt = load22<float>(gate + (size_t)((unsigned)r / (unsigned)rows_per_img) * C + m.c);
                g.lo *= gt.lo; g.hi *= gt.hi;
            }
            if (dpool) {
                const f22 dp = load22<float>(dpool + (size_t)((unsigned)r / (unsigned)rows_per_img) * C + m.c);
                g.lo += dp.lo; g.hi += dp.hi;
            }
            if (act == MIA_ACT_SILU) {
                g.lo *= silu_grad2(y.lo * sc.lo + sh.lo);
                g.hi *= silu_grad2(y.hi * sc.hi + sh.hi);
            } else if (act == MIA_ACT_RELU) {
                const v2f zl = y.lo * sc.lo + sh.lo, zh = y.hi * sc.hi + sh.hi;
                g.lo = v2f{zl.x > 0.f ? g.lo.x : 0.f, zl.y > 0.f ? g.lo.y : 0.f};
                g.hi = v2f{zh.x > 0.f ? g.hi.x : 0.f, zh.y > 0.f ? g.hi.y : 0.f};
            }
            s2.lo += g.lo; s2.hi += g.hi;
            q2.lo += g.lo * ((y.lo - mu.lo) * rs.lo);
            q2.hi += g.hi * ((y.hi - mu.hi) * rs.hi);
            if (writes) store22<T>(Out + (size_t)r * C + m.c, g);
        };
        long long r = (long long)blockIdx.x * m.npt + m.pt;
        // 4 rows per iteration: 8 independent vector loads in flight per lane
        for (; r + 3 * step < R; r += 4 * step) {
            f22 g[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g[u] = load22<T>(G + (size_t)(r + u * step) * C + m.c);
                y[u] = load22<T>(Y + (size_t)(r + u * step) * C + m.c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) body(r + u * step, g[u], y[u]);
        }
        for (; r < R; r += step) body(r, load22<T>(G + (size_t)r * C + m.c), load22<T>(Y + (size_t)r * C + m.c));
        s = to_f4(s2); q = to_f4(q2);
    }
    block_colsum2(red, m, tid, s, q, part + ((size_t)blockIdx.x * 2 + 0) * C, part + ((size_t)blockIdx.x * 2 + 1) * C);
}

extern "C" int mia_bn_act_bwd_reduce(int dtype, const void* G, void* Out, const void* Y, long long R, int C, const float* scale,
                                     const float* shift, const float* mean, const float* rstd, int act,
                                     const float* gate, const float* dpool, int rows_per_img, float* part,
                                     int part_rows, void* stream) {
    MIA_CHECK_ARG(G && Y && R > 0 && C > 0 && C % 4 == 0 && scale && shift && mean && rstd && part,
                  "mia_bn_act_bwd_reduce: bad arguments");
    MIA_CHECK_ARG(Out || !(act != MIA_ACT_NONE || gate || dpool),
                  "mia_bn_act_bwd_reduce: an activation, gate or dpool changes the gradient: Out must be given");
    MIA_CHECK_ARG(part_rows > 0 && part_rows <= 65535, "mia_bn_act_bwd_reduce: bad part_rows");
    MIA_CHECK_ARG(!(gate || dpool) || rows_per_img > 0, "mia_bn_act_bwd_reduce: gate/dpool need rows_per_img");
    MIA_CHECK_ARG(R < (1LL << 32), "mia_bn_act_bwd_reduce: too many rows");
    int ncb, CB;
    host_colblock(C, &ncb, &CB);
    dim3 grid(part_rows, ncb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (rows_per_img <= 0) rows_per_img = 1;
    if (dtype == MIA_BF16)
        MIA_LAUNCH(bn_act_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)G, (bf16_t*)Out, (const bf16_t*)Y, R, C,
                           CB, scale, shift, mean, rstd, act, gate, dpool, rows_per_img, part);
    else
        MIA_LAUNCH(bn_act_bwd_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)G, (float*)Out, (const float*)Y, R, C, CB,
                           scale, shift, mean, rstd, act, gate, dpool, rows_per_img, part);
    MIA_CHECK_LAUNCH("mia_bn_act_bwd_reduce");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// elementwise: out = a[c]*x + b[c]*y + d[c]   (y, b optional)  then act, then + res
//   BN forward apply:  a=scale, d=shift               (act, residual optional)
//   BN backward apply: a=ca, b=cb, d=cd with x=g, y=raw forward tensor
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void affine2_kernel(const T* X, const T* Y, const T* Res, T* Out, long long nvec,
                                                      int C, const float* a, const float* b, const float* d, int act) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const size_t idx = (size_t)i * 4;
        const int c = (int)((unsigned)i % (unsigned)(C >> 2)) << 2;   // 32-bit modulo (host checks the range)
        const f22 x = load22<T>(X + idx);
        const f22 av = load22<float>(a + c), dv = load22<float>(d + c);
        f22 o = {x.lo * av.lo + dv.lo, x.hi * av.hi + dv.hi};
        if (Y) {
            const f22 y = load22<T>(Y + idx);
            const f22 bv = load22<float>(b + c);
            o.lo += y.lo * bv.lo; o.hi += y.hi * bv.hi;
        }
        if (act == MIA_ACT_SILU) {
            o.lo = silu2(o.lo); o.hi = silu2(o.hi);
        } else if (act == MIA_ACT_RELU) {
            o.lo = __builtin_elementwise_max(o.lo, v2f{0.f, 0.f}); o.hi = __builtin_elementwise_max(o.hi, v2f{0.f, 0.f});
        }
        if (Res) {
            const f22 r = load22<T>(Res + idx);
            o.lo += r.lo; o.hi += r.hi;
        }
        store22<T>(Out + idx, o);
    }
}

// the same, eight channels per thread (C % 8 == 0): one 16-byte access per lane and operand for bf16
template <typename T>
__global__ __launch_bounds__(256) void affine2_wide_kernel(const T* X, const T* Y, const T* Res, T* Out, unsigned nvec8,
                                                           int C, const float* a, const float* b, const float* d,
                                                           int act, int fixed_c) {
    const unsigned cvec = (unsigned)C >> 3;
    if (fixed_c) {
        // grid stride = a multiple of the vectors per row (host): fixed channels per thread, coefficients in registers
        unsigned i = blockIdx.x * 256 + threadIdx.x;
        const int c = (int)(i % cvec) << 3;
        const f22 zero = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
        const f22 a0 = load22<float>(a + c), a1 = load22<float>(a + c + 4);
        const f22 d0 = load22<float>(d + c), d1 = load22<float>(d + c + 4);
        const f22 b0 = Y ? load22<float>(b + c) : zero, b1 = Y ? load22<float>(b + c + 4) : zero;
        for (; i < nvec8; i += gridDim.x * 256) {
            const size_t idx = (size_t)i * 8;
            f22 x0, x1;
            load44<T>(X + idx, x0, x1);
            f22 o0 = {x0.lo * a0.lo + d0.lo, x0.hi * a0.hi + d0.hi}, o1 = {x1.lo * a1.lo + d1.lo, x1.hi * a1.hi + d1.hi};
            if (Y) {
                f22 y0, y1;
                load44<T>(Y + idx, y0, y1);
                o0.lo += y0.lo * b0.lo; o0.hi += y0.hi * b0.hi;
                o1.lo += y1.lo * b1.lo; o1.hi += y1.hi * b1.hi;
            }
            if (act == MIA_ACT_SILU) {
                o0.lo = silu2(o0.lo); o0.hi = silu2(o0.hi); o1.lo = silu2(o1.lo); o1.hi = silu2(o1.hi);
            } else if (act == MIA_ACT_RELU) {
                const v2f z = {0.f, 0.f};
                o0.lo = __builtin_elementwise_max(o0.lo, z); o0.hi = __builtin_elementwise_max(o0.hi, z);
                o1.lo = __builtin_elementwise_max(o1.lo, z); o1.hi = __builtin_elementwise_max(o1.hi, z);
            }
            if (Res) {
                f22 r0, r1;
                load44<T>(Res + idx, r0, r1);
                o0.lo += r0.lo; o0.hi += r0.hi; o1.lo += r1.lo; o1.hi += r1.hi;
            }
            store44<T>(Out + idx, o0, o1);
        }
        return;
    }
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < nvec8; i += gridDim.x * 256) {
        const size_t idx = (size_t)i * 8;
        const int c = (int)(i % cvec) << 3;
        f22 x0, x1;
        load44<T>(X + idx, x0, x1);
        const f22 a0 = load22<float>(a + c), a1 = load22<float>(a + c + 4);
        const f22 d0 = load22<float>(d + c), d1 = load22<float>(d + c + 4);
        f22 o0 = {x0.lo * a0.lo + d0.lo, x0.hi * a0.hi + d0.hi}, o1 = {x1.lo * a1.lo + d1.lo, x1.hi * a1.hi + d1.hi};
        if (Y) {
            f22 y0, y1;
            load44<T>(Y + idx, y0, y1);
            const f22 b0 = load22<float>(b + c), b1 = load22<float>(b + c + 4);
            o0.lo += y0.lo * b0.lo; o0.hi += y0.hi * b0.hi;
            o1.lo += y1.lo * b1.lo; o1.hi += y1.hi * b1.hi;
        }
        if (act == MIA_ACT_SILU) {
            o0.lo = silu2(o0.lo); o0.hi = silu2(o0.hi); o1.lo = silu2(o1.lo); o1.hi = silu2(o1.hi);
        } else if (act == MIA_ACT_RELU) {
            const v2f z = {0.f, 0.f};
            o0.lo = __builtin_elementwise_max(o0.lo, z); o0.hi = __builtin_elementwise_max(o0.hi, z);
            o1.lo = __builtin_elementwise_max(o1.lo, z); o1.hi = __builtin_elementwise_max(o1.hi, z);
        }
        if (Res) {
            f22 r0, r1;
            load44<T>(Res + idx, r0, r1);
            o0.lo += r0.lo; o0.hi += r0.hi; o1.lo += r1.lo; o1.hi += r1.hi;
        }
        store44<T>(Out + idx, o0, o1);
    }
}

extern "C" int mia_affine2(int dtype, const void* X, const void* Y, const void* Res, void* Out, long long R, int C,
                           const float* a, const float* b, const float* d, int act, void* stream) {
    MIA_CHECK_ARG(X && Out && a && d && R > 0 && C > 0 && C % 4 == 0, "mia_affine2: bad arguments");
    MIA_CHECK_ARG(!Y || b, "mia_affine2: Y needs coefficient b");
    const long long nvec = R * C / 4;
    MIA_CHECK_ARG(nvec < (1LL << 32), "mia_affine2: tensor too large");
    long long blocks = (nvec + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool aligned16 = (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)Res | (uintptr_t)Out) & 15) == 0;
    if (C % 8 == 0 && aligned16) {
        const long long nvec8 = nvec / 2;
        long long wblocks = (nvec8 + 255) / 256;
        if (wblocks > 16384) wblocks = 16384;
        int fixed_c = 0;
        {
            const long long cvec = C / 8;
            long long a_ = cvec, b_ = 256;
            while (b_) { const long long t = a_ % b_; a_ = b_; b_ = t; }
            const long long unit = cvec / a_;
            if (unit <= wblocks) { wblocks = wblocks / unit * unit; fixed_c = 1; }
        }
        if (dtype == MIA_BF16)
            MIA_LAUNCH(affine2_wide_kernel<bf16_t>, dim3((unsigned)wblocks), dim3(256), 0, s, (const bf16_t*)X,
                               (const bf16_t*)Y, (const bf16_t*)Res, (bf16_t*)Out, (unsigned)nvec8, C, a, b, d, act, fixed_c);
        else
            MIA_LAUNCH(affine2_wide_kernel<float>, dim3((unsigned)wblocks), dim3(256), 0, s, (const float*)X,
                               (const float*)Y, (const float*)Res, (float*)Out, (unsigned)nvec8, C, a, b, d, act, fixed_c);
        MIA_CHECK_LAUNCH("mia_affine2");
        return 0;
    }
    if (dtype == MIA_BF16)
        MIA_LAUNCH(affine2_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)X,
                           (const bf16_t*)Y, (const bf16_t*)Res, (bf16_t*)Out, nvec, C, a, b, d, act);
    else
        MIA_LAUNCH(affine2_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)X,
                           (const float*)Y, (const float*)Res, (float*)Out, nvec, C, a, b, d, act);
    MIA_CHECK_LAUNCH("mia_affine2");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-image channel sums over H*W rows:  out[b][split][c] = sum_rows W(r,c) * silu(y*scale+shift)
//   W = 1           squeeze-excite pooling (forward)
//   W = dZ[r,c]     gradient w.r.t. the gate (backward)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void img_colsum_kernel(const T* Y, const T* Wt, int HW, int C, int CB, int splits,
                                                         const float* scale, const float* shift, int act, float* out) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const RowMap m = make_rowmap(C, CB, blockIdx.y, tid);
    const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int rows_per_split = (HW + splits - 1) / splits;
    const int r_begin = sp * rows_per_split;
    const int r_end = (r_begin + rows_per_split) < HW ? (r_begin + rows_per_split) : HW;
    f4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    if (m.valid) {
        f22 sc = f22_splat(1.f), sh = f22_zero();
        if (scale) { sc = load22<float>(scale + m.c); sh = load22<float>(shift + m.c); }
        const T* yb = Y + (size_t)b * HW * C + m.c;
        const T* wb = Wt ? Wt + (size_t)b * HW * C + m.c : nullptr;
        f22 s2 = f22_zero();
        auto body = [&](f22 y, const f22 w) {
            y.lo = y.lo * sc.lo + sh.lo; y.hi = y.hi * sc.hi + sh.hi;
            if (act == MIA_ACT_SILU) { y.lo = silu2(y.lo); y.hi = silu2(y.hi); }
            if (wb) { y.lo *= w.lo; y.hi *= w.hi; }
            s2.lo += y.lo; s2.hi += y.hi;
        };
        int r = r_begin + m.pt;
        // 4 independent rows per iteration keep several loads in flight per lane
        for (; r + 3 * m.npt < r_end; r += 4 * m.npt) {
            f22 y[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                y[u] = load22<T>(yb + (size_t)(r + u * m.npt) * C);
                w[u] = wb ? load22<T>(wb + (size_t)(r + u * m.npt) * C) : f22_zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) body(y[u], w[u]);
        }
        for (; r < r_end; r += m.npt)
            body(load22<T>(yb + (size_t)r * C), wb ? load22<T>(wb + (size_t)r * C) : f22_zero());
        s = to_f4(s2);
    }
    block_colsum2(red, m, tid, s, q, out + ((size_t)b * splits + sp) * C, nullptr);
}

extern "C" int mia_img_colsum(int dtype, const void* Y, const void* Wt, int B, int HW, int C, const float* scale,
                              const float* shift, int act, int splits, float* out, void* stream) {
    MIA_CHECK_ARG(Y && out && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && splits > 0 && splits <= HW,
                  "mia_img_colsum: bad arguments");
    MIA_CHECK_ARG(!(scale == nullptr) == !(shift == nullptr), "mia_img_colsum: scale/shift must come together");
    // at most 256 channels (64 vectors = one wave per row segment) per workgroup column: >= 4 row-threads
    const int ncb = cdiv(C, 256);
    const int CB = cdiv(cdiv(C, 4), ncb) * 4;
    dim3 grid(B * splits, ncb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(img_colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)Y, (const bf16_t*)Wt, HW, C,
                           CB, splits, scale, shift, act, out);
    else
        MIA_LAUNCH(img_colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)Y, (const float*)Wt, HW, C, CB,
                           splits, scale, shift, act, out);
    MIA_CHECK_LAUNCH("mia_img_colsum");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of  a = silu(z) * gate[b],  z = scale*y + shift  (BatchNorm + SiLU + squeeze-excite), given dZ = dL/da
// summed over... nothing yet: every per-channel sum the backward needs is a per-image sum of products of dZ, y:
//   q0 = sum dZ*silu(z)            gradient w.r.t. the gate
//   q1 = sum silu'(z)*dZ           q2 = sum silu'(z)*dZ*xhat
//   q3 = sum silu'(z)              q4 = sum silu'(z)*xhat          xhat = (y-mean)*rstd
// With g = silu'(z)*(dZ*gate[b] + dpool[b]) (the gradient at the BatchNorm output) this gives
//   sum g = sum_b gate[b]*q1[b] + dpool[b]*q3[b],   sum g*xhat = sum_b gate[b]*q2[b] + dpool[b]*q4[b]
// so ONE pass over (y, dZ) feeds both the squeeze-excite backward and the BatchNorm backward.
// out is [5][B][splits][C].
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void se_bwd_sums_kernel(const T* Y, const T* dZ, int B, int HW, int C, int CB,
                                                          int splits, const float* scale, const float* shift,
                                                          const float* mean, const float* rstd, float* out,
                                                          const float* gate) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x;
    const RowMap m = make_rowmap(C, CB, blockIdx.y, tid);
    const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int rows_per_split = (HW + splits - 1) / splits;
    const int r_begin = sp * rows_per_split;
    const int r_end = (r_begin + rows_per_split) < HW ? (r_begin + rows_per_split) : HW;
    f22 a0 = f22_zero(), a1 = a0, a2 = a0, a3 = a0, a4 = a0;
    if (m.valid) {
        const f22 sc = load22<float>(scale + m.c), sh = load22<float>(shift + m.c);
        const f22 mu = load22<float>(mean + m.c), rs = load22<float>(rstd + m.c);
        const T* yb = Y + (size_t)b * HW * C + m.c;
        const T* wb = dZ + (size_t)b * HW * C + m.c;
        auto half = [&](const v2f y, const v2f w, const v2f sc_, const v2f sh_, const v2f mu_, const v2f rs_, v2f& p0,
                        v2f& p1, v2f& p2, v2f& p3, v2f& p4) {
            const v2f z = y * sc_ + sh_, sg = sigmoid2(z);
            const v2f dsl = sg * (1.0f + z * (1.0f - sg));
            const v2f xh = (y - mu_) * rs_, dd = dsl * w;
            p0 += w * (z * sg);
            p1 += dd;
            p2 += dd * xh;
            p3 += dsl;
            p4 += dsl * xh;
        };
        auto body = [&](const f22 y, const f22 w) {
            half(y.lo, w.lo, sc.lo, sh.lo, mu.lo, rs.lo, a0.lo, a1.lo, a2.lo, a3.lo, a4.lo);
            half(y.hi, w.hi, sc.hi, sh.hi, mu.hi, rs.hi, a0.hi, a1.hi, a2.hi, a3.hi, a4.hi);
        };
        int r = r_begin + m.pt;
        for (; r + 3 * m.npt < r_end; r += 4 * m.npt) {
            f22 y[4], w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                y[u] = load22<T>(yb + (size_t)(r + u * m.npt) * C);
                w[u] = load22<T>(wb + (size_t)(r + u * m.npt) * C);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) body(y[u], w[u]);
        }
        for (; r < r_end; r += m.npt) body(load22<T>(yb + (size_t)r * C), load22<T>(wb + (size_t)r * C));
    }
    const f4 q0 = to_f4(a0), q1 = to_f4(a1), q2 = to_f4(a2), q3 = to_f4(a3), q4 = to_f4(a4);
    const size_t plane = (size_t)B * splits * C, row = ((size_t)b * splits + sp) * C;
    block_colsum2(red, m, tid, q0, q1, out + row, out + plane + row);
    if (gate) {
        // one split per image: plane 0 is then the whole gradient w.r.t. the gate, and the sigmoid's derivative
        // g(1-g) goes in here (each thread revisits the entries it has just written), which makes the plane the
        // dpre2 of mia_se_gate_bwd and saves that entry point a launch
        for (int ch = tid; ch < m.cbv; ch += 256) {
            const float g = gate[(size_t)b * C + m.c0 + ch];
            out[row + m.c0 + ch] *= g * (1.f - g);
        }
    }
    block_colsum2(red, m, tid, q2, q3, out + 2 * plane + row, out + 3 * plane + row);
    block_colsum2(red, m, tid, q4, q4, out + 4 * plane + row, nullptr);
}

extern "C" int mia_se_bwd_sums(int dtype, const void* Y, const void* dZ, int B, int HW, int C, const float* scale,
                               const float* shift, const float* mean, const float* rstd, int splits, float* out,
                               const float* gate, void* stream) {
    MIA_CHECK_ARG(!gate || splits == 1, "mia_se_bwd_sums: the gate factor needs one split per image");
    MIA_CHECK_ARG(Y && dZ && out && scale && shift && mean && rstd && B > 0 && HW > 0 && C > 0 && C % 4 == 0 &&
                      splits > 0 && splits <= HW,
                  "mia_se_bwd_sums: bad arguments");
    const int ncb = cdiv(C, 256);
    const int CB = cdiv(cdiv(C, 4), ncb) * 4;
    dim3 grid(B * splits, ncb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(se_bwd_sums_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)Y, (const bf16_t*)dZ, B, HW,
                           C, CB, splits, scale, shift, mean, rstd, out, gate);
    else
        MIA_LAUNCH(se_bwd_sums_kernel<float>, grid, dim3(256), 0, s, (const float*)Y, (const float*)dZ, B, HW, C,
                           CB, splits, scale, shift, mean, rstd, out, gate);
    MIA_CHECK_LAUNCH("mia_se_bwd_sums");
    return 0;
}

// per-image rows of the BatchNorm-backward partial slab from the five sums: part [B][2][C]
__global__ __launch_bounds__(256) void se_bn_combine_kernel(const float* sums, int B, int splits, int C,
                                                            const float* gate, const float* dpool, float* part) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    const size_t plane = (size_t)B * splits * C;
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) {
        const size_t o = ((size_t)b * splits + k) * C + c;
#pragma unroll
        for (int t = 0; t < 4; ++t) q[t] += sums[(t + 1) * plane + o];
    }
    const float g = gate[i], dp = dpool[i];
    part[((size_t)b * 2 + 0) * C + c] = g * q[0] + dp * q[2];
    part[((size_t)b * 2 + 1) * C + c] = g * q[1] + dp * q[3];
}

extern "C" int mia_se_bn_combine(const float* sums, int B, int splits, int C, const float* gate, const float* dpool,
                                 float* part, void* stream) {
    MIA_CHECK_ARG(sums && gate && dpool && part && B > 0 && splits > 0 && C > 0, "mia_se_bn_combine: bad arguments");
    MIA_LAUNCH(se_bn_combine_kernel, dim3(cdiv((long long)B * C, 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), sums, B, splits, C, gate, dpool, part);
    MIA_CHECK_LAUNCH("mia_se_bn_combine");
    return 0;
}

// dy = ca * [act'(z) * (G*gate[b] + dpool[b])] + cb*y + cd, written over G   (z = scale*y + shift)
// W = channels per thread: 4, or 8 (one 16-byte access per lane for bf16) when C % 8 == 0
template <typename T, int W>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(T* G, const T* Y, unsigned nvec, int C,
                                                               const float* scale, const float* shift, int act,
                                                               const float* gate, const float* dpool,
                                                               unsigned vec_per_img, const float* ca, const float* cb,
                                                               const float* cd, int fixed_c) {
    const unsigned cvec = (unsigned)C / W;
    if (fixed_c) {
        // The grid stride is a multiple of the vectors per row (the host sizes the grid so), hence this thread sees
        // the same channels in every iteration: the five per-channel coefficient vectors are loaded once instead of
        // 160 bytes of them per 32 bytes of tensor, and the channel index needs no division in the loop.
        constexpr int Q = W / 4;
        unsigned i = blockIdx.x * 256 + threadIdx.x;
        const int c = (int)(i % cvec) * W;
        const bool silu = act == MIA_ACT_SILU;
        f22 sc[Q], sh[Q], a[Q], bb[Q], d[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            sc[q] = silu ? load22<float>(scale + c + 4 * q) : f22{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
            sh[q] = silu ? load22<float>(shift + c + 4 * q) : f22{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
            a[q] = load22<float>(ca + c + 4 * q);
            bb[q] = load22<float>(cb + c + 4 * q);
            d[q] = load22<float>(cd + c + 4 * q);
        }
        const float inv_vpi = 1.0f / (float)vec_per_img;
        for (; i < nvec; i += gridDim.x * 256) {
            const size_t idx = (size_t)i * W;
            f22 g[Q], y[Q];
            if constexpr (W == 8) {
                load44<T>(G + idx, g[0], g[1]);
                load44<T>(Y + idx, y[0], y[1]);
            } else {
                g[0] = load22<T>(G + idx);
                y[0] = load22<T>(Y + idx);
            }
            if (gate) {
                // image of this vector: float estimate plus one correction (exact for i < 2^24, else the division)
                unsigned img;
                if (nvec < (1u << 24)) {
                    img = (unsigned)((float)i * inv_vpi);
                    const long long rem = (long long)i - (long long)img * vec_per_img;
                    if (rem < 0) --img;
                    else if (rem >= (long long)vec_per_img) ++img;
                } else {
                    img = i / vec_per_img;
                }
                const size_t go = (size_t)img * C + c;
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const f22 gt = load22<float>(gate + go + 4 * q), dp = load22<float>(dpool + go + 4 * q);
                    g[q].lo = g[q].lo * gt.lo + dp.lo; g[q].hi = g[q].hi * gt.hi + dp.hi;
                }
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (silu) {
                    g[q].lo *= silu_grad2(y[q].lo * sc[q].lo + sh[q].lo);
                    g[q].hi *= silu_grad2(y[q].hi * sc[q].hi + sh[q].hi);
                }
                g[q] = f22{a[q].lo * g[q].lo + bb[q].lo * y[q].lo + d[q].lo, a[q].hi * g[q].hi + bb[q].hi * y[q].hi + d[q].hi};
            }
            if constexpr (W == 8) store44<T>(G + idx, g[0], g[1]);
            else store22<T>(G + idx, g[0]);
        }
        return;
    }
    auto four = [&](f22& g, const f22& y, int c, size_t go) {
        if (gate) {
            const f22 gt = load22<float>(gate + go), dp = load22<float>(dpool + go);
            g.lo = g.lo * gt.lo + dp.lo; g.hi = g.hi * gt.hi + dp.hi;
        }
        if (act == MIA_ACT_SILU) {
            const f22 sc = load22<float>(scale + c), sh = load22<float>(shift + c);
            g.lo *= silu_grad2(y.lo * sc.lo + sh.lo);
            g.hi *= silu_grad2(y.hi * sc.hi + sh.hi);
        }
        const f22 a = load22<float>(ca + c), bb = load22<float>(cb + c), d = load22<float>(cd + c);
        g = f22{a.lo * g.lo + bb.lo * y.lo + d.lo, a.hi * g.hi + bb.hi * y.hi + d.hi};
    };
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        const size_t idx = (size_t)i * W;
        const int c = (int)(i % cvec) * W;
        const size_t go = (size_t)(i / vec_per_img) * C + c;
        if constexpr (W == 8) {
            f22 g0, g1, y0, y1;
            load44<T>(G + idx, g0, g1);
            load44<T>(Y + idx, y0, y1);
            four(g0, y0, c, go);
            four(g1, y1, c + 4, go + 4);
            store44<T>(G + idx, g0, g1);
        } else {
            f22 g = load22<T>(G + idx);
            const f22 y = load22<T>(Y + idx);
            four(g, y, c, go);
            store22<T>(G + idx, g);
        }
    }
}

extern "C" int mia_bn_act_bwd_apply(int dtype, void* G, const void* Y, long long R, int C, const float* scale,
                                    const float* shift, int act, const float* gate, const float* dpool,
                                    int rows_per_img, const float* ca, const float* cb, const float* cd,
                                    void* stream) {
    MIA_CHECK_ARG(G && Y && R > 0 && C > 0 && C % 4 == 0 && ca && cb && cd, "mia_bn_act_bwd_apply: bad arguments");
    MIA_CHECK_ARG(act == MIA_ACT_NONE || (scale && shift), "mia_bn_act_bwd_apply: activation needs scale/shift");
    MIA_CHECK_ARG(!(gate == nullptr) == !(dpool == nullptr), "mia_bn_act_bwd_apply: gate/dpool must come together");
    MIA_CHECK_ARG(!gate || rows_per_img > 0, "mia_bn_act_bwd_apply: gate needs rows_per_img");
    const bool wide = (C % 8 == 0) && ((((uintptr_t)G | (uintptr_t)Y) & 15) == 0);
    const int W = wide ? 8 : 4;
    const long long nvec = R * C / W;
    MIA_CHECK_ARG(nvec < (1LL << 32), "mia_bn_act_bwd_apply: tensor too large");
    long long blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    // a grid whose stride (blocks*256 vectors) is a multiple of the vectors per row keeps every thread on fixed
    // channels: blocks must be a multiple of cvec / gcd(cvec, 256)
    int fixed_c = 0;
    {
        const long long cvec = C / W;
        long long a_ = cvec, b_ = 256;
        while (b_) { const long long t = a_ % b_; a_ = b_; b_ = t; }
        const long long unit = cvec / a_;
        if (unit <= blocks) { blocks = blocks / unit * unit; fixed_c = 1; }
    }
    const unsigned vpi = (unsigned)((long long)(rows_per_img > 0 ? rows_per_img : 1) * C / W);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MIA_APPLY(T_, W_)                                                                                          \
    MIA_LAUNCH((bn_act_bwd_apply_kernel<T_, W_>), dim3((unsigned)blocks), dim3(256), 0, s, (T_*)G, (const T_*)Y, \
                       (unsigned)nvec, C, scale, shift, act, gate, dpool, vpi, ca, cb, cd, fixed_c)
    if (dtype == MIA_BF16) { if (wide) MIA_APPLY(bf16_t, 8); else MIA_APPLY(bf16_t, 4); }
    else { if (wide) MIA_APPLY(float, 8); else MIA_APPLY(float, 4); }
#undef MIA_APPLY
    MIA_CHECK_LAUNCH("mia_bn_act_bwd_apply");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// squeeze-excite gate: pooled -> FC(rd)+SiLU -> FC(C)+sigmoid.  One workgroup per image.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float* pool_part, int splits, float inv_hw, int C,
                                                          int RD, const float* Wr, const float* br, const float* WeT,
                                                          const float* be, float* pooled, float* hpre, float* gate) {
    extern __shared__ float sm[];
    float* pl = sm;      // [C]
    float* h = sm + C;   // [RD]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += pool_part[((size_t)b * splits + k) * C + c];
        s *= inv_hw;
        pl[c] = s;
        pooled[(size_t)b * C + c] = s;
    }
    __syncthreads();
    for (int j = wave; j < RD; j += 4) {
        float d = 0.f;
        for (int c = lane; c < C; c += 64) d += Wr[(size_t)j * C + c] * pl[c];
        d = wave_sum(d);
        if (lane == 0) {
            const float pre = d + br[j];
            hpre[(size_t)b * RD + j] = pre;
            h[j] = siluf_(pre);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float d = be[c];
        for (int j = 0; j < RD; ++j) d += WeT[(size_t)j * C + c] * h[j];
        gate[(size_t)b * C + c] = sigmoidf_(d);
    }
}

// ---- wide layers (C*RD large): the two FCs as batched register-tiled products, so that a weight row fetched
// from L2 serves 8 images and every CU has work (one workgroup per image re-reads both matrices per image).
constexpr int SE_JT = 4, SE_IB = 8;
constexpr int SE_RT = 256;   // threads of the reduction kernel (512 measured slower: 8.6 -> 11.6 us at 2304 channels)

// v[b][c] = inv_hw * sum_k part[b][k][c]                         (mode 0, forward pooling)
// v[b][c] = (sum_k part[b][k][c]) * g[b][c] * (1 - g[b][c])      (mode 1, backward through the sigmoid)
__global__ __launch_bounds__(256) void se_prep_kernel(const float* part, int splits, float inv_hw, int C, int total,
                                                      const float* g, int mode, float* v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[((size_t)b * splits + k) * C + c];
    if (mode == 0) v[i] = s * inv_hw;
    else {
        const float gv = g[i];
        v[i] = s * gv * (1.f - gv);
    }
}

// d[b][j] = sum_c Wm[j][c] * v[b][c]; one workgroup per (4 rows j) x (8 images): its waves split c, lanes stride
// over it, and the partial sums meet in LDS in wave order.
// mode 0: pre = d + bias[j] -> out[b][j]                     (forward hidden pre-activation)
// mode 1: out[b][j] = d * silu'(hpre[b][j])                  (backward through the hidden SiLU)
// vscale: v is used as v*vscale (the forward takes the per-image channel SUMS and folds the 1/HW of the mean in here);
// vout (nullable): the waves of the first row tile also write v*vscale there (the pooled means the backward needs).
__global__ __launch_bounds__(SE_RT) void se_reduce_c_kernel(const float* __restrict__ Wm, const float* __restrict__ v,
                                                          int B, int C, int RD, const float* bias, const float* hpre,
                                                          int mode, float* out, float vscale, float* vout) {
    __shared__ float red[SE_RT / 64][SE_JT * SE_IB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int njt = (RD + SE_JT - 1) / SE_JT;
    const int jt = blockIdx.x % njt, bt = blockIdx.x / njt;
    const int j0 = jt * SE_JT, b0 = bt * SE_IB;
    const float* wp[SE_JT];
    const float* vp[SE_IB];
#pragma unroll
    for (int a = 0; a < SE_JT; ++a) wp[a] = Wm + (size_t)(j0 + a < RD ? j0 + a : RD - 1) * C;
#pragma unroll
    for (int a = 0; a < SE_IB; ++a) vp[a] = v + (size_t)(b0 + a < B ? b0 + a : B - 1) * C;
    float acc[SE_JT][SE_IB];
#pragma unroll
    for (int a = 0; a < SE_JT; ++a)
#pragma unroll
        for (int e = 0; e < SE_IB; ++e) acc[a][e] = 0.f;
    // This kernel sits on every block's critical chain with a handful of waves per CU: its duration is the number of
    // dependent memory round trips, so the loop moves 16 bytes per lane and load, keeps two rounds in flight
    // (4-byte loads, two rounds at a time: 14 us at 2304 channels, of which 12 were waiting) and is split over the
    // workgroup's waves.
    if ((C & 3) == 0) {
#pragma unroll 2
        for (int c = threadIdx.x * 4; c < C; c += SE_RT * 4) {
            float4 w[SE_JT], x[SE_IB];
#pragma unroll
            for (int a = 0; a < SE_JT; ++a) w[a] = *reinterpret_cast<const float4*>(wp[a] + c);
#pragma unroll
            for (int e = 0; e < SE_IB; ++e) x[e] = *reinterpret_cast<const float4*>(vp[e] + c);
#pragma unroll
            for (int a = 0; a < SE_JT; ++a)
#pragma unroll
                for (int e = 0; e < SE_IB; ++e)
                    acc[a][e] += (w[a].x * x[e].x + w[a].y * x[e].y) + (w[a].z * x[e].z + w[a].w * x[e].w);
            if (vout && jt == 0) {
#pragma unroll
                for (int e = 0; e < SE_IB; ++e)
                    if (b0 + e < B)
                        *reinterpret_cast<float4*>(vout + (size_t)(b0 + e) * C + c) =
                            make_float4(x[e].x * vscale, x[e].y * vscale, x[e].z * vscale, x[e].w * vscale);
            }
        }
    } else {
#pragma unroll 2
        for (int c = threadIdx.x; c < C; c += SE_RT) {
            float w[SE_JT], x[SE_IB];
#pragma unroll
            for (int a = 0; a < SE_JT; ++a) w[a] = wp[a][c];
#pragma unroll
            for (int e = 0; e < SE_IB; ++e) x[e] = vp[e][c];
#pragma unroll
            for (int a = 0; a < SE_JT; ++a)
#pragma unroll
                for (int e = 0; e < SE_IB; ++e) acc[a][e] += w[a] * x[e];
            if (vout && jt == 0) {
#pragma unroll
                for (int e = 0; e < SE_IB; ++e)
                    if (b0 + e < B) vout[(size_t)(b0 + e) * C + c] = x[e] * vscale;
            }
        }
    }
    float mine = 0.f;
#pragma unroll
    for (int a = 0; a < SE_JT; ++a)
#pragma unroll
        for (int e = 0; e < SE_IB; ++e) {
            const float t = wave_sum(acc[a][e]);
            if (lane == a * SE_IB + e) mine = t;
        }
    if (lane < SE_JT * SE_IB) red[wave][lane] = mine;
    __syncthreads();
    if (wave == 0 && lane < SE_JT * SE_IB) {
        mine = 0.f;
#pragma unroll
        for (int w = 0; w < SE_RT / 64; ++w) mine += red[w][lane];
        mine *= vscale;
        const int j = j0 + lane / SE_IB, b = b0 + lane % SE_IB;
        if (j < RD && b < B) {
            const size_t o = (size_t)b * RD + j;
            out[o] = mode == 0 ? mine + bias[j] : mine * silu_gradf_(hpre[o]);
        }
    }
}

// d[b][c] = sum_j Wm[j][c] * act(u[b][j]); one thread per channel and 8 images.
// mode 0: u = silu(hpre), out = sigmoid(d + bias[c])   (forward gate);  mode 1: u as is, out = d * oscale
__global__ __launch_bounds__(256) void se_expand_kernel(const float* __restrict__ Wm, const float* __restrict__ u, int B,
                                                        int C, int RD, const float* bias, float oscale, int mode,
                                                        float* out) {
    extern __shared__ float us[];  // [RD][SE_IB]
    const int c = blockIdx.x * 256 + threadIdx.x, b0 = blockIdx.y * SE_IB;
    for (int i = threadIdx.x; i < RD * SE_IB; i += 256) {
        const int j = i / SE_IB, e = i % SE_IB;
        float t = 0.f;
        if (b0 + e < B) {
            t = u[(size_t)(b0 + e) * RD + j];
            if (mode == 0) t = siluf_(t);
        }
        us[i] = t;
    }
    __syncthreads();
    if (c >= C) return;
    float acc[SE_IB];
#pragma unroll
    for (int e = 0; e < SE_IB; ++e) acc[e] = 0.f;
#pragma unroll 8
    for (int j = 0; j < RD; ++j) {
        const float w = Wm[(size_t)j * C + c];
        const float4 u0 = *reinterpret_cast<const float4*>(us + j * SE_IB);
        const float4 u1 = *reinterpret_cast<const float4*>(us + j * SE_IB + 4);
        acc[0] += w * u0.x; acc[1] += w * u0.y; acc[2] += w * u0.z; acc[3] += w * u0.w;
        acc[4] += w * u1.x; acc[5] += w * u1.y; acc[6] += w * u1.z; acc[7] += w * u1.w;
    }
    const float bv = mode == 0 ? bias[c] : 0.f;
#pragma unroll
    for (int e = 0; e < SE_IB; ++e)
        if (b0 + e < B) out[(size_t)(b0 + e) * C + c] = mode == 0 ? sigmoidf_(acc[e] + bv) : acc[e] * oscale;
}

static inline bool se_wide(int C, int RD) { return (long long)C * RD >= 12000; }

extern "C" int mia_se_gate_fwd(const float* pool_part, int B, int splits, int HW, int C, int RD, const float* Wr,
                               const float* br, const float* WeT, const float* be, float* pooled, float* hpre,
                               float* gate, void* stream) {
    MIA_CHECK_ARG(pool_part && Wr && br && WeT && be && pooled && hpre && gate && B > 0 && C > 0 && RD > 0 && HW > 0,
                  "mia_se_gate_fwd: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (se_wide(C, RD)) {
        const int waves = cdiv(RD, SE_JT) * cdiv(B, SE_IB);
        if (splits == 1) {
            // one partial row per image: the "sum of the splits" is the row itself, and the 1/HW of the mean rides in
            // the first FC (which is linear in its input) -- one launch less on the block's critical chain
            MIA_LAUNCH(se_reduce_c_kernel, dim3(waves), dim3(SE_RT), 0, s, Wr, pool_part, B, C, RD, br,
                       (const float*)nullptr, 0, hpre, 1.0f / (float)HW, pooled);
        } else {
            MIA_LAUNCH(se_prep_kernel, dim3(cdiv((long long)B * C, 256)), dim3(256), 0, s, pool_part, splits,
                       1.0f / (float)HW, C, B * C, (const float*)nullptr, 0, pooled);
            MIA_LAUNCH(se_reduce_c_kernel, dim3(waves), dim3(SE_RT), 0, s, Wr, pooled, B, C, RD, br,
                       (const float*)nullptr, 0, hpre, 1.f, (float*)nullptr);
        }
        MIA_LAUNCH(se_expand_kernel, dim3(cdiv(C, 256), cdiv(B, SE_IB)), dim3(256), (size_t)RD * SE_IB * 4, s,
                           WeT, hpre, B, C, RD, be, 1.f, 0, gate);
    } else {
        const size_t lds = (size_t)(C + RD) * 4;
        MIA_LAUNCH(se_gate_fwd_kernel, dim3(B), dim3(256), lds, s, pool_part, splits, 1.0f / (float)HW, C, RD,
                           Wr, br, WeT, be, pooled, hpre, gate);
    }
    MIA_CHECK_LAUNCH("mia_se_gate_fwd");
    return 0;
}

// backward through the gate for one image: dgate (summed partials) -> dpre2 [B,C], dpre1 [B,RD],
// dpool [B,C] (already divided by HW: it is added to every pixel's gradient)
// pre: dgate_part already holds dpre2 (mia_se_bwd_sums multiplied the sigmoid's derivative in): splits == 1 and
// dpre2 is the same array, which is then neither recomputed nor rewritten.
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float* dgate_part, int splits, float inv_hw, int C,
                                                          int RD, const float* Wr, const float* WeT, const float* gate,
                                                          const float* hpre, float* dpre2, float* dpre1, float* dpool,
                                                          int pre) {
    extern __shared__ float sm[];
    float* d2 = sm;      // [C]
    float* d1 = sm + C;  // [RD]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += dgate_part[((size_t)b * splits + k) * C + c];
        if (pre) {
            d2[c] = s;
        } else {
            const float g = gate[(size_t)b * C + c];
            const float v = s * g * (1.f - g);
            d2[c] = v;
            dpre2[(size_t)b * C + c] = v;
        }
    }
    __syncthreads();
    for (int j = wave; j < RD; j += 4) {
        float d = 0.f;
        for (int c = lane; c < C; c += 64) d += WeT[(size_t)j * C + c] * d2[c];
        d = wave_sum(d);
        if (lane == 0) {
            const float v = d * silu_gradf_(hpre[(size_t)b * RD + j]);
            d1[j] = v;
            dpre1[(size_t)b * RD + j] = v;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float d = 0.f;
        for (int j = 0; j < RD; ++j) d += Wr[(size_t)j * C + c] * d1[j];
        dpool[(size_t)b * C + c] = d * inv_hw;
    }
}

// parameter gradients of the two FCs, summed over the batch (fixed order => reproducible).
// 64 (c, j) pairs per workgroup; the 4 waves split the batch and combine through LDS.
__global__ __launch_bounds__(256) void se_param_grad_kernel(int B, int C, int RD, const float* dpre2,
                                                            const float* dpre1, const float* pooled, const float* hpre,
                                                            float* dWr, float* dbr, float* dWe, float* dbe) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool ok = i < C * RD;
    // i = c*RD + j addresses We[c][j]; the same thread also produces Wr[j][c]
    const int c = ok ? i / RD : 0, j = ok ? i % RD : 0;
    float we = 0.f, wr = 0.f, bE = 0.f, bR = 0.f;
    if (ok) {
#pragma unroll 4
        for (int b = wave; b < B; b += 4) {
            const float p2 = dpre2[(size_t)b * C + c], p1 = dpre1[(size_t)b * RD + j];
            we += p2 * siluf_(hpre[(size_t)b * RD + j]);
            wr += p1 * pooled[(size_t)b * C + c];
            bE += p2;
            bR += p1;
        }
    }
    red[0][wave][lane] = we; red[1][wave][lane] = wr; red[2][wave][lane] = bE; red[3][wave][lane] = bR;
    __syncthreads();
    if (wave == 0 && ok) {
        dWe[(size_t)c * RD + j] = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        dWr[(size_t)j * C + c] = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
        if (j == 0) dbe[c] = red[2][0][lane] + red[2][1][lane] + red[2][2][lane] + red[2][3][lane];
        if (c == 0) dbr[j] = red[3][0][lane] + red[3][1][lane] + red[3][2][lane] + red[3][3][lane];
    }
}

extern "C" int mia_se_gate_bwd(const float* dgate_part, int B, int splits, int HW, int C, int RD, const float* Wr,
                               const float* WeT, const float* gate, const float* hpre, const float* pooled,
                               float* dpre2, float* dpre1, float* dpool, float* dWr, float* dbr, float* dWe,
                               float* dbe, void* stream) {
    MIA_CHECK_ARG(dgate_part && Wr && WeT && gate && hpre && pooled && dpre2 && dpre1 && dpool && B > 0 && C > 0 &&
                      RD > 0 && HW > 0 && (!dWr == !dbr) && (!dWr == !dWe) && (!dWr == !dbe),
                  "mia_se_gate_bwd: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // dpre2 == dgate_part: the caller's one partial row per image already is dpre2 (see mia_se_bwd_sums' gate argument)
    const int pre = (dpre2 == dgate_part) ? 1 : 0;
    MIA_CHECK_ARG(!pre || splits == 1, "mia_se_gate_bwd: dpre2 may alias dgate_part only with one split per image");
    if (se_wide(C, RD)) {
        if (!pre)
            MIA_LAUNCH(se_prep_kernel, dim3(cdiv((long long)B * C, 256)), dim3(256), 0, s, dgate_part, splits, 0.f,
                       C, B * C, gate, 1, dpre2);
        const int waves = cdiv(RD, SE_JT) * cdiv(B, SE_IB);
        MIA_LAUNCH(se_reduce_c_kernel, dim3(waves), dim3(SE_RT), 0, s, WeT, dpre2, B, C, RD,
                           (const float*)nullptr, hpre, 1, dpre1, 1.f, (float*)nullptr);
        MIA_LAUNCH(se_expand_kernel, dim3(cdiv(C, 256), cdiv(B, SE_IB)), dim3(256), (size_t)RD * SE_IB * 4, s,
                           Wr, dpre1, B, C, RD, (const float*)nullptr, 1.0f / (float)HW, 1, dpool);
    } else {
        const size_t lds = (size_t)(C + RD) * 4;
        MIA_LAUNCH(se_gate_bwd_kernel, dim3(B), dim3(256), lds, s, dgate_part, splits, 1.0f / (float)HW, C, RD,
                           Wr, WeT, gate, hpre, dpre2, dpre1, dpool, pre);
    }
    if (dWr)
        MIA_LAUNCH(se_param_grad_kernel, dim3(cdiv((long long)C * RD, 64)), dim3(256), 0, s, B, C, RD, dpre2,
                           dpre1, pooled, hpre, dWr, dbr, dWe, dbe);
    MIA_CHECK_LAUNCH("mia_se_gate_bwd");
    return 0;
}

extern "C" int mia_se_param_grad(int B, int C, int RD, const float* dpre2, const float* dpre1, const float* pooled,
                                 const float* hpre, float* dWr, float* dbr, float* dWe, float* dbe, void* stream) {
    MIA_CHECK_ARG(dpre2 && dpre1 && pooled && hpre && dWr && dbr && dWe && dbe && B > 0 && C > 0 && RD > 0,
                  "mia_se_param_grad: bad arguments");
    MIA_LAUNCH(se_param_grad_kernel, dim3(cdiv((long long)C * RD, 64)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), B, C, RD, dpre2, dpre1, pooled, hpre, dWr, dbr, dWe, dbe);
    MIA_CHECK_LAUNCH("mia_se_param_grad");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm backward carried THROUGH a 1x1 convolution instead of being applied to its (wide) output.
// For y = x W^T (x [M,Cin] narrow, y [M,Cm] wide) followed by train-mode BatchNorm, the gradient with respect to y is
// dy = ca*g + cb*y + cd per channel (g = gradient w.r.t. the normalised tensor). Both consumers of dy are linear, so
//   dx = dy W        = g (ca o W) + x S + r,        S = W^T diag(cb) W  [Cin,Cin],  r = cd^T W  [Cin]
//   dW = dy^T x      = ca o (g^T x) + cb o (W G) + cd (x) cs,   G = x^T x  [Cin,Cin],  cs = column sums of x
// and dy itself (three passes over the widest tensor of the block) is never formed. Worth it where M*Cm is large and
// Cin small (the first stages: Cin <= 64); the weight-sized products below are then a few thousand outputs.
// ------------------------------------------------------------------------------------------------
// blocks [0, nS): 16x16 tiles of S (and r from the tiles of the first tile row); blocks [nS, ...): (ca o W)^T
template <typename T>
__global__ __launch_bounds__(256) void bn_lin_prep_kernel(const T* __restrict__ W, int Cm, int Cin,
                                                          const float* __restrict__ ca, const float* __restrict__ cb,
                                                          const float* __restrict__ cd, T* __restrict__ WcaT,
                                                          T* __restrict__ S, float* __restrict__ r, int nS) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nS) {
        const int idx = ((int)blockIdx.x - nS) * 256 + tid;   // element (j, c) of WcaT [Cin][Cm]
        if (idx < Cin * Cm) {
            const int j = idx / Cm, c = idx - j * Cm;
            WcaT[idx] = from_f32<T>(to_f32<T>(W[(size_t)c * Cin + j]) * ca[c]);
        }
        return;
    }
    __shared__ float wi[64][17], wj[64][17], cds[64];
    const int tiles = (Cin + 15) / 16;
    const int i0 = ((int)blockIdx.x / tiles) * 16, j0 = ((int)blockIdx.x % tiles) * 16;
    const int ti = tid >> 4, tj = tid & 15;
    float acc = 0.f, racc = 0.f;
    for (int c0 = 0; c0 < Cm; c0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * 16; e += 256) {
            const int cc = e >> 4, k = e & 15, c = c0 + cc;
            const bool ok = c < Cm;
            const float wiv = (ok && i0 + k < Cin) ? to_f32<T>(W[(size_t)c * Cin + i0 + k]) : 0.f;
            const float wjv = (ok && j0 + k < Cin) ? to_f32<T>(W[(size_t)c * Cin + j0 + k]) : 0.f;
            wi[cc][k] = ok ? wiv * cb[c] : 0.f;
            wj[cc][k] = wjv;
        }
        if (tid < 64) cds[tid] = (c0 + tid < Cm) ? cd[c0 + tid] : 0.f;
        __syncthreads();
#pragma unroll 8
        for (int cc = 0; cc < 64; ++cc) {
            acc += wi[cc][ti] * wj[cc][tj];
            if (ti == 0) racc += cds[cc] * wj[cc][tj];
        }
    }
    if (i0 + ti < Cin && j0 + tj < Cin) S[(size_t)(i0 + ti) * Cin + j0 + tj] = from_f32<T>(acc);
    if (i0 == 0 && ti == 0 && j0 + tj < Cin) r[j0 + tj] = racc;
}

// dW[c][j] = ca[c]*T1[c][j] + cb[c]*sum_i W[c][i]*G[i][j] + cd[c]*cs[j], written over T1 (= g^T x)
template <typename T>
__global__ __launch_bounds__(256) void bn_lin_fix_kernel(const T* __restrict__ W, int Cm, int Cin,
                                                         const float* __restrict__ ca, const float* __restrict__ cb,
                                                         const float* __restrict__ cd, const float* __restrict__ G,
                                                         const float* __restrict__ cs, float* __restrict__ dW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cm * Cin) return;
    const int c = idx / Cin, j = idx - c * Cin;
    float wg = 0.f;
    for (int i = 0; i < Cin; ++i) wg += to_f32<T>(W[(size_t)c * Cin + i]) * G[(size_t)i * Cin + j];
    dW[idx] = ca[c] * dW[idx] + cb[c] * wg + cd[c] * cs[j];
}

extern "C" int mia_bn_lin_prep(int dtype, const void* W, int Cm, int Cin, const float* ca, const float* cb,
                               const float* cd, void* WcaT, void* S, float* r, void* stream) {
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_bn_lin_prep: bad dtype %d", dtype);
    MIA_CHECK_ARG(W && ca && cb && cd && WcaT && S && r && Cm > 0 && Cin > 0, "mia_bn_lin_prep: bad arguments");
    const int tiles = cdiv(Cin, 16), nS = tiles * tiles;
    dim3 grid((unsigned)(nS + cdiv((long long)Cm * Cin, 256)));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(bn_lin_prep_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)W, Cm, Cin, ca, cb, cd,
                           (bf16_t*)WcaT, (bf16_t*)S, r, nS);
    else
        MIA_LAUNCH(bn_lin_prep_kernel<float>, grid, dim3(256), 0, s, (const float*)W, Cm, Cin, ca, cb, cd,
                           (float*)WcaT, (float*)S, r, nS);
    MIA_CHECK_LAUNCH("mia_bn_lin_prep");
    return 0;
}

extern "C" int mia_bn_lin_fix(int dtype, const void* W, int Cm, int Cin, const float* ca, const float* cb,
                              const float* cd, const float* G, const float* cs, float* dW, void* stream) {
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_bn_lin_fix: bad dtype %d", dtype);
    MIA_CHECK_ARG(W && ca && cb && cd && G && cs && dW && Cm > 0 && Cin > 0, "mia_bn_lin_fix: bad arguments");
    dim3 grid((unsigned)cdiv((long long)Cm * Cin, 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(bn_lin_fix_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)W, Cm, Cin, ca, cb, cd, G, cs, dW);
    else
        MIA_LAUNCH(bn_lin_fix_kernel<float>, grid, dim3(256), 0, s, (const float*)W, Cm, Cin, ca, cb, cd, G, cs, dW);
    MIA_CHECK_LAUNCH("mia_bn_lin_fix");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// stem: im2col of the 3x3 stride-2 pad-1 convolution. x NCHW fp32 [B,3,H,W] -> col [B*OH*OW, 32] (T);
// column = ci*9 + kh*3 + kw, columns 27..31 zero. The stem conv is then one pointwise GEMM.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* x, int B, int H, int W, int OH, int OW, T* col) {
    const long long total = (long long)B * OH * OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ow = (int)(i % OW), oh = (int)((i / OW) % OH), b = (int)(i / ((long long)OW * OH));
        float v[32];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                    v[ci * 9 + kh * 3 + kw] = (ih >= 0 && ih < H && iw >= 0 && iw < W)
                                                  ? x[(((size_t)b * 3 + ci) * H + ih) * W + iw]
                                                  : 0.f;
                }
#pragma unroll
        for (int k = 27; k < 32; ++k) v[k] = 0.f;
        T* o = col + (size_t)i * 32;
#pragma unroll
        for (int k = 0; k < 32; k += 4) store4<T>(o + k, f4{v[k], v[k + 1], v[k + 2], v[k + 3]});
    }
}

extern "C" int mia_stem_im2col(int dtype, const float* x, int B, int H, int W, void* col, void* stream) {
    MIA_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0, "mia_stem_im2col: bad arguments");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * OH * OW;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(stem_im2col_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, OH, OW,
                           (bf16_t*)col);
    else
        MIA_LAUNCH(stem_im2col_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, OH, OW,
                           (float*)col);
    MIA_CHECK_LAUNCH("mia_stem_im2col");
    return 0;
}

// The same lowering from a channels-last image already in the compute type: x [B,H,W,3] (T) -> col [B*OH*OW, 32] (T),
// same column order. A tap row of an output pixel is 9 contiguous elements (3 pixels x 3 channels) of the image,
// where the NCHW form gathers 27 strided scalars; no conversion, the values are copied.
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_nhwc_kernel(const T* __restrict__ x, int B, int H, int W, int OH,
                                                               int OW, T* __restrict__ col) {
    const long long total = (long long)B * OH * OW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ow = (int)(i % OW), oh = (int)((i / OW) % OH), b = (int)(i / ((long long)OW * OH));
        T v[32];
        T zero;
        zero = from_f32<T>(0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            const bool rok = ih >= 0 && ih < H;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                const bool ok = rok && iw >= 0 && iw < W;
                const T* px = x + (((size_t)b * H + (rok ? ih : 0)) * W + (ok ? iw : 0)) * 3;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) v[ci * 9 + kh * 3 + kw] = ok ? px[ci] : zero;
            }
        }
#pragma unroll
        for (int k = 27; k < 32; ++k) v[k] = zero;
        T* o = col + (size_t)i * 32;
        constexpr int PER16 = 16 / (int)sizeof(T);
#pragma unroll
        for (int k = 0; k < 32; k += PER16) {
            uint4 w;
            __builtin_memcpy(&w, &v[k], 16);
            *reinterpret_cast<uint4*>(o + k) = w;
        }
    }
}

extern "C" int mia_stem_im2col_nhwc(int dtype, const void* x, int B, int H, int W, void* col, void* stream) {
    MIA_CHECK_ARG(x && col && B > 0 && H > 0 && W > 0, "mia_stem_im2col_nhwc: bad arguments");
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_stem_im2col_nhwc: bad dtype %d", dtype);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)B * OH * OW;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(stem_im2col_nhwc_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, B, H, W, OH,
                   OW, (bf16_t*)col);
    else
        MIA_LAUNCH(stem_im2col_nhwc_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, B, H, W, OH, OW,
                   (float*)col);
    MIA_CHECK_LAUNCH("mia_stem_im2col_nhwc");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weight preparation: fp32 master [R][Cc] -> compute-dtype copy [R][ldd] and/or transpose [Cc][ldt],
// zero-padded to the leading dimensions.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_transpose_kernel(const float* src, int R, int Cc, T* dst, int ldd,
                                                                 T* dstT, int ldt) {
    const long long n1 = dst ? (long long)R * ldd : 0;
    const long long n2 = dstT ? (long long)Cc * ldt : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * 256) {
        if (i < n1) {
            const int r = (int)(i / ldd), c = (int)(i % ldd);
            dst[i] = from_f32<T>(c < Cc ? src[(size_t)r * Cc + c] : 0.f);
        } else {
            const long long k = i - n1;
            const int c = (int)(k / ldt), r = (int)(k % ldt);
            dstT[k] = from_f32<T>(r < R ? src[(size_t)r * Cc + c] : 0.f);
        }
    }
}

extern "C" int mia_cast_pad_transpose(int dtype, const float* src, int R, int Cc, void* dst, int ldd, void* dstT,
                                      int ldt, void* stream) {
    MIA_CHECK_ARG(src && R > 0 && Cc > 0 && (dst || dstT), "mia_cast_pad_transpose: bad arguments");
    MIA_CHECK_ARG((!dst || ldd >= Cc) && (!dstT || ldt >= R), "mia_cast_pad_transpose: leading dims too small");
    const long long n = (dst ? (long long)R * ldd : 0) + (dstT ? (long long)Cc * ldt : 0);
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(cast_pad_transpose_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, src, R, Cc,
                           (bf16_t*)dst, ldd, (bf16_t*)dstT, ldt);
    else
        MIA_LAUNCH(cast_pad_transpose_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, src, R, Cc,
                           (float*)dst, ldd, (float*)dstT, ldt);
    MIA_CHECK_LAUNCH("mia_cast_pad_transpose");
    return 0;
}

// The same for many matrices in one launch (all the 1x1-conv weights of a step): descriptors travel by value in the
// kernel arguments, 48 matrices per launch; a workgroup handles 4096 consecutive output elements of one matrix.
#define CAST_MAXT 48
#define CAST_BLOCK_ELEMS 4096
struct CastChunk {
    const float* src[CAST_MAXT];
    void* dst[CAST_MAXT];
    void* dstT[CAST_MAXT];
    int R[CAST_MAXT], Cc[CAST_MAXT], ldd[CAST_MAXT], ldt[CAST_MAXT];
    int first_block[CAST_MAXT + 1];
    int n;
};

template <typename T>
__global__ __launch_bounds__(256) void cast_pad_transpose_multi_kernel(CastChunk ch) {
    int t = 0;
    while (t + 1 < ch.n && (int)blockIdx.x >= ch.first_block[t + 1]) ++t;   // uniform per workgroup
    const float* src = ch.src[t];
    T* dst = reinterpret_cast<T*>(ch.dst[t]);
    T* dstT = reinterpret_cast<T*>(ch.dstT[t]);
    const int R = ch.R[t], Cc = ch.Cc[t], ldd = ch.ldd[t], ldt = ch.ldt[t];
    const long long n1 = dst ? (long long)R * ldd : 0, n2 = dstT ? (long long)Cc * ldt : 0;
    const long long base = (long long)(blockIdx.x - ch.first_block[t]) * CAST_BLOCK_ELEMS;
    for (int e = threadIdx.x; e < CAST_BLOCK_ELEMS; e += 256) {
        const long long i = base + e;
        if (i >= n1 + n2) break;
        if (i < n1) {
            const int r = (int)(i / ldd), c = (int)(i % ldd);
            dst[i] = from_f32<T>(c < Cc ? src[(size_t)r * Cc + c] : 0.f);
        } else {
            const long long k = i - n1;
            const int c = (int)(k / ldt), r = (int)(k % ldt);
            dstT[k] = from_f32<T>(r < R ? src[(size_t)r * Cc + c] : 0.f);
        }
    }
}

extern "C" int mia_cast_pad_transpose_multi(int dtype, int n, const void* const* src, const int* rows, const int* cols,
                                            void* const* dst, const int* ldd, void* const* dstT, const int* ldt,
                                            void* stream) {
    MIA_CHECK_ARG(n > 0 && src && rows && cols && dst && ldd && dstT && ldt, "mia_cast_pad_transpose_multi: bad arguments");
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_cast_pad_transpose_multi: bad dtype %d", dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < n; t0 += CAST_MAXT) {
        CastChunk ch;
        ch.n = (n - t0) < CAST_MAXT ? (n - t0) : CAST_MAXT;
        int blocks = 0;
        for (int i = 0; i < ch.n; ++i) {
            const int k = t0 + i;
            MIA_CHECK_ARG(src[k] && rows[k] > 0 && cols[k] > 0 && (dst[k] || dstT[k]),
                          "mia_cast_pad_transpose_multi: matrix %d: bad arguments", k);
            MIA_CHECK_ARG((!dst[k] || ldd[k] >= cols[k]) && (!dstT[k] || ldt[k] >= rows[k]),
                          "mia_cast_pad_transpose_multi: matrix %d: leading dims too small", k);
            ch.src[i] = reinterpret_cast<const float*>(src[k]);
            ch.dst[i] = dst[k]; ch.dstT[i] = dstT[k];
            ch.R[i] = rows[k]; ch.Cc[i] = cols[k]; ch.ldd[i] = ldd[k]; ch.ldt[i] = ldt[k];
            ch.first_block[i] = blocks;
            const long long elems = (dst[k] ? (long long)rows[k] * ldd[k] : 0) + (dstT[k] ? (long long)cols[k] * ldt[k] : 0);
            blocks += (int)((elems + CAST_BLOCK_ELEMS - 1) / CAST_BLOCK_ELEMS);
        }
        ch.first_block[ch.n] = blocks;
        if (dtype == MIA_BF16)
            MIA_LAUNCH(cast_pad_transpose_multi_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, ch);
        else
            MIA_LAUNCH(cast_pad_transpose_multi_kernel<float>, dim3(blocks), dim3(256), 0, s, ch);
    }
    MIA_CHECK_LAUNCH("mia_cast_pad_transpose_multi");
    return 0;
}

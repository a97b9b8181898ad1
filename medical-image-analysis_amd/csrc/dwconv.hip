// Depthwise k x k convolutions (k in {3,5}, stride in {1,2}) on NHWC tensors, gfx950.
//
// A workgroup owns a (TH x TW output pixels) x (CB channels) tile of one image. The input halo tile
// is staged once into LDS -- with the producer's BatchNorm+SiLU applied on the way in, so the
// normalised activation never exists in HBM -- and every thread computes strips of 4 output pixels
// for one 4-channel vector, re-using each staged input row across the k taps along W. Lanes run
// fastest over channels, so global loads/stores are contiguous CB*sizeof(T) byte runs per pixel.
//
//   dwconv_fwd        y = dw(pro(x))                      + per-channel sum/sumsq partials (BN stats)
//   data-gradient     stride 1: the same kernel with flipped taps; stride 2: gather kernel
//                     both with the fused epilogue g = dA * silu'(z), partials sum(g), sum(g*xhat)
//   dwconv_bwd_weight dW[c][kh][kw] = sum dY * pro(x)     (per-workgroup partial slab + ordered reduce)
#include "common.h"

#define DW_PW 4  // output pixels per strip

struct DwParams {
    const void* x;
    const float* w;  // [C][K*K]
    void* y;
    int B, H, W, C, OH, OW;
    const float* pscale;
    const float* pshift;
    int flip;
    const void* yfwd;  // epilogue: forward raw tensor at the output positions (nullable)
    const float* escale;
    const float* eshift;
    const float* emean;
    const float* erstd;
    float* stats;  // [gridDim.x][2][C] (nullable)
    int CB, TH, TW, tiles_h, tiles_w;
};

template <typename T>
__device__ __forceinline__ void dw_epilogue_store(const DwParams& p, size_t idx, int c, f4 v, f4 es, f4 eh, f4 em,
                                                  f4 er, f4& s, f4& q) {
    T* y = reinterpret_cast<T*>(p.y);
    if (p.yfwd) {
        f4 yv = load4<T>(reinterpret_cast<const T*>(p.yfwd) + idx);
        f4 g;
        g.x = v.x * silu_gradf_(yv.x * es.x + eh.x);
        g.y = v.y * silu_gradf_(yv.y * es.y + eh.y);
        g.z = v.z * silu_gradf_(yv.z * es.z + eh.z);
        g.w = v.w * silu_gradf_(yv.w * es.w + eh.w);
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        q.x += g.x * ((yv.x - em.x) * er.x);
        q.y += g.y * ((yv.y - em.y) * er.y);
        q.z += g.z * ((yv.z - em.z) * er.z);
        q.w += g.w * ((yv.w - em.w) * er.w);
        store4<T>(y + idx, g);
    } else {
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        store4<T>(y + idx, v);
    }
}

// block-level reduction of per-thread (s, q) over the pixel-threads that share a channel vector
__device__ __forceinline__ void dw_write_stats(const DwParams& p, float* red, int tid, int nvec, int npt, int c0,
                                               int cbv, f4 s, f4 q) {
    __syncthreads();
    red[tid * 8 + 0] = s.x; red[tid * 8 + 1] = s.y; red[tid * 8 + 2] = s.z; red[tid * 8 + 3] = s.w;
    red[tid * 8 + 4] = q.x; red[tid * 8 + 5] = q.y; red[tid * 8 + 6] = q.z; red[tid * 8 + 7] = q.w;
    __syncthreads();
    if (tid < cbv) {
        const int cv = tid >> 2, e = tid & 3;
        float a = 0.f, b = 0.f;
        for (int pt = 0; pt < npt; ++pt) {
            a += red[(pt * nvec + cv) * 8 + e];
            b += red[(pt * nvec + cv) * 8 + 4 + e];
        }
        p.stats[((size_t)blockIdx.x * 2 + 0) * p.C + c0 + tid] = a;
        p.stats[((size_t)blockIdx.x * 2 + 1) * p.C + c0 + tid] = b;
    }
}

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(DwParams p) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NIN = (DW_PW - 1) * S + K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CB = p.CB, TH = p.TH, TW = p.TW;
    const int IHT = (TH - 1) * S + K, IWT = (TW - 1) * S + K;
    float* wsm = reinterpret_cast<float*>(smem);                   // [K*K][CB]
    float* red = wsm + K * K * CB;                                 // [256][8]
    T* tile = reinterpret_cast<T*>(red + 256 * 8);                 // [IHT][IWT][CB]

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * CB;
    const int cbv = (p.C - c0) < CB ? (p.C - c0) : CB;  // valid channels in this block
    const int nvec = CB / 4, npt = 256 / nvec;
    const int cv = tid % nvec, pt = tid / nvec;
    const int c = c0 + cv * 4;
    const bool cvalid = (cv * 4 < cbv) && (pt < npt);
    const bool has_pro = p.pscale != nullptr;

    for (int i = tid; i < K * K * CB; i += 256) {
        const int t = i / CB, cc = i % CB;
        const int ts = p.flip ? (K * K - 1 - t) : t;
        wsm[i] = (cc < cbv) ? p.w[(size_t)(c0 + cc) * (K * K) + ts] : 0.f;
    }
    f4 psc = {1, 1, 1, 1}, psh = {0, 0, 0, 0};
    // the loader's channel vector is (item % nvec) == tid % nvec because 256 % nvec may be != 0;
    // so the loader recomputes its vector per item (below) and reads scale/shift from global/L1.
    f4 es = {0, 0, 0, 0}, eh = es, em = es, er = es;
    if (p.yfwd && cvalid) {
        es = load4<float>(p.escale + c); eh = load4<float>(p.eshift + c);
        em = load4<float>(p.emean + c); er = load4<float>(p.erstd + c);
    }
    f4 st_s = {0, 0, 0, 0}, st_q = {0, 0, 0, 0};

    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    const int strips_w = TW / DW_PW;
    const int nstrips = TH * strips_w;

    for (int work = blockIdx.x; work < total; work += gridDim.x) {
        const int b = work / tiles_per_img;
        const int tr = work % tiles_per_img;
        const int oh0 = (tr / p.tiles_w) * TH, ow0 = (tr % p.tiles_w) * TW;
        const int ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        __syncthreads();  // previous tile fully consumed (and wsm written, first time)
        // ---- stage the input halo tile ----
        const int items = IHT * IWT * nvec;
        for (int it = tid; it < items; it += 256) {
            const int v_ = it % nvec, pix = it / nvec;
            const int ih = ih0 + pix / IWT, iw = iw0 + pix % IWT;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && v_ * 4 < cbv) {
                const int cg = c0 + v_ * 4;
                v = load4<T>(x + (((size_t)b * p.H + ih) * p.W + iw) * p.C + cg);
                if (has_pro) {
                    psc = load4<float>(p.pscale + cg);
                    psh = load4<float>(p.pshift + cg);
                    v.x = siluf_(v.x * psc.x + psh.x);
                    v.y = siluf_(v.y * psc.y + psh.y);
                    v.z = siluf_(v.z * psc.z + psh.z);
                    v.w = siluf_(v.w * psc.w + psh.w);
                }
            }
            store4<T>(tile + (size_t)pix * CB + v_ * 4, v);
        }
        __syncthreads();
        if (cvalid) {
            for (int strip = pt; strip < nstrips; strip += npt) {
                const int ohl = strip / strips_w, owl = (strip % strips_w) * DW_PW;
                f4 acc[DW_PW];
#pragma unroll
                for (int q_ = 0; q_ < DW_PW; ++q_) acc[q_] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int kh = 0; kh < K; ++kh) {
                    const T* row = tile + ((size_t)(ohl * S + kh) * IWT + owl * S) * CB + cv * 4;
                    f4 in[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) in[j] = load4<T>(row + j * CB);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const f4 wv = load4<float>(wsm + (kh * K + kw) * CB + cv * 4);
#pragma unroll
                        for (int q_ = 0; q_ < DW_PW; ++q_) {
                            const f4 a = in[q_ * S + kw];
                            acc[q_].x += a.x * wv.x;
                            acc[q_].y += a.y * wv.y;
                            acc[q_].z += a.z * wv.z;
                            acc[q_].w += a.w * wv.w;
                        }
                    }
                }
                const int oh = oh0 + ohl;
                if (oh < p.OH) {
#pragma unroll
                    for (int q_ = 0; q_ < DW_PW; ++q_) {
                        const int ow = ow0 + owl + q_;
                        if (ow < p.OW) {
                            const size_t idx = (((size_t)b * p.OH + oh) * p.OW + ow) * p.C + c;
                            dw_epilogue_store<T>(p, idx, c, acc[q_], es, eh, em, er, st_s, st_q);
                        }
                    }
                }
            }
        }
    }
    if (p.stats) dw_write_stats(p, red, tid, nvec, npt, c0, cbv, st_s, st_q);
}

// stride-S data gradient by gathering: dA[ih,iw,c] = sum_{kh,kw} dY[(ih+PAD-kh)/S, (iw+PAD-kw)/S, c] * w[c,kh,kw]
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_data_gather_kernel(DwParams p) {
    // here: p.x = dY [B,OH,OW,C]; p.y = dA [B,H,W,C]; p.yfwd = forward raw input-side tensor [B,H,W,C]
    constexpr int PAD = (K - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CB = p.CB;
    float* wsm = reinterpret_cast<float*>(smem);  // [K*K][CB]
    float* red = wsm + K * K * CB;                // [256][8]
    const T* __restrict__ dy = reinterpret_cast<const T*>(p.x);
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * CB;
    const int cbv = (p.C - c0) < CB ? (p.C - c0) : CB;
    const int nvec = CB / 4, npt = 256 / nvec;
    const int cv = tid % nvec, pt = tid / nvec;
    const int c = c0 + cv * 4;
    const bool cvalid = (cv * 4 < cbv) && (pt < npt);
    for (int i = tid; i < K * K * CB; i += 256) {
        const int t = i / CB, cc = i % CB;
        wsm[i] = (cc < cbv) ? p.w[(size_t)(c0 + cc) * (K * K) + t] : 0.f;
    }
    f4 es = {0, 0, 0, 0}, eh = es, em = es, er = es;
    if (p.yfwd && cvalid) {
        es = load4<float>(p.escale + c); eh = load4<float>(p.eshift + c);
        em = load4<float>(p.emean + c); er = load4<float>(p.erstd + c);
    }
    f4 st_s = {0, 0, 0, 0}, st_q = {0, 0, 0, 0};
    __syncthreads();
    const long long npix = (long long)p.B * p.H * p.W;
    if (cvalid) {
        for (long long pix = (long long)blockIdx.x * npt + pt; pix < npix; pix += (long long)gridDim.x * npt) {
            const int iw = (int)(pix % p.W);
            const int ih = (int)((pix / p.W) % p.H);
            const int b = (int)(pix / ((long long)p.W * p.H));
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const int th = ih + PAD - kh;
                if (th < 0 || (th % S) != 0) continue;
                const int oh = th / S;
                if (oh >= p.OH) continue;
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const int tw = iw + PAD - kw;
                    if (tw < 0 || (tw % S) != 0) continue;
                    const int ow = tw / S;
                    if (ow >= p.OW) continue;
                    const f4 g = load4<T>(dy + (((size_t)b * p.OH + oh) * p.OW + ow) * p.C + c);
                    const f4 wv = load4<float>(wsm + (kh * K + kw) * CB + cv * 4);
                    acc.x += g.x * wv.x; acc.y += g.y * wv.y; acc.z += g.z * wv.z; acc.w += g.w * wv.w;
                }
            }
            const size_t idx = (size_t)pix * p.C + c;
            dw_epilogue_store<T>(p, idx, c, acc, es, eh, em, er, st_s, st_q);
        }
    }
    if (p.stats) dw_write_stats(p, red, tid, nvec, npt, c0, cbv, st_s, st_q);
}

// weight gradient: dW[c][kh][kw] += sum_{b,oh,ow} dY[b,oh,ow,c] * pro(x)[b, oh*S+kh-PAD, ow*S+kw-PAD, c]
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(DwParams p, const void* dyv, float* dw) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NIN = (DW_PW - 1) * S + K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CB = p.CB, TH = p.TH, TW = p.TW;
    const int IHT = (TH - 1) * S + K, IWT = (TW - 1) * S + K;
    float* red = reinterpret_cast<float*>(smem);          // [256][4]
    T* tile = reinterpret_cast<T*>(red + 256 * 4);        // [IHT][IWT][CB]
    T* dtile = tile + (size_t)IHT * IWT * CB;             // [TH][TW][CB]

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(dyv);
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * CB;
    const int cbv = (p.C - c0) < CB ? (p.C - c0) : CB;
    const int nvec = CB / 4, npt = 256 / nvec;
    const int cv = tid % nvec, pt = tid / nvec;
    const bool cvalid = (cv * 4 < cbv) && (pt < npt);
    const bool has_pro = p.pscale != nullptr;

    f4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};

    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    const int strips_w = TW / DW_PW;
    const int nstrips = TH * strips_w;

    for (int work = blockIdx.x; work < total; work += gridDim.x) {
        const int b = work / tiles_per_img;
        const int tr = work % tiles_per_img;
        const int oh0 = (tr / p.tiles_w) * TH, ow0 = (tr % p.tiles_w) * TW;
        const int ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        __syncthreads();
        const int items = IHT * IWT * nvec;
        for (int it = tid; it < items; it += 256) {
            const int v_ = it % nvec, pix = it / nvec;
            const int ih = ih0 + pix / IWT, iw = iw0 + pix % IWT;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && v_ * 4 < cbv) {
                const int cg = c0 + v_ * 4;
                v = load4<T>(x + (((size_t)b * p.H + ih) * p.W + iw) * p.C + cg);
                if (has_pro) {
                    const f4 psc = load4<float>(p.pscale + cg);
                    const f4 psh = load4<float>(p.pshift + cg);
                    v.x = siluf_(v.x * psc.x + psh.x);
                    v.y = siluf_(v.y * psc.y + psh.y);
                    v.z = siluf_(v.z * psc.z + psh.z);
                    v.w = siluf_(v.w * psc.w + psh.w);
                }
            }
            store4<T>(tile + (size_t)pix * CB + v_ * 4, v);
        }
        const int ditems = TH * TW * nvec;
        for (int it = tid; it < ditems; it += 256) {
            const int v_ = it % nvec, pix = it / nvec;
            const int oh = oh0 + pix / TW, ow = ow0 + pix % TW;
            f4 v = {0.f, 0.f, 0.f, 0.f};
            if (oh < p.OH && ow < p.OW && v_ * 4 < cbv)
                v = load4<T>(dy + (((size_t)b * p.OH + oh) * p.OW + ow) * p.C + c0 + v_ * 4);
            store4<T>(dtile + (size_t)pix * CB + v_ * 4, v);
        }
        __syncthreads();
        if (cvalid) {
            for (int strip = pt; strip < nstrips; strip += npt) {
                const int ohl = strip / strips_w, owl = (strip % strips_w) * DW_PW;
                f4 g[DW_PW];
#pragma unroll
                for (int q_ = 0; q_ < DW_PW; ++q_)
                    g[q_] = load4<T>(dtile + ((size_t)ohl * TW + owl + q_) * CB + cv * 4);
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const T* row = tile + ((size_t)(ohl * S + kh) * IWT + owl * S) * CB + cv * 4;
                    f4 in[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) in[j] = load4<T>(row + j * CB);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
#pragma unroll
                        for (int q_ = 0; q_ < DW_PW; ++q_) {
                            const f4 a = in[q_ * S + kw];
                            acc[kh * K + kw].x += a.x * g[q_].x;
                            acc[kh * K + kw].y += a.y * g[q_].y;
                            acc[kh * K + kw].z += a.z * g[q_].z;
                            acc[kh * K + kw].w += a.w * g[q_].w;
                        }
                    }
                }
            }
        }
    }
    // reduce over the pixel-threads of each channel vector, one tap at a time
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        __syncthreads();
        red[tid * 4 + 0] = acc[t].x; red[tid * 4 + 1] = acc[t].y;
        red[tid * 4 + 2] = acc[t].z; red[tid * 4 + 3] = acc[t].w;
        __syncthreads();
        if (tid < cbv) {
            const int v_ = tid >> 2, e = tid & 3;
            float a = 0.f;
            for (int q_ = 0; q_ < npt; ++q_) a += red[(q_ * nvec + v_) * 4 + e];
            // slab row of this workgroup: [gridDim.x][C][K*K]; one plain store per output, no contention
            dw[((size_t)blockIdx.x * p.C + c0 + tid) * (K * K) + t] = a;
        }
    }
}

// out[i] = sum over rows of slab[row][i]  (fixed order => reproducible)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, int rows, long long n, float* out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += slab[(size_t)r * n + i];
    out[i] = a;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int pick_cb(int C) {
    if (C % 64 == 0) return 64;
    if (C % 48 == 0) return 48;
    if (C <= 64) return C;
    if (C % 32 == 0) return 32;
    if (C % 16 == 0) return 16;
    return 8;
}

// choose the output tile: maximise useful outputs per staged input pixel within an LDS budget
static void pick_tile(int OH, int OW, int K, int S, int CB, int elem, int extra_out_tile, int lds_budget, int* th,
                      int* tw) {
    static const int ths[] = {4, 5, 8, 10, 16};
    static const int tws[] = {8, 12, 16, 20, 24, 32};
    double best = -1.0;
    *th = 4; *tw = 8;
    for (int a : ths)
        for (int b : tws) {
            const int iht = (a - 1) * S + K, iwt = (b - 1) * S + K;
            long long bytes = (long long)iht * iwt * CB * elem + (extra_out_tile ? (long long)a * b * CB * elem : 0);
            if (bytes > lds_budget) continue;
            const int nth = cdiv(OH, a), ntw = cdiv(OW, b);
            const double useful = (double)OH * OW;
            const double staged = (double)nth * ntw * iht * iwt;
            const double slots = (double)nth * ntw * a * b;
            const double score = useful / staged * (useful / slots);
            if (score > best) { best = score; *th = a; *tw = b; }
        }
}

template <typename T, int K, int S>
static void launch_tile(const DwParams& p, int grid_x, hipStream_t s) {
    const int IHT = (p.TH - 1) * S + K, IWT = (p.TW - 1) * S + K;
    const size_t lds = (size_t)K * K * p.CB * 4 + 256 * 8 * 4 + (size_t)IHT * IWT * p.CB * sizeof(T);
    dim3 grid(grid_x, cdiv(p.C, p.CB));
    hipLaunchKernelGGL((dwconv_tile_kernel<T, K, S>), grid, dim3(256), lds, s, p);
}

template <typename T>
static int dispatch_tile(const DwParams& p, int K, int S, int grid_x, hipStream_t s) {
    if (K == 3 && S == 1) launch_tile<T, 3, 1>(p, grid_x, s);
    else if (K == 3 && S == 2) launch_tile<T, 3, 2>(p, grid_x, s);
    else if (K == 5 && S == 1) launch_tile<T, 5, 1>(p, grid_x, s);
    else if (K == 5 && S == 2) launch_tile<T, 5, 2>(p, grid_x, s);
    else return -1;
    return 0;
}

static int dw_common_checks(const char* name, int dtype, int B, int H, int W, int C, int K, int S) {
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "%s: bad dtype %d", name, dtype);
    MIA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad shape B=%d H=%d W=%d C=%d (C%%4==0)", name,
                  B, H, W, C);
    MIA_CHECK_ARG((K == 3 || K == 5) && (S == 1 || S == 2), "%s: unsupported kernel %d stride %d", name, K, S);
    return 0;
}

extern "C" int mia_dwconv_fwd(int dtype, const void* x, int B, int H, int W, int C, const float* pscale,
                              const float* pshift, const float* w, int K, int S, void* y, float* stats, int part_rows,
                              void* stream) {
    if (dw_common_checks("mia_dwconv_fwd", dtype, B, H, W, C, K, S)) return -1;
    MIA_CHECK_ARG(x && w && y, "mia_dwconv_fwd: null operand");
    MIA_CHECK_ARG(!(pscale == nullptr) == !(pshift == nullptr), "mia_dwconv_fwd: pscale/pshift must come together");
    DwParams p = {};
    p.x = x; p.w = w; p.y = y; p.B = B; p.H = H; p.W = W; p.C = C;
    const int pad = (K - 1) / 2;
    p.OH = (H + 2 * pad - K) / S + 1; p.OW = (W + 2 * pad - K) / S + 1;
    p.pscale = pscale; p.pshift = pshift; p.flip = 0; p.stats = stats;
    p.CB = pick_cb(C);
    const int elem = dtype == MIA_BF16 ? 2 : 4;
    pick_tile(p.OH, p.OW, K, S, p.CB, elem, 0, dtype == MIA_BF16 ? 40960 : 49152, &p.TH, &p.TW);
    p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
    const int total = B * p.tiles_h * p.tiles_w;
    MIA_CHECK_ARG(!stats || (part_rows > 0 && part_rows <= 65535), "mia_dwconv_fwd: stats needs 0 < part_rows <= 65535");
    const int gx = stats ? part_rows : (total < 8192 ? total : 8192);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc = dtype == MIA_BF16 ? dispatch_tile<bf16_t>(p, K, S, gx, s) : dispatch_tile<float>(p, K, S, gx, s);
    MIA_CHECK_ARG(rc == 0, "mia_dwconv_fwd: no kernel for K=%d S=%d", K, S);
    MIA_CHECK_LAUNCH("mia_dwconv_fwd");
    return 0;
}

// dA [B,H,W,C] from dY [B,OH,OW,C]. Optional fused epilogue: dA <- dA * silu'(yfwd*escale+eshift),
// partials sum(g) and sum(g * (yfwd-emean)*erstd) per channel.
extern "C" int mia_dwconv_bwd_data(int dtype, const void* dy, int B, int H, int W, int C, const float* w, int K, int S,
                                   void* dx, const void* yfwd, const float* escale, const float* eshift,
                                   const float* emean, const float* erstd, float* stats, int part_rows, void* stream) {
    if (dw_common_checks("mia_dwconv_bwd_data", dtype, B, H, W, C, K, S)) return -1;
    MIA_CHECK_ARG(dy && w && dx, "mia_dwconv_bwd_data: null operand");
    MIA_CHECK_ARG(!yfwd || (escale && eshift && emean && erstd), "mia_dwconv_bwd_data: epilogue needs all four vectors");
    MIA_CHECK_ARG(!stats || (part_rows > 0 && part_rows <= 65535), "mia_dwconv_bwd_data: bad part_rows");
    const int pad = (K - 1) / 2;
    const int OH = (H + 2 * pad - K) / S + 1, OW = (W + 2 * pad - K) / S + 1;
    DwParams p = {};
    p.w = w; p.B = B; p.C = C; p.yfwd = yfwd; p.escale = escale; p.eshift = eshift; p.emean = emean; p.erstd = erstd;
    p.stats = stats; p.CB = pick_cb(C);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int elem = dtype == MIA_BF16 ? 2 : 4;
    if (S == 1) {
        // correlation with flipped taps: same tiling kernel, input = dY (OHxOW == HxW), output = dA
        p.x = dy; p.y = dx; p.H = OH; p.W = OW; p.OH = H; p.OW = W; p.flip = 1;
        pick_tile(p.OH, p.OW, K, 1, p.CB, elem, 0, dtype == MIA_BF16 ? 40960 : 49152, &p.TH, &p.TW);
        p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
        const int total = B * p.tiles_h * p.tiles_w;
        const int gx = stats ? part_rows : (total < 8192 ? total : 8192);
        int rc = dtype == MIA_BF16 ? dispatch_tile<bf16_t>(p, K, 1, gx, s) : dispatch_tile<float>(p, K, 1, gx, s);
        MIA_CHECK_ARG(rc == 0, "mia_dwconv_bwd_data: no kernel for K=%d", K);
    } else {
        p.x = dy; p.y = dx; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
        const int nvec = p.CB / 4, npt = 256 / nvec;
        const long long npix = (long long)B * H * W;
        long long need = (npix + npt - 1) / npt;
        const int gx = stats ? part_rows : (int)(need < 8192 ? need : 8192);
        const size_t lds = (size_t)K * K * p.CB * 4 + 256 * 8 * 4;
        dim3 grid(gx, cdiv(C, p.CB));
        if (dtype == MIA_BF16) {
            if (K == 3) hipLaunchKernelGGL((dwconv_bwd_data_gather_kernel<bf16_t, 3, 2>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((dwconv_bwd_data_gather_kernel<bf16_t, 5, 2>), grid, dim3(256), lds, s, p);
        } else {
            if (K == 3) hipLaunchKernelGGL((dwconv_bwd_data_gather_kernel<float, 3, 2>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((dwconv_bwd_data_gather_kernel<float, 5, 2>), grid, dim3(256), lds, s, p);
        }
    }
    MIA_CHECK_LAUNCH("mia_dwconv_bwd_data");
    return 0;
}

// dw [C][K*K] fp32 is overwritten. ws: fp32 scratch; every workgroup row writes its partial [C][K*K] there and a
// second kernel sums the rows. The number of rows adapts to ws_floats (at least one row = C*K*K floats).
extern "C" int mia_dwconv_bwd_weight(int dtype, const void* x, int B, int H, int W, int C, const float* pscale,
                                     const float* pshift, const void* dy, int K, int S, float* dw, float* ws,
                                     long long ws_floats, void* stream) {
    if (dw_common_checks("mia_dwconv_bwd_weight", dtype, B, H, W, C, K, S)) return -1;
    MIA_CHECK_ARG(x && dy && dw && ws, "mia_dwconv_bwd_weight: null operand");
    MIA_CHECK_ARG(!(pscale == nullptr) == !(pshift == nullptr), "mia_dwconv_bwd_weight: pscale/pshift come together");
    MIA_CHECK_ARG(ws_floats >= (long long)C * K * K, "mia_dwconv_bwd_weight: workspace smaller than one partial row");
    DwParams p = {};
    p.x = x; p.B = B; p.H = H; p.W = W; p.C = C;
    const int pad = (K - 1) / 2;
    p.OH = (H + 2 * pad - K) / S + 1; p.OW = (W + 2 * pad - K) / S + 1;
    p.pscale = pscale; p.pshift = pshift; p.CB = pick_cb(C);
    const int elem = dtype == MIA_BF16 ? 2 : 4;
    pick_tile(p.OH, p.OW, K, S, p.CB, elem, 1, dtype == MIA_BF16 ? 40960 : 49152, &p.TH, &p.TW);
    p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
    const int total = B * p.tiles_h * p.tiles_w;
    const int ncb = cdiv(C, p.CB);
    long long gx = 1024 / ncb;                      // ~1024 workgroups in all
    if (gx < 8) gx = 8;
    if (gx > 256) gx = 256;
    if (gx > total) gx = total;
    if (gx > ws_floats / ((long long)C * K * K)) gx = ws_floats / ((long long)C * K * K);
    const int IHT = (p.TH - 1) * S + K, IWT = (p.TW - 1) * S + K;
    const size_t lds = 256 * 4 * 4 + ((size_t)IHT * IWT + (size_t)p.TH * p.TW) * p.CB * elem;
    dim3 grid((unsigned)gx, ncb);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MIA_DWW(T_, K_, S_) hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T_, K_, S_>), grid, dim3(256), lds, s, p, dy, ws)
    if (dtype == MIA_BF16) {
        if (K == 3 && S == 1) MIA_DWW(bf16_t, 3, 1);
        else if (K == 3 && S == 2) MIA_DWW(bf16_t, 3, 2);
        else if (K == 5 && S == 1) MIA_DWW(bf16_t, 5, 1);
        else MIA_DWW(bf16_t, 5, 2);
    } else {
        if (K == 3 && S == 1) MIA_DWW(float, 3, 1);
        else if (K == 3 && S == 2) MIA_DWW(float, 3, 2);
        else if (K == 5 && S == 1) MIA_DWW(float, 5, 1);
        else MIA_DWW(float, 5, 2);
    }
#undef MIA_DWW
    const long long nout = (long long)C * K * K;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(nout, 256)), dim3(256), 0, s, ws, (int)gx, nout, dw);
    MIA_CHECK_LAUNCH("mia_dwconv_bwd_weight");
    return 0;
}

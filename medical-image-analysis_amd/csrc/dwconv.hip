// Depthwise k x k convolutions (k in {3,5}, stride in {1,2}) on NHWC tensors, gfx950.
//
// A workgroup owns a (TH x TW output pixels) x (CB channels) tile of one image. The input halo tile is staged
// once into LDS -- with the producer's BatchNorm+SiLU applied on the way in, so the normalised activation never
// exists in HBM -- and every thread computes PH x 4 output patches for ONE 4-channel vector (fixed for the whole
// kernel, so its BN constants live in registers). Each staged input row of a patch is read once and used for
// every tap row that touches it; tap vectors are re-read from LDS (broadcast across pixel-threads). Lanes run fastest over channels, so global
// loads/stores are contiguous CB*sizeof(T)-byte runs per pixel. Index arithmetic in the inner loops is
// incremental: no integer division.
//
//   dwconv_fwd        y = dw(pro(x))                      + per-channel sum/sumsq partials (BN stats)
//   data-gradient     stride 1: the same kernel with flipped taps; stride 2: parity-unrolled tile kernel
//                     both with the fused epilogue g = dA * silu'(z), partials sum(g), sum(g*xhat)
//   dwconv_bwd_weight dW[c][kh][kw] = sum dY * pro(x)     (per-workgroup partial slab + ordered reduce)
#include "common.h"
#include <stdlib.h>

#define DW_PW 4  // output pixels per patch row

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
    int gx, ncb;  // matrix-core kernels: 1-D grid of gx tile walkers x ncb channel blocks (see dw_decode_block)
};

// Four channels as two explicit 2-vectors (f22): each 2-vector op is one packed-f32 instruction on a naturally
// paired register couple. (Left to itself the vectoriser pairs values across taps and spends a move per packed FMA.)
__device__ __forceinline__ void fma22(f22& acc, const f22& a, const f22& b) {
    acc.lo = a.lo * b.lo + acc.lo;
    acc.hi = a.hi * b.hi + acc.hi;
}

template <typename T>
__device__ __forceinline__ void dw_epilogue_store(const DwParams& p, unsigned idx, const f22& v, const f22& es,
                                                  const f22& eh, const f22& em, const f22& er, f22& s, f22& q) {
    T* y = reinterpret_cast<T*>(p.y);
    if (p.yfwd) {
        const f22 yv = load22<T>(reinterpret_cast<const T*>(p.yfwd) + idx);
        f22 g;
        g.lo = v.lo * silu_grad2(yv.lo * es.lo + eh.lo);
        g.hi = v.hi * silu_grad2(yv.hi * es.hi + eh.hi);
        s.lo += g.lo; s.hi += g.hi;
        q.lo += g.lo * ((yv.lo - em.lo) * er.lo);
        q.hi += g.hi * ((yv.hi - em.hi) * er.hi);
        store22<T>(y + idx, g);
    } else {
        s.lo += v.lo; s.hi += v.hi;
        q.lo += v.lo * v.lo; q.hi += v.hi * v.hi;
        store22<T>(y + idx, v);
    }
}

// The same epilogue with the forward value already in registers. The data-gradient kernels load it well before the
// point of use (at the top of a patch, or one pixel group ahead): issued where it is needed, the load's full memory
// latency is exposed once per patch, and these kernels do not have the occupancy to cover it.
template <typename T>
__device__ __forceinline__ void dw_epilogue_store_pre(const DwParams& p, unsigned idx, const f22& v, const f22& yv,
                                                      const f22& es, const f22& eh, const f22& em, const f22& er, f22& s,
                                                      f22& q) {
    T* y = reinterpret_cast<T*>(p.y);
    f22 g;
    g.lo = v.lo * silu_grad2(yv.lo * es.lo + eh.lo);
    g.hi = v.hi * silu_grad2(yv.hi * es.hi + eh.hi);
    s.lo += g.lo; s.hi += g.hi;
    q.lo += g.lo * ((yv.lo - em.lo) * er.lo);
    q.hi += g.hi * ((yv.hi - em.hi) * er.hi);
    store22<T>(y + idx, g);
}

// block-level reduction of per-thread (s, q) over the pixel-threads that share a channel vector
__device__ __forceinline__ void dw_write_stats(const DwParams& p, float* red, int tid, int nvec, int npt, int c0,
                                               int cbv, const f22& s, const f22& q, int slot) {
    __syncthreads();
    red[tid * 8 + 0] = s.lo.x; red[tid * 8 + 1] = s.lo.y; red[tid * 8 + 2] = s.hi.x; red[tid * 8 + 3] = s.hi.y;
    red[tid * 8 + 4] = q.lo.x; red[tid * 8 + 5] = q.lo.y; red[tid * 8 + 6] = q.hi.x; red[tid * 8 + 7] = q.hi.y;
    __syncthreads();
    if (tid < cbv) {
        const int cv = tid >> 2, e = tid & 3;
        float a = 0.f, b = 0.f;
        for (int pt = 0; pt < npt; ++pt) {
            a += red[(pt * nvec + cv) * 8 + e];
            b += red[(pt * nvec + cv) * 8 + 4 + e];
        }
        p.stats[((size_t)slot * 2 + 0) * p.C + c0 + tid] = a;
        p.stats[((size_t)slot * 2 + 1) * p.C + c0 + tid] = b;
    }
}

// Staging of a [rows x cols] pixel window of one image for this thread's channel vector, split in two so that the
// global loads of the NEXT tile are in flight while the current tile is being computed from LDS:
//   dw_tile_fetch   issues up to NV vector loads into registers (zero outside the image / channel range)
//   dw_tile_commit  applies the optional BatchNorm+SiLU prologue and writes the vectors to the LDS tile
// Thread (pt, cv) walks pixels pt, pt+npt, ... keeping (row, col) incrementally. The host sizes tiles so that a
// thread owns at most NV pixels.
template <typename T>
struct Raw4T;
template <>
struct Raw4T<bf16_t> {
    typedef uint2 type;
    static __device__ __forceinline__ type zero() { return make_uint2(0, 0); }
    static __device__ __forceinline__ f22 unpack22(type v) { return {unpack_bf16x2(v.x), unpack_bf16x2(v.y)}; }
};
template <>
struct Raw4T<float> {
    typedef float4 type;
    static __device__ __forceinline__ type zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ f22 unpack22(type v) { return {v2f{v.x, v.y}, v2f{v.z, v.w}}; }
};

template <typename T, int NV>
struct TileRegs {
    typename Raw4T<T>::type v[NV];
    unsigned inb;  // bit u: v[u] was loaded from inside the image
};

template <typename T, int NV>
__device__ __forceinline__ void dw_tile_fetch(TileRegs<T, NV>& t, const T* __restrict__ img, int rows, int cols, int r0,
                                              int c0px, int H, int W, int C, int cg, bool chan_ok, int pt, int npt) {
    t.inb = 0u;
    if (pt >= npt) return;
    // The walk below depends only on (pt, tile shape), not on the tile's position. Its start (one division) is left to
    // the optimiser to hoist out of the caller's tile loop; everything derived from it is hidden behind an opaque copy,
    // so that the few adds are redone per tile instead of 3*NV hoisted index registers staying alive across the
    // compute phase (which would halve the occupancy).
    int row = pt / cols, col = pt - row * cols;
    asm volatile("" : "+v"(pt), "+v"(row), "+v"(col));
    const int npix = rows * cols;
    const int drow = npt / cols, dcol = npt - drow * cols;
    // position and element offset advance together: one add per step and one more where the column wraps, instead of
    // two integer multiplications per loaded vector (quarter-rate instructions: they were a third of the staging)
    int pix = pt, ih = r0 + row, iw = c0px + col;
    const int iw_end = c0px + cols;
    unsigned off = (unsigned)((ih * W + iw) * C + cg);
    const unsigned step = (unsigned)((drow * W + dcol) * C), wrap = (unsigned)((W - cols) * C);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const bool ok = pix < npix && chan_ok && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        t.v[u] = Raw4T<T>::zero();
        if (ok) {
            t.v[u] = *reinterpret_cast<const typename Raw4T<T>::type*>(img + off);
            t.inb |= 1u << u;
        }
        pix += npt;
        ih += drow;
        iw += dcol;
        off += step;
        if (iw >= iw_end) { iw -= cols; ++ih; off += wrap; }
    }
}

template <typename T, int NV>
__device__ __forceinline__ void dw_tile_commit(const TileRegs<T, NV>& t, T* tile, int npix, int cv, int pt, int npt,
                                               int CB, bool has_pro, const f22& psc, const f22& psh) {
    if (pt >= npt) return;
    asm volatile("" : "+v"(pt));  // as in dw_tile_fetch: recompute the LDS offsets per tile, do not keep them
    int pix = pt;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (pix < npix) {
            T* dst = tile + (unsigned)(pix * CB + cv * 4);
            if (has_pro && ((t.inb >> u) & 1u)) {
                f22 v = Raw4T<T>::unpack22(t.v[u]);
                v.lo = silu2(v.lo * psc.lo + psh.lo);
                v.hi = silu2(v.hi * psc.hi + psh.hi);
                store22<T>(dst, v);
            } else {
                *reinterpret_cast<typename Raw4T<T>::type*>(dst) = t.v[u];
            }
        }
        pix += npt;
        // four vectors at a time: left alone, the scheduler interleaves all NV activations (4*NV live floats)
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// Which tiles a workgroup walks. Consecutive block ids land on different XCDs (each with its own L2), so the tile
// list is cut into eight contiguous ranges, one per XCD, and the XCD's workgroups stride through their range
// together: tiles that share halo rows are then in flight on the same L2. Falls back to a plain grid stride when
// the grid is not a multiple of eight.
struct TileWalk {
    int base, count, first, stride;
};
__device__ __forceinline__ TileWalk make_tile_walk(int total, int slot, int nslots) {
    TileWalk w;
    if (nslots & 7) {
        w.base = 0; w.count = total; w.first = slot; w.stride = nslots;
    } else {
        const int per = (total + 7) >> 3, xcd = slot & 7;
        w.base = xcd * per;
        w.count = total - w.base < per ? total - w.base : per;
        if (w.count < 0) w.count = 0;
        w.first = slot >> 3;
        w.stride = nslots >> 3;
    }
    return w;
}
__device__ __forceinline__ TileWalk make_tile_walk(int total) { return make_tile_walk(total, blockIdx.x, gridDim.x); }

// 1-D grid of gx walkers x ncb channel blocks. A channel block's 64- or 128-byte run inside a pixel is not aligned
// to the 128-byte cache line when the pixel stride (2*C bytes) is not a multiple of it (C = 288, 816, 1392), so
// neighbouring channel blocks share lines; workgroup ids are dealt to the XCDs round-robin, so the blocks of one
// walker are given ids with the same id%8 and adjacent dispatch order: they run on one XCD at about the same time
// and the shared lines come out of its L2 instead of being fetched once per XCD.
__device__ __forceinline__ void dw_decode_block(const DwParams& p, int& slot, int& cb) {
    const int id = blockIdx.x;
    if (p.gx & 7) {
        cb = id % p.ncb;
        slot = id / p.ncb;
    } else {
        const int r = id >> 3;
        cb = r % p.ncb;
        slot = (r / p.ncb) * 8 + (id & 7);
    }
}

constexpr int DW_NV = 16;   // pixels per thread in a staged input tile (forward / data-gradient kernel)

constexpr int DW_NVX = 12;  // weight-gradient kernel: input tile
constexpr int DW_NVD = 8;   //                          output-gradient tile

// EPI: p.yfwd != nullptr (the data gradient's fused epilogue). A compile-time choice: the forward instance then carries
// neither the epilogue's four channel vectors nor its branch (158 -> 142 registers at 3x3 stride 1, 142 -> 126 = four
// waves per SIMD at stride 2; forward launches 3-11 % faster). Forcing the stride-1 instance down to 128 spills.
// NV: staging slots per thread (12 or 16). Every slot costs its address walk and bounds tests whether or not the tile
// reaches it; the tiles of the large maps fill 10-12 of 16 (180 pixels over 16 pixel-threads at 75x75x192, 264 over 25
// at 150x150x40), so the host picks the 12-slot instance when the tile fits it.
template <typename T, int K, int S, int PH, bool EPI, int NV>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(DwParams p) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int NIN = (DW_PW - 1) * S + K;   // staged input pixels per patch row
    constexpr int NIR = (PH - 1) * S + K;      // staged input rows per patch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CB = p.CB, TH = p.TH, TW = p.TW;
    const int IHT = (TH - 1) * S + K, IWT = (TW - 1) * S + K;
    float* wsm = reinterpret_cast<float*>(smem);                   // [K*K][CB]
    float* red = wsm + K * K * CB;                                 // [256][8]
    T* tile = reinterpret_cast<T*>(red + 256 * 8);                 // [IHT][IWT][CB]

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const int tid = threadIdx.x;
    int slot, cblk;
    dw_decode_block(p, slot, cblk);
    const int c0 = cblk * CB;
    const int cbv = (p.C - c0) < CB ? (p.C - c0) : CB;  // valid channels in this block
    const int nvec = CB / 4, npt = 256 / nvec;
    const int cv = tid % nvec, pt = tid / nvec;
    const int c = c0 + cv * 4;
    const bool chan_ok = cv * 4 < cbv;
    const bool cvalid = chan_ok && (pt < npt);
    const bool has_pro = p.pscale != nullptr;

    for (int i = tid; i < K * K * CB; i += 256) {
        const int t = i / CB, cc = i - t * CB;
        const int ts = p.flip ? (K * K - 1 - t) : t;
        wsm[i] = (cc < cbv) ? p.w[(size_t)(c0 + cc) * (K * K) + ts] : 0.f;
    }
    const f22 zero22 = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    f22 psc = {v2f{1.f, 1.f}, v2f{1.f, 1.f}}, psh = zero22;
    if (has_pro && chan_ok) { psc = load22<float>(p.pscale + c); psh = load22<float>(p.pshift + c); }
    f22 es = zero22, eh = zero22, em = zero22, er = zero22;
    if constexpr (EPI) {
        if (cvalid) {
            es = load22<float>(p.escale + c); eh = load22<float>(p.eshift + c);
            em = load22<float>(p.emean + c); er = load22<float>(p.erstd + c);
        }
    }
    f22 st_s = zero22, st_q = zero22;
    __syncthreads();
    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    const int patches_w = TW / DW_PW, patches_h = TH / PH;
    const int npatch = patches_h * patches_w;
    const int dph = npt / patches_w, dpw = npt - dph * patches_w;

    TileRegs<T, NV> regs;
    auto fetch = [&](int work) {
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int oh0 = trh * TH, ow0 = (tr - trh * p.tiles_w) * TW;
        dw_tile_fetch<T, NV>(regs, x + (size_t)b * p.H * p.W * p.C, IHT, IWT, oh0 * S - PAD, ow0 * S - PAD, p.H, p.W,
                                p.C, c, chan_ok, pt, npt);
    };
    const TileWalk walk = make_tile_walk(total, slot, p.gx);
    if (walk.first < walk.count) fetch(walk.base + walk.first);
    for (int wi = walk.first; wi < walk.count; wi += walk.stride) {
        const int work = walk.base + wi;
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int oh0 = trh * TH, ow0 = (tr - trh * p.tiles_w) * TW;
        __syncthreads();  // previous tile fully consumed
        dw_tile_commit<T, NV>(regs, tile, IHT * IWT, cv, pt, npt, CB, has_pro, psc, psh);
        __syncthreads();
        if (wi + walk.stride < walk.count) fetch(work + walk.stride);  // in flight while this tile is computed
        if (cvalid) {
            int ptc = pt;
            asm volatile("" : "+v"(ptc));  // patch walk recomputed per tile, not hoisted (registers)
            int ph = ptc / patches_w, pw = ptc - ph * patches_w;
            for (int patch = ptc; patch < npatch; patch += npt) {
                const int ohl = ph * PH, owl = pw * DW_PW;
                // the epilogue's operand, requested before the taps are computed (addresses clamped into the image so
                // that the loads need no branches; the stores below keep their edge tests)
                typename Raw4T<T>::type yraw[EPI ? PH : 1][DW_PW];
                if constexpr (EPI) {
                    const T* yimg = reinterpret_cast<const T*>(p.yfwd) + (size_t)b * p.OH * p.OW * p.C + c;
#pragma unroll
                    for (int a = 0; a < PH; ++a) {
                        int oh = oh0 + ohl + a;
                        oh = oh < p.OH ? oh : p.OH - 1;
#pragma unroll
                        for (int q_ = 0; q_ < DW_PW; ++q_) {
                            int ow = ow0 + owl + q_;
                            ow = ow < p.OW ? ow : p.OW - 1;
                            yraw[a][q_] = *reinterpret_cast<const typename Raw4T<T>::type*>(
                                yimg + (unsigned)((oh * p.OW + ow) * p.C));
                        }
                    }
                }
                f22 acc[PH][DW_PW];
#pragma unroll
                for (int a = 0; a < PH; ++a)
#pragma unroll
                    for (int q_ = 0; q_ < DW_PW; ++q_) acc[a][q_] = f22{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
                // one staged input row at a time (a real loop: keeps the live set to one row + the accumulators);
                // the row feeds every output row of the patch whose tap row kh = ir - a*S is in range
                const T* row = tile + (unsigned)(((ohl * S) * IWT + owl * S) * CB + cv * 4);
#pragma unroll 1
                for (int ir = 0; ir < NIR; ++ir, row += IWT * CB) {
                    f22 in[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) in[j] = load22<T>(row + j * CB);
#pragma unroll
                    for (int a = 0; a < PH; ++a) {
                        const int kh = ir - a * S;
                        if (kh >= 0 && kh < K) {
                            const float* wrow = wsm + (unsigned)(kh * K * CB + cv * 4);
#pragma unroll
                            for (int kw = 0; kw < K; ++kw) {
                                const f22 wv = load22<float>(wrow + kw * CB);
#pragma unroll
                                for (int q_ = 0; q_ < DW_PW; ++q_) fma22(acc[a][q_], in[q_ * S + kw], wv);
                            }
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < PH; ++a) {
                    const int oh = oh0 + ohl + a;
                    if (oh < p.OH) {
#pragma unroll
                        for (int q_ = 0; q_ < DW_PW; ++q_) {
                            const int ow = ow0 + owl + q_;
                            if (ow < p.OW) {
                                const unsigned idx = (unsigned)(((b * p.OH + oh) * p.OW + ow) * p.C + c);
                                if constexpr (EPI)
                                    dw_epilogue_store_pre<T>(p, idx, acc[a][q_], Raw4T<T>::unpack22(yraw[a][q_]), es, eh,
                                                             em, er, st_s, st_q);
                                else {
                                    // no epilogue operand: the value itself and its column statistics
                                    const f22 v = acc[a][q_];
                                    st_s.lo += v.lo; st_s.hi += v.hi;
                                    st_q.lo += v.lo * v.lo; st_q.hi += v.hi * v.hi;
                                    store22<T>(reinterpret_cast<T*>(p.y) + idx, v);
                                }
                            }
                        }
                    }
                }
                ph += dph;
                pw += dpw;
                if (pw >= patches_w) { pw -= patches_w; ++ph; }
            }
        }
    }
    if (p.stats) dw_write_stats(p, red, tid, nvec, npt, c0, cbv, st_s, st_q, slot);
}

// ------------------------------------------------------------------------------------------------
// bf16 depthwise convolution on the matrix cores (forward, and the stride-1 data gradient with flipped taps).
// A depthwise tap is a product with a DIAGONAL matrix: for 16 channels and 16 pixels,
//   D[ch][pix] += diag(w_tap)[ch][ch'] * X_tap[ch'][pix],   X_tap = the staged input shifted by the tap.
// One v_mfma_f32_16x16x16 per tap does that for 256 outputs; 15/16 of its multiplies hit zeros, but the matrix pipe
// has the headroom (a whole 5x5 layer is ~10 us of MFMA time) and the vector ALU is left with staging and the
// epilogue only, which is what bounded the per-thread-patch kernel above.
// Operand layouts make this elementwise from the lane's point of view: lane (pix = lane%16, quad = lane/16) supplies
// channels 4*quad..+3 of pixel `pix` as the B operand -- one 8-byte LDS read at a per-lane base plus a per-tap
// offset -- and receives the four outputs of exactly those channels at that pixel. The diagonal weights are the A
// operand, fixed for the kernel: wave w of the workgroup owns channels 16w..16w+15 of the 64-channel block.
// Any 16 output pixels of the tile form a group (linear index over the tile), so odd map sizes waste no MFMA rows.
// The LDS pixel stride is padded to 136 bytes: the 32 lanes of a half-wave then cover every bank pair exactly twice.
// Staging (register prefetch of the next tile, BatchNorm+SiLU on the way into LDS) is shared with the kernel above.
// Weights are rounded to bf16 here (the per-thread-patch kernel multiplies by the fp32 masters).
typedef __attribute__((ext_vector_type(4))) float dw_f32x4;
typedef __attribute__((ext_vector_type(4))) short dw_s16x4;
// Channels per workgroup: 64 (wave w owns channel group w) or 32 (waves w and w+2 share channel group w%2 and take
// alternate pixel groups). The narrower block halves the LDS per staged pixel, so that twice as large a window fits:
// whole 19x19 maps with their 5x5 halo instead of four 10x10 quarters that stage every pixel 2.2 times.
constexpr int DWM_NV = 17;    // staged pixels per thread
// LDS pixel padding in elements. Measured at 4, 8 and 12 (pixel strides of 72/80/88 and 136/144/152 bytes): 4 is the
// fastest on every 5x5 and 3x3 layer of the network, by 5-15 % on the 19x19 and 38x38 maps.
constexpr int DWM_PAD = 4;

template <int K, int S, int DWM_CB, bool EPI>   // EPI: p.yfwd != nullptr, as in dwconv_tile_kernel
__global__ __launch_bounds__(256, (EPI ? (K == 5 ? 3 : 4) : 1)) void dwconv_mfma_kernel(DwParams p) {
    typedef bf16_t T;
    constexpr int PAD = (K - 1) / 2;
    constexpr int DWM_CBP = DWM_CB + DWM_PAD;   // LDS pixel stride in elements
    constexpr int NG = DWM_CB / 16;       // channel groups
    constexpr int WPG = 4 / NG;           // waves per channel group
    constexpr int nvec = DWM_CB / 4, npt = 256 / nvec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TH = p.TH, TW = p.TW;
    const int IHT = (TH - 1) * S + K, IWT = (TW - 1) * S + K;
    const int npos = TH * TW, npos16 = (npos + 15) & ~15;
    int4* ptab = reinterpret_cast<int4*>(smem);      // [npos16]: per output pixel of the tile, see below
    T* tile = reinterpret_cast<T*>(ptab + npos16);   // [IHT*IWT][DWM_CBP]
    float* red = reinterpret_cast<float*>(tile);     // [256][8], used after the last tile only

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int slot, cblk;
    dw_decode_block(p, slot, cblk);
    const int c0 = cblk * DWM_CB;
    const int cbv = (p.C - c0) < DWM_CB ? (p.C - c0) : DWM_CB;
    // staging role: thread (pt, cv) moves the 4-channel vector cv of pixels pt, pt+npt, ...
    const int cv = tid % nvec, pt = tid / nvec;
    const int cst = c0 + cv * 4;
    const bool st_ok = cv * 4 < cbv;
    // compute role
    const int pix_l = lane & 15, quad = lane >> 4;
    const int grp = wave % NG, sub = wave / NG;
    const int cl = grp * 16 + quad * 4;
    const int c = c0 + cl;
    const bool cvalid = cl < cbv;
    const bool wave_on = grp * 16 < cbv;
    const bool has_pro = p.pscale != nullptr;

    const f22 zero22 = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    f22 st_s = zero22, st_q = zero22;
    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    // The walk over a tile's output pixels is the same for every tile: tabulate, once, for the q-th pixel its byte
    // offset in the staged window, its element offset in the output image relative to the tile's origin, and its
    // (row, col) for the edge test. The pixel-group loop then costs one LDS read instead of a division per group.
    for (int q = tid; q < npos16; q += 256) {
        const int ohl = q / TW, owl = q - ohl * TW;
        ptab[q] = q < npos ? make_int4(((ohl * S) * IWT + owl * S) * DWM_CBP * 2, (ohl * p.OW + owl) * p.C, ohl, owl)
                           : make_int4(0, 0, 1 << 20, 1 << 20);
    }

    TileRegs<T, DWM_NV> regs;
    auto fetch = [&](int work) {
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int oh0 = trh * TH, ow0 = (tr - trh * p.tiles_w) * TW;
        dw_tile_fetch<T, DWM_NV>(regs, x + (size_t)b * p.H * p.W * p.C, IHT, IWT, oh0 * S - PAD, ow0 * S - PAD, p.H,
                                 p.W, p.C, cst, st_ok, pt, npt);
    };
    const TileWalk walk = make_tile_walk(total, slot, p.gx);
    if (walk.first < walk.count) fetch(walk.base + walk.first);
    // constants are loaded AFTER the first tile's loads are issued: one memory latency for both
    dw_s16x4 wA[K * K];
    {
        const int ch = grp * 16 + pix_l;   // channel of the A row this lane holds
        const bool mine = (pix_l >> 2) == quad && ch < cbv;
        const int j = pix_l & 3;
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            const int ts = p.flip ? (K * K - 1 - t) : t;
            const short wv = mine ? (short)f32_to_bf16_bits(p.w[(size_t)(c0 + ch) * (K * K) + ts]) : (short)0;
            wA[t] = dw_s16x4{j == 0 ? wv : (short)0, j == 1 ? wv : (short)0, j == 2 ? wv : (short)0,
                             j == 3 ? wv : (short)0};
        }
    }
    f22 psc = {v2f{1.f, 1.f}, v2f{1.f, 1.f}}, psh = zero22;
    if (has_pro && st_ok) { psc = load22<float>(p.pscale + cst); psh = load22<float>(p.pshift + cst); }
    f22 es = zero22, eh = zero22, em = zero22, er = zero22;
    if constexpr (EPI) {
        if (cvalid) {
            es = load22<float>(p.escale + c); eh = load22<float>(p.eshift + c);
            em = load22<float>(p.emean + c); er = load22<float>(p.erstd + c);
        }
    }
    for (int wi = walk.first; wi < walk.count; wi += walk.stride) {
        const int work = walk.base + wi;
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int oh0 = trh * TH, ow0 = (tr - trh * p.tiles_w) * TW;
        __syncthreads();  // previous tile fully consumed
        dw_tile_commit<T, DWM_NV>(regs, tile, IHT * IWT, cv, pt, npt, DWM_CBP, has_pro, psc, psh);
        __syncthreads();
        if (wi + walk.stride < walk.count) fetch(work + walk.stride);  // in flight while this tile is computed
        if (wave_on) {
            const int lim_h = p.OH - oh0, lim_w = p.OW - ow0;
            const unsigned out0 = (unsigned)(((b * p.OH + oh0) * p.OW + ow0) * p.C + c);
            // Data gradient: the epilogue's forward value runs one pixel group ahead of the arithmetic. Every lane always
            // loads (lanes without an output read element 0 of the tensor), so the loads are never under a branch and the
            // wait before the epilogue can leave the newer request in flight. (Only in the instance with the epilogue:
            // the forward instance must not carry these registers, it sits at an occupancy step.)
            const T* yimg = reinterpret_cast<const T*>(p.yfwd);
            uint2 y_n = make_uint2(0u, 0u);
            if constexpr (EPI) {
                const int4 pf = ptab[(sub * 16 < npos ? sub * 16 : 0) + pix_l];
                const bool ok = cvalid && pf.z < lim_h && pf.w < lim_w;
                y_n = *reinterpret_cast<const uint2*>(yimg + (ok ? out0 + (unsigned)pf.y : 0u));
            }
            for (int g0 = sub * 16; g0 < npos; g0 += 16 * WPG) {
                const int4 pe = ptab[g0 + pix_l];
                const uint2 y_c = y_n;
                if constexpr (EPI) {
                    const int gn = g0 + 16 * WPG < npos ? g0 + 16 * WPG : g0;
                    const int4 pf = ptab[gn + pix_l];
                    const bool ok = cvalid && pf.z < lim_h && pf.w < lim_w;
                    y_n = *reinterpret_cast<const uint2*>(yimg + (ok ? out0 + (unsigned)pf.y : 0u));
                }
                const T* bp = reinterpret_cast<const T*>(reinterpret_cast<const char*>(tile) + pe.x) + cl;
                dw_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // one tap row of B operands ahead of the MFMAs that consume it, so the LDS latency is paid once per
                // pixel group and not once per row
                dw_s16x4 cur[K], nxt[K];
#pragma unroll
                for (int kw = 0; kw < K; ++kw) cur[kw] = *reinterpret_cast<const dw_s16x4*>(bp + kw * DWM_CBP);
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    if (kh + 1 < K) {
                        const T* rp = bp + (unsigned)((kh + 1) * IWT * DWM_CBP);
#pragma unroll
                        for (int kw = 0; kw < K; ++kw) nxt[kw] = *reinterpret_cast<const dw_s16x4*>(rp + kw * DWM_CBP);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the loads above the MFMAs (the scheduler sinks them)
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
                        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wA[kh * K + kw], cur[kw], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) cur[kw] = nxt[kw];
                }
                if (cvalid && pe.z < lim_h && pe.w < lim_w) {
                    const unsigned idx = out0 + (unsigned)pe.y;
                    const f22 v = {v2f{acc[0], acc[1]}, v2f{acc[2], acc[3]}};
                    if constexpr (EPI) {
                        dw_epilogue_store_pre<T>(p, idx, v, Raw4T<T>::unpack22(y_c), es, eh, em, er, st_s, st_q);
                    } else {
                        st_s.lo += v.lo; st_s.hi += v.hi;
                        st_q.lo += v.lo * v.lo; st_q.hi += v.hi * v.hi;
                        store22<T>(reinterpret_cast<T*>(p.y) + idx, v);
                    }
                }
            }
        }
    }
    if (p.stats) {
        __syncthreads();
        red[tid * 8 + 0] = st_s.lo.x; red[tid * 8 + 1] = st_s.lo.y; red[tid * 8 + 2] = st_s.hi.x; red[tid * 8 + 3] = st_s.hi.y;
        red[tid * 8 + 4] = st_q.lo.x; red[tid * 8 + 5] = st_q.lo.y; red[tid * 8 + 6] = st_q.hi.x; red[tid * 8 + 7] = st_q.hi.y;
        __syncthreads();
        if (tid < cbv) {   // channel c0 + tid: group tid/16, quad (tid%16)/4, element tid%4; sum over its lanes
            const int e = tid & 3;
            float a = 0.f, bsum = 0.f;
            for (int sb = 0; sb < WPG; ++sb) {
                const int src = (sb * NG + (tid >> 4)) * 64 + ((tid >> 2) & 3) * 16;
                for (int pl = 0; pl < 16; ++pl) {
                    a += red[(src + pl) * 8 + e];
                    bsum += red[(src + pl) * 8 + 4 + e];
                }
            }
            p.stats[((size_t)slot * 2 + 0) * p.C + c0 + tid] = a;
            p.stats[((size_t)slot * 2 + 1) * p.C + c0 + tid] = bsum;
        }
    }
}

// Stride-2 data gradient on LDS tiles. A "cell" is one dY position (r, c); it owns the 2x2 block of dA pixels
// (2r+pr, 2c+pc). For an output of parity pr only the taps kh = (pr+PAD) mod 2 (+2, +4) contribute, from dY row
// r + (pr+PAD-kh)/2, i.e. r-1, r or r+1 -- all static once the parity is fixed, so the tap loops unroll with no
// parity tests, no bounds tests (the halo is staged as zeros) and each dY vector is read from LDS once per patch.
// A thread computes 1 x 2 cells = 2 x 4 output pixels for one 4-channel vector.
//   p.x = dY [B,OH,OW,C], p.y = dA [B,H,W,C], p.yfwd = forward tensor at the dA positions (fused silu' epilogue)
template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_bwd_data_s2_kernel(DwParams p) {
    constexpr int PAD = (K - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CB = p.CB, TH = p.TH, TW = p.TW;       // tile in cells
    const int IHT = TH + 2, IWT = TW + 2;            // staged dY window: one cell of halo on every side
    float* wsm = reinterpret_cast<float*>(smem);     // [K*K][CB]
    float* red = wsm + K * K * CB;                   // [256][8]
    T* tile = reinterpret_cast<T*>(red + 256 * 8);   // [IHT][IWT][CB]

    const T* __restrict__ dy = reinterpret_cast<const T*>(p.x);
    const int tid = threadIdx.x;
    int slot, cblk;
    dw_decode_block(p, slot, cblk);
    const int c0 = cblk * CB;
    const int cbv = (p.C - c0) < CB ? (p.C - c0) : CB;
    const int nvec = CB / 4, npt = 256 / nvec;
    const int cv = tid % nvec, pt = tid / nvec;
    const int c = c0 + cv * 4;
    const bool chan_ok = cv * 4 < cbv;
    const bool cvalid = chan_ok && (pt < npt);
    for (int i = tid; i < K * K * CB; i += 256) {
        const int t = i / CB, cc = i - t * CB;
        wsm[i] = (cc < cbv) ? p.w[(size_t)(c0 + cc) * (K * K) + t] : 0.f;
    }
    const f22 zero22 = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    f22 es = zero22, eh = zero22, em = zero22, er = zero22;
    if (p.yfwd && cvalid) {
        es = load22<float>(p.escale + c); eh = load22<float>(p.eshift + c);
        em = load22<float>(p.emean + c); er = load22<float>(p.erstd + c);
    }
    f22 st_s = zero22, st_q = zero22;
    __syncthreads();
    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    const int patches_w = TW / 2;
    const int npatch = TH * patches_w;
    const int dph = npt / patches_w, dpw = npt - dph * patches_w;

    TileRegs<T, DW_NV> regs;
    auto fetch = [&](int work) {
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int r0 = trh * TH, q0 = (tr - trh * p.tiles_w) * TW;
        dw_tile_fetch<T, DW_NV>(regs, dy + (size_t)b * p.OH * p.OW * p.C, IHT, IWT, r0 - 1, q0 - 1, p.OH, p.OW, p.C, c,
                                chan_ok, pt, npt);
    };
    const TileWalk walk = make_tile_walk(total, slot, p.gx);
    if (walk.first < walk.count) fetch(walk.base + walk.first);
    for (int wi = walk.first; wi < walk.count; wi += walk.stride) {
        const int work = walk.base + wi;
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int r0 = trh * TH, q0 = (tr - trh * p.tiles_w) * TW;
        __syncthreads();
        dw_tile_commit<T, DW_NV>(regs, tile, IHT * IWT, cv, pt, npt, CB, false, zero22, zero22);
        __syncthreads();
        if (wi + walk.stride < walk.count) fetch(work + walk.stride);
        if (cvalid) {
            int ptc = pt;
            asm volatile("" : "+v"(ptc));
            int ph = ptc / patches_w, pw = ptc - ph * patches_w;
            for (int patch = ptc; patch < npatch; patch += npt) {
                const int rl = ph, ql = pw * 2;   // first cell of the patch, tile-local
                typename Raw4T<T>::type yraw[2][4];   // the epilogue's operand, requested ahead of the tap arithmetic
                if (p.yfwd) {
                    const T* yimg = reinterpret_cast<const T*>(p.yfwd) + (size_t)b * p.H * p.W * p.C + c;
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        int ih = (r0 + rl) * 2 + pr;
                        ih = ih < p.H ? ih : p.H - 1;
#pragma unroll
                        for (int x_ = 0; x_ < 4; ++x_) {
                            int iw = (q0 + ql) * 2 + x_;
                            iw = iw < p.W ? iw : p.W - 1;
                            yraw[pr][x_] = *reinterpret_cast<const typename Raw4T<T>::type*>(
                                yimg + (unsigned)((ih * p.W + iw) * p.C));
                        }
                    }
                }
                // dY window of the patch: rows rl-1..rl+1, cols ql-1..ql+2 (tile-local + 1 for the halo)
                f22 in[3][4];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        in[a][j] = load22<T>(tile + (unsigned)(((rl + a) * IWT + ql + j) * CB + cv * 4));
                f22 acc[2][4];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int x_ = 0; x_ < 4; ++x_) acc[pr][x_] = zero22;
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int pr = (kh + PAD) & 1;             // output row parity this tap row serves
                    const int dh = (pr + PAD - kh) / 2;        // exact: the numerator is even; in {-1, 0, 1}
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const int pc = (kw + PAD) & 1;
                        const int dw = (pc + PAD - kw) / 2;
                        const f22 wv = load22<float>(wsm + (unsigned)((kh * K + kw) * CB + cv * 4));
#pragma unroll
                        for (int cell = 0; cell < 2; ++cell)
                            fma22(acc[pr][cell * 2 + pc], in[dh + 1][cell + dw + 1], wv);
                    }
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int ih = (r0 + rl) * 2 + pr;
                    if (ih < p.H) {
#pragma unroll
                        for (int x_ = 0; x_ < 4; ++x_) {
                            const int iw = (q0 + ql) * 2 + x_;
                            if (iw < p.W) {
                                const unsigned idx = (unsigned)(((b * p.H + ih) * p.W + iw) * p.C + c);
                                if (p.yfwd)
                                    dw_epilogue_store_pre<T>(p, idx, acc[pr][x_], Raw4T<T>::unpack22(yraw[pr][x_]), es,
                                                             eh, em, er, st_s, st_q);
                                else
                                    dw_epilogue_store<T>(p, idx, acc[pr][x_], es, eh, em, er, st_s, st_q);
                            }
                        }
                    }
                }
                ph += dph;
                pw += dpw;
                if (pw >= patches_w) { pw -= patches_w; ++ph; }
            }
        }
    }
    if (p.stats) dw_write_stats(p, red, tid, nvec, npt, c0, cbv, st_s, st_q, slot);
}

// ------------------------------------------------------------------------------------------------
// Stride-2 data gradient of an inverted-residual block whose BatchNorm backward is carried through the expansion
// convolution. There the gradient g = dA * silu'(z) has exactly two consumers, both linear in it:
//     dxp = g (ca o Wexp)     [pixels x Cin]     ca = gamma * rstd of the expansion's BatchNorm
//     T1  = g^T x             [C x Cin]
// so g is never written to memory. A workgroup owns ALL C channels of a tile of 4 x 6 cells (96 dA pixels), keeps
// the tile's g in LDS as bf16 (the rounding the stored tensor would have had) and feeds the matrix cores from there:
// k-contiguous rows for the first product, transposing LDS reads for the second, whose accumulator stays in
// registers for the whole kernel and goes to a per-workgroup slab at the end (summed in a fixed order afterwards).
// Per tile: stage dY and x, compute g with the vector ALUs exactly as dwconv_bwd_data_s2_kernel does, then
// 15 + 15 (C = 144) matrix instructions per wave. HBM sees dY (with halo), the forward tensor, x and dxp -- the
// expanded gradient's write and its two re-reads are gone.
// ------------------------------------------------------------------------------------------------
struct DwLinParams {
    const bf16_t* x;      // block input [B*H*W][Cin]
    const bf16_t* wexp;   // expansion weights [C][Cin]
    const float* gamma;   // [C]
    bf16_t* dxp;          // [B*H*W][Cin]
    float* slab;          // [gx][C][Cin]
    int Cin;
};
constexpr int DWL_TH = 4, DWL_TW = 6;                 // tile in cells
constexpr int DWL_NPX = 4 * DWL_TH * DWL_TW;          // dA pixels per tile (rows of the g tile)
constexpr int DWL_LDX = 40;                           // LDS row stride of the x tile (32 columns + 8)
constexpr int DWL_LDC = 24;                           // LDS row stride of a wave's output strip (16 columns + 8)
typedef __attribute__((ext_vector_type(8))) __bf16 dw_bf16x8;
typedef __attribute__((ext_vector_type(8))) short dw_s16x8;

// reduction-major LDS image ([red][col]) -> matrix operand: base = &tile[k0][col0], LD = row stride in elements
__device__ __forceinline__ dw_bf16x8 dwl_load_t(const bf16_t* base, int LD, int lane) {
    typedef __attribute__((address_space(3))) dw_s16x4 lds_s16x4;
    const int i = lane & 15, g = lane >> 4;
    const bf16_t* p0 = base + (8 * g + (i >> 2)) * LD + 4 * (i & 3);
    const dw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    const dw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * LD));
    const dw_s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(dw_bf16x8, r);
}

static size_t dwl_lds_bytes(int K, int C) {
    const int LDG = ((C + 31) & ~31) + 8;
    return (size_t)K * K * C * 4 + (size_t)(DWL_TH + 2) * (DWL_TW + 2) * C * 2 + (size_t)DWL_NPX * LDG * 2 +
           (size_t)DWL_NPX * DWL_LDX * 2 + (size_t)32 * LDG * 2 + (size_t)4 * 16 * DWL_LDC * 2;
}

template <int K, int C>
__global__ __launch_bounds__(256, 2) void dwconv_bwd_lin_s2_kernel(DwParams p, DwLinParams q) {
    typedef bf16_t T;
    constexpr int PAD = (K - 1) / 2;
    constexpr int TH = DWL_TH, TW = DWL_TW, IHT = TH + 2, IWT = TW + 2, NPX = DWL_NPX, PW2 = 2 * TW;
    constexpr int CP = (C + 31) & ~31, LDG = CP + 8, LDX = DWL_LDX;
    constexpr int CT = C / 16, MTW = (CT + 1) / 2;     // 16-channel row tiles of T1; per wave
    constexpr int nvec = C / 4, npt = 256 / nvec;
    static_assert(C % 16 == 0 && C <= 256, "channel count");
    constexpr int NV = (IHT * IWT + npt - 1) / npt;   // staged dY pixels per thread
    static_assert(256 * 8 * 4 <= NPX * LDG * 2, "the end-of-kernel reduction aliases the g tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wsm = reinterpret_cast<float*>(smem);       // [K*K][C]
    T* tile = reinterpret_cast<T*>(wsm + K * K * C);   // [IHT*IWT][C]
    T* Gs = tile + IHT * IWT * C;                      // [NPX][LDG]   g of the tile; columns C.. stay zero
    T* Xs = Gs + NPX * LDG;                            // [NPX][LDX]   x of the tile; columns Cin.. stay zero
    T* Wc = Xs + NPX * LDX;                            // [32][LDG]    (ca o Wexp)^T; rows Cin.. and columns C.. zero
    T* Cw = Wc + 32 * LDG;                             // [4 waves][16][LDC]
    float* red = reinterpret_cast<float*>(Gs);

    const T* __restrict__ dy = reinterpret_cast<const T*>(p.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = blockIdx.x;
    const int Cin = q.Cin, XC = Cin >> 3;              // 16-byte chunks per x row
    const int cv = tid % nvec, pt = tid / nvec;
    const int c = cv * 4;
    const bool cvalid = pt < npt;
    for (int i = tid; i < K * K * C; i += 256) {
        const int t = i / C, cc = i - t * C;
        wsm[i] = p.w[(size_t)cc * (K * K) + t];
    }
    {
        uint4* z = reinterpret_cast<uint4*>(Gs);
        constexpr int n16 = (NPX * LDG + NPX * LDX + 32 * LDG + 4 * 16 * DWL_LDC) / 8;
        for (int i = tid; i < n16; i += 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    for (int i = tid; i < C * Cin; i += 256) {
        const int cc = i / Cin, j = i - cc * Cin;
        Wc[j * LDG + cc] = from_f32<T>(to_f32<T>(q.wexp[i]) * (q.gamma[cc] * p.erstd[cc]));
    }
    const f22 zero22 = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
    f22 es = zero22, eh = zero22, em = zero22, er = zero22;
    if (cvalid) {
        es = load22<float>(p.escale + c); eh = load22<float>(p.eshift + c);
        em = load22<float>(p.emean + c); er = load22<float>(p.erstd + c);
    }
    f22 st_s = zero22, st_q = zero22;
    dw_f32x4 acc_t[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) acc_t[i] = dw_f32x4{0.f, 0.f, 0.f, 0.f};

    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    constexpr int patches_w = TW / 2, npatch = TH * patches_w;

    TileRegs<T, NV> regs;
    uint4 xr[2];
    auto fetch = [&](int work) {
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int r0 = trh * TH, q0 = (tr - trh * p.tiles_w) * TW;
        dw_tile_fetch<T, NV>(regs, dy + (size_t)b * p.OH * p.OW * C, IHT, IWT, r0 - 1, q0 - 1, p.OH, p.OW, C, c, true,
                                 pt, npt);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256;
            const int row = e / XC, ch = e - row * XC;
            const int lh = row / PW2, lw = row - lh * PW2;
            const int ih = r0 * 2 + lh, iw = q0 * 2 + lw;
            xr[u] = make_uint4(0u, 0u, 0u, 0u);
            if (row < NPX && ih < p.H && iw < p.W)
                xr[u] = *reinterpret_cast<const uint4*>(q.x + ((size_t)(b * p.H + ih) * p.W + iw) * Cin + ch * 8);
        }
    };
    // The forward tensor at this thread's output pixels of a tile (the epilogue's operand), loaded a phase ahead:
    // issued before the matrix phase of the previous tile, used in the vector phase of this one. With two workgroups
    // per CU there are too few waves to hide a global load issued where it is needed. Addresses are clamped into
    // the image so that the loads need no branches; values outside are ignored by the epilogue.
    constexpr int NIT = (npatch + npt - 1) / npt;      // patches per thread and tile
    uint2 yr[NIT][2][4];
    auto fetch_y = [&](int work) {
        if (!cvalid) return;
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int r0 = trh * TH, q0 = (tr - trh * p.tiles_w) * TW;
        const T* yimg = reinterpret_cast<const T*>(p.yfwd) + (size_t)b * p.H * p.W * C + c;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int patch = pt + u * npt;
            const int ph = patch / patches_w, pw = patch - ph * patches_w;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                int ih = (r0 + ph) * 2 + pr;
                ih = ih < p.H ? ih : p.H - 1;
#pragma unroll
                for (int x_ = 0; x_ < 4; ++x_) {
                    int iw = (q0 + pw * 2) * 2 + x_;
                    iw = iw < p.W ? iw : p.W - 1;
                    yr[u][pr][x_] = *reinterpret_cast<const uint2*>(yimg + (unsigned)((ih * p.W + iw) * C));
                }
            }
        }
    };
    const TileWalk walk = make_tile_walk(total, slot, gridDim.x);
    if (walk.first < walk.count) {
        fetch(walk.base + walk.first);
        fetch_y(walk.base + walk.first);
    }
    for (int wi = walk.first; wi < walk.count; wi += walk.stride) {
        const int work = walk.base + wi;
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int r0 = trh * TH, q0 = (tr - trh * p.tiles_w) * TW;
        __syncthreads();   // every wave is past the previous tile's matrix phase: all three tiles are free
        dw_tile_commit<T, NV>(regs, tile, IHT * IWT, cv, pt, npt, C, false, zero22, zero22);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * 256;
            const int row = e / XC, ch = e - row * XC;
            if (row < NPX) *reinterpret_cast<uint4*>(Xs + row * LDX + ch * 8) = xr[u];
        }
        __syncthreads();
        if (wi + walk.stride < walk.count) fetch(work + walk.stride);
        // ---- vector phase: g of the tile into LDS ----
        if (cvalid) {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int patch = pt + u * npt;
                if (patch >= npatch) break;
                const int ph = patch / patches_w, pw = patch - ph * patches_w;
                const int rl = ph, ql = pw * 2;
                f22 in[3][4];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        in[a][j] = load22<T>(tile + (unsigned)(((rl + a) * IWT + ql + j) * C + c));
                f22 acc[2][4];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int x_ = 0; x_ < 4; ++x_) acc[pr][x_] = zero22;
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const int pr = (kh + PAD) & 1;
                    const int dh = (pr + PAD - kh) / 2;
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const int pc = (kw + PAD) & 1;
                        const int dw = (pc + PAD - kw) / 2;
                        const f22 wv = load22<float>(wsm + (unsigned)((kh * K + kw) * C + c));
#pragma unroll
                        for (int cell = 0; cell < 2; ++cell)
                            fma22(acc[pr][cell * 2 + pc], in[dh + 1][cell + dw + 1], wv);
                    }
                }
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int lh = rl * 2 + pr;
                    const int ih = r0 * 2 + lh;
#pragma unroll
                    for (int x_ = 0; x_ < 4; ++x_) {
                        const int lw = ql * 2 + x_;
                        const int iw = q0 * 2 + lw;
                        const f22 yv = Raw4T<T>::unpack22(yr[u][pr][x_]);
                        const f22 v = acc[pr][x_];
                        f22 g;
                        g.lo = v.lo * silu_grad2(yv.lo * es.lo + eh.lo);
                        g.hi = v.hi * silu_grad2(yv.hi * es.hi + eh.hi);
                        if (!(ih < p.H && iw < p.W)) g = zero22;
                        st_s.lo += g.lo; st_s.hi += g.hi;
                        st_q.lo += g.lo * ((yv.lo - em.lo) * er.lo);
                        st_q.hi += g.hi * ((yv.hi - em.hi) * er.hi);
                        store22<T>(Gs + (unsigned)((lh * PW2 + lw) * LDG + c), g);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // one patch at a time: interleaving two doubles the live registers
            }
        }
        if (wi + walk.stride < walk.count) fetch_y(work + walk.stride);
        __syncthreads();
        // ---- matrix phase ----
        const int nt = wave & 1, mh = wave >> 1;
        {
            // dxp rows of the tile = Gs [96 x CP] . Wc^T [CP x 32]; this wave: column tile nt, row tiles mh, mh+2, mh+4
            dw_f32x4 acc[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
            const T* bbase = Wc + (nt * 16 + (lane & 15)) * LDG + 8 * (lane >> 4);
            const T* abase = Gs + (mh * 16 + (lane & 15)) * LDG + 8 * (lane >> 4);
#pragma unroll
            for (int kk = 0; kk < CP / 32; ++kk) {
                const dw_bf16x8 bf = *reinterpret_cast<const dw_bf16x8*>(bbase + kk * 32);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const dw_bf16x8 af = *reinterpret_cast<const dw_bf16x8*>(abase + i * 32 * LDG + kk * 32);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[i], 0, 0, 0);
                }
            }
            T* cw = Cw + wave * 16 * DWL_LDC;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cw[((lane >> 4) * 4 + r) * DWL_LDC + (lane & 15)] = from_f32<T>(acc[i][r]);
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) {
                    const int row = lane >> 1, half = lane & 1;
                    const int pp = (mh + 2 * i) * 16 + row;
                    const int lh = pp / PW2, lw = pp - lh * PW2;
                    const int ih = r0 * 2 + lh, iw = q0 * 2 + lw;
                    const int col = nt * 16 + half * 8;
                    if (ih < p.H && iw < p.W && col < Cin)
                        *reinterpret_cast<uint4*>(q.dxp + ((size_t)(b * p.H + ih) * p.W + iw) * Cin + col) =
                            *reinterpret_cast<const uint4*>(cw + row * DWL_LDC + half * 8);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        {
            // T1 += Gs^T [C x 96] . Xs [96 x 32]; this wave: column tile nt, row tiles mh, mh+2, ...
            const T* xb = Xs + nt * 16;
#pragma unroll
            for (int kk = 0; kk < NPX / 32; ++kk) {
                const dw_bf16x8 bf = dwl_load_t(xb + kk * 32 * LDX, LDX, lane);
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    if (mh + 2 * i < CT) {
                        const dw_bf16x8 af = dwl_load_t(Gs + kk * 32 * LDG + (mh + 2 * i) * 16, LDG, lane);
                        acc_t[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc_t[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    {
        const int nt = wave & 1, mh = wave >> 1;
        const int j = nt * 16 + (lane & 15);
        float* dst = q.slab + (size_t)slot * C * Cin;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
            if (mh + 2 * i < CT && j < Cin) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((mh + 2 * i) * 16 + (lane >> 4) * 4 + r) * Cin + j] = acc_t[i][r];
            }
    }
    dw_write_stats(p, red, tid, nvec, npt, 0, C, st_s, st_q, slot);
}

// weight gradient: dW[c][kh][kw] = sum_{b,oh,ow} dY[b,oh,ow,c] * pro(x)[b, oh*S+kh-PAD, ow*S+kw-PAD, c]
// A thread owns KR tap rows of one 4-channel vector (KR*K accumulators). KR = 1 keeps the register footprint small for
// the 5x5 kernels (the K tap rows of a channel vector are K different threads reading the same staged tiles); KR = K
// for 3x3: the four gradient vectors of a patch are read and unpacked once for all nine taps instead of once per tap
// row, and the patch walk's address arithmetic is paid once instead of three times (per 72 packed FMAs: 22 LDS reads
// instead of 30, 88 unpack instructions instead of 120).
// SW (stride 1 with a prologue): the operands change roles. dW[t] = sum_q dY(q) * a(q + t - PAD) is also
// sum_p a(p) * dY(p - t + PAD): the activated tensor is staged as the CENTRE tile and the gradient as the window with
// the halo, and the result lands on the mirrored tap. The BatchNorm+SiLU prologue, which is what the staging costs,
// then runs on TH*TW pixels per tile instead of (TH+K-1)*(TW+K-1) (8x16 tile, 3x3: 128 instead of 180).
template <typename T, int K, int S, int KR, bool SW>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(DwParams p, const void* dyv, float* slab) {
    static_assert(KR == 1 || KR == K, "a thread owns one tap row or all of them");
    static_assert(!SW || S == 1, "the operands can change roles at stride 1 only");
    constexpr int KG = K / KR;   // thread groups along the tap rows
    constexpr int PAD = (K - 1) / 2;
    constexpr int NIN = (DW_PW - 1) * S + K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int CB = p.CB, TH = p.TH, TW = p.TW;
    const int IHT = (TH - 1) * S + K, IWT = (TW - 1) * S + K;
    float* red = reinterpret_cast<float*>(smem);          // [256][4]
    T* tile = reinterpret_cast<T*>(red + 256 * 4);        // [IHT][IWT][CB]
    T* dtile = tile + (size_t)IHT * IWT * CB;             // [TH][TW][CB]

    // x: the operand staged with the halo, dy: the centre tile (the roles of the caller's tensors, see SW)
    const T* __restrict__ x = reinterpret_cast<const T*>(SW ? dyv : p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(SW ? p.x : dyv);
    const int tid = threadIdx.x;
    int slot, cblk;
    dw_decode_block(p, slot, cblk);
    const int c0 = cblk * CB;
    const int cbv = (p.C - c0) < CB ? (p.C - c0) : CB;
    const int nvec = CB / 4;
    // staging map: (pixel-thread, channel vector)
    const int s_npt = 256 / nvec, s_cv = tid % nvec, s_pt = tid / nvec;
    const bool s_chan_ok = s_cv * 4 < cbv;
    const bool has_pro = p.pscale != nullptr;
    const f22 zero = {v2f{0.f, 0.f}, v2f{0.f, 0.f}}, one = {v2f{1.f, 1.f}, v2f{1.f, 1.f}};
    f22 psc = one, psh = zero;
    if (has_pro && s_chan_ok) { psc = load22<float>(p.pscale + c0 + s_cv * 4); psh = load22<float>(p.pshift + c0 + s_cv * 4); }
    // compute map: (pixel-thread, tap row, channel vector)
    const int nkc = KG * nvec, npt = 256 / nkc;
    const int cv = tid % nvec, kh0 = ((tid / nvec) % KG) * KR, pt = tid / nkc;
    const bool cvalid = (cv * 4 < cbv) && (pt < npt);

    f22 acc[KR][K];
#pragma unroll
    for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int t = 0; t < K; ++t) acc[r][t] = f22{v2f{0.f, 0.f}, v2f{0.f, 0.f}};

    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    const int patches_w = TW / DW_PW;
    const int npatch = TH * patches_w;
    const int dph = npt / patches_w, dpw = npt - dph * patches_w;

    TileRegs<T, DW_NVX> xregs;
    TileRegs<T, DW_NVD> dregs;
    auto fetch = [&](int work) {
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int oh0 = trh * TH, ow0 = (tr - trh * p.tiles_w) * TW;
        dw_tile_fetch<T, DW_NVX>(xregs, x + (size_t)b * p.H * p.W * p.C, IHT, IWT, oh0 * S - PAD, ow0 * S - PAD, p.H,
                                 p.W, p.C, c0 + s_cv * 4, s_chan_ok, s_pt, s_npt);
        dw_tile_fetch<T, DW_NVD>(dregs, dy + (size_t)b * p.OH * p.OW * p.C, TH, TW, oh0, ow0, p.OH, p.OW, p.C,
                                 c0 + s_cv * 4, s_chan_ok, s_pt, s_npt);
    };
    const TileWalk walk = make_tile_walk(total, slot, p.gx);
    if (walk.first < walk.count) fetch(walk.base + walk.first);
    for (int wi = walk.first; wi < walk.count; wi += walk.stride) {
        const int work = walk.base + wi;
        __syncthreads();
        dw_tile_commit<T, DW_NVX>(xregs, tile, IHT * IWT, s_cv, s_pt, s_npt, CB, SW ? false : has_pro, psc, psh);
        dw_tile_commit<T, DW_NVD>(dregs, dtile, TH * TW, s_cv, s_pt, s_npt, CB, SW ? has_pro : false, psc, psh);
        __syncthreads();
        if (wi + walk.stride < walk.count) fetch(work + walk.stride);  // in flight while this tile is reduced
        if (cvalid) {
            int ph = pt / patches_w, pw = pt - ph * patches_w;
            for (int patch = pt; patch < npatch; patch += npt) {
                const int owl = pw * DW_PW;
                const T* grow = dtile + (unsigned)((ph * TW + owl) * CB + cv * 4);
                const T* row = tile + (unsigned)(((ph * S + kh0) * IWT + owl * S) * CB + cv * 4);
                f22 g[DW_PW];
#pragma unroll
                for (int q_ = 0; q_ < DW_PW; ++q_) g[q_] = load22<T>(grow + q_ * CB);
#pragma unroll
                for (int r = 0; r < KR; ++r, row += IWT * CB) {
                    f22 in[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) in[j] = load22<T>(row + j * CB);
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
#pragma unroll
                        for (int q_ = 0; q_ < DW_PW; ++q_) fma22(acc[r][kw], in[q_ * S + kw], g[q_]);
                }
                ph += dph;
                pw += dpw;
                if (pw >= patches_w) { pw -= patches_w; ++ph; }
            }
        }
    }
    // reduce over the pixel-threads of each (tap-row group, channel vector), one tap at a time
#pragma unroll
    for (int r = 0; r < KR; ++r)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
            __syncthreads();
            red[tid * 4 + 0] = acc[r][kw].lo.x; red[tid * 4 + 1] = acc[r][kw].lo.y;
            red[tid * 4 + 2] = acc[r][kw].hi.x; red[tid * 4 + 3] = acc[r][kw].hi.y;
            __syncthreads();
            for (int t = tid; t < KG * cbv; t += 256) {
                const int kg = t / cbv, ch = t - kg * cbv;
                const int v_ = ch >> 2, e = ch & 3;
                float a = 0.f;
                for (int q_ = 0; q_ < npt; ++q_) a += red[((q_ * KG + kg) * nvec + v_) * 4 + e];
                // slab row of this workgroup: [gridDim.x][C][K*K]; one plain store per output, no contention
                const int tap = (kg * KR + r) * K + kw;
                slab[((size_t)slot * p.C + c0 + ch) * (K * K) + (SW ? K * K - 1 - tap : tap)] = a;
            }
        }
}

// bf16 weight gradient on the matrix cores. With channels on both matrix axes and output pixels on the reduction
// axis,  D_t[c][c'] = sum_pix dY[pix][c] * X[pix + t][c'],  and the gradient of tap t is the diagonal of D_t. Both
// operands need 4 consecutive PIXELS of one channel per lane while the staged tiles are pixel-major ([pixel][channel],
// as they arrive from NHWC memory): gfx950's transposing LDS read delivers exactly that (a block of 4 pixel rows x
// 16 channels per 16-lane group, each lane receiving one channel's four pixels). Per 16 pixels x 16 channels: one read
// of dY, and per tap one read of X at the tap's offset plus one v_mfma_f32_16x16x16 into that tap's accumulator.
// A wave owns 16 channels of the 64-channel block for the whole kernel, so the K*K accumulators never leave its
// registers and no cross-wave reduction is needed; pixels beyond the tile read zeros of dY. The LDS pixel stride of
// 160 bytes keeps the eight pixel rows of a 32-lane half on disjoint banks.
constexpr int DWG_CB = 64;
constexpr int DWG_CBP = 80;   // LDS pixel stride in elements
constexpr int DWG_NVX = 16;   // staged input pixels per thread
constexpr int DWG_NVD = 9;    // staged output-gradient pixels per thread

template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwdw_mfma_kernel(DwParams p, const void* dyv, float* slab) {
    typedef bf16_t T;
    typedef __attribute__((address_space(3))) dw_s16x4 lds_s16x4;
    constexpr int PAD = (K - 1) / 2;
    constexpr int nvec = DWG_CB / 4, npt = 256 / nvec;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int TH = p.TH, TW = p.TW;
    const int IHT = (TH - 1) * S + K, IWT = (TW - 1) * S + K;
    const int npos = TH * TW, npos16 = (npos + 15) & ~15;
    int* ptab = reinterpret_cast<int*>(smem);                      // [npos16] byte offset of the pixel's window origin
    T* tile = reinterpret_cast<T*>(ptab + npos16);                 // [IHT*IWT][DWG_CBP]  pro(x)
    T* dtile = tile + (size_t)IHT * IWT * DWG_CBP;                 // [npos16][DWG_CBP]   dY, zeros past the tile

    const T* __restrict__ x = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(dyv);
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
    int slot, cblk;
    dw_decode_block(p, slot, cblk);
    const int c0 = cblk * DWG_CB;
    const int cbv = (p.C - c0) < DWG_CB ? (p.C - c0) : DWG_CB;
    const int cv = tid % nvec, pt = tid / nvec;
    const bool st_ok = cv * 4 < cbv;
    const bool has_pro = p.pscale != nullptr;
    const bool wave_on = grp * 16 < cbv;
    const f22 zero = {v2f{0.f, 0.f}, v2f{0.f, 0.f}}, one = {v2f{1.f, 1.f}, v2f{1.f, 1.f}};

    for (int q = tid; q < npos16; q += 256) {
        const int ohl = q / TW, owl = q - ohl * TW;
        ptab[q] = q < npos ? ((ohl * S) * IWT + owl * S) * DWG_CBP * 2 : 0;
    }
    for (int i = npos * (DWG_CBP / 2) + tid; i < npos16 * (DWG_CBP / 2); i += 256)
        reinterpret_cast<unsigned*>(dtile)[i] = 0u;   // never written again

    dw_f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = dw_f32x4{0.f, 0.f, 0.f, 0.f};

    const int tiles_per_img = p.tiles_h * p.tiles_w;
    const int total = p.B * tiles_per_img;
    TileRegs<T, DWG_NVX> xregs;
    TileRegs<T, DWG_NVD> dregs;
    auto fetch = [&](int work) {
        const int b = work / tiles_per_img;
        const int tr = work - b * tiles_per_img;
        const int trh = tr / p.tiles_w;
        const int oh0 = trh * TH, ow0 = (tr - trh * p.tiles_w) * TW;
        dw_tile_fetch<T, DWG_NVX>(xregs, x + (size_t)b * p.H * p.W * p.C, IHT, IWT, oh0 * S - PAD, ow0 * S - PAD, p.H,
                                  p.W, p.C, c0 + cv * 4, st_ok, pt, npt);
        dw_tile_fetch<T, DWG_NVD>(dregs, dy + (size_t)b * p.OH * p.OW * p.C, TH, TW, oh0, ow0, p.OH, p.OW, p.C,
                                  c0 + cv * 4, st_ok, pt, npt);
    };
    const TileWalk walk = make_tile_walk(total, slot, p.gx);
    if (walk.first < walk.count) fetch(walk.base + walk.first);
    f22 psc = one, psh = zero;
    if (has_pro && st_ok) { psc = load22<float>(p.pscale + c0 + cv * 4); psh = load22<float>(p.pshift + c0 + cv * 4); }
    // transposed-read role: lane 4q+r of 16-lane group g addresses pixel row 4g+q of a 16-pixel chunk, channels 4r..4r+3
    const int tg = lane >> 4, tq = (lane >> 2) & 3, tr4 = lane & 3;
    const int chan_off = (grp * 16 + tr4 * 4) * 2;
    for (int wi = walk.first; wi < walk.count; wi += walk.stride) {
        const int work = walk.base + wi;
        __syncthreads();
        dw_tile_commit<T, DWG_NVX>(xregs, tile, IHT * IWT, cv, pt, npt, DWG_CBP, has_pro, psc, psh);
        dw_tile_commit<T, DWG_NVD>(dregs, dtile, npos, cv, pt, npt, DWG_CBP, false, one, zero);
        __syncthreads();
        if (wi + walk.stride < walk.count) fetch(work + walk.stride);  // in flight while this tile is reduced
        if (wave_on) {
            for (int q0 = 0; q0 < npos; q0 += 16) {
                const int q = q0 + tg * 4 + tq;
                const char* dptr = reinterpret_cast<const char*>(dtile) + q * (DWG_CBP * 2) + chan_off;
                const char* xptr = reinterpret_cast<const char*>(tile) + ptab[q] + chan_off;
                const dw_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)dptr);
#pragma unroll
                for (int kh = 0; kh < K; ++kh) {
                    const char* rp = xptr + kh * IWT * (DWG_CBP * 2);
                    dw_s16x4 bv[K];
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
                        bv[kw] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(rp + kw * (DWG_CBP * 2)));
#pragma unroll
                    for (int kw = 0; kw < K; ++kw)
                        acc[kh * K + kw] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bv[kw], acc[kh * K + kw], 0, 0, 0);
                }
            }
        }
    }
    // the diagonal: lane l holds D[4*(l/16)+i][l%16]; rows equal columns where (l%16)/4 == l/16, at i = l%4
    if (wave_on) {
        const int n = lane & 15;
        const int ch = grp * 16 + n;
        if ((n >> 2) == tg && ch < cbv) {
            float* out = slab + ((size_t)slot * p.C + c0 + ch) * (K * K);
            const int i = n & 3;
#pragma unroll
            for (int t = 0; t < K * K; ++t) out[t] = i == 0 ? acc[t][0] : i == 1 ? acc[t][1] : i == 2 ? acc[t][2] : acc[t][3];
        }
    }
}

// out[i] = sum over rows of slab[row][i]  (fixed order => reproducible)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slab, int rows, long long n, float* out) {
    // 32 columns x 8 row groups per workgroup: the slab has few columns (C*K*K), so the rows carry the parallelism
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + col;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
        int r = rg;
        for (; r + 24 < rows; r += 32) {
            a0 += slab[(size_t)r * n + i];
            a1 += slab[(size_t)(r + 8) * n + i];
            a2 += slab[(size_t)(r + 16) * n + i];
            a3 += slab[(size_t)(r + 24) * n + i];
        }
        for (; r < rows; r += 8) a0 += slab[(size_t)r * n + i];
    }
    red[rg][col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rg == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += red[g][col];
        out[i] = t;
    }
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

// Choose the output tile (TH multiple of ph, TW multiple of 4): few wasted output slots, little halo, patches
// dividing evenly over the 256/nvec pixel-threads, at most max_in (max_out) staged input (output-gradient) pixels
// per thread, within an LDS budget.
static void pick_tile(int OH, int OW, int K, int S, int CB, int elem, int ph, int max_in, int max_out, int lds_budget,
                      int* th, int* tw) {
    static const int ths[] = {2, 4, 6, 8, 10, 12, 16, 20};
    static const int tws[] = {4, 8, 12, 16, 20, 24, 32, 40};
    const int npt = 256 / (CB / 4);
    double best = -1.0;
    *th = ph; *tw = 4;
    for (int a : ths)
        for (int b : tws) {
            if (a % ph) continue;
            const int iht = (a - 1) * S + K, iwt = (b - 1) * S + K;
            if (cdiv(iht * iwt, npt) > max_in) continue;
            if (max_out && cdiv(a * b, npt) > max_out) continue;
            const long long bytes = (long long)iht * iwt * CB * elem + (max_out ? (long long)a * b * CB * elem : 0);
            if (bytes > lds_budget) continue;
            const int nth = cdiv(OH, a), ntw = cdiv(OW, b);
            const double useful = (double)OH * OW;
            const double staged = (double)nth * ntw * iht * iwt;
            const double slots = (double)nth * ntw * a * b;
            const int patches = (a / ph) * (b / 4);
            const double balance = (double)patches / (double)(cdiv(patches, npt) * npt);
            const double score = (useful / staged) * (useful / slots) * balance;
            if (score > best) { best = score; *th = a; *tw = b; }
        }
}

template <typename T, int K, int S, int PH>
static void launch_tile(const DwParams& p, int grid_x, hipStream_t s) {
    const int IHT = (p.TH - 1) * S + K, IWT = (p.TW - 1) * S + K;
    const size_t lds = (size_t)K * K * p.CB * 4 + 256 * 8 * 4 + (size_t)IHT * IWT * p.CB * sizeof(T);
    // 1-D grid decoded by dw_decode_block: the channel blocks of one tile walker run on one XCD at about the same time
    DwParams q = p;
    q.gx = grid_x; q.ncb = cdiv(p.C, p.CB);
    dim3 grid((unsigned)(q.gx * q.ncb));
    const bool few = cdiv(IHT * IWT, 256 / (p.CB / 4)) <= 12;   // staged pixels per pixel-thread
    if (p.yfwd) {
        if (few) MIA_LAUNCH((dwconv_tile_kernel<T, K, S, PH, true, 12>), grid, dim3(256), lds, s, q);
        else MIA_LAUNCH((dwconv_tile_kernel<T, K, S, PH, true, DW_NV>), grid, dim3(256), lds, s, q);
    } else {
        if (few) MIA_LAUNCH((dwconv_tile_kernel<T, K, S, PH, false, 12>), grid, dim3(256), lds, s, q);
        else MIA_LAUNCH((dwconv_tile_kernel<T, K, S, PH, false, DW_NV>), grid, dim3(256), lds, s, q);
    }
}

template <typename T>
static int dispatch_tile(const DwParams& p, int K, int S, int grid_x, hipStream_t s) {
    if (K == 3 && S == 1) launch_tile<T, 3, 1, 2>(p, grid_x, s);
    else if (K == 3 && S == 2) launch_tile<T, 3, 2, 1>(p, grid_x, s);
    else if (K == 5 && S == 1) launch_tile<T, 5, 1, 2>(p, grid_x, s);
    else if (K == 5 && S == 2) launch_tile<T, 5, 2, 1>(p, grid_x, s);
    else return -1;
    return 0;
}
static int tile_ph(int S) { return S == 1 ? 2 : 1; }

// Output tile for the matrix-core kernel: any TH x TW whose halo window fits the per-thread staging registers and the
// LDS budget; scored by halo overhead, wasted output slots and how well TH*TW fills groups of 16 pixels.
static double pick_tile_mfma(int OH, int OW, int K, int S, int cb, int lds_budget, int* th, int* tw) {
    const int max_px = (256 / (cb / 4)) * DWM_NV;
    double best = -1.0;
    *th = 1; *tw = 1;
    for (int a = 1; a <= 32 && a <= OH + 3; ++a)
        for (int b = 1; b <= 32 && b <= OW + 3; ++b) {
            const int iht = (a - 1) * S + K, iwt = (b - 1) * S + K;
            if (iht * iwt > max_px) continue;
            if ((long long)iht * iwt * (cb + DWM_PAD) * 2 > lds_budget) continue;
            const int nth = cdiv(OH, a), ntw = cdiv(OW, b);
            const double useful = (double)OH * OW;
            const double staged = (double)nth * ntw * iht * iwt;
            const double slots = (double)nth * ntw * a * b;
            const double groups = (double)(a * b) / (double)(cdiv(a * b, 16) * 16);
            const double score = (useful / staged) * (useful / slots) * groups;
            if (score > best) { best = score; *th = a; *tw = b; }
        }
    return best;
}

// 64 channels per workgroup unless 32 stages markedly less halo (or wastes fewer channels of the last block).
static int pick_cb_mfma(int OH, int OW, int C, int K, int S, int* th, int* tw) {
    int th64, tw64, th32, tw32;
    const double s64 = pick_tile_mfma(OH, OW, K, S, 64, 38 * 1024, &th64, &tw64) * C / (cdiv(C, 64) * 64.0);
    const double s32 = pick_tile_mfma(OH, OW, K, S, 32, 38 * 1024, &th32, &tw32) * C / (cdiv(C, 32) * 32.0);
    if (s32 > 1.15 * s64) { *th = th32; *tw = tw32; return 32; }
    *th = th64; *tw = tw64;
    return 64;
}

// Workgroups per channel block: about as many as the chip keeps resident (3 per CU at the 5x5 kernel's register
// count, 4 for 3x3), each walking an equal number of tiles, so that the per-workgroup set-up is paid once and the next
// tile's loads overlap the current tile's arithmetic.
static int dw_mfma_grid(int total, int chan_blocks, int K) {
    const int resident = 256 * (K == 5 ? 3 : 4);
    int target = resident / chan_blocks;
    if (target < 8) target = 8;
    if (K == 3) { const int g = total < 8192 ? total : 8192; return g >= 16 ? (g & ~7) : g; }
    const int per_wg = cdiv(total, target);
    int gx = cdiv(total, per_wg);
    if (gx >= 16) gx &= ~7;   // a multiple of eight lets dw_decode_block keep a walker's channel blocks on one XCD
    return gx;
}

// bf16 only. Measured at batch 128 against the per-thread-patch kernel: ahead on maps up to 38x38 (1.2-1.7x on the
// 5x5 layers), behind on the 75x75 and 150x150 maps (streaming-bound there, and the 16-pixel staging tiles of this
// kernel re-read more halo), on layers narrower than 64 channels and on small stride-2 maps that tile badly.
static bool dw_use_mfma(int dtype, int C, int S, int OH, int OW) {
    // documented switch (include/mia_hip.h): keeps the fp32-tap per-thread kernels, read once per process
    static const bool off = getenv("MIA_DW_NO_MFMA") != nullptr;
    return !off && dtype == MIA_BF16 && C % 4 == 0 && C >= 64 && OH * OW <= 1600 && !(S == 2 && OH * OW < 1024);
}

// measured: ahead of the per-thread kernel on the 5x5 layers (15-20 %), level or behind on 3x3
static bool dw_use_mfma_w(int dtype, int C, int K, int S, int OH, int OW) {
    static const bool off = getenv("MIA_DWW_NO_MFMA") != nullptr;   // documented switch, as above
    return !off && dtype == MIA_BF16 && C % 4 == 0 && C >= 64 && OH * OW <= 1600 && K == 5 && !(S == 2 && OH * OW < 1024);
}

// p.x/p.y/p.H/p.W/p.OH/p.OW/flip/prologue/epilogue/stats set by the caller; fills the tile fields and launches.
static int launch_mfma(DwParams& p, int K, int S, float* stats, int part_rows, hipStream_t s, int* gx_out) {
    p.CB = pick_cb_mfma(p.OH, p.OW, p.C, K, S, &p.TH, &p.TW);
    p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
    const int total = p.B * p.tiles_h * p.tiles_w;
    int gx = dw_mfma_grid(total, cdiv(p.C, p.CB), K);
    if (stats && gx > part_rows) gx = part_rows;
    *gx_out = gx;
    const int IHT = (p.TH - 1) * S + K, IWT = (p.TW - 1) * S + K;
    size_t tile_bytes = (size_t)IHT * IWT * (p.CB + DWM_PAD) * 2;
    if (tile_bytes < 256 * 8 * 4) tile_bytes = 256 * 8 * 4;   // the statistics reduction reuses the tile
    const size_t lds = (size_t)((p.TH * p.TW + 15) & ~15) * 16 + tile_bytes;
    p.gx = gx; p.ncb = cdiv(p.C, p.CB);
    dim3 grid((unsigned)(p.gx * p.ncb));
#define MIA_DWM(K_, S_) \
    do { \
        if (p.yfwd) { \
            if (p.CB == 64) MIA_LAUNCH((dwconv_mfma_kernel<K_, S_, 64, true>), grid, dim3(256), lds, s, p); \
            else MIA_LAUNCH((dwconv_mfma_kernel<K_, S_, 32, true>), grid, dim3(256), lds, s, p); \
        } else { \
            if (p.CB == 64) MIA_LAUNCH((dwconv_mfma_kernel<K_, S_, 64, false>), grid, dim3(256), lds, s, p); \
            else MIA_LAUNCH((dwconv_mfma_kernel<K_, S_, 32, false>), grid, dim3(256), lds, s, p); \
        } \
    } while (0)
    if (K == 3 && S == 1) MIA_DWM(3, 1);
    else if (K == 3 && S == 2) MIA_DWM(3, 2);
    else if (K == 5 && S == 1) MIA_DWM(5, 1);
    else if (K == 5 && S == 2) MIA_DWM(5, 2);
    else return -1;
#undef MIA_DWM
    return 0;
}

// grid size launch_mfma will use (for clearing the statistics rows no workgroup owns)
static int mfma_grid_for(const DwParams& p, int K, int S) {
    int th, tw;
    const int cb = pick_cb_mfma(p.OH, p.OW, p.C, K, S, &th, &tw);
    return dw_mfma_grid(p.B * cdiv(p.OH, th) * cdiv(p.OW, tw), cdiv(p.C, cb), K);
}

static int dw_common_checks(const char* name, int dtype, int B, int H, int W, int C, int K, int S) {
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "%s: bad dtype %d", name, dtype);
    MIA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "%s: bad shape B=%d H=%d W=%d C=%d (C%%4==0)", name,
                  B, H, W, C);
    MIA_CHECK_ARG((K == 3 || K == 5) && (S == 1 || S == 2), "%s: unsupported kernel %d stride %d", name, K, S);
    MIA_CHECK_ARG((long long)B * H * W * C < (1LL << 31), "%s: tensor too large for 32-bit element offsets", name);
    return 0;
}

// Number of statistics rows a launch fills (= workgroups per channel block) when the caller allows `cap` rows: a caller
// that passes exactly this many rows as part_rows gets every row written and no zero-fill launch for unowned rows.
// kind 0: mia_dwconv_fwd, 1: mia_dwconv_bwd_data (H, W are the INPUT map of the forward convolution in both cases).
extern "C" int mia_dwconv_part_rows(int dtype, int kind, int B, int H, int W, int C, int K, int S, int cap) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (K != 3 && K != 5) || (S != 1 && S != 2) || cap <= 0) return -1;
    const int pad = (K - 1) / 2;
    const int OH = (H + 2 * pad - K) / S + 1, OW = (W + 2 * pad - K) / S + 1;
    const int elem = dtype == MIA_BF16 ? 2 : 4;
    const int lds = dtype == MIA_BF16 ? 32768 : 65536;
    DwParams p = {};
    p.B = B; p.C = C;
    int gx;
    int mh, mw, ks = K, ss = S, ph = tile_ph(S);   // map the tiles are counted on, kernel/stride of the tiling
    if (kind == 0) { mh = OH; mw = OW; }
    else if (S == 1) { mh = H; mw = W; }
    else { mh = OH; mw = OW; ks = 3; ss = 1; ph = 1; }   // stride-2 data gradient: tiles of dY cells
    if (!(kind == 1 && S == 2) && dw_use_mfma(dtype, C, kind == 0 ? S : 1, mh, mw)) {
        p.OH = mh; p.OW = mw;
        gx = mfma_grid_for(p, K, kind == 0 ? S : 1);
    } else {
        int th, tw;
        pick_tile(mh, mw, ks, ss, pick_cb(C), elem, ph, DW_NV, 0, lds, &th, &tw);
        const int total = B * cdiv(mh, th) * cdiv(mw, tw);
        gx = total < 8192 ? total : 8192;
        if (gx >= 16) gx &= ~7;
    }
    return gx < cap ? gx : cap;
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
    MIA_CHECK_ARG(!stats || (part_rows > 0 && part_rows <= 65535), "mia_dwconv_fwd: stats needs 0 < part_rows <= 65535");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dw_use_mfma(dtype, C, S, p.OH, p.OW)) {
        int gx = 0;
        // rows no workgroup owns must still read as zero: the row count is known only after the tile is picked, so
        // clear all of them first when the grid may be smaller
        const int gxm = mfma_grid_for(p, K, S);
        if (stats && gxm < part_rows) {
            hipError_t e = mia_zero_words(stats + (size_t)gxm * 2 * C, (long long)(part_rows - gxm) * 2 * C, s);
            MIA_CHECK_ARG(e == hipSuccess, "mia_dwconv_fwd: zero fill failed: %s", hipGetErrorString(e));
        }
        MIA_CHECK_ARG(launch_mfma(p, K, S, stats, part_rows, s, &gx) == 0, "mia_dwconv_fwd: no kernel for K=%d S=%d", K, S);
        MIA_CHECK_LAUNCH("mia_dwconv_fwd");
        return 0;
    }
    p.CB = pick_cb(C);
    const int elem = dtype == MIA_BF16 ? 2 : 4;
    pick_tile(p.OH, p.OW, K, S, p.CB, elem, tile_ph(S), DW_NV, 0, dtype == MIA_BF16 ? 32768 : 65536, &p.TH, &p.TW);
    p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
    const int total = B * p.tiles_h * p.tiles_w;
    int gx = total < 8192 ? total : 8192;
    if (gx >= 16) gx &= ~7;   // a multiple of eight: sibling channel blocks share an XCD (dw_decode_block)
    if (stats) {
        if (gx > part_rows) gx = part_rows;
        if (gx < part_rows) {  // rows no workgroup owns must still read as zero
            hipError_t e = mia_zero_words(stats + (size_t)gx * 2 * C, (long long)(part_rows - gx) * 2 * C, s);
            MIA_CHECK_ARG(e == hipSuccess, "mia_dwconv_fwd: zero fill failed: %s", hipGetErrorString(e));
        }
    }
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
        if (dw_use_mfma(dtype, C, 1, p.OH, p.OW)) {
            const int gxm = mfma_grid_for(p, K, 1);
            if (stats && gxm < part_rows) {
                hipError_t e = mia_zero_words(stats + (size_t)gxm * 2 * C, (long long)(part_rows - gxm) * 2 * C, s);
                MIA_CHECK_ARG(e == hipSuccess, "mia_dwconv_bwd_data: zero fill failed: %s", hipGetErrorString(e));
            }
            int gx = 0;
            MIA_CHECK_ARG(launch_mfma(p, K, 1, stats, part_rows, s, &gx) == 0, "mia_dwconv_bwd_data: no kernel for K=%d", K);
            MIA_CHECK_LAUNCH("mia_dwconv_bwd_data");
            return 0;
        }
        pick_tile(p.OH, p.OW, K, 1, p.CB, elem, tile_ph(1), DW_NV, 0, dtype == MIA_BF16 ? 32768 : 65536, &p.TH, &p.TW);
        p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
        const int total = B * p.tiles_h * p.tiles_w;
        int gx = total < 8192 ? total : 8192;
        if (gx >= 16) gx &= ~7;   // a multiple of eight: sibling channel blocks share an XCD (dw_decode_block)
        if (stats) {
            if (gx > part_rows) gx = part_rows;
            if (gx < part_rows) {
                hipError_t e = mia_zero_words(stats + (size_t)gx * 2 * C, (long long)(part_rows - gx) * 2 * C, s);
                MIA_CHECK_ARG(e == hipSuccess, "mia_dwconv_bwd_data: zero fill failed: %s", hipGetErrorString(e));
            }
        }
        int rc = dtype == MIA_BF16 ? dispatch_tile<bf16_t>(p, K, 1, gx, s) : dispatch_tile<float>(p, K, 1, gx, s);
        MIA_CHECK_ARG(rc == 0, "mia_dwconv_bwd_data: no kernel for K=%d", K);
    } else {
        p.x = dy; p.y = dx; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
        // tiles are counted in dY cells (each cell = 2x2 dA pixels); the staged window has one cell of halo a side
        pick_tile(OH, OW, 3, 1, p.CB, elem, 1, DW_NV, 0, dtype == MIA_BF16 ? 32768 : 65536, &p.TH, &p.TW);
        p.tiles_h = cdiv(OH, p.TH); p.tiles_w = cdiv(OW, p.TW);
        const int total = B * p.tiles_h * p.tiles_w;
        int gx = total < 8192 ? total : 8192;
        if (gx >= 16) gx &= ~7;   // a multiple of eight: sibling channel blocks share an XCD (dw_decode_block)
        if (stats) {
            if (gx > part_rows) gx = part_rows;
            if (gx < part_rows) {
                hipError_t e = mia_zero_words(stats + (size_t)gx * 2 * C, (long long)(part_rows - gx) * 2 * C, s);
                MIA_CHECK_ARG(e == hipSuccess, "mia_dwconv_bwd_data: zero fill failed: %s", hipGetErrorString(e));
            }
        }
        const size_t lds = (size_t)K * K * p.CB * 4 + 256 * 8 * 4 + (size_t)(p.TH + 2) * (p.TW + 2) * p.CB * elem;
        p.gx = gx; p.ncb = cdiv(C, p.CB);
        dim3 grid((unsigned)(p.gx * p.ncb));
        if (dtype == MIA_BF16) {
            if (K == 3) MIA_LAUNCH((dwconv_bwd_data_s2_kernel<bf16_t, 3>), grid, dim3(256), lds, s, p);
            else MIA_LAUNCH((dwconv_bwd_data_s2_kernel<bf16_t, 5>), grid, dim3(256), lds, s, p);
        } else {
            if (K == 3) MIA_LAUNCH((dwconv_bwd_data_s2_kernel<float, 3>), grid, dim3(256), lds, s, p);
            else MIA_LAUNCH((dwconv_bwd_data_s2_kernel<float, 5>), grid, dim3(256), lds, s, p);
        }
    }
    MIA_CHECK_LAUNCH("mia_dwconv_bwd_data");
    return 0;
}

// Shapes dwconv_bwd_lin_s2_kernel is built for (1 = yes).
extern "C" int mia_dwconv_bwd_data_lin_supported(int dtype, int C, int K, int S, int Cin) {
    // (A 5x5 / 192-channel build for block 2.0 was measured too: its 100 KB of LDS leave one workgroup per CU and the
    // fused kernel then takes 582 us against 433 us for the three separate ones.)
    return dtype == MIA_BF16 && S == 2 && K == 3 && C == 144 && Cin > 0 && Cin <= 32 && Cin % 8 == 0;
}

// Workgroups (= statistics rows written = slab rows) the fused launch uses when the caller allows `cap` of them.
extern "C" int mia_dwconv_bwd_data_lin_rows(int B, int H, int W, int K, int C, int cap) {
    if (B <= 0 || H <= 0 || W <= 0 || cap <= 0 || (K != 3 && K != 5)) return -1;
    const int pad = (K - 1) / 2;
    const int OH = (H + 2 * pad - K) / 2 + 1, OW = (W + 2 * pad - K) / 2 + 1;
    const int total = B * cdiv(OH, DWL_TH) * cdiv(OW, DWL_TW);
    const int per_cu = (int)(160 * 1024 / dwl_lds_bytes(K, C));
    int gx = 256 * (per_cu < 1 ? 1 : per_cu);
    if (gx > total) gx = total;
    if (gx > cap) gx = cap;
    if (gx >= 16) gx &= ~7;
    return gx;
}

// Stride-2 data gradient with the fused silu' epilogue of mia_dwconv_bwd_data, but instead of the gradient g itself
// its two linear images leave the kernel (see dwconv_bwd_lin_s2_kernel):
//   dxp [B*H*W, Cin] = g . ((gamma*erstd) o wexp)       t1 [C, Cin] fp32 = g^T x
// wexp [C, Cin] and x [B*H*W, Cin] in bf16. stats [part_rows][2][C] as in mia_dwconv_bwd_data. ws: fp32 scratch of
// at least rows*C*Cin floats, rows = mia_dwconv_bwd_data_lin_rows(..., part_rows).
extern "C" int mia_dwconv_bwd_data_lin(int dtype, const void* dy, int B, int H, int W, int C, const float* w, int K,
                                       int S, const void* yfwd, const float* escale, const float* eshift,
                                       const float* emean, const float* erstd, const float* gamma, const void* wexp,
                                       const void* x, int Cin, void* dxp, float* t1, float* stats, int part_rows,
                                       float* ws, long long ws_floats, void* stream) {
    if (dw_common_checks("mia_dwconv_bwd_data_lin", dtype, B, H, W, C, K, S)) return -1;
    MIA_CHECK_ARG(mia_dwconv_bwd_data_lin_supported(dtype, C, K, S, Cin),
                  "mia_dwconv_bwd_data_lin: no kernel for dtype=%d C=%d K=%d S=%d Cin=%d", dtype, C, K, S, Cin);
    MIA_CHECK_ARG(dy && w && yfwd && escale && eshift && emean && erstd && gamma && wexp && x && dxp && t1 && stats && ws,
                  "mia_dwconv_bwd_data_lin: null operand");
    MIA_CHECK_ARG(part_rows > 0 && part_rows <= 65535, "mia_dwconv_bwd_data_lin: bad part_rows");
    MIA_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dxp % 16) == 0, "mia_dwconv_bwd_data_lin: x and dxp must be 16-byte aligned");
    MIA_CHECK_ARG(((uintptr_t)dy % 8) == 0 && ((uintptr_t)yfwd % 8) == 0, "mia_dwconv_bwd_data_lin: dy and yfwd must be 8-byte aligned");
    const int pad = (K - 1) / 2;
    DwParams p = {};
    p.x = dy; p.w = w; p.B = B; p.H = H; p.W = W; p.C = C;
    p.OH = (H + 2 * pad - K) / 2 + 1; p.OW = (W + 2 * pad - K) / 2 + 1;
    p.yfwd = yfwd; p.escale = escale; p.eshift = eshift; p.emean = emean; p.erstd = erstd; p.stats = stats;
    p.CB = C; p.TH = DWL_TH; p.TW = DWL_TW; p.tiles_h = cdiv(p.OH, DWL_TH); p.tiles_w = cdiv(p.OW, DWL_TW);
    const int gx = mia_dwconv_bwd_data_lin_rows(B, H, W, K, C, part_rows);
    MIA_CHECK_ARG(gx > 0 && (long long)gx * C * Cin <= ws_floats, "mia_dwconv_bwd_data_lin: workspace of %lld floats is too small for %d rows",
                  ws_floats, gx);
    p.gx = gx; p.ncb = 1;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (gx < part_rows) {
        hipError_t e = mia_zero_words(stats + (size_t)gx * 2 * C, (long long)(part_rows - gx) * 2 * C, s);
        MIA_CHECK_ARG(e == hipSuccess, "mia_dwconv_bwd_data_lin: zero fill failed: %s", hipGetErrorString(e));
    }
    DwLinParams q = {};
    q.x = (const bf16_t*)x; q.wexp = (const bf16_t*)wexp; q.gamma = gamma; q.dxp = (bf16_t*)dxp; q.slab = ws; q.Cin = Cin;
    const size_t lds = dwl_lds_bytes(K, C);
    {
        auto kern = dwconv_bwd_lin_s2_kernel<3, 144>;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        MIA_LAUNCH(kern, dim3((unsigned)gx), dim3(256), lds, s, p, q);
    }
    MIA_CHECK_LAUNCH("mia_dwconv_bwd_data_lin");
    const long long n = (long long)C * Cin;
    MIA_LAUNCH(slab_reduce_kernel, dim3((unsigned)cdiv(n, 32)), dim3(256), 0, s, ws, gx, n, t1);
    MIA_CHECK_LAUNCH("mia_dwconv_bwd_data_lin");
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
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dw_use_mfma_w(dtype, C, K, S, p.OH, p.OW)) {
        p.CB = DWG_CB;
        // largest tile whose window and gradient pixels fit the staging registers; scored as in the forward kernel
        double best = -1.0;
        p.TH = 1; p.TW = 1;
        for (int a = 1; a <= 16 && a <= p.OH + 3; ++a)
            for (int b = 1; b <= 16 && b <= p.OW + 3; ++b) {
                const int iht = (a - 1) * S + K, iwt = (b - 1) * S + K;
                if (iht * iwt > 16 * DWG_NVX || a * b > 16 * DWG_NVD) continue;
                // dynamic LDS of a launch without an opt-in attribute is limited to 64 KiB
                if ((size_t)((a * b + 15) & ~15) * 4 + ((size_t)iht * iwt + ((a * b + 15) & ~15)) * DWG_CBP * 2 > 64 * 1024) continue;
                const int nth = cdiv(p.OH, a), ntw = cdiv(p.OW, b);
                const double useful = (double)p.OH * p.OW;
                const double score = (useful / ((double)nth * ntw * iht * iwt)) * (useful / ((double)nth * ntw * a * b)) *
                                     ((double)(a * b) / (cdiv(a * b, 16) * 16.0));
                if (score > best) { best = score; p.TH = a; p.TW = b; }
            }
        p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
        const int total = B * p.tiles_h * p.tiles_w;
        const int ncb = cdiv(C, DWG_CB);
        long long gx = 512 / ncb;                   // two workgroups per CU fit (registers, LDS)
        if (gx < 8) gx = 8;
        if (gx > total) gx = total;
        if (gx > ws_floats / ((long long)C * K * K)) gx = ws_floats / ((long long)C * K * K);
        const int IHT = (p.TH - 1) * S + K, IWT = (p.TW - 1) * S + K;
        const int npos16 = (p.TH * p.TW + 15) & ~15;
        const size_t lds = (size_t)npos16 * 4 + ((size_t)IHT * IWT + npos16) * DWG_CBP * 2;
        p.gx = (int)gx; p.ncb = ncb;
        dim3 grid((unsigned)(gx * ncb));
        if (K == 3 && S == 1) MIA_LAUNCH((dwconv_bwdw_mfma_kernel<3, 1>), grid, dim3(256), lds, s, p, dy, ws);
        else if (K == 3 && S == 2) MIA_LAUNCH((dwconv_bwdw_mfma_kernel<3, 2>), grid, dim3(256), lds, s, p, dy, ws);
        else if (K == 5 && S == 1) MIA_LAUNCH((dwconv_bwdw_mfma_kernel<5, 1>), grid, dim3(256), lds, s, p, dy, ws);
        else MIA_LAUNCH((dwconv_bwdw_mfma_kernel<5, 2>), grid, dim3(256), lds, s, p, dy, ws);
        const long long nout = (long long)C * K * K;
        MIA_LAUNCH(slab_reduce_kernel, dim3(cdiv(nout, 32)), dim3(256), 0, s, ws, (int)gx, nout, dw);
        MIA_CHECK_LAUNCH("mia_dwconv_bwd_weight");
        return 0;
    }
    pick_tile(p.OH, p.OW, K, S, p.CB, elem, 1, DW_NVX, DW_NVD, dtype == MIA_BF16 ? 40960 : 81920, &p.TH, &p.TW);
    p.tiles_h = cdiv(p.OH, p.TH); p.tiles_w = cdiv(p.OW, p.TW);
    const int total = B * p.tiles_h * p.tiles_w;
    const int ncb = cdiv(C, p.CB);
    long long gx = 1024 / ncb;                      // ~1024 workgroups in all
    if (gx < 8) gx = 8;
    if (gx > 256) gx = 256;
    if (gx > total) gx = total;
    if (gx > ws_floats / ((long long)C * K * K)) gx = ws_floats / ((long long)C * K * K);
    if (gx >= 16) gx &= ~7ll;   // a multiple of eight lets dw_decode_block keep a walker's channel blocks on one XCD
    const int IHT = (p.TH - 1) * S + K, IWT = (p.TW - 1) * S + K;
    const size_t lds = 256 * 4 * 4 + ((size_t)IHT * IWT + (size_t)p.TH * p.TW) * p.CB * elem;
    p.gx = (int)gx; p.ncb = ncb;
    dim3 grid((unsigned)(gx * ncb));
    // all tap rows per thread: measured ahead on the large 3x3 stride-1 maps only (75x75x192 268 -> 246 us; stride 2 and
    // the 19x19 / 10x10 maps 3-20 % behind, where the 154 registers cost a wave per SIMD)
    const bool rows3 = K == 3 && S == 1 && p.OH * p.OW > 1600;
    const bool sw = S == 1 && pscale != nullptr;
#define MIA_DWW_(T_, K_, S_, KR_, SW_) \
    MIA_LAUNCH((dwconv_bwd_weight_kernel<T_, K_, S_, KR_, SW_>), grid, dim3(256), lds, s, p, dy, ws)
#define MIA_DWW(T_, K_, S_, PH_)                                                              \
    do {                                                                                      \
        if (S_ == 1 && sw) {                                                                  \
            if (K_ == 3 && rows3) MIA_DWW_(T_, K_, 1, (K_ == 3 ? 3 : 1), true);               \
            else MIA_DWW_(T_, K_, 1, 1, true);                                                \
        } else {                                                                              \
            if (K_ == 3 && rows3) MIA_DWW_(T_, K_, S_, (K_ == 3 ? 3 : 1), false);             \
            else MIA_DWW_(T_, K_, S_, 1, false);                                              \
        }                                                                                     \
    } while (0)
    if (dtype == MIA_BF16) {
        if (K == 3 && S == 1) MIA_DWW(bf16_t, 3, 1, 2);
        else if (K == 3 && S == 2) MIA_DWW(bf16_t, 3, 2, 1);
        else if (K == 5 && S == 1) MIA_DWW(bf16_t, 5, 1, 1);
        else MIA_DWW(bf16_t, 5, 2, 1);
    } else {
        if (K == 3 && S == 1) MIA_DWW(float, 3, 1, 2);
        else if (K == 3 && S == 2) MIA_DWW(float, 3, 2, 1);
        else if (K == 5 && S == 1) MIA_DWW(float, 5, 1, 1);
        else MIA_DWW(float, 5, 2, 1);
    }
#undef MIA_DWW
#undef MIA_DWW_
    const long long nout = (long long)C * K * K;
    MIA_LAUNCH(slab_reduce_kernel, dim3(cdiv(nout, 32)), dim3(256), 0, s, ws, (int)gx, nout, dw);
    MIA_CHECK_LAUNCH("mia_dwconv_bwd_weight");
    return 0;
}

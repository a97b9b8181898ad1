// Bandwidth-bound preprocessing kernels (gfx950): 8-bit bilinear resize + normalise, and the
// zero-intensity slice filter (mask, ordered compaction, gather).
//
// The resize follows OpenCV's 8-bit INTER_LINEAR arithmetic exactly (11-bit fixed-point weights,
// int32 horizontal pass, ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2 vertical pass), so results are
// integers and can be held bit-exact to a CPU restatement; the normalise step is the two fp32
// operations (x - 255*mean) * (1/(255*std)). Floating-point contraction is disabled in this file:
// an fma in the coordinate computation would change which source pixel a destination maps to.
#include "common.h"
#include <mutex>

#pragma clang fp contract(off)

struct LinCoef {
    int s0, s1, w0, w1;
};

__device__ __forceinline__ LinCoef lin_coef(int d, int src, int dst) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    LinCoef c;
    c.s0 = s;
    c.s1 = (s + 1) < src ? (s + 1) : (src - 1);
    c.w1 = (int)rintf(f * 2048.0f);
    c.w0 = (int)rintf((1.0f - f) * 2048.0f);
    return c;
}

// src u8 [N,h,w,ch] (ch = 1 or 3) -> dst f32 [N,3,oh,ow] (OUT 0), bf16 [N,oh,ow,3] (OUT 1) or f32 [N,oh,ow,3] (OUT 2)
template <int OUT>
__global__ __launch_bounds__(256) void resize_normalize_kernel(const uint8_t* src, int N, int h, int w, int ch,
                                                               void* dstv, int oh, int ow, float m0, float m1, float m2,
                                                               float r0, float r1, float r2, int mode) {
    float* dst = reinterpret_cast<float*>(dstv);
    const long long total = (long long)N * oh * ow;
    const float mm[3] = {m0, m1, m2}, rr[3] = {r0, r1, r2};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh), n = (int)(i / ((long long)ow * oh));
        const LinCoef cx = lin_coef(ox, w, ow), cy = lin_coef(oy, h, oh);
        const uint8_t* img = src + (size_t)n * h * w * ch;
        const uint8_t* p00 = img + ((size_t)cy.s0 * w + cx.s0) * ch;
        const uint8_t* p01 = img + ((size_t)cy.s0 * w + cx.s1) * ch;
        const uint8_t* p10 = img + ((size_t)cy.s1 * w + cx.s0) * ch;
        const uint8_t* p11 = img + ((size_t)cy.s1 * w + cx.s1) * ch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = ch == 3 ? c : 0;
            const int S0 = (int)p00[cs] * cx.w0 + (int)p01[cs] * cx.w1;
            const int S1 = (int)p10[cs] * cx.w0 + (int)p11[cs] * cx.w1;
            int v = ((cy.w0 * (S0 >> 4)) >> 16) + ((cy.w1 * (S1 >> 4)) >> 16);
            v = (v + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            float o;
            if (mode == 0) {
                o = (float)v - mm[c];
                o = o * rr[c];
            } else {
                o = (float)v / 255.0f;
            }
            if constexpr (OUT == 0) dst[(((size_t)n * 3 + c) * oh + oy) * ow + ox] = o;
            else if constexpr (OUT == 1) reinterpret_cast<bf16_t*>(dstv)[(size_t)i * 3 + c] = from_f32<bf16_t>(o);
            else dst[(size_t)i * 3 + c] = o;
        }
    }
}

// The same arithmetic, tiled: one workgroup produces RZ_ROWS output rows of one image. The x and y coefficient
// tables are computed once per workgroup, the source rows the tile needs are staged in LDS with 16-byte loads, and
// every thread then produces four consecutive output pixels per channel (16-byte stores). Global memory sees only
// streaming vector accesses; the twelve byte-sized taps per pixel are LDS reads.
// OUT: 0 = float [N,3,oh,ow] (the tensor ToTensorV2 hands to the model); 1 = bf16 [N,oh,ow,3]; 2 = float [N,oh,ow,3]:
// the channels-last forms are what the trunk's stem reads directly (mia_stem_im2col_nhwc), in the trunk's compute
// type, so the normalised image is written once at 6 (or 12) bytes a pixel instead of 12 and never converted again.
#define RZ_ROWS 20
template <int OUT>
__global__ __launch_bounds__(256) void resize_normalize_tile_kernel(const uint8_t* src, int h, int w, int ch, void* dstv,
                                                                    int oh, int ow, float m0, float m1, float m2, float r0,
                                                                    float r1, float r2, int mode) {
    float* dst = reinterpret_cast<float*>(dstv);
    extern __shared__ __align__(16) uint8_t rz_lds[];
    int4* cxs = reinterpret_cast<int4*>(rz_lds);   // per output column: byte offset of tap 0, of tap 1, weights
    int4* cys = cxs + ow;                          // per output row of the tile: source rows, weights
    uint8_t* rows = reinterpret_cast<uint8_t*>(cys + RZ_ROWS);
    const int tid = threadIdx.x, n = blockIdx.y, oy0 = blockIdx.x * RZ_ROWS;
    const int nr = (oh - oy0) < RZ_ROWS ? (oh - oy0) : RZ_ROWS;
    for (int i = tid; i < ow; i += 256) {
        const LinCoef c = lin_coef(i, w, ow);
        cxs[i] = make_int4(c.s0 * ch, c.s1 * ch, c.w0, c.w1);
    }
    for (int i = tid; i < nr; i += 256) {
        const LinCoef c = lin_coef(oy0 + i, h, oh);
        cys[i] = make_int4(c.s0, c.s1, c.w0, c.w1);
    }
    __syncthreads();
    const int y_first = cys[0].x, y_last = cys[nr - 1].y;
    const size_t row_bytes = (size_t)w * ch;
    const size_t total = (size_t)(y_last - y_first + 1) * row_bytes;
    const uint8_t* sbase = src + ((size_t)n * h + y_first) * row_bytes;
    if ((reinterpret_cast<uintptr_t>(sbase) & 15) == 0) {
        const size_t vec_end = total & ~(size_t)15;
        for (size_t i = (size_t)tid * 16; i < vec_end; i += 256 * 16)
            *reinterpret_cast<uint4*>(rows + i) = *reinterpret_cast<const uint4*>(sbase + i);
        for (size_t i = vec_end + tid; i < total; i += 256) rows[i] = sbase[i];
    } else {
        for (size_t i = tid; i < total; i += 256) rows[i] = sbase[i];
    }
    __syncthreads();
    const float mm[3] = {m0, m1, m2}, rr[3] = {r0, r1, r2};
    const int quads = (ow + 3) >> 2;
    const bool vec_out = (ow & 3) == 0;
    for (int item = tid; item < nr * quads; item += 256) {
        const int r = item / quads, ox = (item % quads) * 4, oy = oy0 + r;
        const int4 cy = cys[r];
        const uint8_t* p0 = rows + (size_t)(cy.x - y_first) * row_bytes;
        const uint8_t* p1 = rows + (size_t)(cy.y - y_first) * row_bytes;
        float o[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int4 cx = cxs[(ox + j) < ow ? (ox + j) : (ow - 1)];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int cs = ch == 3 ? c : 0;
                const int S0 = (int)p0[cx.x + cs] * cx.z + (int)p0[cx.y + cs] * cx.w;
                const int S1 = (int)p1[cx.x + cs] * cx.z + (int)p1[cx.y + cs] * cx.w;
                int v = ((cy.z * (S0 >> 4)) >> 16) + ((cy.w * (S1 >> 4)) >> 16);
                v = (v + 2) >> 2;
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                float f;
                if (mode == 0) {
                    f = (float)v - mm[c];
                    f = f * rr[c];
                } else {
                    f = (float)v / 255.0f;
                }
                o[c][j] = f;
            }
        }
        if constexpr (OUT == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float* d = dst + (((size_t)n * 3 + c) * oh + oy) * ow + ox;
                if (vec_out) {
                    *reinterpret_cast<float4*>(d) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ox + j < ow) d[j] = o[c][j];
                }
            }
        } else if constexpr (OUT == 1) {
            // four pixels x three channels = 24 contiguous bytes, 8-byte aligned when ow % 4 == 0
            bf16_t* d = reinterpret_cast<bf16_t*>(dstv) + (((size_t)n * oh + oy) * ow + ox) * 3;
            if (vec_out) {
                uint2* d2 = reinterpret_cast<uint2*>(d);
                d2[0] = make_uint2(pack_bf16x2(v2f{o[0][0], o[1][0]}), pack_bf16x2(v2f{o[2][0], o[0][1]}));
                d2[1] = make_uint2(pack_bf16x2(v2f{o[1][1], o[2][1]}), pack_bf16x2(v2f{o[0][2], o[1][2]}));
                d2[2] = make_uint2(pack_bf16x2(v2f{o[2][2], o[0][3]}), pack_bf16x2(v2f{o[1][3], o[2][3]}));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ox + j < ow)
#pragma unroll
                        for (int c = 0; c < 3; ++c) d[j * 3 + c] = from_f32<bf16_t>(o[c][j]);
            }
        } else {
            float* d = dst + (((size_t)n * oh + oy) * ow + ox) * 3;
            if (vec_out) {
                float4* d4 = reinterpret_cast<float4*>(d);
                d4[0] = make_float4(o[0][0], o[1][0], o[2][0], o[0][1]);
                d4[1] = make_float4(o[1][1], o[2][1], o[0][2], o[1][2]);
                d4[2] = make_float4(o[2][2], o[0][3], o[1][3], o[2][3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ox + j < ow)
#pragma unroll
                        for (int c = 0; c < 3; ++c) d[j * 3 + c] = o[c][j];
            }
        }
    }
}

// mode 0: (x - 255*mean) * (1/(255*std));  mode 1: x / 255
// out_form 0: float [N,3,oh,ow]; 1: bf16 [N,oh,ow,3]; 2: float [N,oh,ow,3]
static int resize_normalize_impl(const char* name, const uint8_t* src, int N, int h, int w, int ch, void* dst,
                                 int out_form, int oh, int ow, const float* mean3, const float* std3, int mode,
                                 void* stream) {
    MIA_CHECK_ARG(src && dst && N > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "%s: bad arguments", name);
    MIA_CHECK_ARG(ch == 1 || ch == 3, "%s: channels must be 1 or 3, got %d", name, ch);
    MIA_CHECK_ARG(mode == 1 || (mean3 && std3), "%s: normalise mode needs mean and std", name);
    float m[3] = {0, 0, 0}, r[3] = {1, 1, 1};
    if (mode == 0) {
        for (int i = 0; i < 3; ++i) {
            volatile float mv = mean3[i] * 255.0f;
            volatile float sv = std3[i] * 255.0f;
            volatile float rv = 1.0f / sv;
            m[i] = mv;
            r[i] = rv;
        }
    }
    // tiled kernel when the tile's source rows fit in LDS: at most (RZ_ROWS-1)*h/oh + 3 rows (two taps per output
    // row, monotone in the row index), one more for slack
    const long long src_rows = ((long long)(RZ_ROWS - 1) * h + oh - 1) / oh + 4;
    const long long lds = (long long)ow * 16 + RZ_ROWS * 16 + src_rows * (long long)w * ch + 16;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (lds <= 64 * 1024 && N <= 65535 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
        dim3 grid((unsigned)((oh + RZ_ROWS - 1) / RZ_ROWS), (unsigned)N);
        if (out_form == 0)
            MIA_LAUNCH(resize_normalize_tile_kernel<0>, grid, dim3(256), (size_t)lds, s, src, h, w, ch, dst, oh, ow, m[0],
                       m[1], m[2], r[0], r[1], r[2], mode);
        else if (out_form == 1)
            MIA_LAUNCH(resize_normalize_tile_kernel<1>, grid, dim3(256), (size_t)lds, s, src, h, w, ch, dst, oh, ow, m[0],
                       m[1], m[2], r[0], r[1], r[2], mode);
        else
            MIA_LAUNCH(resize_normalize_tile_kernel<2>, grid, dim3(256), (size_t)lds, s, src, h, w, ch, dst, oh, ow, m[0],
                       m[1], m[2], r[0], r[1], r[2], mode);
        MIA_CHECK_LAUNCH(name);
        return 0;
    }
    const long long total = (long long)N * oh * ow;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (out_form == 0)
        MIA_LAUNCH(resize_normalize_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, src, N, h, w, ch, dst, oh, ow, m[0],
                   m[1], m[2], r[0], r[1], r[2], mode);
    else if (out_form == 1)
        MIA_LAUNCH(resize_normalize_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, src, N, h, w, ch, dst, oh, ow, m[0],
                   m[1], m[2], r[0], r[1], r[2], mode);
    else
        MIA_LAUNCH(resize_normalize_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, src, N, h, w, ch, dst, oh, ow, m[0],
                   m[1], m[2], r[0], r[1], r[2], mode);
    MIA_CHECK_LAUNCH(name);
    return 0;
}

extern "C" int mia_resize_normalize_u8(const uint8_t* src, int N, int h, int w, int ch, float* dst, int oh, int ow,
                                       const float* mean3, const float* std3, int mode, void* stream) {
    return resize_normalize_impl("mia_resize_normalize_u8", src, N, h, w, ch, dst, 0, oh, ow, mean3, std3, mode, stream);
}

extern "C" int mia_resize_normalize_u8_nhwc(const uint8_t* src, int N, int h, int w, int ch, void* dst, int dtype, int oh,
                                            int ow, const float* mean3, const float* std3, int mode, void* stream) {
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_resize_normalize_u8_nhwc: bad dtype %d", dtype);
    return resize_normalize_impl("mia_resize_normalize_u8_nhwc", src, N, h, w, ch, dst, dtype == MIA_BF16 ? 1 : 2, oh, ow,
                                 mean3, std3, mode, stream);
}

// 8-bit result of the same resize, HWC with 3 channels (gray replicated): the input of the augmentation kernels
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* src, int N, int h, int w, int ch, uint8_t* dst,
                                                        int oh, int ow) {
    const long long total = (long long)N * oh * ow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh), n = (int)(i / ((long long)ow * oh));
        const LinCoef cx = lin_coef(ox, w, ow), cy = lin_coef(oy, h, oh);
        const uint8_t* img = src + (size_t)n * h * w * ch;
        const uint8_t* p00 = img + ((size_t)cy.s0 * w + cx.s0) * ch;
        const uint8_t* p01 = img + ((size_t)cy.s0 * w + cx.s1) * ch;
        const uint8_t* p10 = img + ((size_t)cy.s1 * w + cx.s0) * ch;
        const uint8_t* p11 = img + ((size_t)cy.s1 * w + cx.s1) * ch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = ch == 3 ? c : 0;
            const int S0 = (int)p00[cs] * cx.w0 + (int)p01[cs] * cx.w1;
            const int S1 = (int)p10[cs] * cx.w0 + (int)p11[cs] * cx.w1;
            int v = ((cy.w0 * (S0 >> 4)) >> 16) + ((cy.w1 * (S1 >> 4)) >> 16);
            v = (v + 2) >> 2;
            dst[(size_t)i * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

extern "C" int mia_resize_u8(const uint8_t* src, int N, int h, int w, int ch, uint8_t* dst, int oh, int ow, void* stream) {
    MIA_CHECK_ARG(src && dst && N > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "mia_resize_u8: bad arguments");
    MIA_CHECK_ARG(ch == 1 || ch == 3, "mia_resize_u8: channels must be 1 or 3, got %d", ch);
    long long blocks = ((long long)N * oh * ow + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    MIA_LAUNCH(resize_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src,
                       N, h, w, ch, dst, oh, ow);
    MIA_CHECK_LAUNCH("mia_resize_u8");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// zero-intensity slice filter
// ------------------------------------------------------------------------------------------------
// flags[z] |= (any element of slice z != 0). grid = (splits, N). kind: 0 integer bytes, 1 float32
// (-0.0 equals zero, NaN does not).
__global__ __launch_bounds__(256) void slice_nonzero_kernel(const uint8_t* vol, long long slice_bytes, int kind,
                                                            int* flags) {
    const long long z = blockIdx.y;
    const uint8_t* base = vol + z * slice_bytes;
    const uint32_t fmask = kind == 1 ? 0x7fffffffu : 0xffffffffu;
    uint32_t acc = 0;
    // head bytes up to 16-byte alignment, 16-byte body, tail
    const uintptr_t addr = reinterpret_cast<uintptr_t>(base);
    long long head = (16 - (long long)(addr & 15)) & 15;
    if (head > slice_bytes) head = slice_bytes;
    const long long body = (slice_bytes - head) / 16;
    const uint4* vb = reinterpret_cast<const uint4*>(base + head);
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < body; i += stride) {
        const uint4 v = vb[i];
        acc |= (v.x & fmask) | (v.y & fmask) | (v.z & fmask) | (v.w & fmask);
    }
    if (blockIdx.x == 0) {
        // head [0, head) and tail [tail_begin, slice_bytes): fewer than 32 bytes in total, element-wise
        const long long tail_begin = head + body * 16;
        const int esz = kind == 1 ? 4 : 1;
        for (long long o = (long long)threadIdx.x * esz; o < head; o += 256 * esz) {
            if (kind == 1) acc |= (*reinterpret_cast<const uint32_t*>(base + o)) & fmask;
            else acc |= base[o];
        }
        for (long long o = tail_begin + (long long)threadIdx.x * esz; o < slice_bytes; o += 256 * esz) {
            if (kind == 1) acc |= (*reinterpret_cast<const uint32_t*>(base + o)) & fmask;
            else acc |= base[o];
        }
    }
    const unsigned long long any = __ballot(acc != 0);
    if ((threadIdx.x & 63) == 0 && any) atomicOr(flags + z, 1);
}

// keep[z] = flag != 0; idx = ascending positions of kept slices; count[0] = how many. One workgroup.
__global__ __launch_bounds__(256) void slice_compact_kernel(const int* flags, long long N, uint8_t* keep, int* idx,
                                                            int* count) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (long long z0 = 0; z0 < N; z0 += 256) {
        const long long z = z0 + tid;
        const int f = (z < N) ? (flags[z] != 0) : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        const int base = base_s;
        if (z < N) {
            keep[z] = (uint8_t)f;
            if (f) idx[base + woff + before] = (int)z;
        }
        __syncthreads();
        if (tid == 0) base_s = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) count[0] = base_s;
}

// ------------------------------------------------------------------------------------------------
// One launch for slices a single workgroup can scan (the configured volumes: 256x256 slices, 32..512 of them, where
// three launches -- zero fill, OR-reduce, compaction -- were a fixed 23 us whatever N): workgroup z scans slice z
// and stores its flag word (no zero fill, no atomic OR), then takes a ticket; the workgroup that draws the last ticket
// runs the compaction over all flag words and puts the ticket counter back to zero.
//
// Hand-over between workgroups (they may sit on different XCDs, whose L2 caches are not coherent with each other for
// ordinary accesses): the flag word is stored and loaded with agent scope, the ticket is an agent-scope
// read-modify-write issued, after that store has completed, by the thread that stored the flag, and every thread of
// the last workgroup passes an agent-scope acquire fence before it reads flag words.
//
// Ticket counters live in a zero-initialised device array built into the library; each call takes the next of 64
// slots, so calls on different streams of one GPU do not share a counter unless more than 64 are in flight at once.
// ------------------------------------------------------------------------------------------------
#define MIA_ZSF_TICKETS 64
static __device__ __align__(256) unsigned zsf_tickets[MIA_ZSF_TICKETS];
static unsigned* g_ticket_ptr[64];
static unsigned g_ticket_next[64];
static std::mutex g_ticket_mu;

// -> device address of the ticket slot for this call (nullptr + error message on failure)
static unsigned* zsf_ticket_slot() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        mia_set_error("mia_zero_slice_filter: hipGetDevice failed or device %d out of range: %s", dev, hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_ticket_mu);
    if (!g_ticket_ptr[dev]) {
        unsigned* ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(zsf_tickets));
        if (e != hipSuccess || !ptr) {
            mia_set_error("mia_zero_slice_filter: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return nullptr;
        }
        g_ticket_ptr[dev] = ptr;
    }
    return g_ticket_ptr[dev] + (g_ticket_next[dev]++ % MIA_ZSF_TICKETS);
}

__global__ __launch_bounds__(256) void slice_filter_fused_kernel(const uint8_t* __restrict__ vol, long long slice_bytes,
                                                                 int kind, int* flags, unsigned* ticket, int N,
                                                                 uint8_t* __restrict__ keep, int* __restrict__ idx,
                                                                 int* __restrict__ count) {
    __shared__ int wsum[4];
    __shared__ int base_s, last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long z = blockIdx.x;
    const uint8_t* base = vol + z * slice_bytes;
    const uint32_t fmask = kind == 1 ? 0x7fffffffu : 0xffffffffu;
    uint32_t acc = 0;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(base);
    long long head = (16 - (long long)(addr & 15)) & 15;
    if (head > slice_bytes) head = slice_bytes;
    const long long body = (slice_bytes - head) / 16;
    const uint4* vb = reinterpret_cast<const uint4*>(base + head);
    long long i = tid;
    // eight independent 16-byte loads in flight per lane
    for (; i + 7 * 256 < body; i += 8 * 256) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(vb + i + u * 256));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc |= (v[u].x & fmask) | (v[u].y & fmask) | (v[u].z & fmask) | (v[u].w & fmask);
    }
    for (; i < body; i += 256) {
        const uint4 v = vb[i];
        acc |= (v.x & fmask) | (v.y & fmask) | (v.z & fmask) | (v.w & fmask);
    }
    {
        // head [0, head) and tail [tail_begin, slice_bytes): fewer than 32 bytes in total, element-wise
        const long long tail_begin = head + body * 16;
        const int esz = kind == 1 ? 4 : 1;
        for (long long o = (long long)tid * esz; o < head; o += 256 * esz) {
            if (kind == 1) acc |= (*reinterpret_cast<const uint32_t*>(base + o)) & fmask;
            else acc |= base[o];
        }
        for (long long o = tail_begin + (long long)tid * esz; o < slice_bytes; o += 256 * esz) {
            if (kind == 1) acc |= (*reinterpret_cast<const uint32_t*>(base + o)) & fmask;
            else acc |= base[o];
        }
    }
    const unsigned long long any = __ballot(acc != 0);
    if (lane == 0) wsum[wave] = any != 0ull;
    __syncthreads();
    if (tid == 0) {
        const int f = wsum[0] | wsum[1] | wsum[2] | wsum[3];
        // The flag goes out as an agent-scope store (written through to where every XCD sees it); the wait makes this
        // thread hold the ticket back until that store has completed. Nothing else of this kernel is dirty
        // in the L2, so no cache write-back is asked for, and the workgroups that are not last invalidate nothing: a
        // full acquire-release ticket costs every workgroup an L2 write-back + invalidate (measured: 512 slices
        // 30 us against 12 us).
        __hip_atomic_store(flags + z, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = (t == gridDim.x - 1);
        base_s = 0;
    }
    __syncthreads();
    if (!last_s) return;
    // ---- the last workgroup to arrive: every flag word has been stored and released before its ticket was drawn
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int z0 = 0; z0 < N; z0 += 256) {
        const int zz = z0 + tid;
        const int f = (zz < N) ? (__hip_atomic_load(flags + zz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wave; ++k) woff += wsum[k];
        const int b0 = base_s;
        if (zz < N) {
            keep[zz] = (uint8_t)f;
            if (f) idx[b0 + woff + before] = zz;
        }
        __syncthreads();
        if (tid == 0) base_s = b0 + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) {
        count[0] = base_s;
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the slot's next call
    }
}

// slices up to this size are scanned by one workgroup each (one launch); larger ones are split over several
#define MIA_ZSF_FUSED_MAX_SLICE_BYTES (1ll << 20)

extern "C" int mia_zero_slice_filter(const void* vol, int elem_bytes, int is_float, long long n_slices,
                                     long long slice_elems, int* flags_ws, uint8_t* keep, int* idx, int* count,
                                     void* stream) {
    MIA_CHECK_ARG(vol && flags_ws && keep && idx && count, "mia_zero_slice_filter: null argument");
    MIA_CHECK_ARG(n_slices > 0 && n_slices <= 65535 && slice_elems > 0, "mia_zero_slice_filter: bad shape");
    MIA_CHECK_ARG((elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4) && (!is_float || elem_bytes == 4),
                  "mia_zero_slice_filter: unsupported element type (%d bytes, float=%d)", elem_bytes, is_float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long slice_bytes = slice_elems * elem_bytes;
    if (slice_bytes <= MIA_ZSF_FUSED_MAX_SLICE_BYTES) {
        unsigned* ticket = zsf_ticket_slot();
        if (!ticket) return -3;
        MIA_LAUNCH(slice_filter_fused_kernel, dim3((unsigned)n_slices), dim3(256), 0, s, (const uint8_t*)vol,
                           slice_bytes, is_float ? 1 : 0, flags_ws, ticket, (int)n_slices, keep, idx, count);
        MIA_CHECK_LAUNCH("mia_zero_slice_filter");
        return 0;
    }
    hipError_t e = mia_zero_words(flags_ws, n_slices, s);
    MIA_CHECK_ARG(e == hipSuccess, "mia_zero_slice_filter: zero fill failed: %s", hipGetErrorString(e));
    long long splits = slice_bytes / (16 * 256 * 8);  // >= 8 vector loads per thread
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    MIA_LAUNCH(slice_nonzero_kernel, dim3((unsigned)splits, (unsigned)n_slices), dim3(256), 0, s,
                       (const uint8_t*)vol, slice_bytes, is_float ? 1 : 0, flags_ws);
    MIA_LAUNCH(slice_compact_kernel, dim3(1), dim3(256), 0, s, flags_ws, n_slices, keep, idx, count);
    MIA_CHECK_LAUNCH("mia_zero_slice_filter");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The same filter for callers that cannot allocate device memory (code under HIP graph capture, tight serving
// loops): the per-slice flag words live in a file-scope device array instead of a caller-provided workspace.
// Each GPU holds its own copy of the array (256 KiB, until the process ends) at its own address, so the address is
// resolved once per device with that device current and kept; host code only ever uses the resolved pointer.
// Calls on one GPU share the array: two calls that overlap on the same GPU (different streams) race.
// ------------------------------------------------------------------------------------------------
#define MIA_ZSF_STATIC_MAX_SLICES 65535
#define MIA_MAX_DEVICES 64
static __device__ __align__(256) int scratch[MIA_ZSF_STATIC_MAX_SLICES + 1];
static int* g_scratch_ptr[MIA_MAX_DEVICES];
static std::mutex g_scratch_mu;

// Device address of `scratch` on the current device (nullptr + error message on failure).
static int* zsf_scratch_ptr() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= MIA_MAX_DEVICES) {
        mia_set_error("mia_zero_slice_filter_static: hipGetDevice failed or device %d out of range: %s", dev,
                      hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        int* ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            mia_set_error("mia_zero_slice_filter_static: hipGetSymbolAddress failed on device %d: %s", dev,
                          hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = ptr;
    }
    return g_scratch_ptr[dev];
}

// Resolves the built-in arrays' addresses on the current device. Called from mia_check_device(), so a process that
// checks its device at start-up (as the Python package's users do) never meets the first-use symbol lookup -- which
// may load the code object and takes a host lock -- inside a HIP graph capture.
int mia_preprocess_init_device() {
    if (!zsf_scratch_ptr()) return -3;
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -3;
        std::lock_guard<std::mutex> lock(g_ticket_mu);
        if (!g_ticket_ptr[dev]) {
            unsigned* ptr = nullptr;
            if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(zsf_tickets)) != hipSuccess || !ptr) {
                mia_set_error("mia_check_device: hipGetSymbolAddress(zsf_tickets) failed on device %d", dev);
                return -3;
            }
            g_ticket_ptr[dev] = ptr;
        }
    }
    return 0;
}

extern "C" int mia_zero_slice_filter_static(const void* vol, int elem_bytes, int is_float, long long n_slices,
                                            long long slice_elems, uint8_t* keep, int* idx, int* count,
                                            void* stream) {
    MIA_CHECK_ARG(vol && keep && idx && count, "mia_zero_slice_filter_static: null argument");
    MIA_CHECK_ARG(n_slices > 0 && slice_elems > 0, "mia_zero_slice_filter_static: bad shape");
    MIA_CHECK_ARG(n_slices <= MIA_ZSF_STATIC_MAX_SLICES,
                  "mia_zero_slice_filter_static: %lld slices exceed the built-in scratch (%d slices per call); "
                  "split the volume or use mia_zero_slice_filter with a workspace",
                  n_slices, MIA_ZSF_STATIC_MAX_SLICES);
    int* ptr = zsf_scratch_ptr();
    if (!ptr) return -3;
    // the array keeps the previous call's flags: mia_zero_slice_filter clears the first n_slices words first
    return mia_zero_slice_filter(vol, elem_bytes, is_float, n_slices, slice_elems, ptr, keep, idx, count, stream);
}

// out[k] = vol[idx[k]] for k < count[0]; grid = (chunks, n_slices) and rows beyond count exit at once.
__global__ __launch_bounds__(256) void gather_slices_kernel(const uint8_t* vol, long long slice_bytes, const int* idx,
                                                            const int* count, uint8_t* out) {
    const int k = blockIdx.y;
    if (k >= count[0]) return;
    const uint8_t* src = vol + (long long)idx[k] * slice_bytes;
    uint8_t* dst = out + (long long)k * slice_bytes;
    const long long stride = (long long)gridDim.x * 256;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        const long long nv = slice_bytes / 16;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride)
            reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
        for (long long o = nv * 16 + (long long)blockIdx.x * 256 + threadIdx.x; o < slice_bytes; o += stride)
            dst[o] = src[o];
    } else {
        for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < slice_bytes; o += stride) dst[o] = src[o];
    }
}

extern "C" int mia_gather_slices(const void* vol, int elem_bytes, long long n_slices, long long slice_elems,
                                 const int* idx, const int* count, void* out, void* stream) {
    MIA_CHECK_ARG(vol && idx && count && out, "mia_gather_slices: null argument");
    MIA_CHECK_ARG(n_slices > 0 && n_slices <= 65535 && slice_elems > 0 && elem_bytes > 0, "mia_gather_slices: bad shape");
    const long long slice_bytes = slice_elems * elem_bytes;
    long long chunks = slice_bytes / (16 * 256 * 4);
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    MIA_LAUNCH(gather_slices_kernel, dim3((unsigned)chunks, (unsigned)n_slices), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (const uint8_t*)vol, slice_bytes, idx, count,
                       (uint8_t*)out);
    MIA_CHECK_LAUNCH("mia_gather_slices");
    return 0;
}

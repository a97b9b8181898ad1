// Shared device/host helpers for the gfx950 kernels of libmia_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define MIA_WAVE 64

// Every kernel launch of the library goes through MIA_LAUNCH, which also counts it (mia_launch_count): the number of
// launches per training step is a figure the benchmark reports, and the only place to count it without a profiler.
extern unsigned long long g_mia_launches;
#define MIA_LAUNCH(...)                    \
    do {                                   \
        ++g_mia_launches;                  \
        hipLaunchKernelGGL(__VA_ARGS__);   \
    } while (0)

// ----------------------------------------------------------------------------------------------
// error handling: every extern "C" entry returns 0 or a negative code and records a message.
// ----------------------------------------------------------------------------------------------
void mia_set_error(const char* fmt, ...);

#define MIA_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            mia_set_error(__VA_ARGS__);          \
            return -1;                           \
        }                                        \
    } while (0)

#define MIA_CHECK_LAUNCH(name)                                                     \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            mia_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return -2;                                                             \
        }                                                                          \
    } while (0)

// ----------------------------------------------------------------------------------------------
// bf16 storage type (raw 16 bits) and conversions. Round-to-nearest-even through the compiler's
// __bf16 cast (v_cvt_pk_bf16_f32 on gfx950), which keeps NaN a NaN.
// ----------------------------------------------------------------------------------------------
struct bf16_t {
    uint16_t v;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}

typedef float v2f __attribute__((ext_vector_type(2)));
// two floats -> two bf16 in one dword with ONE v_cvt_pk_bf16_f32 (round to nearest even)
__device__ __forceinline__ uint32_t pack_bf16x2(v2f v) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 h = __builtin_convertvector(v, bf2);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ v2f unpack_bf16x2(uint32_t w) {
    return v2f{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
}

template <typename T>
__device__ __forceinline__ float to_f32(T x);
template <>
__device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return bf16_bits_to_f32(x.v); }

template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) {
    bf16_t r;
    r.v = f32_to_bf16_bits(x);
    return r;
}

// 4 consecutive channel values as fp32, loaded/stored with one vector access
// (16 B for float, 8 B for bf16). Pointers must be aligned to that width.
struct f4 {
    float x, y, z, w;
};

template <typename T>
__device__ __forceinline__ f4 load4(const T* p);
template <>
__device__ __forceinline__ f4 load4<float>(const float* p) {
    float4 v = *reinterpret_cast<const float4*>(p);
    return {v.x, v.y, v.z, v.w};
}
template <>
__device__ __forceinline__ f4 load4<bf16_t>(const bf16_t* p) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    return {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
            __uint_as_float(v.y & 0xffff0000u)};
}
template <typename T>
__device__ __forceinline__ void store4(T* p, f4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f4 v) {
    *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, f4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v2f{v.x, v.y}), pack_bf16x2(v2f{v.z, v.w}));
}

// ----------------------------------------------------------------------------------------------
// activations
// ----------------------------------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (about 1 ulp each): an IEEE division here would cost ~10 more VALU instructions per
// element in kernels that are otherwise bandwidth-bound.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// d/dx silu(x) = s(x) * (1 + x * (1 - s(x)))
__device__ __forceinline__ float silu_gradf_(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}

// ----------------------------------------------------------------------------------------------
// Two channels at a time. Arithmetic on a 2-vector compiles to the packed-f32 instructions (v_pk_fma_f32,
// v_pk_mul_f32, v_pk_add_f32: two lanes' worth of work per issue slot); only exp and rcp stay scalar. The
// activation-heavy kernels are bound by vector-ALU issue, so this halves their cost per element.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ v2f sigmoid2(v2f z) {
    const v2f t = z * -1.4426950408889634f;
    v2f e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const v2f d = e + 1.0f;
    v2f r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return r;
}
__device__ __forceinline__ v2f silu2(v2f z) { return z * sigmoid2(z); }
__device__ __forceinline__ v2f silu_grad2(v2f z) {
    const v2f s = sigmoid2(z);
    return s * (1.0f + z * (1.0f - s));
}
// four channels = two 2-vectors
struct f22 {
    v2f lo, hi;
};
template <typename T>
__device__ __forceinline__ f22 load22(const T* p);
template <>
__device__ __forceinline__ f22 load22<float>(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return {v2f{v.x, v.y}, v2f{v.z, v.w}};
}
template <>
__device__ __forceinline__ f22 load22<bf16_t>(const bf16_t* p) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return {v2f{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)},
            v2f{__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)}};
}
template <typename T>
__device__ __forceinline__ void store22(T* p, const f22& v);
template <>
__device__ __forceinline__ void store22<float>(float* p, const f22& v) {
    *reinterpret_cast<float4*>(p) = make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y);
}
template <>
__device__ __forceinline__ void store22<bf16_t>(bf16_t* p, const f22& v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.lo), pack_bf16x2(v.hi));
}
// eight consecutive channels as two f22: ONE 16-byte access per lane for bf16 (the width at which HBM streams best),
// two for fp32
template <typename T>
__device__ __forceinline__ void load44(const T* p, f22& a, f22& b);
template <>
__device__ __forceinline__ void load44<float>(const float* p, f22& a, f22& b) {
    a = load22<float>(p);
    b = load22<float>(p + 4);
}
template <>
__device__ __forceinline__ void load44<bf16_t>(const bf16_t* p, f22& a, f22& b) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    a = {unpack_bf16x2(v.x), unpack_bf16x2(v.y)};
    b = {unpack_bf16x2(v.z), unpack_bf16x2(v.w)};
}
template <typename T>
__device__ __forceinline__ void store44(T* p, const f22& a, const f22& b);
template <>
__device__ __forceinline__ void store44<float>(float* p, const f22& a, const f22& b) {
    store22<float>(p, a);
    store22<float>(p + 4, b);
}
template <>
__device__ __forceinline__ void store44<bf16_t>(bf16_t* p, const f22& a, const f22& b) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(a.lo), pack_bf16x2(a.hi), pack_bf16x2(b.lo), pack_bf16x2(b.hi));
}
__device__ __forceinline__ f4 to_f4(const f22& v) { return {v.lo.x, v.lo.y, v.hi.x, v.hi.y}; }
__device__ __forceinline__ f22 to_f22(const f4& v) { return {v2f{v.x, v.y}, v2f{v.z, v.w}}; }

// ----------------------------------------------------------------------------------------------
// wave (64-lane) reductions. Inside a row of 16 lanes the partners come through DPP operand modifiers (quad
// permutes, then the half-row and full-row mirrors): four VALU instructions, no LDS crossbar trip as a lane shuffle
// (ds_bpermute) would take. The four row results are then read as scalars and combined, which also broadcasts the
// total to every lane.
// ----------------------------------------------------------------------------------------------
#define MIA_DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define MIA_DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]
#define MIA_DPP_ROW_HALF_MIRROR 0x141
#define MIA_DPP_ROW_MIRROR 0x140
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<MIA_DPP_QUAD_XOR1>(v);
    v += dpp_f32<MIA_DPP_QUAD_XOR2>(v);
    v += dpp_f32<MIA_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f32<MIA_DPP_ROW_MIRROR>(v);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<MIA_DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<MIA_DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f32<MIA_DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<MIA_DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Zero fill of n 32-bit words by a kernel. hipMemsetAsync is not used on the kernels' stream: inside a captured HIP
// graph it becomes a memset node, and the replays of such nodes were observed not to clear the range again (results
// were right on the first replay only, while the memory was still fresh), whereas a kernel node replays like the rest.
static __global__ __launch_bounds__(256) void mia_zero_words_kernel(unsigned* __restrict__ p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0u;
}

static inline hipError_t mia_zero_words(void* p, long long n_words, hipStream_t s) {
    if (n_words <= 0) return hipSuccess;
    long long blocks = (n_words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    MIA_LAUNCH(mia_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned*)p, n_words);
    return hipGetLastError();
}

enum { MIA_F32 = 0, MIA_BF16 = 1 };
enum { MIA_ACT_NONE = 0, MIA_ACT_SILU = 1, MIA_ACT_RELU = 2 };

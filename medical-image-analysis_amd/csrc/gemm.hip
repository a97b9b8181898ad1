// MFMA GEMMs for the 1x1 convolutions and linear layers (gfx950).
//
//   gemm_nt : C[M,N] = pro(A)[M,K] . B[N,K]^T (+bias) (act) (+D)      forward and data-gradient
//             optional per-column sum / sum-of-squares partials for train-mode BatchNorm
//   gemm_tn : O[N,K] += G[M,N]^T . pro(X)[M,K]                          weight-gradient (fp32 out)
//
// pro(.) is the fused BatchNorm+SiLU(+squeeze-excite gate) applied while the A/X tile is staged:
//   a' = act(a * pscale[k] + pshift[k]) * gate[row / rows_per_img][k]
// Rows are NHWC pixels (b, h, w) so `row / rows_per_img` is the image index.
//
// Storage type T is float (exact-f32 MFMA 16x16x4) or bf16 (MFMA 16x16x32); accumulation is fp32.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <typename T>
struct Mma;

template <>
struct Mma<bf16_t> {
    static constexpr int KSTEP = 32;  // reduction elements per MFMA
    static constexpr int BK = 64;     // reduction elements per LDS stage
    static constexpr int PAD = 8;     // LDS row padding, elements
    typedef bf16x8v Frag;
    // k-contiguous operand image: row_k points at &tile[row][kk*KSTEP]
    static __device__ __forceinline__ Frag load(const bf16_t* row_k, int lane) {
        return *reinterpret_cast<const Frag*>(row_k + 8 * (lane >> 4));
    }
    // reduction-major image ([red][col]): base points at &tile[kk*KSTEP][col0]; LD = row stride (elements)
    static __device__ __forceinline__ Frag load_t(const bf16_t* base, int LD, int lane) {
        const int i = lane & 15, g = lane >> 4;
        const bf16_t* p0 = base + (8 * g + (i >> 2)) * LD + 4 * (i & 3);
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * LD));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(Frag, r);
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <>
struct Mma<float> {
    static constexpr int KSTEP = 4;
    static constexpr int BK = 32;
    static constexpr int PAD = 4;
    typedef float Frag;
    static __device__ __forceinline__ Frag load(const float* row_k, int lane) { return row_k[lane >> 4]; }
    static __device__ __forceinline__ Frag load_t(const float* base, int LD, int lane) {
        return base[(lane >> 4) * LD + (lane & 15)];
    }
    static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// 16-byte vector of T as floats and back
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* o) {
        float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* o) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
};
template <>
struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float* o) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[2 * i] = __uint_as_float(w[i] << 16);
            o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* o) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (uint32_t)f32_to_bf16_bits(o[2 * i]) | ((uint32_t)f32_to_bf16_bits(o[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// a 16-byte chunk held in registers between the global load and the LDS store
typedef uint4 raw16;
template <typename T>
__device__ __forceinline__ void raw_unpack(const raw16& r, float* o);
template <>
__device__ __forceinline__ void raw_unpack<float>(const raw16& r, float* o) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
template <>
__device__ __forceinline__ void raw_unpack<bf16_t>(const raw16& r, float* o) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <typename T>
__device__ __forceinline__ raw16 raw_pack(const float* o);
template <>
__device__ __forceinline__ raw16 raw_pack<float>(const float* o) {
    return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
}
template <>
__device__ __forceinline__ raw16 raw_pack<bf16_t>(const float* o) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = (uint32_t)f32_to_bf16_bits(o[2 * i]) | ((uint32_t)f32_to_bf16_bits(o[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

struct GemmNTParams {
    const void* A;
    const void* B;
    void* C;
    const void* D;
    int M, N, K, lda, ldb, ldc;
    const float* pscale;
    const float* pshift;
    const float* gate;
    int rows_per_img;
    int pro_act;
    const float* bias;
    int epi_act;
    float* stats;  // [row_slots][2][N] or null
    int row_slots;  // persistent row slots (= partial-sum rows when stats != null)
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNTParams p) {
    using M_ = Mma<T>;
    using V_ = Vec16<T>;
    constexpr int BK = M_::BK, LDK = BK + M_::PAD;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int VEC = V_::N;
    constexpr int CPR = BK / VEC;  // 16-byte chunks per staged row
    constexpr int LDC = BN + VEC;  // C staging row stride (elements)
    constexpr int STAGE_BYTES = (BM + BN) * LDK * (int)sizeof(T);
    constexpr int CS_BYTES = BM * LDC * (int)sizeof(T);
    constexpr int MAIN_BYTES = STAGE_BYTES > CS_BYTES ? STAGE_BYTES : CS_BYTES;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    static_assert(256 % CPR == 0, "a thread keeps one k-chunk column");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + BM * LDK;
    T* Cs = reinterpret_cast<T*>(smem);
    float* sred = reinterpret_cast<float*>(smem + MAIN_BYTES);  // [WAVES_M][BN][2]

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(p.B);
    T* __restrict__ C = reinterpret_cast<T*>(p.C);
    const T* __restrict__ D = reinterpret_cast<const T*>(p.D);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int M = p.M, N = p.N, K = p.K;
    // Workgroup -> (row slot, column tile). Blocks b and b+8 share an XCD (and its L2), so the column tiles of
    // one row slot are placed on consecutive slots of ONE XCD: they stream the same A rows at about the same
    // time and all but the first read them from that L2. Purely a speed choice.
    const int ny = (N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    const int rowslot = (sidx / ny) * 8 + xcd;
    if (rowslot >= p.row_slots) return;
    const int n0 = (sidx % ny) * BN;
    const bool has_pro = p.pscale != nullptr;
    const bool has_gate = p.gate != nullptr;
    const int kc = (tid % CPR) * VEC;  // this thread's k-chunk offset inside a stage
    const int rbase = tid / CPR;       // and its first row; further chunks are 256/CPR rows apart
    constexpr int RSTEP = 256 / CPR;
    constexpr int ACH = BM / RSTEP, BCH = BN / RSTEP;

    float st_sum = 0.f, st_sq = 0.f;
    const int tiles_m = (M + BM - 1) / BM;

    raw16 ra[ACH], rb[BCH];
    auto prefetch = [&](int m0, int k0) {
        const int gk = k0 + kc;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int gm = m0 + rbase + i * RSTEP;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (gm < M && gk < K) ra[i] = *reinterpret_cast<const raw16*>(A + (size_t)gm * p.lda + gk);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int gn = n0 + rbase + i * RSTEP;
            rb[i] = make_uint4(0, 0, 0, 0);
            if (gn < N && gk < K) rb[i] = *reinterpret_cast<const raw16*>(B + (size_t)gn * p.ldb + gk);
        }
    };
    auto commit = [&](int m0, int k0) {
        const int gk = k0 + kc;
        if ((has_pro || has_gate) && gk < K) {
            float sc[VEC], sh[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                sc[i] = has_pro ? p.pscale[gk + i] : 1.f;
                sh[i] = has_pro ? p.pshift[gk + i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int gm = m0 + rbase + i * RSTEP;
                if (gm < M) {
                    float v[VEC];
                    raw_unpack<T>(ra[i], v);
                    if (has_pro) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float z = v[e] * sc[e] + sh[e];
                            v[e] = (p.pro_act == MIA_ACT_SILU) ? siluf_(z) : z;
                        }
                    }
                    if (has_gate) {
                        const float* g = p.gate + (size_t)(gm / p.rows_per_img) * K + gk;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) v[e] *= g[e];
                    }
                    ra[i] = raw_pack<T>(v);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < ACH; ++i)
            *reinterpret_cast<raw16*>(As + (rbase + i * RSTEP) * LDK + kc) = ra[i];
#pragma unroll
        for (int i = 0; i < BCH; ++i)
            *reinterpret_cast<raw16*>(Bs + (rbase + i * RSTEP) * LDK + kc) = rb[i];
    };

    f32x4 acc[MT][NT];
    int tm = rowslot, k0 = 0;
    if (tm < tiles_m) prefetch(tm * BM, 0);
    while (tm < tiles_m) {
        const int m0 = tm * BM;
        if (k0 == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();   // LDS (staging or the C image of the previous tile) is free
        commit(m0, k0);
        __syncthreads();
        // the next stage's global loads fly while this stage's MFMAs (and the epilogue) run
        int nk0 = k0 + BK, ntm = tm;
        if (nk0 >= K) { nk0 = 0; ntm = tm + p.row_slots; }
        if (ntm < tiles_m) prefetch(ntm * BM, nk0);
        {
            const int krem = (K - k0) < BK ? (K - k0) : BK;
            const int ksteps = (krem + M_::KSTEP - 1) / M_::KSTEP;
            for (int kk = 0; kk < ksteps; ++kk) {
                typename M_::Frag a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    a[i] = M_::load(As + (wr * WM + i * 16 + (lane & 15)) * LDK + kk * M_::KSTEP, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    b[j] = M_::load(Bs + (wc * WN + j * 16 + (lane & 15)) * LDK + kk * M_::KSTEP, lane);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]);
            }
        }
        const bool last_k = (k0 + BK >= K);
        const int cur_k0 = k0;
        tm = ntm;
        k0 = nk0;
        if (!last_k) continue;
        (void)cur_k0;

        // ---------------------------------- epilogue ----------------------------------
        __syncthreads();  // every wave is done with As/Bs before Cs overwrites them
        const int lrow = (lane >> 4) * 4, lcol = lane & 15;
        if (p.stats) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r];
                        s += v;
                        q += v * v;
                    }
                s += __shfl_xor(s, 16, 64);
                q += __shfl_xor(q, 16, 64);
                s += __shfl_xor(s, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (lane < 16) {
                    const int col = wc * WN + j * 16 + lane;
                    sred[(wr * BN + col) * 2 + 0] = s;
                    sred[(wr * BN + col) * 2 + 1] = q;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = wc * WN + j * 16 + lcol;
            const int gn = n0 + col;
            const float bv = (p.bias && gn < N) ? p.bias[gn] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bv;
                    if (p.epi_act == MIA_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.epi_act == MIA_ACT_SILU) v = siluf_(v);
                    Cs[(wr * WM + i * 16 + lrow + r) * LDC + col] = from_f32<T>(v);
                }
        }
        __syncthreads();
        if (p.stats && tid < BN) {
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) {
                st_sum += sred[(w * BN + tid) * 2 + 0];
                st_sq += sred[(w * BN + tid) * 2 + 1];
            }
        }
        if ((N % VEC) == 0 && (p.ldc % VEC) == 0) {
            constexpr int CPC = BN / VEC;
            for (int c = tid; c < BM * CPC; c += 256) {
                const int r = c / CPC, cc = (c % CPC) * VEC;
                const int gm = m0 + r, gn = n0 + cc;
                if (gm < M && gn < N) {
                    float v[VEC];
                    V_::load(Cs + r * LDC + cc, v);
                    if (D) {
                        float d[VEC];
                        V_::load(D + (size_t)gm * p.ldc + gn, d);
#pragma unroll
                        for (int i = 0; i < VEC; ++i) v[i] += d[i];
                    }
                    V_::store(C + (size_t)gm * p.ldc + gn, v);
                }
            }
        } else {
            for (int c = tid; c < BM * BN; c += 256) {
                const int r = c / BN, cc = c % BN;
                const int gm = m0 + r, gn = n0 + cc;
                if (gm < M && gn < N) {
                    float v = to_f32<T>(Cs[r * LDC + cc]);
                    if (D) v += to_f32<T>(D[(size_t)gm * p.ldc + gn]);
                    C[(size_t)gm * p.ldc + gn] = from_f32<T>(v);
                }
            }
        }
    }
    if (p.stats && tid < BN && n0 + tid < N) {
        p.stats[((size_t)rowslot * 2 + 0) * N + n0 + tid] = st_sum;
        p.stats[((size_t)rowslot * 2 + 1) * N + n0 + tid] = st_sq;
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_nt(const GemmNTParams& p, int part_rows, hipStream_t stream) {
    using M_ = Mma<T>;
    constexpr int LDK = M_::BK + M_::PAD;
    constexpr int VEC = Vec16<T>::N;
    constexpr int STAGE_BYTES = (BM + BN) * LDK * (int)sizeof(T);
    constexpr int CS_BYTES = BM * (BN + VEC) * (int)sizeof(T);
    constexpr int MAIN_BYTES = STAGE_BYTES > CS_BYTES ? STAGE_BYTES : CS_BYTES;
    constexpr int LDS_BYTES = MAIN_BYTES + WAVES_M * BN * 2 * (int)sizeof(float);
    const int tiles_m = cdiv(p.M, BM);
    GemmNTParams q = p;
    q.row_slots = p.stats ? part_rows : (tiles_m < 1024 ? tiles_m : 1024);
    dim3 grid((unsigned)(cdiv(q.row_slots, 8) * 8 * cdiv(p.N, BN)));
    auto kern = gemm_nt_kernel<T, BM, BN, WAVES_M, WAVES_N>;
    static bool attr_set = false;
    if (!attr_set && LDS_BYTES > 65536) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS_BYTES, stream, q);
    return 0;
}

template <typename T>
static int dispatch_nt(const GemmNTParams& p, int part_rows, hipStream_t stream) {
    if (p.N <= 32) return launch_nt<T, 128, 32, 4, 1>(p, part_rows, stream);
    if (p.N <= 64) return launch_nt<T, 128, 64, 4, 1>(p, part_rows, stream);
    return launch_nt<T, 128, 128, 2, 2>(p, part_rows, stream);
}

extern "C" int mia_gemm_nt(int dtype, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* D,
                           int M, int N, int K, const float* pscale, const float* pshift, int pro_act,
                           const float* gate, int rows_per_img, const float* bias, int epi_act, float* stats,
                           int part_rows, void* stream) {
    const int vec = dtype == MIA_BF16 ? 8 : 4;
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_gemm_nt: bad dtype %d", dtype);
    MIA_CHECK_ARG(A && B && C, "mia_gemm_nt: null operand");
    MIA_CHECK_ARG(M > 0 && N > 0 && K > 0, "mia_gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    MIA_CHECK_ARG(K % vec == 0 && lda % vec == 0 && ldb % vec == 0 && lda >= K && ldb >= K,
                  "mia_gemm_nt: K=%d lda=%d ldb=%d must be multiples of %d (16-byte rows)", K, lda, ldb, vec);
    MIA_CHECK_ARG(ldc >= N, "mia_gemm_nt: ldc=%d < N=%d", ldc, N);
    MIA_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0 &&
                      ((uintptr_t)D % 16) == 0,
                  "mia_gemm_nt: operands must be 16-byte aligned");
    MIA_CHECK_ARG(!(pscale == nullptr) == !(pshift == nullptr), "mia_gemm_nt: pscale/pshift must come together");
    MIA_CHECK_ARG(!gate || rows_per_img > 0, "mia_gemm_nt: gate needs rows_per_img > 0");
    MIA_CHECK_ARG(!stats || (part_rows > 0 && part_rows <= 65535), "mia_gemm_nt: stats needs 0 < part_rows <= 65535");
    GemmNTParams p;
    p.A = A; p.B = B; p.C = C; p.D = D;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.pscale = pscale; p.pshift = pshift; p.gate = gate; p.rows_per_img = rows_per_img > 0 ? rows_per_img : 1;
    p.pro_act = pro_act; p.bias = bias; p.epi_act = epi_act; p.stats = stats;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16) dispatch_nt<bf16_t>(p, part_rows, s);
    else dispatch_nt<float>(p, part_rows, s);
    MIA_CHECK_LAUNCH("mia_gemm_nt");
    return 0;
}

// ================================================================================================
// weight gradient: O[N,K] (fp32) += sum_m G[m,N] * pro(X)[m,K]
// ================================================================================================
struct GemmTNParams {
    const void* G;
    const void* X;
    float* O;
    int M, N, K, ldg, ldx, ldo;
    const float* pscale;
    const float* pshift;
    const float* gate;
    int rows_per_img;
    int pro_act;
    int chunks_per_split;
};

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNParams p) {
    using M_ = Mma<T>;
    using V_ = Vec16<T>;
    constexpr int BT = 128;          // output tile edge (n and k)
    constexpr int BR = M_::BK;       // reduction rows per stage
    constexpr int VEC = V_::N;
    constexpr int LD = BT + VEC;     // LDS row stride, elements
    constexpr int CPR = BT / VEC;
    constexpr int MT = 4, NT = 4;    // 64x64 per wave

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Gs = reinterpret_cast<T*>(smem);  // [BR][LD]
    T* Xs = Gs + BR * LD;                // [BR][LD]

    const T* __restrict__ G = reinterpret_cast<const T*>(p.G);
    const T* __restrict__ X = reinterpret_cast<const T*>(p.X);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * BT, k0 = blockIdx.y * BT;
    const int M = p.M, N = p.N, K = p.K;
    const bool has_pro = p.pscale != nullptr;
    const bool has_gate = p.gate != nullptr;

    // tiles of this wave that hold any valid output
    const int nvalid = N - (n0 + wr * 64), kvalid = K - (k0 + wc * 64);
    const int mt_cnt = nvalid <= 0 ? 0 : (nvalid >= 64 ? 4 : (nvalid + 15) / 16);
    const int nt_cnt = kvalid <= 0 ? 0 : (kvalid >= 64 ? 4 : (kvalid + 15) / 16);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cc = (tid % CPR) * VEC;  // column chunk of this thread (same for G and X tiles)
    float sc[VEC], sh[VEC];
    const int gkx = k0 + cc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        sc[i] = (has_pro && gkx < K) ? p.pscale[gkx + i] : 1.f;
        sh[i] = (has_pro && gkx < K) ? p.pshift[gkx + i] : 0.f;
    }

    const long long chunk_begin = (long long)blockIdx.z * p.chunks_per_split;
    long long chunk_end = chunk_begin + p.chunks_per_split;
    const long long total_chunks = ((long long)M + BR - 1) / BR;
    if (chunk_end > total_chunks) chunk_end = total_chunks;

    constexpr int RSTEP = 256 / CPR;
    constexpr int CH = BR / RSTEP;
    const int rbase = tid / CPR;
    raw16 rg[CH], rx[CH];
    auto prefetch = [&](long long ch) {
        const long long m0 = ch * BR;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const long long gm = m0 + rbase + i * RSTEP;
            rg[i] = make_uint4(0, 0, 0, 0);
            rx[i] = make_uint4(0, 0, 0, 0);
            if (gm < M && n0 + cc < N) rg[i] = *reinterpret_cast<const raw16*>(G + (size_t)gm * p.ldg + n0 + cc);
            if (gm < M && gkx < K) rx[i] = *reinterpret_cast<const raw16*>(X + (size_t)gm * p.ldx + gkx);
        }
    };
    auto commit = [&](long long ch) {
        const long long m0 = ch * BR;
        if ((has_pro || has_gate) && gkx < K) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const long long gm = m0 + rbase + i * RSTEP;
                if (gm < M) {
                    float v[VEC];
                    raw_unpack<T>(rx[i], v);
                    if (has_pro) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float z = v[e] * sc[e] + sh[e];
                            v[e] = (p.pro_act == MIA_ACT_SILU) ? siluf_(z) : z;
                        }
                    }
                    if (has_gate) {
                        const float* g = p.gate + (size_t)(gm / p.rows_per_img) * K + gkx;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) v[e] *= g[e];
                    }
                    rx[i] = raw_pack<T>(v);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            *reinterpret_cast<raw16*>(Gs + (rbase + i * RSTEP) * LD + cc) = rg[i];
            *reinterpret_cast<raw16*>(Xs + (rbase + i * RSTEP) * LD + cc) = rx[i];
        }
    };

    if (chunk_begin < chunk_end) prefetch(chunk_begin);
    for (long long ch = chunk_begin; ch < chunk_end; ++ch) {
        __syncthreads();
        commit(ch);
        __syncthreads();
        if (ch + 1 < chunk_end) prefetch(ch + 1);  // in flight during the MFMAs below
        if (mt_cnt > 0 && nt_cnt > 0) {
#pragma unroll
            for (int kk = 0; kk < BR / M_::KSTEP; ++kk) {
                typename M_::Frag a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    a[i] = M_::load_t(Gs + kk * M_::KSTEP * LD + wr * 64 + i * 16, LD, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    b[j] = M_::load_t(Xs + kk * M_::KSTEP * LD + wc * 64 + j * 16, LD, lane);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        if (i < mt_cnt && j < nt_cnt) acc[i][j] = M_::mma(a[i], b[j], acc[i][j]);
            }
        }
    }

    const int lrow = (lane >> 4) * 4, lcol = lane & 15;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (i < mt_cnt && j < nt_cnt) {
                const int gk = k0 + wc * 64 + j * 16 + lcol;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gn = n0 + wr * 64 + i * 16 + lrow + r;
                    if (gn < N && gk < K) atomicAdd(p.O + (size_t)gn * p.ldo + gk, acc[i][j][r]);
                }
            }
        }
}

extern "C" int mia_gemm_tn(int dtype, const void* G, int ldg, const void* X, int ldx, float* O, int ldo, int M, int N,
                           int K, const float* pscale, const float* pshift, int pro_act, const float* gate,
                           int rows_per_img, void* stream) {
    const int vec = dtype == MIA_BF16 ? 8 : 4;
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_gemm_tn: bad dtype %d", dtype);
    MIA_CHECK_ARG(G && X && O, "mia_gemm_tn: null operand");
    MIA_CHECK_ARG(M > 0 && N > 0 && K > 0, "mia_gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
    MIA_CHECK_ARG(N % vec == 0 && K % vec == 0 && ldg % vec == 0 && ldx % vec == 0 && ldg >= N && ldx >= K,
                  "mia_gemm_tn: N=%d K=%d ldg=%d ldx=%d must be multiples of %d", N, K, ldg, ldx, vec);
    MIA_CHECK_ARG(ldo >= K, "mia_gemm_tn: ldo=%d < K=%d", ldo, K);
    MIA_CHECK_ARG(((uintptr_t)G % 16) == 0 && ((uintptr_t)X % 16) == 0, "mia_gemm_tn: operands must be 16-byte aligned");
    MIA_CHECK_ARG(!(pscale == nullptr) == !(pshift == nullptr), "mia_gemm_tn: pscale/pshift must come together");
    MIA_CHECK_ARG(!gate || rows_per_img > 0, "mia_gemm_tn: gate needs rows_per_img > 0");
    GemmTNParams p;
    p.G = G; p.X = X; p.O = O; p.M = M; p.N = N; p.K = K; p.ldg = ldg; p.ldx = ldx; p.ldo = ldo;
    p.pscale = pscale; p.pshift = pshift; p.gate = gate; p.rows_per_img = rows_per_img > 0 ? rows_per_img : 1;
    p.pro_act = pro_act;
    const int BR = dtype == MIA_BF16 ? 64 : 32;
    const int tiles = cdiv(N, 128) * cdiv(K, 128);
    const long long chunks = ((long long)M + BR - 1) / BR;
    // aim for ~1536 blocks, at least 4 reduction chunks per block
    long long splits = 1536 / tiles;
    if (splits < 1) splits = 1;
    if (splits > (chunks + 3) / 4) splits = (chunks + 3) / 4;
    if (splits < 1) splits = 1;
    p.chunks_per_split = (int)((chunks + splits - 1) / splits);
    splits = (chunks + p.chunks_per_split - 1) / p.chunks_per_split;
    dim3 grid(cdiv(N, 128), cdiv(K, 128), (unsigned)splits);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MIA_BF16) {
        constexpr int LDS = 2 * 64 * (128 + 8) * 2;
        hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, grid, dim3(256), LDS, s, p);
    } else {
        constexpr int LDS = 2 * 32 * (128 + 4) * 4;
        hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), LDS, s, p);
    }
    MIA_CHECK_LAUNCH("mia_gemm_tn");
    return 0;
}

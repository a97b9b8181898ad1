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
#include <stdlib.h>

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
        for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v2f{o[2 * i], o[2 * i + 1]});
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// a 16-byte chunk held in registers between the global load and the LDS store
typedef uint4 raw16;
// BatchNorm + SiLU (+ gate) on one 16-byte chunk, two channels per instruction.
// sc/sh/g point at VEC consecutive floats (registers after inlining); g may be null.
template <typename T>
struct Pro16;
template <>
struct Pro16<bf16_t> {
    template <bool SILU, bool AFF, bool GATE>
    static __device__ __forceinline__ raw16 apply(const raw16& r, const float* sc, const float* sh, const float* g) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v2f v = unpack_bf16x2(w[i]);
            if (AFF) v = v * v2f{sc[2 * i], sc[2 * i + 1]} + v2f{sh[2 * i], sh[2 * i + 1]};
            if (SILU) v = silu2(v);
            if (GATE) v = v * v2f{g[2 * i], g[2 * i + 1]};
            o[i] = pack_bf16x2(v);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};
template <>
struct Pro16<float> {
    template <bool SILU, bool AFF, bool GATE>
    static __device__ __forceinline__ raw16 apply(const raw16& r, const float* sc, const float* sh, const float* g) {
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            v2f v = {__uint_as_float(w[2 * i]), __uint_as_float(w[2 * i + 1])};
            if (AFF) v = v * v2f{sc[2 * i], sc[2 * i + 1]} + v2f{sh[2 * i], sh[2 * i + 1]};
            if (SILU) v = silu2(v);
            if (GATE) v = v * v2f{g[2 * i], g[2 * i + 1]};
            o[2 * i] = __float_as_uint(v.x);
            o[2 * i + 1] = __float_as_uint(v.y);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};
// runtime flags -> one of the specialisations (wave-uniform branches, taken once per chunk)
template <typename T>
__device__ __forceinline__ raw16 pro16(const raw16& r, bool silu, bool aff, bool gate, const float* sc, const float* sh,
                                       const float* g) {
    if (aff && silu && gate) return Pro16<T>::template apply<true, true, true>(r, sc, sh, g);
    if (aff && silu) return Pro16<T>::template apply<true, true, false>(r, sc, sh, g);
    if (aff && gate) return Pro16<T>::template apply<false, true, true>(r, sc, sh, g);
    if (aff) return Pro16<T>::template apply<false, true, false>(r, sc, sh, g);
    if (gate) return Pro16<T>::template apply<false, false, true>(r, sc, sh, g);
    return r;
}

template <int V>
struct IntTag {
    static constexpr int value = V;
};

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
    float* stats;   // [row_slots][2][N] or null
    int row_slots;  // persistent row slots (= partial-sum rows when stats != null)
    int bk_log2;    // reduction elements per LDS stage (a power of two, at most Mma<T>::BK)
    int alias_cs;   // 1: the C staging rows reuse the operand stage (one more barrier per tile, less LDS)
    int dump_off;   // byte offset of the per-thread 16-byte dump slots
    int pro_off;    // byte offset of the prologue vectors (scale[K], shift[K]) staged once per workgroup (PRO == 2)
};

// Guarded 16-byte loads through a buffer descriptor: an offset at or beyond the descriptor's size returns zeros,
// so out-of-range rows/columns need no branch. Branch-free loads matter twice: no exec-mask juggling, and the
// compiler can count the loads in flight, which lets a commit wait for ITS stage only (s_waitcnt vmcnt(N), N > 0)
// while the next stage's loads stay outstanding. A branchy load sequence degrades every wait to vmcnt(0).
#define MIA_BUF_OOB 0xffffffffu
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// stream = true: non-temporal hint for tiles read once (keeps them from evicting the weights in L2)
template <bool STREAM>
__device__ __forceinline__ raw16 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, STREAM ? 2 : 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// One pipeline slot: the registers a thread holds for one (tile, k-stage) between its global loads and its commit.
template <int ACH, int BCH, int VEC, int PRO>
struct NtSlot {
    raw16 ra[ACH], rb[BCH];
    float pg[PRO ? ACH : 1][PRO ? VEC : 1];         // gate of each chunk row's image (row 0 only when the tile lies in one image)
    float psc[PRO ? VEC : 1], psh[PRO ? VEC : 1];   // BatchNorm scale / shift of this thread's k-chunk
};

// Four waves stacked along M; each wave owns 16*MT complete rows of the BM x BN tile (BM = 64*MT, BN = 16*NT), so
// the epilogue is wave-local: 16 rows at a time go through a private LDS strip and leave as 16-byte row chunks.
//
// Pipeline: TWO stages of global loads are in flight per workgroup (slots R0/R1). When a stage is committed to
// LDS its slot is refilled with the stage after next, so loads overlap the prologue arithmetic of the commit as
// well as the MFMAs; with one stage in flight the kernel is bound by the load round trip.
// PRO: 0 no prologue; 1 prologue with its vectors read from memory per stage and one gate vector per chunk row; 2 the
// form for large maps (row tiles inside one image, workgroups that walk many tiles): BatchNorm vectors staged in LDS
// once, one gate vector per tile where the tile lies in one image. The load INSTRUCTIONS, not their bytes, bound the
// prologue there: 17 buffer loads per thread and stage in form 1, 7 in form 2 (24-wide projection of the 150x150 maps:
// 192 -> 131 us). On the small maps form 2 measured slower (start-up round trip, two images in most tiles).
// EPI: 1 general; 0 the launch has neither column statistics nor a bias (data gradients): their per-column registers
// (3 per 16 columns) are not allocated, which takes the 128-row plain tiles from two to three workgroups per CU.
template <typename T, int MT, int NT, int PRO, int DEPTH, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNTParams p) {
    using M_ = Mma<T>;
    using V_ = Vec16<T>;
    constexpr int BM = 64 * MT, BN = 16 * NT;
    constexpr int VEC = V_::N;
    constexpr int LDC = BN + VEC;  // C staging row stride (elements)
    constexpr int ACH = BM / 32, BCH = (BN + 31) / 32;  // chunk rows per thread at the deepest stage
    typedef NtSlot<ACH, BCH, VEC, PRO> Slot;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bk = 1 << p.bk_log2, ldk = bk + M_::PAD;
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + BM * ldk;
    const int stage_bytes = (BM + BN) * ldk * (int)sizeof(T);

    const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
    const T* __restrict__ B = reinterpret_cast<const T*>(p.B);
    T* __restrict__ C = reinterpret_cast<T*>(p.C);
    const T* __restrict__ D = reinterpret_cast<const T*>(p.D);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T* Cw = reinterpret_cast<T*>(smem + (p.alias_cs ? 0 : stage_bytes)) + wave * 16 * LDC;
    T* dump = reinterpret_cast<T*>(smem + p.dump_off + tid * 16);
    // [2][K]: BatchNorm scale and shift (prologue variants), staged once per workgroup. Every thread needs 2*VEC of them
    // per stage; read straight from memory that is four more buffer loads per thread and stage, and the load
    // instructions (not their bytes) bound the prologue variants on the large maps. LDS reads do not pass that unit.
    float* Ps = reinterpret_cast<float*>(smem + p.pro_off);
    const int M = p.M, N = p.N, K = p.K;
    // Workgroup -> (row slot, column tile). Blocks b and b+8 share an XCD (and its L2), so the column tiles of
    // one row slot are placed on consecutive slots of ONE XCD: they stream the same A rows at about the same
    // time and all but the first read them from that L2. Purely a speed choice.
    const int ny = (N + BN - 1) / BN;
    const int xcd = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    const int rowslot = (sidx / ny) * 8 + xcd;
    if (rowslot >= p.row_slots) return;
    const int n0 = (sidx % ny) * BN;
    const bool has_pro = PRO && p.pscale != nullptr;
    const bool has_gate = PRO && p.gate != nullptr;
    const bool pro_silu = has_pro && p.pro_act == MIA_ACT_SILU;
    if constexpr (PRO == 2) {
        if (has_pro) {
            for (int i = tid; i < K; i += 256) { Ps[i] = p.pscale[i]; Ps[K + i] = p.pshift[i]; }
            __syncthreads();
        }
    }
    const int cpr_log2 = p.bk_log2 - (VEC == 8 ? 3 : 2);  // 16-byte chunks per staged row
    const int kc = (tid & ((1 << cpr_log2) - 1)) * VEC;   // this thread's k-chunk offset inside a stage
    const int rbase = tid >> cpr_log2;                    // and its first row; further chunks are rstep rows apart
    const int rstep = 256 >> cpr_log2;
    const float inv_rpi = 1.0f / (float)p.rows_per_img;

    constexpr int NE = EPI ? NT : 1;
    float st_sum[NE], st_sq[NE], bias_r[NE];   // bias of this lane's columns, read once (no loads in the epilogue)
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        st_sum[j] = st_sq[j] = 0.f;
        const int gn = n0 + j * 16 + (lane & 15);
        bias_r[j] = (EPI && p.bias && gn < N) ? p.bias[gn] : 0.f;
    }
    float* const stats = EPI ? p.stats : nullptr;
    const int tiles_m = (M + BM - 1) / BM;

    // row / rows_per_img without an integer division: float estimate, then one correction step either way
    // (rows < 2^24 are exact in float; larger row counts take the division)
    auto image_of = [&](int row) -> int {
        if (M >= (1 << 24)) return row / p.rows_per_img;
        int q = (int)((float)row * inv_rpi);
        const int rem = row - q * p.rows_per_img;
        if (rem < 0) --q;
        else if (rem >= p.rows_per_img) ++q;
        return q;
    };
    // every row of the tile starting at m0 (whose first row lies in image img0) belongs to that image
    auto single_image = [&](int m0, int img0) -> bool {
        const int last = m0 + BM - 1 < M ? m0 + BM - 1 : M - 1;
        return last < (img0 + 1) * p.rows_per_img;
    };
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (unsigned)(((size_t)(M - 1) * p.lda + K) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (unsigned)(((size_t)(N - 1) * p.ldb + K) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rsSc = make_rsrc(p.pscale, has_pro ? (unsigned)K * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rsSh = make_rsrc(p.pshift, has_pro ? (unsigned)K * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rsG = make_rsrc(p.gate, has_gate ? (unsigned)(((size_t)(M - 1) / p.rows_per_img + 1) * K * 4) : 0u);
    auto fetch = [&](Slot& R, int t, int k0) {
        const int m0 = t * BM;
        const int gk = k0 + kc;
        const bool kin = gk < K;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int r = rbase + i * rstep, gm = m0 + r;
            const unsigned off = (r < BM && gm < M && kin) ? (unsigned)(gm * p.lda + gk) * (unsigned)sizeof(T) : MIA_BUF_OOB;
            // read-once rows get the streaming hint; with several column tiles the siblings re-read them from L2
            R.ra[i] = ny == 1 ? buf_load16<true>(rsA, off) : buf_load16<false>(rsA, off);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = rbase + i * rstep, gn = n0 + r;
            const unsigned off = (r < BN && gn < N && kin) ? (unsigned)(gn * p.ldb + gk) * (unsigned)sizeof(T) : MIA_BUF_OOB;
            R.rb[i] = buf_load16<false>(rsB, off);
        }
        if constexpr (PRO) {
            if (has_pro && kin) {
#pragma unroll
                for (int e = 0; e < VEC; e += 4) {
                    if constexpr (PRO == 2) {
                        const float4 a4 = *reinterpret_cast<const float4*>(Ps + gk + e);
                        const float4 b4 = *reinterpret_cast<const float4*>(Ps + K + gk + e);
                        R.psc[e] = a4.x; R.psc[e + 1] = a4.y; R.psc[e + 2] = a4.z; R.psc[e + 3] = a4.w;
                        R.psh[e] = b4.x; R.psh[e + 1] = b4.y; R.psh[e + 2] = b4.z; R.psh[e + 3] = b4.w;
                    } else {
                        const raw16 a4 = buf_load16<false>(rsSc, (unsigned)(gk + e) * 4u);
                        const raw16 b4 = buf_load16<false>(rsSh, (unsigned)(gk + e) * 4u);
                        R.psc[e] = __uint_as_float(a4.x); R.psc[e + 1] = __uint_as_float(a4.y);
                        R.psc[e + 2] = __uint_as_float(a4.z); R.psc[e + 3] = __uint_as_float(a4.w);
                        R.psh[e] = __uint_as_float(b4.x); R.psh[e + 1] = __uint_as_float(b4.y);
                        R.psh[e + 2] = __uint_as_float(b4.z); R.psh[e + 3] = __uint_as_float(b4.w);
                    }
                }
            }
            if (has_gate) {
                const int img0 = PRO == 2 ? image_of(m0) : 0;
                const bool one_img = PRO == 2 && single_image(m0, img0);
#pragma unroll
                for (int i = 0; i < ACH; ++i) {
                    if (PRO == 1 || i == 0 || !one_img) {
                        const int gm = m0 + rbase + i * rstep;
                        const int img = one_img ? img0 : image_of(gm < M ? gm : M - 1);
                        const unsigned goff = kin ? (unsigned)(img * K + gk) * 4u : MIA_BUF_OOB;
#pragma unroll
                        for (int e = 0; e < VEC; e += 4) {
                            const raw16 g4 = buf_load16<false>(rsG, goff + e * 4);
                            R.pg[i][e] = __uint_as_float(g4.x); R.pg[i][e + 1] = __uint_as_float(g4.y);
                            R.pg[i][e + 2] = __uint_as_float(g4.z); R.pg[i][e + 3] = __uint_as_float(g4.w);
                        }
                    }
                }
            }
        }
    };
    auto commit = [&](Slot& R, int m0, int k0) {
        const int gk = k0 + kc;
        if constexpr (PRO) {
            if ((has_pro || has_gate) && gk < K) {
                // two copies of the loop, chosen by a workgroup-uniform branch: a per-element select between pg[0] and
                // pg[i] is folded into an indexed read of the slot, which moves the whole slot into scratch memory
                if (PRO == 2 && has_gate && single_image(m0, image_of(m0))) {
#pragma unroll
                    for (int i = 0; i < ACH; ++i) {
                        const int r = rbase + i * rstep, gm = m0 + r;
                        if (r < BM && gm < M) R.ra[i] = pro16<T>(R.ra[i], pro_silu, has_pro, true, R.psc, R.psh, R.pg[0]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < ACH; ++i) {
                        const int r = rbase + i * rstep, gm = m0 + r;
                        if (r < BM && gm < M) R.ra[i] = pro16<T>(R.ra[i], pro_silu, has_pro, has_gate, R.psc, R.psh, R.pg[i]);
                    }
                }
            }
        }
        // chunk rows past the tile (short stages, BN not a multiple of the row step) land in a per-thread dump slot:
        // unconditional stores keep the commit free of branches (see buf_load16)
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int r = rbase + i * rstep;
            *reinterpret_cast<raw16*>(r < BM ? As + r * ldk + kc : dump) = R.ra[i];
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int r = rbase + i * rstep;
            *reinterpret_cast<raw16*>(r < BN ? Bs + r * ldk + kc : dump) = R.rb[i];
        }
    };

    f32x4 acc[MT][NT];
    int tm = rowslot, k0 = 0;      // stage being consumed
    int f_tm = rowslot, f_k0 = 0;  // next stage to fetch
    auto advance = [&](int& t, int& k) {
        k += bk;
        if (k >= K) { k = 0; t += p.row_slots; }
    };
    const T* a_frag = As + (wave * 16 * MT + (lane & 15)) * ldk;
    const T* b_frag = Bs + (lane & 15) * ldk;

    auto step = [&](Slot& R) {
        const int m0 = tm * BM;
        if (k0 == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();  // the operand stage is free (every wave is past its MFMAs and, if aliased, its epilogue)
        commit(R, m0, k0);
        __syncthreads();
        // refill this slot with the stage after next: its loads fly during the MFMAs, the epilogue and the NEXT commit
        if (f_tm < tiles_m) {
            fetch(R, f_tm, f_k0);
            advance(f_tm, f_k0);
        }
        {
            const int krem = (K - k0) < bk ? (K - k0) : bk;
            const int ksteps = (krem + M_::KSTEP - 1) / M_::KSTEP;
            for (int kk = 0; kk < ksteps; ++kk) {
                typename M_::Frag a[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = M_::load(a_frag + i * 16 * ldk + kk * M_::KSTEP, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const typename M_::Frag b = M_::load(b_frag + j * 16 * ldk + kk * M_::KSTEP, lane);
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][j] = M_::mma(a[i], b, acc[i][j]);
                }
            }
        }
        const bool last_k = (k0 + bk >= K);
        advance(tm, k0);
        if (!last_k) return;

        // ---------------------------------- epilogue (wave-local) ----------------------------------
        if (p.alias_cs) __syncthreads();  // every wave is done with As/Bs before the strips overwrite them
        const int lrow = (lane >> 4) * 4, lcol = lane & 15;
        if (EPI && stats) {
#pragma unroll
            for (int j = 0; j < NE; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[i][j][r];  // rows past M hold exact zeros (zero-filled A rows)
                        st_sum[j] += v;
                        st_sq[j] += v * v;
                    }
        }
        const bool vec_ok = (N % VEC) == 0 && (p.ldc % VEC) == 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // the activation is chosen once per strip, not per element: left inside the loops, the test compiles to
            // four scalar branches per output value, which cost more than the rest of the epilogue together
            auto to_strip = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = EPI ? acc[i][j][r] + bias_r[EPI ? j : 0] : acc[i][j][r];
                        if (ACT == MIA_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (ACT == MIA_ACT_SILU) v = siluf_(v);
                        Cw[(lrow + r) * LDC + j * 16 + lcol] = from_f32<T>(v);
                    }
                }
            };
            if (p.epi_act == MIA_ACT_RELU) to_strip(IntTag<MIA_ACT_RELU>{});
            else if (p.epi_act == MIA_ACT_SILU) to_strip(IntTag<MIA_ACT_SILU>{});
            else to_strip(IntTag<MIA_ACT_NONE>{});
            __builtin_amdgcn_wave_barrier();
            const int mrow = m0 + wave * 16 * MT + i * 16;
            if (vec_ok) {
                constexpr int CPC = BN / VEC;
                for (int c = lane; c < 16 * CPC; c += 64) {
                    const int r = c / CPC, cc = (c % CPC) * VEC;
                    const int gm = mrow + r, gn = n0 + cc;
                    if (gm < M && gn < N) {
                        float v[VEC];
                        V_::load(Cw + r * LDC + cc, v);
                        if (D) {
                            float d[VEC];
                            V_::load(D + (size_t)gm * p.ldc + gn, d);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) v[e] += d[e];
                        }
                        V_::store(C + (size_t)gm * p.ldc + gn, v);
                    }
                }
            } else {
                for (int c = lane; c < 16 * BN; c += 64) {
                    const int r = c / BN, cc = c % BN;
                    const int gm = mrow + r, gn = n0 + cc;
                    if (gm < M && gn < N) {
                        float v = to_f32<T>(Cw[r * LDC + cc]);
                        if (D) v += to_f32<T>(D[(size_t)gm * p.ldc + gn]);
                        C[(size_t)gm * p.ldc + gn] = from_f32<T>(v);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    if constexpr (DEPTH == 2) {
        Slot R0, R1;
        if (f_tm < tiles_m) { fetch(R0, f_tm, f_k0); advance(f_tm, f_k0); }
        if (f_tm < tiles_m) { fetch(R1, f_tm, f_k0); advance(f_tm, f_k0); }
        while (tm < tiles_m) {
            step(R0);
            if (tm >= tiles_m) break;
            step(R1);
        }
    } else {
        Slot R0;
        if (f_tm < tiles_m) { fetch(R0, f_tm, f_k0); advance(f_tm, f_k0); }
        while (tm < tiles_m) step(R0);
    }

    if (EPI && stats) {
        // lanes l, l+16, l+32, l+48 hold the same column: fold them, then fold the four waves through LDS
        __syncthreads();
        float* sred = reinterpret_cast<float*>(smem);  // [4][BN][2]
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            float s = st_sum[j], q = st_sq[j];
            s += __shfl_xor(s, 16, 64);
            q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (lane < 16) {
                sred[(wave * BN + j * 16 + lane) * 2 + 0] = s;
                sred[(wave * BN + j * 16 + lane) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                s += sred[(w * BN + tid) * 2 + 0];
                q += sred[(w * BN + tid) * 2 + 1];
            }
            stats[((size_t)rowslot * 2 + 0) * N + n0 + tid] = s;
            stats[((size_t)rowslot * 2 + 1) * N + n0 + tid] = q;
        }
    }
}

template <typename T, int MT, int NT, int PRO, int EPI>
static int launch_nt_(const GemmNTParams& p, int part_rows, hipStream_t stream);

static int nt_stage_depth(int K, int full, int kstep) {
    // the smallest power of two covering K, between one MFMA step (at least 16 elements) and the full depth
    int bk = full;
    while (bk / 2 >= K && bk / 2 >= (kstep > 16 ? kstep : 16)) bk /= 2;
    return bk;
}

template <typename T, int MT, int NT>
static int launch_nt(const GemmNTParams& p, int part_rows, hipStream_t stream) {
    if (p.pscale || p.gate) {
        // maps of 32x32 and more (the narrow tiles only: the wide ones never see such maps in this network)
        if (NT <= 3 && p.rows_per_img >= 1024 && (2 * p.K * 4) <= 32 * 1024) return launch_nt_<T, MT, NT, (NT <= 3 ? 2 : 1), 1>(p, part_rows, stream);
        return launch_nt_<T, MT, NT, 1, 1>(p, part_rows, stream);
    }
    if (!p.stats && !p.bias) return launch_nt_<T, MT, NT, 0, 0>(p, part_rows, stream);
    return launch_nt_<T, MT, NT, 0, 1>(p, part_rows, stream);
}

template <typename T, int MT, int NT, int PRO, int EPI>
static int launch_nt_(const GemmNTParams& p, int part_rows, hipStream_t stream) {
    using M_ = Mma<T>;
    constexpr int BM = 64 * MT, BN = 16 * NT;
    constexpr int VEC = Vec16<T>::N;
    GemmNTParams q = p;
    const int bk = nt_stage_depth(p.K, M_::BK, M_::KSTEP);
    q.bk_log2 = 0;
    while ((1 << q.bk_log2) < bk) ++q.bk_log2;
    const int stage_bytes = (BM + BN) * (bk + M_::PAD) * (int)sizeof(T);
    const int strip_bytes = 4 * 16 * (BN + VEC) * (int)sizeof(T);
    const int red_bytes = 4 * BN * 2 * (int)sizeof(float);
    // private strips cost LDS: keep them separate only while three workgroups still fit a CU
    q.alias_cs = (stage_bytes + strip_bytes) * 3 > 160 * 1024 ? 1 : 0;
    int lds = q.alias_cs ? (stage_bytes > strip_bytes ? stage_bytes : strip_bytes) : stage_bytes + strip_bytes;
    if (lds < red_bytes) lds = red_bytes;
    lds = (lds + 15) & ~15;
    q.dump_off = lds;
    lds += 256 * 16;
    const int tiles_m = cdiv(p.M, BM);
    q.row_slots = p.stats ? part_rows : (tiles_m < 1024 ? tiles_m : 1024);
    q.pro_off = lds;
    if (PRO == 2) lds += (2 * p.K * 4 + 15) & ~15;
    dim3 grid((unsigned)(cdiv(q.row_slots, 8) * 8 * cdiv(p.N, BN)));
    // wide tiles and prologue variants are short of registers: one stage in flight; narrow plain ones stream: two
    auto kern = gemm_nt_kernel<T, MT, NT, PRO, ((NT <= 6 && !PRO) ? 2 : 1), EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    MIA_LAUNCH(kern, grid, dim3(256), lds, stream, q);
    return 0;
}

// ================================================================================================
// Streaming form of gemm_nt for the projections of the large maps (bf16, BatchNorm+SiLU(+gate) prologue, column
// statistics): few channels on both sides, millions of rows. There the whole weight matrix is a handful of MFMA
// operands, so every wave keeps it in REGISTERS for the whole kernel and streams rows past it: no LDS staging, no
// barriers, no tiles shared between waves (the tiled kernel spends these launches in its commit/barrier cycle: 2.2-3.3
// TB/s where this form reaches 3.6-4.3). The product is computed transposed, D[n][m] = W[n][:] . act[m][:], with the
// weights as the first MFMA operand: a lane then
//   - supplies activations as 16-byte global loads already in operand layout (lane = m + 16*g holds 8 consecutive k
//     of row m) and runs the prologue on those registers,
//   - receives four consecutive output channels of ONE row: an 8-byte store straight from the accumulator.
// The squeeze-excite gate multiplies the WEIGHTS instead of the activations, (a*g).w = a.(g*w): a wave walks a
// contiguous range of 16-row blocks image by image, blocks never cross an image, and the weight operands are re-derived
// when the range enters the next image (once in hundreds of blocks). Rounding therefore happens on g*w instead of on
// a*g: one bf16 rounding either way.
// Loads run two blocks ahead of the arithmetic; all loads and stores of the inner loop are unconditional buffer
// accesses (masked through out-of-range offsets), so its waits are counted ones that leave the younger requests in
// flight. Column statistics: per-lane sums, folded once per workgroup in a fixed order (one row of partials each).
// (The same kernel without prologue was measured on the expansions and data gradients of these maps: the 8-byte
// stores lose to the tiled kernel's 16-byte row chunks wherever the output is the wide side -- 128 vs 67 us on the
// 192-channel data gradient -- and without the prologue's arithmetic to hide there is nothing to gain on the rest.)
// ================================================================================================
template <int NTW, int KS, int MT>
__global__ __launch_bounds__(256, 2) void gemm_skinny_kernel(GemmNTParams p) {
    typedef bf16_t T;
    constexpr bool PS_LDS = KS > 2;     // BatchNorm vectors from LDS (too many k slots per lane for registers)
    constexpr int KP = KS * 32;
    __shared__ float Ps[PS_LDS ? 2 * KP : 4];
    __shared__ float red[4 * 2 * NTW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ml = lane & 15, g = lane >> 4;
    const int M = p.M, N = p.N, K = p.K;
    // BatchNorm + SiLU on the activations is fixed at compile time (a run-time choice of the activation inside the
    // loop makes the compiler wait for every load in flight before the first use); the gate is optional
    const bool has_gate = p.gate != nullptr;
    const int rpi = has_gate ? p.rows_per_img : M;           // blocks never cross an image (one "image" without a gate)
    const int nimg = (M + rpi - 1) / rpi;
    constexpr int RB = 16 * MT;                              // rows per block
    const int bpi = (rpi + RB - 1) / RB;
    const int total = nimg * bpi;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int per = (total + nw - 1) / nw;
    const int blk0 = gw * per;
    const int blk_end = blk0 + per < total ? blk0 + per : total;

    int kb[KS];       // first k of (MFMA step s, lane group g)
    bool kok[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) {
        kb[s_] = s_ * 32 + g * 8;
        kok[s_] = kb[s_] < K;
    }
    if constexpr (PS_LDS) {
        for (int i = tid; i < KP; i += 256) {
            Ps[i] = i < K ? p.pscale[i] : 0.f;
            Ps[KP + i] = i < K ? p.pshift[i] : 0.f;
        }
        __syncthreads();
    }
    float psc[!PS_LDS ? KS : 1][8], psh[!PS_LDS ? KS : 1][8];
    if constexpr (!PS_LDS) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                psc[s_][e] = kok[s_] ? p.pscale[kb[s_] + e] : 0.f;
                psh[s_][e] = kok[s_] ? p.pshift[kb[s_] + e] : 0.f;
            }
    }

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, (unsigned)(((size_t)(M - 1) * p.lda + K) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.B, (unsigned)(((size_t)(N - 1) * p.ldb + K) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rsC = make_rsrc(p.C, (unsigned)(((size_t)(M - 1) * p.ldc + N) * sizeof(T)));
    const __amdgpu_buffer_rsrc_t rsG = make_rsrc(p.gate, has_gate ? (unsigned)((size_t)nimg * K * 4) : 0u);

    // weight operands: lane (ml, g) holds W[nt*16 + ml][kb[s] .. +7], times the image's gate
    bf16x8v wf[NTW][KS];
    auto load_w = [&](int img) {
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            float gt[8];
            if (has_gate) {
                const unsigned goff = kok[s_] ? (unsigned)(img * K + kb[s_]) * 4u : MIA_BUF_OOB;
                const raw16 g0 = buf_load16<false>(rsG, goff), g1 = buf_load16<false>(rsG, goff == MIA_BUF_OOB ? goff : goff + 16u);
                gt[0] = __uint_as_float(g0.x); gt[1] = __uint_as_float(g0.y); gt[2] = __uint_as_float(g0.z); gt[3] = __uint_as_float(g0.w);
                gt[4] = __uint_as_float(g1.x); gt[5] = __uint_as_float(g1.y); gt[6] = __uint_as_float(g1.z); gt[7] = __uint_as_float(g1.w);
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int n = nt * 16 + ml;
                const unsigned off = (n < N && kok[s_]) ? (unsigned)(n * p.ldb + kb[s_]) * 2u : MIA_BUF_OOB;
                raw16 w = buf_load16<false>(rsW, off);
                if (has_gate) w = Pro16<T>::template apply<false, false, true>(w, nullptr, nullptr, gt);
                wf[nt][s_] = __builtin_bit_cast(bf16x8v, w);
            }
        }
    };

    struct Frag {
        raw16 a[MT][KS];
    };
    // A wave's range is walked image by image (a "segment" = its blocks inside one image): the weight operands are
    // fixed inside a segment, and the inner loop holds nothing but counted loads, arithmetic and stores.
    int seg_end = 0, row0 = 0, bi_base = 0;   // current segment: end block, first row and first block of its image
    auto fetch = [&](Frag& f, int blk) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int r = ((blk - bi_base) * MT + t) * 16 + ml;
            const int m = row0 + r;
            const bool ok = blk < seg_end && r < rpi && m < M;
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
                f.a[t][s_] = buf_load16<true>(rsA, (ok && kok[s_]) ? (unsigned)(m * p.lda + kb[s_]) * 2u : MIA_BUF_OOB);
        }
    };

    float st_s[NTW][4], st_q[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st_s[nt][r] = st_q[nt][r] = 0.f;

    auto process = [&](const Frag& f, int blk) {
        // the BatchNorm vectors are re-read from LDS for every block: hoisted out of the loop (they do not change)
        // they would occupy 16 registers per k step and push the kernel into scratch memory
        int ps_off = 0;
        asm volatile("" : "+v"(ps_off));
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int r = ((blk - bi_base) * MT + t) * 16 + ml;
            const int m = row0 + r;
            const bool ok = r < rpi && m < M;
            bf16x8v af[KS];
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                raw16 a = f.a[t][s_];
                if constexpr (PS_LDS) {
                    float sc[8], sh[8];
                    const float* ps = Ps + ps_off + kb[s_];
                    const float4 a0 = *reinterpret_cast<const float4*>(ps);
                    const float4 a1 = *reinterpret_cast<const float4*>(ps + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(ps + KP);
                    const float4 b1 = *reinterpret_cast<const float4*>(ps + KP + 4);
                    sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
                    sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
                    a = Pro16<T>::template apply<true, true, false>(a, sc, sh, nullptr);
                } else {
                    a = Pro16<T>::template apply<true, true, false>(a, psc[s_], psh[s_], nullptr);
                }
                // rows outside the block contribute nothing (their loads returned zeros, but silu(shift) != 0)
                if (!ok) a = make_uint4(0u, 0u, 0u, 0u);
                af[s_] = __builtin_bit_cast(bf16x8v, a);
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][s_], af[s_], acc, 0, 0, 0);
                // this lane: row m, channels nt*16 + 4g .. +3
#pragma unroll
                for (int r_ = 0; r_ < 4; ++r_) {
                    st_s[nt][r_] += acc[r_];
                    st_q[nt][r_] += acc[r_] * acc[r_];
                }
                const int n = nt * 16 + g * 4;
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u o = {pack_bf16x2(v2f{acc[0], acc[1]}), pack_bf16x2(v2f{acc[2], acc[3]})};
                __builtin_amdgcn_raw_buffer_store_b64(o, rsC, (ok && n < N) ? (unsigned)(m * p.ldc + n) * 2u : MIA_BUF_OOB, 0, 0);
            }
        }
    };
    // Three register sets in rotation, by name: a copy between sets would have to wait for the loads it copies.
    Frag f0, f1, f2;
    for (int seg = blk0; seg < blk_end; seg = seg_end) {
        const int img = seg / bpi;
        bi_base = img * bpi;
        row0 = img * rpi;
        seg_end = bi_base + bpi < blk_end ? bi_base + bpi : blk_end;
        load_w(img);
        fetch(f0, seg);
        fetch(f1, seg + 1);
        // three blocks per turn; the first turn is peeled off so that the loop is entered with the same requests in
        // flight as its back edge has
        auto turn = [&](int blk) -> bool {
            fetch(f2, blk + 2);
            process(f0, blk);
            if (blk + 1 >= seg_end) return false;
            fetch(f0, blk + 3);
            process(f1, blk + 1);
            if (blk + 2 >= seg_end) return false;
            fetch(f1, blk + 4);
            process(f2, blk + 2);
            return blk + 3 < seg_end;
        };
        int blk = seg;
        if (turn(blk)) {
            blk += 3;
            while (turn(blk)) blk += 3;
        }
    }
    // lanes ml = 0..15 of a group g hold the same channels: fold them, then the waves of the workgroup in order
    float* mine = red + wave * 2 * NTW * 16;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) {
            float a = st_s[nt][r_], b = st_q[nt][r_];
            a += dpp_f32<MIA_DPP_QUAD_XOR1>(a); b += dpp_f32<MIA_DPP_QUAD_XOR1>(b);
            a += dpp_f32<MIA_DPP_QUAD_XOR2>(a); b += dpp_f32<MIA_DPP_QUAD_XOR2>(b);
            a += dpp_f32<MIA_DPP_ROW_HALF_MIRROR>(a); b += dpp_f32<MIA_DPP_ROW_HALF_MIRROR>(b);
            a += dpp_f32<MIA_DPP_ROW_MIRROR>(a); b += dpp_f32<MIA_DPP_ROW_MIRROR>(b);
            if (ml == 0) {
                mine[nt * 16 + g * 4 + r_] = a;
                mine[NTW * 16 + nt * 16 + g * 4 + r_] = b;
            }
        }
    __syncthreads();
    if (tid < NTW * 16 && tid < N) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            a += red[w * 2 * NTW * 16 + tid];
            b += red[w * 2 * NTW * 16 + NTW * 16 + tid];
        }
        p.stats[((size_t)blockIdx.x * 2 + 0) * N + tid] = a;
        p.stats[((size_t)blockIdx.x * 2 + 1) * N + tid] = b;
    }
}

// Shapes the streaming kernel is built for (the 24- and 32-channel projections of the 150x150 and 75x75 maps), and
// its launch. Returns 1 when it took the product.
static int try_skinny(const GemmNTParams& p, int part_rows, hipStream_t stream) {
    static const bool off = getenv("MIA_GEMM_NO_STREAM") != nullptr;   // documented A/B switch, read once per process
    if (off) return 0;
    if (!p.pscale || p.pro_act != MIA_ACT_SILU || !p.stats || p.bias || p.D || p.epi_act != MIA_ACT_NONE) return 0;
    if (p.M < 400000 || p.N % 4 || p.ldc % 4 || p.K % 8 || p.K > 192 || p.N > 32 || part_rows < 64) return 0;
    if (p.gate && (p.rows_per_img < 1024 || p.M % p.rows_per_img)) return 0;
    const dim3 grid((unsigned)part_rows);
    switch (cdiv(p.K, 32)) {
        case 1: MIA_LAUNCH((gemm_skinny_kernel<2, 1, 2>), grid, dim3(256), 0, stream, p); return 1;
        case 2: MIA_LAUNCH((gemm_skinny_kernel<2, 2, 2>), grid, dim3(256), 0, stream, p); return 1;
        case 5: MIA_LAUNCH((gemm_skinny_kernel<2, 5, 1>), grid, dim3(256), 0, stream, p); return 1;
        case 6: MIA_LAUNCH((gemm_skinny_kernel<2, 6, 1>), grid, dim3(256), 0, stream, p); return 1;
        default: return 0;
    }
}

// Column-tile widths on offer (multiples of 16). 144 = 6*24 divides most EfficientNet-B3 expanded widths.
template <typename T>
static int dispatch_nt(const GemmNTParams& p, int part_rows, hipStream_t stream) {
    static const int widths[] = {32, 48, 96, 128, 144, 192, 240};
    int best = 128;
    double best_cost = -1.0;
    for (int w : widths) {
        // columns computed (padding included), with a per-tile overhead that favours wide tiles: every column tile
        // re-streams the A rows and re-runs the prologue, and narrow tiles feed the MFMAs poorly
        const double cost = (double)cdiv(p.N, w) * w * (1.0 + 32.0 / w);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = w; }
    }
    // the 240-wide tile is register-bound (one row tile per wave); with few rows to stream (the 10x10 stages) two
    // 128-wide column tiles of 128 rows were measured faster: 1392-wide expansion 37 -> 31 us, 232-wide projection 61 -> 52 us
    if (best == 240 && p.M <= 16384) best = 128;
    // few row tiles: halve the tile height so that the grid still covers the chip
    const long long tiles128 = (long long)cdiv(p.M, 128) * cdiv(p.N, best);
    const bool small_m = tiles128 < 384 && p.M > 64;
    // the 144-wide tile reads 9 B fragments per A fragment from LDS at 64 rows, 4.5 at 128: the taller tile wins from
    // 256 tiles on (19x19 maps, 136 <-> 816 channels: data gradient 44.7 -> 36.3 us, projection 60.1 -> 56.4 us)
    const bool small_m144 = tiles128 < 256 && p.M > 64;
    switch (best) {
        case 32: return launch_nt<T, 2, 2>(p, part_rows, stream);
        case 48: return launch_nt<T, 2, 3>(p, part_rows, stream);
        case 96: return small_m ? launch_nt<T, 1, 6>(p, part_rows, stream) : launch_nt<T, 2, 6>(p, part_rows, stream);
        case 144: return small_m144 ? launch_nt<T, 1, 9>(p, part_rows, stream) : launch_nt<T, 2, 9>(p, part_rows, stream);
        case 192:
            // 128-row tiles of this width with the prologue's registers would spill
            return (small_m || p.pscale || p.gate) ? launch_nt<T, 1, 12>(p, part_rows, stream)
                                                   : launch_nt<T, 2, 12>(p, part_rows, stream);
        case 240: return launch_nt<T, 1, 15>(p, part_rows, stream);
        default: return small_m ? launch_nt<T, 1, 8>(p, part_rows, stream) : launch_nt<T, 2, 8>(p, part_rows, stream);
    }
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
    MIA_CHECK_ARG((long long)M * lda * (dtype == MIA_BF16 ? 2 : 4) < (1ll << 32) &&
                      (long long)N * ldb * (dtype == MIA_BF16 ? 2 : 4) < (1ll << 32),
                  "mia_gemm_nt: operands must be smaller than 4 GiB (32-bit buffer offsets)");
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
    if (dtype == MIA_BF16) {
        if (!try_skinny(p, part_rows, s)) dispatch_nt<bf16_t>(p, part_rows, s);
    } else {
        dispatch_nt<float>(p, part_rows, s);
    }
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
    // deterministic split reduction: every split stores its partial tile into its own slot of ws, and a second
    // kernel (tn_fold_kernel) adds the slots of each tile in split order and writes O. The order of the additions is
    // fixed, so the result is bit-reproducible whatever order the workgroups ran in; no zero-filled output, no float
    // atomics. The hand-over between the two kernels is a kernel boundary, the one ordering every GPU guarantees.
    int det;        // 1: partial tiles to ws (O written directly when splits == 1), 0: float atomics into O
    float* ws;      // [tiles][splits][slot], unused when splits == 1
    int splits;
    int slot;       // floats per slot = rows x cols of the largest tile
};

// Second kernel of the deterministic weight gradient: O tile = sum over the splits' partial tiles, in split order.
// A workgroup owns 256/ZL float4 columns of one tile; its ZL "z-lanes" each add every ZL-th slot (strided, so all of
// them stream at once), and the ZL partial sums are then added in lane order. The order of the additions depends only
// on (splits, ZL), never on timing. grid = (column chunks, tiles).
template <int ZL>
__global__ __launch_bounds__(256) void tn_fold_kernel(const float* __restrict__ ws, int splits, int slot, int N, int K,
                                                      int TN, int TK, int tiles_n, float* __restrict__ O, int ldo) {
    constexpr int COLS = 256 / ZL;
    __shared__ float4 red[ZL][COLS];
    const int tile = blockIdx.y;
    const int tn_i = tile % tiles_n, tk_i = tile / tiles_n;
    const int n0 = tn_i * TN, k0 = tk_i * TK;
    const int nh = N - n0 < TN ? N - n0 : TN, kw = K - k0 < TK ? K - k0 : TK;
    const int cnt4 = nh * kw / 4;
    const int col = threadIdx.x % COLS, zl = threadIdx.x / COLS;
    const int e = blockIdx.x * COLS + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < cnt4) {
        const float4* src = reinterpret_cast<const float4*>(ws + (size_t)tile * splits * slot) + e;
        const size_t zstride = (size_t)slot / 4;
#pragma unroll 4
        for (int z = zl; z < splits; z += ZL) {
            const float4 b = src[(size_t)z * zstride];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    }
    if (ZL > 1) {
        red[zl][col] = a;
        __syncthreads();
        if (zl != 0) return;
#pragma unroll
        for (int z = 1; z < ZL; ++z) {
            const float4 b = red[z][col];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
    }
    if (e < cnt4) {
        const int ln = (e * 4) / kw, lk = (e * 4) - ln * kw;
        *reinterpret_cast<float4*>(O + (size_t)(n0 + ln) * ldo + k0 + lk) = a;
    }
}

// Output tile TN (columns of G = rows of O) x TK (columns of X), shared by 4 waves laid out WR (along n) x WC (along
// k). The tile extents on offer follow the EfficientNet-B3 channel counts, so that the narrow side of a weight matrix
// (24..384 channels) is ONE tile wherever it fits: the wide operand is then streamed once instead of once per tile of
// the narrow side, and there are fewer, fuller partial tiles to reduce.
//   128x128 (2x2 waves)                                general
//   128x64  (4x1)   K <= 64                            expansion weights of the first stages (24..48 input channels)
//   64x128  (1x4)   N <= 64                            projection weights onto 24..48 channels
//   144x128 (1x4)   64 < N <= 144, K > N               the 136-channel projections
//   128x144 (4x1)   64 < K <= 144, N > K               the 136-channel expansions
// (240- and 192-wide tiles for the 232- and 384-channel layers were measured too: their 30- and 24-fragment wave
// tiles need more than 256 registers, which leaves one wave per SIMD, and at M = 12800 they ran 1.3-1.6x slower.)
template <typename T, bool PRO, int TN, int TK, int WR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNParams p) {
    using M_ = Mma<T>;
    using V_ = Vec16<T>;
    constexpr int BR = M_::BK;       // reduction rows per stage
    constexpr int VEC = V_::N;
    constexpr int LDG = TN + VEC, LDX = TK + VEC;     // LDS row strides, elements
    constexpr int CHG = (BR * (TN / VEC) + 255) / 256;  // 16-byte chunks per thread at full tile width
    constexpr int CHX = (BR * (TK / VEC) + 255) / 256;
    constexpr int WC = 4 / WR;
    constexpr int WN = TN / WR, WK = TK / WC;   // wave tile
    constexpr int MT = WN / 16, NT = WK / 16;
    static_assert(WN % 16 == 0 && WK % 16 == 0 && TN % VEC == 0 && TK % VEC == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Gs = reinterpret_cast<T*>(smem);  // [BR][LDG]
    T* Xs = Gs + BR * LDG;               // [BR][LDX]
    float* ssm = reinterpret_cast<float*>(Xs + BR * LDX);  // [2][TK] prologue scale, shift of this k tile

    const T* __restrict__ G = reinterpret_cast<const T*>(p.G);
    const T* __restrict__ X = reinterpret_cast<const T*>(p.X);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
    const int M = p.M, N = p.N, K = p.K;
    const bool has_pro = PRO && p.pscale != nullptr;
    const bool has_gate = PRO && p.gate != nullptr;

    // tiles of this wave that hold any valid output
    const int nvalid = N - (n0 + wr * WN), kvalid = K - (k0 + wc * WK);
    const int mt_cnt = nvalid <= 0 ? 0 : (nvalid >= WN ? MT : (nvalid + 15) / 16);
    const int nt_cnt = kvalid <= 0 ? 0 : (kvalid >= WK ? NT : (kvalid + 15) / 16);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The (row, 16-byte chunk) items of a stage are dealt to the threads over the VALID width of the tile, so a
    // narrow operand (24 or 40 channels) still keeps every thread loading and activating.
    const int cprg = ((N - n0 < TN ? N - n0 : TN) + VEC - 1) / VEC;
    const int cprx = ((K - k0 < TK ? K - k0 : TK) + VEC - 1) / VEC;
    int grow[CHG], gcol[CHG], xrow[CHX], xcol[CHX];
#pragma unroll
    for (int i = 0; i < CHG; ++i) {
        const int idx = tid + i * 256;
        grow[i] = idx / cprg; gcol[i] = (idx - grow[i] * cprg) * VEC;
    }
#pragma unroll
    for (int i = 0; i < CHX; ++i) {
        const int idx = tid + i * 256;
        xrow[i] = idx / cprx; xcol[i] = (idx - xrow[i] * cprx) * VEC;
    }
    if (PRO && has_pro && tid < TK) {
        const bool in = k0 + tid < K;
        ssm[tid] = in ? p.pscale[k0 + tid] : 1.f;
        ssm[TK + tid] = in ? p.pshift[k0 + tid] : 0.f;
    }

    const long long chunk_begin = (long long)blockIdx.z * p.chunks_per_split;
    long long chunk_end = chunk_begin + p.chunks_per_split;
    const long long total_chunks = ((long long)M + BR - 1) / BR;
    if (chunk_end > total_chunks) chunk_end = total_chunks;

    raw16 rg[CHG], rx[CHX];
    float pg[PRO ? CHX : 1][PRO ? VEC : 1];
    // element offsets of this thread's chunks inside a stage (the API bounds operands to < 2^31 elements)
    unsigned goff[CHG], xoff[CHX];
    int ximg[PRO ? CHX : 1], xrem[PRO ? CHX : 1];  // image index and row-in-image of each X chunk row, kept incrementally
#pragma unroll
    for (int i = 0; i < CHG; ++i) goff[i] = (unsigned)(grow[i] * p.ldg + gcol[i]);
#pragma unroll
    for (int i = 0; i < CHX; ++i) {
        xoff[i] = (unsigned)(xrow[i] * p.ldx + xcol[i]);
        if constexpr (PRO) {
            const long long mf = chunk_begin * BR + xrow[i];
            ximg[i] = (int)(mf / p.rows_per_img);
            xrem[i] = (int)(mf - (long long)ximg[i] * p.rows_per_img);
        }
    }
    // chunks are prefetched in increasing order, one call per chunk
    auto prefetch = [&](long long ch) {
        const int m0 = (int)(ch * BR);
        const T* gp = G + (size_t)m0 * p.ldg + n0;
        const T* xp = X + (size_t)m0 * p.ldx + k0;
#pragma unroll
        for (int i = 0; i < CHG; ++i) {
            rg[i] = make_uint4(0, 0, 0, 0);
            if (grow[i] < BR && m0 + grow[i] < M) rg[i] = *reinterpret_cast<const raw16*>(gp + goff[i]);
        }
#pragma unroll
        for (int i = 0; i < CHX; ++i) {
            rx[i] = make_uint4(0, 0, 0, 0);
            if (xrow[i] < BR && m0 + xrow[i] < M) {
                rx[i] = *reinterpret_cast<const raw16*>(xp + xoff[i]);
                if constexpr (PRO) if (has_gate) {
                    // the gate row rides along with the tile so that its latency is hidden like the tile's
                    const float* g = p.gate + (unsigned)(ximg[i] * K + k0 + xcol[i]);
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) {
                        const float4 g4 = *reinterpret_cast<const float4*>(g + e);
                        pg[i][e] = g4.x; pg[i][e + 1] = g4.y; pg[i][e + 2] = g4.z; pg[i][e + 3] = g4.w;
                    }
                }
            }
            if constexpr (PRO) {
                xrem[i] += BR;
                while (xrem[i] >= p.rows_per_img) { xrem[i] -= p.rows_per_img; ++ximg[i]; }
            }
        }
    };
    auto commit = [&](long long ch) {
        const long long m0 = ch * BR;
        if constexpr (PRO) if (has_pro || has_gate) {
#pragma unroll
            for (int i = 0; i < CHX; ++i) {
                if (xrow[i] < BR && m0 + xrow[i] < M) {
                    float sc[VEC], sh[VEC];
                    if (has_pro) {
#pragma unroll
                        for (int e = 0; e < VEC; e += 4) {
                            const float4 a4 = *reinterpret_cast<const float4*>(ssm + xcol[i] + e);
                            const float4 b4 = *reinterpret_cast<const float4*>(ssm + TK + xcol[i] + e);
                            sc[e] = a4.x; sc[e + 1] = a4.y; sc[e + 2] = a4.z; sc[e + 3] = a4.w;
                            sh[e] = b4.x; sh[e + 1] = b4.y; sh[e + 2] = b4.z; sh[e + 3] = b4.w;
                        }
                    }
                    rx[i] = pro16<T>(rx[i], has_pro && p.pro_act == MIA_ACT_SILU, has_pro, has_gate, sc, sh, pg[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CHG; ++i)
            if (grow[i] < BR) *reinterpret_cast<raw16*>(Gs + grow[i] * LDG + gcol[i]) = rg[i];
#pragma unroll
        for (int i = 0; i < CHX; ++i)
            if (xrow[i] < BR) *reinterpret_cast<raw16*>(Xs + xrow[i] * LDX + xcol[i]) = rx[i];
    };

    if (chunk_begin < chunk_end) prefetch(chunk_begin);
    for (long long ch = chunk_begin; ch < chunk_end; ++ch) {
        __syncthreads();
        commit(ch);
        __syncthreads();
        if (ch + 1 < chunk_end) prefetch(ch + 1);  // in flight during the MFMAs below
        // The shorter side of the wave tile is held in registers per reduction step, the longer side is read one
        // fragment at a time (a 15-fragment side kept whole would cost 60 registers).
        if (mt_cnt == MT && nt_cnt == NT) {
            // full wave tile: no per-MFMA tests (left in, each compiles to a scalar branch around its MFMA and the
            // branches cost more than the MFMAs)
#pragma unroll
            for (int kk = 0; kk < BR / M_::KSTEP; ++kk) {
                if constexpr (MT <= NT) {
                    typename M_::Frag a[MT];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        a[i] = M_::load_t(Gs + kk * M_::KSTEP * LDG + wr * WN + i * 16, LDG, lane);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const typename M_::Frag b = M_::load_t(Xs + kk * M_::KSTEP * LDX + wc * WK + j * 16, LDX, lane);
#pragma unroll
                        for (int i = 0; i < MT; ++i) acc[i][j] = M_::mma(a[i], b, acc[i][j]);
                    }
                } else {
                    typename M_::Frag b[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        b[j] = M_::load_t(Xs + kk * M_::KSTEP * LDX + wc * WK + j * 16, LDX, lane);
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const typename M_::Frag a = M_::load_t(Gs + kk * M_::KSTEP * LDG + wr * WN + i * 16, LDG, lane);
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = M_::mma(a, b[j], acc[i][j]);
                    }
                }
            }
        } else if (mt_cnt > 0 && nt_cnt > 0) {
#pragma unroll
            for (int kk = 0; kk < BR / M_::KSTEP; ++kk) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (i < mt_cnt) {
                        const typename M_::Frag a = M_::load_t(Gs + kk * M_::KSTEP * LDG + wr * WN + i * 16, LDG, lane);
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            if (j < nt_cnt) {
                                const typename M_::Frag b = M_::load_t(Xs + kk * M_::KSTEP * LDX + wc * WK + j * 16, LDX, lane);
                                acc[i][j] = M_::mma(a, b, acc[i][j]);
                            }
                    }
                }
            }
        }
    }

    const int lrow = (lane >> 4) * 4, lcol = lane & 15;
    if (!p.det) {
        // accumulate into O with float atomics (callers without a workspace; the order of the additions varies)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (i < mt_cnt && j < nt_cnt) {
                    const int gk = k0 + wc * WK + j * 16 + lcol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gn = n0 + wr * WN + i * 16 + lrow + r;
                        if (gn < N && gk < K) atomicAdd(p.O + (size_t)gn * p.ldo + gk, acc[i][j][r]);
                    }
                }
            }
        return;
    }
    // ---- deterministic path: this split's partial tile to its slot (or straight to O when there is one split) ----
    const int nh = N - n0 < TN ? N - n0 : TN, kw = K - k0 < TK ? K - k0 : TK;   // valid extent of this tile
    const int S = p.splits;
    const int tile = blockIdx.y * gridDim.x + blockIdx.x;
    float* dst = S == 1 ? p.O + (size_t)n0 * p.ldo + k0 : p.ws + ((size_t)tile * S + blockIdx.z) * p.slot;
    const int dld = S == 1 ? p.ldo : kw;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (i < mt_cnt && j < nt_cnt) {
                const int lk = wc * WK + j * 16 + lcol;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ln = wr * WN + i * 16 + lrow + r;
                    if (ln < nh && lk < kw) dst[(size_t)ln * dld + lk] = acc[i][j][r];
                }
            }
        }
}

// The tile extents for a product (see gemm_tn_kernel): index into TN_SHAPES.
struct TnShape { int tn, tk, wr; };
static const TnShape TN_SHAPES[] = {{128, 128, 2}, {128, 64, 4}, {64, 128, 1}, {144, 128, 1}, {128, 144, 4}};
static int tn_pick_shape(int dtype, int N, int K) {
    if (K <= 64) return 1;
    if (N <= 64) return 2;
    if (dtype != MIA_BF16) return 0;
    // the narrow side in one tile where a shape fits it; the other side stays 128 wide
    if (N <= K && N > 128 && N <= 144) return 3;
    if (K < N && K > 128 && K <= 144) return 4;
    return 0;
}

// How the reduction over M is split, shared by the launch and by the workspace query.
struct TnPlan {
    int shape;   // index into TN_SHAPES
    int chunks_per_split, splits, tiles, slot;
    long long ws_floats;
};
static TnPlan tn_plan(int dtype, int M, int N, int K, bool pro) {
    TnPlan t;
    const int BR = dtype == MIA_BF16 ? 64 : 32;
    t.shape = tn_pick_shape(dtype, N, K);
    const int tn = TN_SHAPES[t.shape].tn, tk = TN_SHAPES[t.shape].tk;
    t.tiles = cdiv(N, tn) * cdiv(K, tk);
    const long long chunks = ((long long)M + BR - 1) / BR;
    // aim for ~512 blocks (two per CU), but keep splits long enough to amortise their partial-tile epilogue: at least
    // 16 reduction chunks per split for the plain variant, 4 for the prologue variant, whose chunks carry the
    // activation work and prefer the parallelism
    const int min_chunks = pro ? 4 : 16;
    long long splits = 512 / t.tiles;
    if (splits < 1) splits = 1;
    if (splits > (chunks + min_chunks - 1) / min_chunks) splits = (chunks + min_chunks - 1) / min_chunks;
    if (splits < 1) splits = 1;
    t.chunks_per_split = (int)((chunks + splits - 1) / splits);
    t.splits = (int)((chunks + t.chunks_per_split - 1) / t.chunks_per_split);
    t.slot = (N < tn ? N : tn) * (K < tk ? K : tk);
    t.ws_floats = t.splits == 1 ? 0 : (long long)t.tiles * t.splits * t.slot;
    return t;
}

extern "C" long long mia_gemm_tn_ws_floats(int dtype, int M, int N, int K, int has_prologue) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    return tn_plan(dtype, M, N, K, has_prologue != 0).ws_floats;
}

extern "C" int mia_gemm_tn(int dtype, const void* G, int ldg, const void* X, int ldx, float* O, int ldo, int M, int N,
                           int K, const float* pscale, const float* pshift, int pro_act, const float* gate,
                           int rows_per_img, float* ws, long long ws_floats, void* stream) {
    const int vec = dtype == MIA_BF16 ? 8 : 4;
    MIA_CHECK_ARG(dtype == MIA_F32 || dtype == MIA_BF16, "mia_gemm_tn: bad dtype %d", dtype);
    MIA_CHECK_ARG(G && X && O, "mia_gemm_tn: null operand");
    MIA_CHECK_ARG(M > 0 && N > 0 && K > 0, "mia_gemm_tn: bad shape M=%d N=%d K=%d", M, N, K);
    MIA_CHECK_ARG(N % vec == 0 && K % vec == 0 && ldg % vec == 0 && ldx % vec == 0 && ldg >= N && ldx >= K,
                  "mia_gemm_tn: N=%d K=%d ldg=%d ldx=%d must be multiples of %d", N, K, ldg, ldx, vec);
    MIA_CHECK_ARG(ldo >= K, "mia_gemm_tn: ldo=%d < K=%d", ldo, K);
    MIA_CHECK_ARG((long long)M * ldg < (1ll << 31) && (long long)M * ldx < (1ll << 31),
                  "mia_gemm_tn: operands must hold fewer than 2^31 elements");
    MIA_CHECK_ARG(((uintptr_t)G % 16) == 0 && ((uintptr_t)X % 16) == 0, "mia_gemm_tn: operands must be 16-byte aligned");
    MIA_CHECK_ARG(!(pscale == nullptr) == !(pshift == nullptr), "mia_gemm_tn: pscale/pshift must come together");
    MIA_CHECK_ARG(!gate || rows_per_img > 0, "mia_gemm_tn: gate needs rows_per_img > 0");
    GemmTNParams p;
    p.G = G; p.X = X; p.O = O; p.M = M; p.N = N; p.K = K; p.ldg = ldg; p.ldx = ldx; p.ldo = ldo;
    p.pscale = pscale; p.pshift = pshift; p.gate = gate; p.rows_per_img = rows_per_img > 0 ? rows_per_img : 1;
    p.pro_act = pro_act;
    const bool pro = pscale || gate;
    const TnPlan t = tn_plan(dtype, M, N, K, pro);
    p.chunks_per_split = t.chunks_per_split;
    p.det = 0; p.ws = nullptr; p.splits = t.splits; p.slot = t.slot;
    if (ws != nullptr || ws_floats < 0) {
        // deterministic mode (ws_floats < 0 selects it for products that need no workspace): O is overwritten
        MIA_CHECK_ARG(t.ws_floats == 0 || (ws != nullptr && ws_floats >= t.ws_floats),
                      "mia_gemm_tn: workspace of %lld floats needed (mia_gemm_tn_ws_floats), %lld given", t.ws_floats,
                      ws_floats);
        MIA_CHECK_ARG(ldo % 4 == 0 && ((uintptr_t)O % 16) == 0, "mia_gemm_tn: O must be 16-byte aligned with ldo %% 4 == 0");
        MIA_CHECK_ARG(((uintptr_t)ws % 16) == 0, "mia_gemm_tn: the workspace must be 16-byte aligned (it is read as float4)");
        p.det = 1; p.ws = ws;
    }
    const long long splits = t.splits;
    const TnShape sh = TN_SHAPES[t.shape];
    dim3 grid(cdiv(N, sh.tn), cdiv(K, sh.tk), (unsigned)splits);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MIA_TN_GO(T_, TN_, TK_, WR_)                                                                             \
    do {                                                                                                         \
        const int esz = (int)sizeof(T_), vec = 16 / esz, br = Mma<T_>::BK;                                       \
        const int lds = br * ((TN_ + vec) + (TK_ + vec)) * esz + 2 * TK_ * 4;                                    \
        if (pro) MIA_LAUNCH((gemm_tn_kernel<T_, true, TN_, TK_, WR_>), grid, dim3(256), lds, s, p);      \
        else MIA_LAUNCH((gemm_tn_kernel<T_, false, TN_, TK_, WR_>), grid, dim3(256), lds, s, p);         \
    } while (0)
    if (dtype == MIA_BF16) {
        switch (t.shape) {
            case 1: MIA_TN_GO(bf16_t, 128, 64, 4); break;
            case 2: MIA_TN_GO(bf16_t, 64, 128, 1); break;
            case 3: MIA_TN_GO(bf16_t, 144, 128, 1); break;
            case 4: MIA_TN_GO(bf16_t, 128, 144, 4); break;
            default: MIA_TN_GO(bf16_t, 128, 128, 2); break;
        }
    } else {
        switch (t.shape) {
            case 1: MIA_TN_GO(float, 128, 64, 4); break;
            case 2: MIA_TN_GO(float, 64, 128, 1); break;
            default: MIA_TN_GO(float, 128, 128, 2); break;
        }
    }
#undef MIA_TN_GO
    if (p.det && t.splits > 1) {
        // z-lanes: enough workgroups to fill the chip when the tiles are few and the splits many
        const int cnt4 = t.slot / 4;
        int zl = 1;
        while (zl < 16 && zl * 2 <= t.splits / 4 && (long long)t.tiles * cdiv(cnt4, 256 / zl) < 512) zl *= 2;
        dim3 fgrid(cdiv(cnt4, 256 / zl), t.tiles);
        const int tiles_n = cdiv(N, sh.tn);
#define MIA_TN_FOLD(ZL_) MIA_LAUNCH((tn_fold_kernel<ZL_>), fgrid, dim3(256), 0, s, ws, t.splits, t.slot, N, K, \
                                            sh.tn, sh.tk, tiles_n, O, ldo)
        if (zl == 1) MIA_TN_FOLD(1);
        else if (zl == 2) MIA_TN_FOLD(2);
        else if (zl == 4) MIA_TN_FOLD(4);
        else if (zl == 8) MIA_TN_FOLD(8);
        else MIA_TN_FOLD(16);
#undef MIA_TN_FOLD
    }
    MIA_CHECK_LAUNCH("mia_gemm_tn");
    return 0;
}

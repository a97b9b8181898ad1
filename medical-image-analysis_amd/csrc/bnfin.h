// Train-mode BatchNorm finalize fused into the kernel that produces the partial sums (gfx950).
//
// A producer (1x1-conv GEMM, depthwise conv, depthwise data gradient) leaves one row of per-channel partial sums per
// workgroup in a slab part[P][2][C]. Turning the slab into the per-channel vectors the consumers need used to be a
// kernel of its own (mia_bn_finalize_fwd / _bwd): 155 launches a training step, each a few microseconds of work
// behind a kernel boundary. Here the workgroups that share a channel range (a column tile of the GEMM, a channel
// block of the depthwise kernels) count their arrivals on a ticket, and the one that arrives last reduces that
// range's P rows -- in row order, in double precision, so the result does not depend on which workgroup it was --
// and writes the vectors.
//
// Hand-over between workgroups, which may sit on different XCDs whose L2 caches are not coherent with each other for
// ordinary accesses:
//   * partial sums are stored with agent scope (written through to where every XCD sees them);
//   * every thread waits for its stores to complete, the workgroup meets at a barrier, then ONE thread draws the
//     ticket with an agent-scope read-modify-write. No cache write-back is requested: the tensor tile the workgroup
//     has written is not part of the hand-over, and a write-back + invalidate per workgroup costs more than the
//     launch it saves (measured on the slice filter in preprocess.hip: 30 us against 17 us at 512 workgroups);
//   * the last workgroup passes an agent-scope acquire fence (invalidates its view of the caches) before it reads
//     the other workgroups' rows.
// Tickets are caller-provided words that are zero on entry; the last arriver puts its ticket back to zero, so a
// ring of them can be reused launch after launch on one stream without a clearing pass.
#pragma once
#include "common.h"

struct BnFinFwd {
    unsigned* tickets;        // one word per channel range, zero on entry; null = no fused finalize
    const float* gamma;       // nullable (1)
    const float* beta;        // nullable (0)
    float* rmean;             // running statistics, nullable
    float* rvar;
    long long* nbt;           // num_batches_tracked, nullable
    float momentum, eps;
    double count;             // rows the statistics are taken over
    const float* conv_bias;   // bias added before the normalisation but not stored with the tensor, nullable
    float* scale;             // outputs [C]
    float* shift;
    float* mean;
    float* rstd;
};

struct BnFinBwd {
    unsigned* tickets;
    const float* gamma;       // nullable (1)
    const float* mean;
    const float* rstd;
    int training;
    double count;
    float* dgamma;            // nullable
    float* dbeta;             // nullable
    float* ca;
    float* cb;
    float* cd;
    float* dconv_bias;        // nullable
};

#define MIA_FIN_LDS_BYTES (64 + 256 * 8 * 8 + 2 * 256 * 8)   // flag, per-thread partials, per-column sums

__device__ __forceinline__ void store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by EVERY thread of the workgroup after its stores to the partial slab. True, for all of them, in the
// workgroup that arrives last of `expected`; that workgroup has also passed the acquire fence.
__device__ __forceinline__ bool mia_last_arriver(unsigned* ticket, unsigned expected, int* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t + 1 == expected;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag_lds = last ? 1 : 0;
    }
    __syncthreads();
    const bool last = *flag_lds != 0;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return last;
}

// Column sums of rows 0..P-1 of part[P][2][C] over columns [col0, col0 + ncols), ncols <= 256, by one workgroup of 256
// threads: out[c] = sum of part[r][0][col0+c], out[ncols + c] = sum of part[r][1][col0+c], as doubles in LDS.
// Threads are dealt (column quad, row group); a thread adds its rows in increasing order, the row groups are then
// added in increasing order: the order of additions is a function of (P, ncols) alone.
// lds: MIA_FIN_LDS_BYTES, 16-byte aligned; its first 64 bytes are left alone (the arrival flag lives there).
__device__ __forceinline__ void mia_slab_colsum(const float* __restrict__ part, int P, int C, int col0, int ncols,
                                                char* lds, double*& out) {
    double* scratch = reinterpret_cast<double*>(lds + 64);    // [256][8]
    out = scratch + 256 * 8;                                  // [2][ncols]
    const int tid = threadIdx.x;
    const bool vec = ((C & 3) == 0) && ((col0 & 3) == 0) && ((ncols & 3) == 0);
    if (vec) {
        const int quads = ncols >> 2;
        const int RG = 256 / quads;
        const int q = tid % quads, rg = tid / quads;
        double s[4] = {0.0, 0.0, 0.0, 0.0}, sq[4] = {0.0, 0.0, 0.0, 0.0};
        if (rg < RG) {
            const float* base = part + col0 + q * 4;
            int r = rg;
            // four rows (eight 16-byte loads) in flight per thread
            for (; r + 3 * RG < P; r += 4 * RG) {
                float4 a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const float4*>(base + ((size_t)(r + u * RG) * 2 + 0) * C);
                    b[u] = *reinterpret_cast<const float4*>(base + ((size_t)(r + u * RG) * 2 + 1) * C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s[0] += (double)a[u].x; s[1] += (double)a[u].y; s[2] += (double)a[u].z; s[3] += (double)a[u].w;
                    sq[0] += (double)b[u].x; sq[1] += (double)b[u].y; sq[2] += (double)b[u].z; sq[3] += (double)b[u].w;
                }
            }
            for (; r < P; r += RG) {
                const float4 a = *reinterpret_cast<const float4*>(base + ((size_t)r * 2 + 0) * C);
                const float4 b = *reinterpret_cast<const float4*>(base + ((size_t)r * 2 + 1) * C);
                s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
                sq[0] += (double)b.x; sq[1] += (double)b.y; sq[2] += (double)b.z; sq[3] += (double)b.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { scratch[tid * 8 + e] = s[e]; scratch[tid * 8 + 4 + e] = sq[e]; }
        __syncthreads();
        for (int c = tid; c < ncols; c += 256) {
            const int qq = c >> 2, e = c & 3;
            double a = 0.0, b = 0.0;
            for (int g = 0; g < RG; ++g) {
                a += scratch[(g * quads + qq) * 8 + e];
                b += scratch[(g * quads + qq) * 8 + 4 + e];
            }
            out[c] = a;
            out[ncols + c] = b;
        }
    } else {
        for (int c = tid; c < ncols; c += 256) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < P; ++r) {
                a += (double)part[((size_t)r * 2 + 0) * C + col0 + c];
                b += (double)part[((size_t)r * 2 + 1) * C + col0 + c];
            }
            out[c] = a;
            out[ncols + c] = b;
        }
    }
    __syncthreads();
}

// sums -> scale / shift / mean / rstd and the running-statistics update for columns [col0, col0 + ncols)
// (the arithmetic of mia_bn_finalize_fwd in training mode). bump_nbt: this range also counts the batch.
__device__ __forceinline__ void mia_bn_fin_fwd_cols(const BnFinFwd& f, const double* sums, int col0, int ncols,
                                                    bool bump_nbt) {
    for (int i = threadIdx.x; i < ncols; i += 256) {
        const int c = col0 + i;
        const float cbias = f.conv_bias ? f.conv_bias[c] : 0.f;
        const double m = sums[i] / f.count;
        double v = sums[ncols + i] / f.count - m * m;
        if (v < 0.0) v = 0.0;
        const float mu = (float)m, var = (float)v;
        if (f.rmean) {
            const double unb = f.count > 1.0 ? v * f.count / (f.count - 1.0) : v;
            f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * (mu + cbias);
            f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unb;
        }
        const float rs = 1.0f / sqrtf(var + f.eps);
        const float g = f.gamma ? f.gamma[c] : 1.f, b = f.beta ? f.beta[c] : 0.f;
        f.scale[c] = g * rs;
        f.shift[c] = b - mu * g * rs;
        f.mean[c] = mu;
        f.rstd[c] = rs;
    }
    if (bump_nbt && threadIdx.x == 0 && f.nbt) *f.nbt += 1;
}

// sums of (g, g*xhat) -> dgamma, dbeta and the coefficients of dy = ca*g + cb*y + cd (mia_bn_finalize_bwd)
__device__ __forceinline__ void mia_bn_fin_bwd_cols(const BnFinBwd& f, const double* sums, int col0, int ncols) {
    for (int i = threadIdx.x; i < ncols; i += 256) {
        const int c = col0 + i;
        const double s = sums[i], q = sums[ncols + i];
        const double g = f.gamma ? (double)f.gamma[c] : 1.0;
        const double rs = f.rstd[c], mu = f.mean[c];
        if (f.dgamma) f.dgamma[c] = (float)q;
        if (f.dbeta) f.dbeta[c] = (float)s;
        if (f.dconv_bias) f.dconv_bias[c] = f.training ? 0.f : (float)(g * rs * s);
        if (f.training) {
            f.ca[c] = (float)(g * rs);
            f.cb[c] = (float)(-g * rs * rs * q / f.count);
            f.cd[c] = (float)(g * rs * (-s / f.count + mu * rs * q / f.count));
        } else {
            f.ca[c] = (float)(g * rs);
            f.cb[c] = 0.f;
            f.cd[c] = 0.f;
        }
    }
}

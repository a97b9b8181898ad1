// Classification-head kernels (gfx950): CBAM channel + spatial attention, multi-head self-attention core,
// LayerNorm/dropout, cross-entropy. The feature map is NHWC [B, HW, C] in the compute dtype T; everything
// internal is fp32. The four Linear layers of the attention head and the MLP run on the MFMA GEMMs of gemm.hip.
//
// Reductions use wavefront shuffles (DPP) for per-pixel channel pooling and LDS for per-image pooling.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// counter-based dropout mask: keep iff hash(seed, idx) >= p.  Recomputed, never stored.
// ------------------------------------------------------------------------------------------------
// Effective seed of a launch: the host's seed plus an optional counter that lives in device memory. A captured HIP
// graph replays the same kernel arguments every time; the counter (bumped once per replay by a kernel of the graph)
// is what makes each replay draw new masks, while forward and backward of one replay still agree.
__device__ __forceinline__ unsigned long long seed_with_offset(unsigned long long seed, const unsigned long long* off) {
    return off ? seed + off[0] * 0x9E3779B97F4A7C15ull : seed;
}

__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx, float p) {
    if (p <= 0.f) return 1.f;
    unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.f;
}

// ------------------------------------------------------------------------------------------------
// CBAM channel gate: per-image average and max over H*W (with the arg-max pixel for the backward pass)
// grid (B, ceil(C/256)), one thread per channel, rows streamed with channel-contiguous (coalesced) loads
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cbam_channel_pool_kernel(const T* x, int HW, int C, float* avg, float* mx,
                                                                int* amax) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const T* p = x + (size_t)b * HW * C + c;
    float s = 0.f, m = -INFINITY;
    int am = 0;
    for (int r = 0; r < HW; ++r) {
        const float v = to_f32<T>(p[(size_t)r * C]);
        s += v;
        if (v > m) { m = v; am = r; }
    }
    avg[(size_t)b * C + c] = s / (float)HW;
    mx[(size_t)b * C + c] = m;
    amax[(size_t)b * C + c] = am;
}

// The same pooling for large maps (decoder levels: up to 150x150 pixels, few channels): one workgroup per
// (image, row split), 256 threads = row lanes x 4-channel lanes, then a per-image combine that keeps the FIRST
// arg-max in row order (what max_pool2d's backward uses).
template <typename T>
__global__ __launch_bounds__(256) void cbam_channel_pool_split_kernel(const T* x, int HW, int C, int splits,
                                                                      float* psum, float* pmax, int* parg) {
    __shared__ f4 rs[256];
    __shared__ f4 rm[256];
    __shared__ int ra[256][4];
    const int tid = threadIdx.x, cvn = C >> 2, npl = 256 / cvn;
    const int cvl = tid % cvn, pl = tid / cvn;
    const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int per = (HW + splits - 1) / splits;
    const int r0 = sp * per, r1 = (r0 + per) < HW ? (r0 + per) : HW;
    f4 s = {0.f, 0.f, 0.f, 0.f}, m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int a0 = r0, a1 = r0, a2 = r0, a3 = r0;
    if (pl < npl)
        for (int r = r0 + pl; r < r1; r += npl) {
            const f4 v = load4<T>(x + ((size_t)b * HW + r) * C + cvl * 4);
            s = {s.x + v.x, s.y + v.y, s.z + v.z, s.w + v.w};
            if (v.x > m.x) { m.x = v.x; a0 = r; }
            if (v.y > m.y) { m.y = v.y; a1 = r; }
            if (v.z > m.z) { m.z = v.z; a2 = r; }
            if (v.w > m.w) { m.w = v.w; a3 = r; }
        }
    rs[tid] = s; rm[tid] = m;
    ra[tid][0] = a0; ra[tid][1] = a1; ra[tid][2] = a2; ra[tid][3] = a3;
    __syncthreads();
    if (tid < cvn) {
        for (int q = 1; q < npl; ++q) {
            const int o = q * cvn + tid;
            const f4 os = rs[o], om = rm[o];
            s = {s.x + os.x, s.y + os.y, s.z + os.z, s.w + os.w};
            if (om.x > m.x || (om.x == m.x && ra[o][0] < a0)) { m.x = om.x; a0 = ra[o][0]; }
            if (om.y > m.y || (om.y == m.y && ra[o][1] < a1)) { m.y = om.y; a1 = ra[o][1]; }
            if (om.z > m.z || (om.z == m.z && ra[o][2] < a2)) { m.z = om.z; a2 = ra[o][2]; }
            if (om.w > m.w || (om.w == m.w && ra[o][3] < a3)) { m.w = om.w; a3 = ra[o][3]; }
        }
        const size_t o = ((size_t)b * splits + sp) * C + tid * 4;
        store4<float>(psum + o, s);
        store4<float>(pmax + o, m);
        parg[o] = a0; parg[o + 1] = a1; parg[o + 2] = a2; parg[o + 3] = a3;
    }
}

__global__ __launch_bounds__(256) void cbam_channel_pool_combine_kernel(const float* psum, const float* pmax,
                                                                        const int* parg, int HW, int C, int splits,
                                                                        float* avg, float* mx, int* amax) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f, m = -INFINITY;
    int am = 0;
    for (int sp = 0; sp < splits; ++sp) {   // splits are in row order: strict > keeps the first arg-max
        const size_t o = ((size_t)b * splits + sp) * C + c;
        s += psum[o];
        if (pmax[o] > m) { m = pmax[o]; am = parg[o]; }
    }
    avg[(size_t)b * C + c] = s / (float)HW;
    mx[(size_t)b * C + c] = m;
    amax[(size_t)b * C + c] = am;
}

// shared MLP C -> HID -> C on both pooled vectors, summed, sigmoid. One workgroup per image.
__global__ __launch_bounds__(256) void cbam_channel_mlp_kernel(const float* avg, const float* mx, int C, int HID,
                                                               const float* W1, const float* b1, const float* W2,
                                                               const float* b2, float* ha, float* hm, float* cscale) {
    extern __shared__ float sm[];
    float* sa = sm;           // [C]
    float* sx = sm + C;       // [C]
    float* hs = sm + 2 * C;   // [HID] (ha + hm)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        sa[c] = avg[(size_t)b * C + c];
        sx[c] = mx[(size_t)b * C + c];
    }
    __syncthreads();
    for (int j = wave; j < HID; j += 4) {
        float da = 0.f, dm = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float w = W1[(size_t)j * C + c];
            da += w * sa[c];
            dm += w * sx[c];
        }
        da = wave_sum(da);
        dm = wave_sum(dm);
        if (lane == 0) {
            const float a = fmaxf(da + b1[j], 0.f), m = fmaxf(dm + b1[j], 0.f);
            ha[(size_t)b * HID + j] = a;
            hm[(size_t)b * HID + j] = m;
            hs[j] = a + m;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float d = 2.f * b2[c];
        for (int j = 0; j < HID; ++j) d += W2[(size_t)c * HID + j] * hs[j];
        cscale[(size_t)b * C + c] = sigmoidf_(d);
    }
}

// channel pool of x*cscale per pixel: comp[b][p] = (max_c, mean_c), arg-max channel kept. One wave per pixel.
template <typename T>
__global__ __launch_bounds__(256) void cbam_spatial_pool_kernel(const T* x, const float* cscale, int B, int HW, int C,
                                                                float* comp, int* cmax) {
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long long)B * HW) return;
    const int b = (int)(pix / HW);
    const T* p = x + (size_t)pix * C;
    const float* cs = cscale + (size_t)b * C;
    float s = 0.f, m = -INFINITY;
    int am = 0;
    for (int c = lane; c < C; c += 64) {
        const float v = to_f32<T>(p[c]) * cs[c];
        s += v;
        if (v > m) { m = v; am = c; }
    }
    s = wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oa = __shfl_xor(am, o, 64);
        if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    if (lane == 0) {
        comp[(size_t)pix * 2 + 0] = m;
        comp[(size_t)pix * 2 + 1] = s / (float)C;
        cmax[pix] = am;
    }
}

// 7x7 conv (2 -> 1, pad 3, no bias) over comp [B,H,W,2]; partial sums of pre and pre^2 for BatchNorm(1)
__global__ __launch_bounds__(256) void cbam_spatial_conv_kernel(const float* comp, int B, int H, int W, const float* w7,
                                                                float* pre, float* part) {
    __shared__ float ws[98];
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    if (tid < 98) ws[tid] = w7[tid];  // [ch][kh][kw], ch 0 = max, 1 = mean
    __syncthreads();
    const long long total = (long long)B * H * W;
    float s = 0.f, q = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
        const int xw = (int)(i % W), yh = (int)((i / W) % H);
        const long long b = i / ((long long)W * H);
        float acc = 0.f;
        for (int kh = 0; kh < 7; ++kh) {
            const int ih = yh + kh - 3;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const int iw = xw + kw - 3;
                if (iw < 0 || iw >= W) continue;
                const float* cp = comp + ((size_t)(b * H + ih) * W + iw) * 2;
                acc += cp[0] * ws[kh * 7 + kw] + cp[1] * ws[49 + kh * 7 + kw];
            }
        }
        pre[i] = acc;
        s += acc;
        q += acc * acc;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) {
        part[(size_t)blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[(size_t)blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

// x2 = x * cscale[b,c] * sigmoid(pre*s + t)[b,p]; also stores the spatial gate
template <typename T>
__global__ __launch_bounds__(256) void cbam_apply_kernel(const T* x, const float* cscale, const float* pre,
                                                         const float* bnst, int B, int HW, int C, T* out, float* sgate) {
    const long long nvec = (long long)B * HW * C / 4;
    const float s = bnst[0], t = bnst[1];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const size_t idx = (size_t)i * 4;
        const long long pix = idx / C;
        const int c = (int)(idx % C), b = (int)(pix / HW);
        const float g = sigmoidf_(pre[pix] * s + t);
        if (c == 0) sgate[pix] = g;
        const f4 v = load4<T>(x + idx), cs = load4<float>(cscale + (size_t)b * C + c);
        store4<T>(out + idx, f4{v.x * cs.x * g, v.y * cs.y * g, v.z * cs.z * g, v.w * cs.w * g});
    }
}

// ---- CBAM backward ----
// dsg[b,p] = sum_c dx2 * x * cscale. One wave per pixel.
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd_dsgate_kernel(const T* dx2, const T* x, const float* cscale, int B,
                                                              int HW, int C, float* dsg) {
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (long long)B * HW) return;
    const int b = (int)(pix / HW);
    float s = 0.f;
    for (int c = lane; c < C; c += 64)
        s += to_f32<T>(dx2[(size_t)pix * C + c]) * to_f32<T>(x[(size_t)pix * C + c]) * cscale[(size_t)b * C + c];
    s = wave_sum(s);
    if (lane == 0) dsg[pix] = s;
}

// BatchNorm(1)+sigmoid backward over all N = B*HW values. One workgroup (N is small: 12,800 at batch 128).
// in: dsg, sgate, pre, st = {scale, shift, mean, rstd}, gamma. out: dpre (gradient of the conv output), dgamma, dbeta
__global__ __launch_bounds__(1024) void cbam_bwd_spatial_bn_kernel(const float* dsg, const float* sgate,
                                                                   const float* pre, const float* st,
                                                                   const float* gamma, int training, long long N,
                                                                   float* dpre, float* dgamma, float* dbeta) {
    __shared__ float red[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float mean = st[2], rstd = st[3];
    float s = 0.f, q = 0.f;
    for (long long i = tid; i < N; i += 1024) {
        const float sg = sgate[i];
        const float g = dsg[i] * sg * (1.f - sg);
        s += g;
        q += g * ((pre[i] - mean) * rstd);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    float S = 0.f, Q = 0.f;
    for (int k = 0; k < 16; ++k) { S += red[0][k]; Q += red[1][k]; }
    const float ga = gamma[0];
    const float invn = 1.0f / (float)N;
    for (long long i = tid; i < N; i += 1024) {
        const float sg = sgate[i];
        const float g = dsg[i] * sg * (1.f - sg);
        const float xh = (pre[i] - mean) * rstd;
        dpre[i] = training ? ga * rstd * (g - S * invn - xh * Q * invn) : ga * rstd * g;
    }
    if (tid == 0) { dgamma[0] = Q; dbeta[0] = S; }
}

// The same in two launches for large N: per-workgroup partial sums, then every workgroup folds the partials and
// applies the formula to its own slice.
__global__ __launch_bounds__(256) void cbam_bwd_spatial_bn_part_kernel(const float* dsg, const float* sgate,
                                                                       const float* pre, const float* st, long long N,
                                                                       float* part) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const float mean = st[2], rstd = st[3];
    float s = 0.f, q = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < N; i += (long long)gridDim.x * 256) {
        const float sg = sgate[i];
        const float g = dsg[i] * sg * (1.f - sg);
        s += g;
        q += g * ((pre[i] - mean) * rstd);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) {
        part[blockIdx.x * 2 + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[blockIdx.x * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(256) void cbam_bwd_spatial_bn_apply_kernel(const float* dsg, const float* sgate,
                                                                        const float* pre, const float* st,
                                                                        const float* gamma, int training, long long N,
                                                                        const float* part, int nparts, float* dpre,
                                                                        float* dgamma, float* dbeta) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    float s = 0.f, q = 0.f;
    for (int k = tid; k < nparts; k += 256) { s += part[k * 2 + 0]; q += part[k * 2 + 1]; }
    s = wave_sum(s);
    q = wave_sum(q);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
    __syncthreads();
    const float S = red[0][0] + red[0][1] + red[0][2] + red[0][3], Q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float mean = st[2], rstd = st[3], ga = gamma[0], invn = 1.0f / (float)N;
    for (long long i = (long long)blockIdx.x * 256 + tid; i < N; i += (long long)gridDim.x * 256) {
        const float sg = sgate[i];
        const float g = dsg[i] * sg * (1.f - sg);
        const float xh = (pre[i] - mean) * rstd;
        dpre[i] = training ? ga * rstd * (g - S * invn - xh * Q * invn) : ga * rstd * g;
    }
    if (blockIdx.x == 0 && tid == 0) { dgamma[0] = Q; dbeta[0] = S; }
}

// weight gradient of the 7x7 conv: dw[ch][kh][kw] = sum_{b,y,x} dpre[b,y,x] * comp[b,y+kh-3,x+kw-3,ch].
// grid = (98 taps, pixel chunks); chunk partials land in out[chunk][98].
__global__ __launch_bounds__(256) void cbam_bwd_conv_weight_kernel(const float* dpre, const float* comp, int B, int H,
                                                                   int W, float* out) {
    __shared__ float red[4];
    const int tap = blockIdx.x, ch = tap / 49, kh = (tap % 49) / 7, kw = tap % 7;
    const long long total = (long long)B * H * W;
    const long long per = (total + gridDim.y - 1) / gridDim.y;
    const long long i0 = (long long)blockIdx.y * per, i1 = (i0 + per) < total ? (i0 + per) : total;
    float s = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const int xw = (int)(i % W), yh = (int)((i / W) % H);
        const long long b = i / ((long long)W * H);
        const int ih = yh + kh - 3, iw = xw + kw - 3;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) s += dpre[i] * comp[((size_t)(b * H + ih) * W + iw) * 2 + ch];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * 98 + tap] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(128) void cbam_bwd_conv_weight_fold_kernel(const float* part, int chunks, float* dw7) {
    const int tap = threadIdx.x;
    if (tap >= 98) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * 98 + tap];
    dw7[tap] = s;
}

// dcomp[b,p,ch] = sum_taps dpre[b, p - tap] * w[ch][tap]  (data gradient of the 7x7 conv)
__global__ __launch_bounds__(256) void cbam_bwd_conv_data_kernel(const float* dpre, int B, int H, int W, const float* w7,
                                                                 float* dcomp) {
    __shared__ float ws[98];
    if (threadIdx.x < 98) ws[threadIdx.x] = w7[threadIdx.x];
    __syncthreads();
    const long long total = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xw = (int)(i % W), yh = (int)((i / W) % H);
        const long long b = i / ((long long)W * H);
        float a0 = 0.f, a1 = 0.f;
        for (int kh = 0; kh < 7; ++kh) {
            const int oh = yh - kh + 3;
            if (oh < 0 || oh >= H) continue;
            for (int kw = 0; kw < 7; ++kw) {
                const int ow = xw - kw + 3;
                if (ow < 0 || ow >= W) continue;
                const float g = dpre[(size_t)(b * H + oh) * W + ow];
                a0 += g * ws[kh * 7 + kw];
                a1 += g * ws[49 + kh * 7 + kw];
            }
        }
        dcomp[(size_t)i * 2 + 0] = a0;
        dcomp[(size_t)i * 2 + 1] = a1;
    }
}

// pass 1 of the input gradient: dx1 = dx2*sg + dcomp_mean/C + dcomp_max*[c == cmax]; dcs[b,c] = sum_p dx1 * x.
// grid (B, ceil(C/256)), one thread per channel.
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd_dcs_kernel(const T* dx2, const T* x, const float* sgate,
                                                           const float* dcomp, const int* cmax, int HW, int C,
                                                           float* dcs) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    const float invc = 1.0f / (float)C;
    for (int r = 0; r < HW; ++r) {
        const size_t pix = (size_t)b * HW + r;
        float d = to_f32<T>(dx2[pix * C + c]) * sgate[pix] + dcomp[pix * 2 + 1] * invc;
        if (cmax[pix] == c) d += dcomp[pix * 2 + 0];
        s += d * to_f32<T>(x[pix * C + c]);
    }
    dcs[(size_t)b * C + c] = s;
}

// The same for large maps: one workgroup per (image, row split); dcs_part[b][split][c].
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd_dcs_split_kernel(const T* dx2, const T* x, const float* sgate,
                                                                 const float* dcomp, const int* cmax, int HW, int C,
                                                                 int splits, float* dcs_part) {
    __shared__ f4 rs[256];
    const int tid = threadIdx.x, cvn = C >> 2, npl = 256 / cvn;
    const int cvl = tid % cvn, pl = tid / cvn, c = cvl * 4;
    const int b = blockIdx.x / splits, sp = blockIdx.x % splits;
    const int per = (HW + splits - 1) / splits;
    const int r0 = sp * per, r1 = (r0 + per) < HW ? (r0 + per) : HW;
    const float invc = 1.0f / (float)C;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    if (pl < npl)
        for (int r = r0 + pl; r < r1; r += npl) {
            const size_t pix = (size_t)b * HW + r;
            const f4 g = load4<T>(dx2 + pix * C + c), v = load4<T>(x + pix * C + c);
            const float sg = sgate[pix], dm = dcomp[pix * 2 + 1] * invc, dmx = dcomp[pix * 2 + 0];
            const int cm = cmax[pix] - c;
            s.x += (g.x * sg + dm + (cm == 0 ? dmx : 0.f)) * v.x;
            s.y += (g.y * sg + dm + (cm == 1 ? dmx : 0.f)) * v.y;
            s.z += (g.z * sg + dm + (cm == 2 ? dmx : 0.f)) * v.z;
            s.w += (g.w * sg + dm + (cm == 3 ? dmx : 0.f)) * v.w;
        }
    rs[tid] = s;
    __syncthreads();
    if (tid < cvn) {
        for (int q = 1; q < npl; ++q) {
            const f4 o = rs[q * cvn + tid];
            s = {s.x + o.x, s.y + o.y, s.z + o.z, s.w + o.w};
        }
        store4<float>(dcs_part + ((size_t)b * splits + sp) * C + c, s);
    }
}

// channel-MLP backward for one image: datt, dpre_a, dpre_m (dcs summed over its row splits) kept for the parameter gradients; davg, dmax out
__global__ __launch_bounds__(256) void cbam_bwd_mlp_kernel(const float* dcs, int splits, const float* cscale,
                                                           const float* ha, const float* hm, int C, int HID,
                                                           const float* W1,
                                                           const float* W2, float* datt, float* dpa, float* dpm,
                                                           float* davg, float* dmax) {
    extern __shared__ float sm[];
    float* da = sm;            // [C]
    float* pa = sm + C;        // [HID]
    float* pm = pa + HID;      // [HID]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        const float cs = cscale[(size_t)b * C + c];
        float d = 0.f;
        for (int sp = 0; sp < splits; ++sp) d += dcs[((size_t)b * splits + sp) * C + c];
        const float v = d * cs * (1.f - cs);
        da[c] = v;
        datt[(size_t)b * C + c] = v;
    }
    __syncthreads();
    for (int j = wave; j < HID; j += 4) {
        float d = 0.f;
        for (int c = lane; c < C; c += 64) d += W2[(size_t)c * HID + j] * da[c];
        d = wave_sum(d);
        if (lane == 0) {
            const float a = ha[(size_t)b * HID + j] > 0.f ? d : 0.f;
            const float m = hm[(size_t)b * HID + j] > 0.f ? d : 0.f;
            pa[j] = a; pm[j] = m;
            dpa[(size_t)b * HID + j] = a;
            dpm[(size_t)b * HID + j] = m;
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float a = 0.f, m = 0.f;
        for (int j = 0; j < HID; ++j) {
            const float w = W1[(size_t)j * C + c];
            a += w * pa[j];
            m += w * pm[j];
        }
        davg[(size_t)b * C + c] = a;
        dmax[(size_t)b * C + c] = m;
    }
}

// parameter gradients of the channel MLP summed over the batch; thread per (c, j)
__global__ __launch_bounds__(256) void cbam_bwd_mlp_param_kernel(int B, int C, int HID, const float* datt,
                                                                 const float* dpa, const float* dpm, const float* ha,
                                                                 const float* hm, const float* avg, const float* mx,
                                                                 float* dW1, float* db1, float* dW2, float* db2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * HID) return;
    const int c = i / HID, j = i % HID;
    float w2 = 0.f, w1 = 0.f, bb2 = 0.f, bb1 = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = datt[(size_t)b * C + c];
        const float a = dpa[(size_t)b * HID + j], m = dpm[(size_t)b * HID + j];
        w2 += d * (ha[(size_t)b * HID + j] + hm[(size_t)b * HID + j]);
        w1 += a * avg[(size_t)b * C + c] + m * mx[(size_t)b * C + c];
        bb2 += 2.f * d;
        bb1 += a + m;
    }
    dW2[(size_t)c * HID + j] = w2;
    dW1[(size_t)j * C + c] = w1;
    if (j == 0) db2[c] = bb2;
    if (c == 0) db1[j] = bb1;
}

// pass 2: dx = dx1 * cscale + davg/HW + dmax*[p == amax]
template <typename T>
__global__ __launch_bounds__(256) void cbam_bwd_dx_kernel(const T* dx2, const float* sgate, const float* dcomp,
                                                          const int* cmax, const float* cscale, const float* davg,
                                                          const float* dmax, const int* amax, int B, int HW, int C,
                                                          T* dx) {
    const long long n = (long long)B * HW * C;
    const float invc = 1.0f / (float)C, invhw = 1.0f / (float)HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long pix = i / C;
        const int c = (int)(i % C), b = (int)(pix / HW), r = (int)(pix % HW);
        float d = to_f32<T>(dx2[i]) * sgate[pix] + dcomp[pix * 2 + 1] * invc;
        if (cmax[pix] == c) d += dcomp[pix * 2 + 0];
        const size_t bc = (size_t)b * C + c;
        float o = d * cscale[bc] + davg[bc] * invhw;
        if (amax[bc] == r) o += dmax[bc];
        dx[i] = from_f32<T>(o);
    }
}

// ------------------------------------------------------------------------------------------------
// multi-head self-attention core on qkv [B*HW, 3C] (q | k | v, each head a contiguous HD slice)
// one workgroup per (image, head); K and V rows live in LDS and are read as broadcasts
// ------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const T* qkv, int HW, int C, int heads, float scale,
                                                            float pdrop, unsigned long long seed, const unsigned long long* seed_off, float* P, T* out) {
    seed = seed_with_offset(seed, seed_off);
    extern __shared__ float sm[];
    float* Ks = sm;             // [HW][HD]
    float* Vs = sm + HW * HD;   // [HW][HD]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t row0 = (size_t)b * HW;
    const int ld = 3 * C;
    for (int i = threadIdx.x; i < HW * HD; i += 256) {
        const int r = i / HD, d = i % HD;
        Ks[i] = to_f32<T>(qkv[(row0 + r) * ld + C + h * HD + d]);
        Vs[i] = to_f32<T>(qkv[(row0 + r) * ld + 2 * C + h * HD + d]);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < HW; r += 256) {
        float q[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] = to_f32<T>(qkv[(row0 + r) * ld + h * HD + d]) * scale;
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < HW; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s += q[d] * Ks[j * HD + d];
            const float mn = fmaxf(m, s);
            l = l * __expf(m - mn) + __expf(s - mn);
            m = mn;
        }
        const float invl = 1.0f / l;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
        float* prow = P + (((size_t)b * heads + h) * HW + r) * HW;
        for (int j = 0; j < HW; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s += q[d] * Ks[j * HD + d];
            const float p = __expf(s - m) * invl;
            prow[j] = p;
            const float pd = p * drop_scale(seed, (((unsigned long long)b * heads + h) * HW + r) * HW + j, pdrop);
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] += pd * Vs[j * HD + d];
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) out[(row0 + r) * C + h * HD + d] = from_f32<T>(o[d]);
    }
}

// backward: dqkv from dO, O, P and qkv. Phase A (thread = query row): D_r = dO_r . O_r (the row sum of P o dP, by
// O = P_dropped V) and dQ, with P staged through LDS in key chunks so its global reads are row segments; phase B
// (thread = key row): dK, dV.
#define ATT_JC 25
template <typename T, int HD>
__global__ __launch_bounds__(256) void attention_bwd_kernel(const T* qkv, const float* P, const T* dout, const T* out,
                                                            int HW, int C, int heads, float scale, float pdrop,
                                                            unsigned long long seed, const unsigned long long* seed_off, T* dqkv) {
    seed = seed_with_offset(seed, seed_off);
    extern __shared__ float sm[];
    float* As = sm;             // phase A: K ; phase B: Q
    float* Bs = sm + HW * HD;   // phase A: V ; phase B: dO
    float* Dr = sm + 2 * HW * HD;  // [HW]
    float* Pl = Dr + HW;           // [ATT_JC][HW] chunk of P, key-major
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t row0 = (size_t)b * HW;
    const int ld = 3 * C;
    const unsigned long long pbase = ((unsigned long long)b * heads + h) * HW;
    for (int i = threadIdx.x; i < HW * HD; i += 256) {
        const int r = i / HD, d = i % HD;
        As[i] = to_f32<T>(qkv[(row0 + r) * ld + C + h * HD + d]);
        Bs[i] = to_f32<T>(qkv[(row0 + r) * ld + 2 * C + h * HD + d]);
    }
    // rows are handled in passes of 256 (HW <= 256: one pass); every thread joins the chunk loads and barriers
    const int r = threadIdx.x;
    const bool live = r < HW;
    float g[HD], dq[HD];
    float D = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        g[d] = live ? to_f32<T>(dout[(row0 + r) * C + h * HD + d]) : 0.f;
        dq[d] = 0.f;
        if (live) D += g[d] * to_f32<T>(out[(row0 + r) * C + h * HD + d]);
    }
    if (live) Dr[r] = D;
    for (int j0 = 0; j0 < HW; j0 += ATT_JC) {
        const int jc = (HW - j0) < ATT_JC ? (HW - j0) : ATT_JC;
        __syncthreads();   // previous chunk consumed (first pass: K, V staged)
        for (int i = threadIdx.x; i < HW * jc; i += 256) {
            const int rr = i / jc, jj = i % jc;
            Pl[jj * HW + rr] = P[(pbase + rr) * HW + j0 + jj];
        }
        __syncthreads();
        if (live)
            for (int jj = 0; jj < jc; ++jj) {
                const int j = j0 + jj;
                float dp = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) dp += g[d] * Bs[j * HD + d];
                dp *= drop_scale(seed, (pbase + r) * HW + j, pdrop);
                const float ds = Pl[jj * HW + r] * (dp - D) * scale;
#pragma unroll
                for (int d = 0; d < HD; ++d) dq[d] += ds * As[j * HD + d];
            }
    }
    if (live) {
#pragma unroll
        for (int d = 0; d < HD; ++d) dqkv[(row0 + r) * ld + h * HD + d] = from_f32<T>(dq[d]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW * HD; i += 256) {
        const int r = i / HD, d = i % HD;
        As[i] = to_f32<T>(qkv[(row0 + r) * ld + h * HD + d]);      // Q
        Bs[i] = to_f32<T>(dout[(row0 + r) * C + h * HD + d]);      // dO
    }
    __syncthreads();
    for (int j = threadIdx.x; j < HW; j += 256) {
        float vj[HD], dk[HD], dv[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            vj[d] = to_f32<T>(qkv[(row0 + j) * ld + 2 * C + h * HD + d]);
            dk[d] = 0.f;
            dv[d] = 0.f;
        }
        for (int r = 0; r < HW; ++r) {
            const float p = P[(pbase + r) * HW + j];
            const float dsc = drop_scale(seed, (pbase + r) * HW + j, pdrop);
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dp += Bs[r * HD + d] * vj[d];
            dp *= dsc;
            const float ds = p * (dp - Dr[r]) * scale;
            const float pd = p * dsc;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dk[d] += ds * As[r * HD + d];
                dv[d] += pd * Bs[r * HD + d];
            }
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            dqkv[(row0 + j) * ld + C + h * HD + d] = from_f32<T>(dk[d]);
            dqkv[(row0 + j) * ld + 2 * C + h * HD + d] = from_f32<T>(dv[d]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same attention core on the matrix cores: bf16, up to 128 tokens, head widths 16/32/48/64 (the B3 head: 100
// tokens, 8 heads of 48). One workgroup per (image, head); Q, K, V (and dO) of that head are staged in LDS as bf16,
// row-major [token][d] and/or transposed [d][token], tokens padded with zero rows to a multiple of 16.
//
// Everything is computed TRANSPOSED so that no product has to leave the registers between two MFMAs:
// v_mfma_f32_16x16x16_bf16 hands lane (c = lane%16, g = lane/16) the results D[4g..4g+3][c] and takes as B operand
// B[k = 4g..4g+3][c] -- the same ownership. So S^T = K Q^T (rows keys, columns queries) leaves each lane with four
// consecutive keys of one query, which after the softmax is exactly the B operand of O^T = V^T P^T; the softmax
// reductions over keys are in-register sums plus two cross-lane steps (lanes c, c+16, c+32, c+48). The backward pass
// does the same twice: with queries as columns for dQ^T = K^T dS^T, with keys as columns for dK^T = Q^T dS and
// dV^T = dO^T P (dP is recomputed in each orientation: the matrix cores have the time). P (fp32, written by the
// forward pass for the backward pass, as in the scalar kernels) is rounded to bf16 only as an MFMA operand.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float att_f32x4;
typedef __attribute__((ext_vector_type(4))) short att_s16x4;
#define ATT_MAXB 8   // blocks of 16 tokens

__device__ __forceinline__ att_f32x4 att_mma(att_s16x4 a, att_s16x4 b, att_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ att_s16x4 att_ld4(const uint16_t* p) { return *reinterpret_cast<const att_s16x4*>(p); }
__device__ __forceinline__ att_s16x4 att_pack4(float a, float b, float c, float d) {
    return __builtin_bit_cast(att_s16x4, make_uint2(pack_bf16x2(v2f{a, b}), pack_bf16x2(v2f{c, d})));
}
__device__ __forceinline__ float att_xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float att_xor_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// src: first element of the head's slice in row 0 of the image; rows are `ld` elements apart. Rows [T, TP) become zeros.
template <int HD>
__device__ __forceinline__ void att_stage(const uint16_t* src, int ld, int T, int TP, uint16_t* rm, int ldr, uint16_t* tr,
                                          int ldt) {
    constexpr int CPR = HD / 4;
    for (int i = threadIdx.x; i < TP * CPR; i += 256) {
        const int r = i / CPR, c4 = (i - r * CPR) * 4;
        uint2 v = make_uint2(0u, 0u);
        if (r < T) v = *reinterpret_cast<const uint2*>(src + (size_t)r * ld + c4);
        if (rm) *reinterpret_cast<uint2*>(rm + r * ldr + c4) = v;
        if (tr) {
            tr[(c4 + 0) * ldt + r] = (uint16_t)(v.x & 0xffffu);
            tr[(c4 + 1) * ldt + r] = (uint16_t)(v.x >> 16);
            tr[(c4 + 2) * ldt + r] = (uint16_t)(v.y & 0xffffu);
            tr[(c4 + 3) * ldt + r] = (uint16_t)(v.y >> 16);
        }
    }
}

__device__ __forceinline__ void att_store4(bf16_t* dst, const att_f32x4& v) {
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v2f{v[0], v[1]}), pack_bf16x2(v2f{v[2], v[3]}));
}

template <int HD>
__global__ __launch_bounds__(256) void attention_fwd_mfma_kernel(const bf16_t* qkv, int T, int C, int heads, float scale,
                                                                 float pdrop, unsigned long long seed,
                                                                 const unsigned long long* seed_off, float* P, bf16_t* out) {
    seed = seed_with_offset(seed, seed_off);
    extern __shared__ __attribute__((aligned(16))) char att_sm[];
    constexpr int LDR = HD + 4, DB = HD / 16;
    const int TP = (T + 15) & ~15, NB = TP >> 4, LDT = TP + 4;
    uint16_t* Qs = reinterpret_cast<uint16_t*>(att_sm);   // [TP][LDR]
    uint16_t* Ks = Qs + TP * LDR;                          // [TP][LDR]
    uint16_t* Vt = Ks + TP * LDR;                          // [HD][LDT]
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t row0 = (size_t)b * T;
    const int ld = 3 * C;
    const uint16_t* g = reinterpret_cast<const uint16_t*>(qkv) + row0 * ld + h * HD;
    att_stage<HD>(g, ld, T, TP, Qs, LDR, nullptr, 0);
    att_stage<HD>(g + C, ld, T, TP, Ks, LDR, nullptr, 0);
    att_stage<HD>(g + 2 * C, ld, T, TP, nullptr, 0, Vt, LDT);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, quad = lane >> 4;
    const unsigned long long pbase = ((unsigned long long)b * heads + h) * T;
    const bool vec_p = (T & 3) == 0;
    for (int qb = wave; qb < NB; qb += 4) {
        const int query = qb * 16 + l16;
        const bool qok = query < T;
        att_s16x4 qf[DB];
#pragma unroll
        for (int ks = 0; ks < DB; ++ks) qf[ks] = att_ld4(Qs + query * LDR + ks * 16 + quad * 4);
        att_f32x4 s[ATT_MAXB];
        float m = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < ATT_MAXB; ++kb) {
            s[kb] = att_f32x4{0.f, 0.f, 0.f, 0.f};
            if (kb < NB) {
#pragma unroll
                for (int ks = 0; ks < DB; ++ks)
                    s[kb] = att_mma(att_ld4(Ks + (kb * 16 + l16) * LDR + ks * 16 + quad * 4), qf[ks], s[kb]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = kb * 16 + quad * 4 + i;
                    s[kb][i] = key < T ? s[kb][i] * scale : -INFINITY;
                    m = fmaxf(m, s[kb][i]);
                }
            }
        }
        m = att_xor_max(m);
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < ATT_MAXB; ++kb)
            if (kb < NB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[kb][i] = __expf(s[kb][i] - m);
                    l += s[kb][i];
                }
            }
        l = att_xor_sum(l);
        const float inv = 1.0f / l;
        att_s16x4 pf[ATT_MAXB];
#pragma unroll
        for (int kb = 0; kb < ATT_MAXB; ++kb) {
            pf[kb] = att_s16x4{0, 0, 0, 0};
            if (kb < NB) {
                const int key0 = kb * 16 + quad * 4;
                float pv[4], pd[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pv[i] = s[kb][i] * inv;
                    pd[i] = (qok && key0 + i < T) ? pv[i] * drop_scale(seed, (pbase + query) * T + key0 + i, pdrop) : 0.f;
                }
                if (qok && key0 < T) {
                    float* prow = P + (pbase + query) * T + key0;
                    if (vec_p) {
                        *reinterpret_cast<float4*>(prow) = make_float4(pv[0], pv[1], pv[2], pv[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (key0 + i < T) prow[i] = pv[i];
                    }
                }
                pf[kb] = att_pack4(pd[0], pd[1], pd[2], pd[3]);
            }
        }
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            att_f32x4 o = att_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < ATT_MAXB; ++kb)
                if (kb < NB) o = att_mma(att_ld4(Vt + (db * 16 + l16) * LDT + kb * 16 + quad * 4), pf[kb], o);
            if (qok) att_store4(out + (row0 + query) * C + h * HD + db * 16 + quad * 4, o);
        }
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_mfma_kernel(const bf16_t* qkv, const float* P, const bf16_t* dout,
                                                                 const bf16_t* out, int T, int C, int heads, float scale,
                                                                 float pdrop, unsigned long long seed,
                                                                 const unsigned long long* seed_off, bf16_t* dqkv) {
    seed = seed_with_offset(seed, seed_off);
    extern __shared__ __attribute__((aligned(16))) char att_sm[];
    constexpr int LDR = HD + 4, DB = HD / 16;
    const int TP = (T + 15) & ~15, NB = TP >> 4, LDT = TP + 4;
    uint16_t* Vs = reinterpret_cast<uint16_t*>(att_sm);   // [TP][LDR]
    uint16_t* Gs = Vs + TP * LDR;                          // [TP][LDR]  dO
    uint16_t* Kt = Gs + TP * LDR;                          // [HD][LDT]
    uint16_t* Qt = Kt + HD * LDT;                          // [HD][LDT]
    uint16_t* Gt = Qt + HD * LDT;                          // [HD][LDT]  dO^T
    float* Ds = reinterpret_cast<float*>(Gt + HD * LDT);   // [TP]  D_r = dO_r . O_r
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const size_t row0 = (size_t)b * T;
    const int ld = 3 * C;
    const uint16_t* g = reinterpret_cast<const uint16_t*>(qkv) + row0 * ld + h * HD;
    const uint16_t* go = reinterpret_cast<const uint16_t*>(dout) + row0 * C + h * HD;
    att_stage<HD>(g, ld, T, TP, nullptr, 0, Qt, LDT);
    att_stage<HD>(g + C, ld, T, TP, nullptr, 0, Kt, LDT);
    att_stage<HD>(g + 2 * C, ld, T, TP, Vs, LDR, nullptr, 0);
    att_stage<HD>(go, C, T, TP, Gs, LDR, Gt, LDT);
    for (int r = threadIdx.x; r < TP; r += 256) {
        float d = 0.f;
        if (r < T) {
            const bf16_t* a = dout + (row0 + r) * C + h * HD;
            const bf16_t* o = out + (row0 + r) * C + h * HD;
#pragma unroll
            for (int c = 0; c < HD; c += 4) {
                const f22 x = load22<bf16_t>(a + c), y = load22<bf16_t>(o + c);
                d += x.lo.x * y.lo.x + x.lo.y * y.lo.y + x.hi.x * y.hi.x + x.hi.y * y.hi.y;
            }
        }
        Ds[r] = d;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, quad = lane >> 4;
    const unsigned long long pbase = ((unsigned long long)b * heads + h) * T;
    const bool vec_p = (T & 3) == 0;
    uint16_t* dq_out = reinterpret_cast<uint16_t*>(dqkv) + row0 * ld + h * HD;

    // queries as columns: dP^T = V dO^T, dS^T = P^T o (dP^T - D) * scale, dQ^T = K^T dS^T
    for (int qb = wave; qb < NB; qb += 4) {
        const int query = qb * 16 + l16;
        const bool qok = query < T;
        const float Dq = Ds[query];
        att_s16x4 gf[DB];
#pragma unroll
        for (int ks = 0; ks < DB; ++ks) gf[ks] = att_ld4(Gs + query * LDR + ks * 16 + quad * 4);
        att_s16x4 dsf[ATT_MAXB];
#pragma unroll
        for (int kb = 0; kb < ATT_MAXB; ++kb) {
            dsf[kb] = att_s16x4{0, 0, 0, 0};
            if (kb < NB) {
                att_f32x4 dp = att_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < DB; ++ks)
                    dp = att_mma(att_ld4(Vs + (kb * 16 + l16) * LDR + ks * 16 + quad * 4), gf[ks], dp);
                const int key0 = kb * 16 + quad * 4;
                float pv[4] = {0.f, 0.f, 0.f, 0.f};
                if (qok && key0 < T) {
                    const float* prow = P + (pbase + query) * T + key0;
                    if (vec_p) {
                        const float4 t = *reinterpret_cast<const float4*>(prow);
                        pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[3] = t.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (key0 + i < T) pv[i] = prow[i];
                    }
                }
                float ds[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dsc = (qok && key0 + i < T) ? drop_scale(seed, (pbase + query) * T + key0 + i, pdrop) : 0.f;
                    ds[i] = pv[i] * (dp[i] * dsc - Dq) * scale;
                }
                dsf[kb] = att_pack4(ds[0], ds[1], ds[2], ds[3]);
            }
        }
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            att_f32x4 acc = att_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < ATT_MAXB; ++kb)
                if (kb < NB) acc = att_mma(att_ld4(Kt + (db * 16 + l16) * LDT + kb * 16 + quad * 4), dsf[kb], acc);
            if (qok) att_store4(reinterpret_cast<bf16_t*>(dq_out + (size_t)query * ld + db * 16 + quad * 4), acc);
        }
    }

    // keys as columns: dP = dO V^T, dS = P o (dP - D) * scale, dK^T = Q^T dS, dV^T = dO^T P
    for (int kb = wave; kb < NB; kb += 4) {
        const int key = kb * 16 + l16;
        const bool kok = key < T;
        att_s16x4 vf[DB];
#pragma unroll
        for (int ks = 0; ks < DB; ++ks) vf[ks] = att_ld4(Vs + key * LDR + ks * 16 + quad * 4);
        att_s16x4 dsf[ATT_MAXB], pdf[ATT_MAXB];
#pragma unroll
        for (int qb = 0; qb < ATT_MAXB; ++qb) {
            dsf[qb] = att_s16x4{0, 0, 0, 0};
            pdf[qb] = att_s16x4{0, 0, 0, 0};
            if (qb < NB) {
                att_f32x4 dp = att_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < DB; ++ks)
                    dp = att_mma(att_ld4(Gs + (qb * 16 + l16) * LDR + ks * 16 + quad * 4), vf[ks], dp);
                float ds[4], pd[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int q = qb * 16 + quad * 4 + i;
                    const bool ok = kok && q < T;
                    const float pv = ok ? P[(pbase + q) * T + key] : 0.f;
                    const float dsc = ok ? drop_scale(seed, (pbase + q) * T + key, pdrop) : 0.f;
                    ds[i] = pv * (dp[i] * dsc - Ds[q]) * scale;
                    pd[i] = pv * dsc;
                }
                dsf[qb] = att_pack4(ds[0], ds[1], ds[2], ds[3]);
                pdf[qb] = att_pack4(pd[0], pd[1], pd[2], pd[3]);
            }
        }
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            att_f32x4 dk = att_f32x4{0.f, 0.f, 0.f, 0.f}, dv = att_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qb = 0; qb < ATT_MAXB; ++qb)
                if (qb < NB) {
                    dk = att_mma(att_ld4(Qt + (db * 16 + l16) * LDT + qb * 16 + quad * 4), dsf[qb], dk);
                    dv = att_mma(att_ld4(Gt + (db * 16 + l16) * LDT + qb * 16 + quad * 4), pdf[qb], dv);
                }
            if (kok) {
                att_store4(reinterpret_cast<bf16_t*>(dq_out + (size_t)key * ld + C + db * 16 + quad * 4), dk);
                att_store4(reinterpret_cast<bf16_t*>(dq_out + (size_t)key * ld + 2 * C + db * 16 + quad * 4), dv);
            }
        }
    }
}

static bool attention_use_mfma(int dtype, int HW, int HD) {
    return dtype == MIA_BF16 && HW <= 16 * ATT_MAXB && HD % 16 == 0 && HD <= 64;
}

template <int HD>
static void launch_attention_mfma_(bool fwd, const void* qkv, const float* P, const void* dout, const void* o, int B, int HW,
                                   int C, int heads, float scale, float pdrop, unsigned long long seed,
                                   const unsigned long long* seed_off, float* Pout, void* out, hipStream_t s) {
    const int TP = (HW + 15) & ~15;
    const size_t rm = (size_t)TP * (HD + 4) * 2, tr = (size_t)HD * (TP + 4) * 2;
    const size_t lds = fwd ? 2 * rm + tr : 2 * rm + 3 * tr + (size_t)TP * 4;
    dim3 grid(B * heads);
    if (fwd) {
        auto k = attention_fwd_mfma_kernel<HD>;
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MIA_LAUNCH(k, grid, dim3(256), lds, s, (const bf16_t*)qkv, HW, C, heads, scale, pdrop, seed, seed_off, Pout, (bf16_t*)out);
    } else {
        auto k = attention_bwd_mfma_kernel<HD>;
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MIA_LAUNCH(k, grid, dim3(256), lds, s, (const bf16_t*)qkv, P, (const bf16_t*)dout, (const bf16_t*)o, HW, C, heads, scale, pdrop, seed, seed_off, (bf16_t*)out);
    }
}

// ------------------------------------------------------------------------------------------------
// pooled features -> (LayerNorm) -> dropout. One wave per image row. in_part [B][S][C] holds sums over H*W.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_ln_dropout_fwd_kernel(const float* in_part, int B, int S, int C,
                                                                  float inv_hw, const float* gamma, const float* beta,
                                                                  float eps, float pdrop, unsigned long long seed, const unsigned long long* seed_off,
                                                                  float* out, float* xhat, float* rstd_out) {
    seed = seed_with_offset(seed, seed_off);
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float s = 0.f, q = 0.f;
    for (int c = lane; c < C; c += 64) {
        float v = 0.f;
        for (int k = 0; k < S; ++k) v += in_part[((size_t)b * S + k) * C + c];
        v *= inv_hw;
        xhat[(size_t)b * C + c] = v;  // pooled value for now
        s += v;
    }
    s = wave_sum(s);
    const float mean = s / (float)C;
    float rs = 1.f;
    if (gamma) {
        for (int c = lane; c < C; c += 64) {
            const float d = xhat[(size_t)b * C + c] - mean;
            q += d * d;
        }
        q = wave_sum(q);
        rs = 1.0f / sqrtf(q / (float)C + eps);
    }
    for (int c = lane; c < C; c += 64) {
        float v = xhat[(size_t)b * C + c];
        if (gamma) {
            const float xh = (v - mean) * rs;
            xhat[(size_t)b * C + c] = xh;
            v = xh * gamma[c] + beta[c];
        }
        out[(size_t)b * C + c] = v * drop_scale(seed, (unsigned long long)b * C + c, pdrop);
    }
    if (lane == 0) rstd_out[b] = rs;
}

// backward: dpooled [B,C] from dout [B,C]; per-image partial dgamma/dbeta rows are summed by a colsum afterwards
__global__ __launch_bounds__(256) void pool_ln_dropout_bwd_kernel(const float* dout, int B, int C, const float* gamma,
                                                                  const float* xhat, const float* rstd, float pdrop,
                                                                  unsigned long long seed, const unsigned long long* seed_off, float* dpooled,
                                                                  float* dgamma_rows, float* dbeta_rows) {
    seed = seed_with_offset(seed, seed_off);
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    if (!gamma) {
        for (int c = lane; c < C; c += 64)
            dpooled[(size_t)b * C + c] = dout[(size_t)b * C + c] * drop_scale(seed, (unsigned long long)b * C + c, pdrop);
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float dy = dout[(size_t)b * C + c] * drop_scale(seed, (unsigned long long)b * C + c, pdrop);
        const float xh = xhat[(size_t)b * C + c];
        dgamma_rows[(size_t)b * C + c] = dy * xh;
        dbeta_rows[(size_t)b * C + c] = dy;
        const float dxh = dy * gamma[c];
        s1 += dxh;
        s2 += dxh * xh;
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    const float rs = rstd[b];
    for (int c = lane; c < C; c += 64) {
        const float dy = dout[(size_t)b * C + c] * drop_scale(seed, (unsigned long long)b * C + c, pdrop);
        const float xh = xhat[(size_t)b * C + c];
        dpooled[(size_t)b * C + c] = rs * (dy * gamma[c] - s1 - xh * s2);
    }
}

// out[b, p, c] = v[b, c] * mult (+ res): gradient of the average pool broadcast back over H*W
template <typename T>
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* v, int B, int HW, int C, float mult,
                                                             const T* res, T* out) {
    const long long nvec = (long long)B * HW * C / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const size_t idx = (size_t)i * 4;
        const int c = (int)(idx % C), b = (int)(idx / ((size_t)HW * C));
        f4 a = load4<float>(v + (size_t)b * C + c);
        a.x *= mult; a.y *= mult; a.z *= mult; a.w *= mult;
        if (res) {
            const f4 r = load4<T>(res + idx);
            a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
        }
        store4<T>(out + idx, a);
    }
}

// y = x * dropout_mask (in place), optionally gated by (ref > 0): dropout forward, or ReLU+dropout backward
__global__ __launch_bounds__(256) void dropout_f32_kernel(float* x, const float* relu_ref, long long n, float pdrop,
                                                          unsigned long long seed, const unsigned long long* seed_off) {
    seed = seed_with_offset(seed, seed_off);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i] * drop_scale(seed, (unsigned long long)i, pdrop);
        if (relu_ref && !(relu_ref[i] > 0.f)) v = 0.f;
        x[i] = v;
    }
}

// column sums of an [R, C] matrix (bias gradients). grid = (ceil(C/64), row_chunks); every row chunk writes its
// partial sums to its own row of `part` ([row_chunks][C]; the output itself with one chunk), and colsum_fold adds the
// rows in order: no float atomics, so the result does not depend on the order the workgroups ran in.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long long R, int C, int ld, float* part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const long long rows_per = (R + gridDim.y - 1) / gridDim.y;
    const long long r0 = (long long)blockIdx.y * rows_per;
    const long long r1 = (r0 + rows_per) < R ? (r0 + rows_per) : R;
    float s = 0.f;
    if (c < C)
        for (long long r = r0 + w; r < r1; r += 4) s += to_f32<T>(x[(size_t)r * ld + c]);
    red[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C)
        part[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* part, int chunks, int C, float* out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * C + c];
    out[c] = s;
}

// softmax cross-entropy, mean over the batch, with its gradient (d loss / d logits). One workgroup.
// Labels follow nn.CrossEntropyLoss: -100 (its default ignore_index) drops the row from the sum and from the divisor
// (all rows ignored: 0/0 = NaN, as in torch). Any other label outside [0, n) is an error there (a device-side assert);
// here the row contributes no gradient, is never used as an index, and the loss comes back NaN so the mistake shows.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* logits, const long long* labels, int B, int n,
                                                         int ld, float* loss, float* dlogits, int ldd) {
    __shared__ float red[4], cnt[4], bad[4];
    float valid = 0.f, invalid = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const long long y = labels[b];
        if (y >= 0 && y < n) valid += 1.f;
        else if (y != -100) invalid += 1.f;
    }
    valid = wave_sum(valid);
    invalid = wave_sum(invalid);
    if ((threadIdx.x & 63) == 0) { cnt[threadIdx.x >> 6] = valid; bad[threadIdx.x >> 6] = invalid; }
    __syncthreads();
    const float count = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    const float nbad = bad[0] + bad[1] + bad[2] + bad[3];
    const float inv = 1.0f / count;
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* z = logits + (size_t)b * ld;
        const long long y = labels[b];
        const bool use = y >= 0 && y < n;
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) m = fmaxf(m, z[j]);
        float l = 0.f;
        for (int j = 0; j < n; ++j) l += expf(z[j] - m);
        const float lse = m + logf(l);
        if (use) acc += lse - z[y];
        for (int j = 0; j < ldd; ++j)
            dlogits[(size_t)b * ldd + j] = (use && j < n) ? (expf(z[j] - lse) - (j == y ? 1.f : 0.f)) * inv : 0.f;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = nbad > 0.f ? __builtin_nanf("") : (red[0] + red[1] + red[2] + red[3]) / count;
}

// =================================================================================================
// C ABI
// =================================================================================================
#define MIA_STREAM reinterpret_cast<hipStream_t>(stream)
static inline unsigned grid1d(long long n, int per = 256, long long cap = 16384) {
    long long b = (n + per - 1) / per;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// Row splits per image for the pooled reductions of a map with HW pixels: about 1024 workgroups in total, at
// least 64 rows each; 1 = the one-thread-per-channel kernels (small maps, or more than 1024 channels).
static int cbam_row_splits(int B, int HW, int C) {
    if (C % 4 != 0 || C / 4 > 256) return 1;
    int s = HW / 64, cap = 1024 / B;
    if (cap < 1) cap = 1;
    if (s > cap) s = cap;
    return s < 2 ? 1 : s;
}
static int cbam_bn_parts(long long npix) { return npix > 32768 ? (int)((npix + 4095) / 4096 > 256 ? 256 : (npix + 4095) / 4096) : 0; }
static int cbam_conv_chunks(long long npix) { return npix > 32768 ? (int)((npix + 8191) / 8192 > 64 ? 64 : (npix + 8191) / 8192) : 1; }

extern "C" int mia_cbam_ws_floats(int B, int H, int W, int C, int HID, int backward) {
    const long long npix = (long long)B * H * W;
    const int splits = cbam_row_splits(B, H * W, C);
    long long n;
    if (!backward) n = splits > 1 ? 3LL * B * splits * C : 0;
    else
        n = 4 * npix + 4LL * B * C + 2LL * B * HID + (splits > 1 ? (long long)B * splits * C : 0) +
            2LL * cbam_bn_parts(npix) + (cbam_conv_chunks(npix) > 1 ? 98LL * cbam_conv_chunks(npix) : 0);
    return n > 0x7fffffffLL ? -1 : (int)n;
}

extern "C" int mia_cbam_channel_fwd(int dtype, const void* x, int B, int HW, int C, int HID, const float* W1,
                                    const float* b1, const float* W2, const float* b2, float* avg, float* mx,
                                    int* amax, float* ha, float* hm, float* cscale, float* ws, void* stream) {
    MIA_CHECK_ARG(x && W1 && b1 && W2 && b2 && avg && mx && amax && ha && hm && cscale, "mia_cbam_channel_fwd: null");
    MIA_CHECK_ARG(B > 0 && HW > 0 && C > 0 && HID > 0, "mia_cbam_channel_fwd: bad shape");
    const int splits = cbam_row_splits(B, HW, C);
    if (splits > 1) {
        MIA_CHECK_ARG(ws, "mia_cbam_channel_fwd: this shape needs the workspace of mia_cbam_ws_floats");
        float* psum = ws;
        float* pmax = psum + (size_t)B * splits * C;
        int* parg = reinterpret_cast<int*>(pmax + (size_t)B * splits * C);
        if (dtype == MIA_BF16)
            MIA_LAUNCH(cbam_channel_pool_split_kernel<bf16_t>, dim3(B * splits), dim3(256), 0, MIA_STREAM,
                               (const bf16_t*)x, HW, C, splits, psum, pmax, parg);
        else
            MIA_LAUNCH(cbam_channel_pool_split_kernel<float>, dim3(B * splits), dim3(256), 0, MIA_STREAM,
                               (const float*)x, HW, C, splits, psum, pmax, parg);
        MIA_LAUNCH(cbam_channel_pool_combine_kernel, dim3(B, cdiv(C, 256)), dim3(256), 0, MIA_STREAM, psum, pmax,
                           parg, HW, C, splits, avg, mx, amax);
    } else {
        dim3 g(B, cdiv(C, 256));
        if (dtype == MIA_BF16)
            MIA_LAUNCH(cbam_channel_pool_kernel<bf16_t>, g, dim3(256), 0, MIA_STREAM, (const bf16_t*)x, HW, C, avg, mx, amax);
        else
            MIA_LAUNCH(cbam_channel_pool_kernel<float>, g, dim3(256), 0, MIA_STREAM, (const float*)x, HW, C, avg, mx, amax);
    }
    MIA_LAUNCH(cbam_channel_mlp_kernel, dim3(B), dim3(256), (size_t)(2 * C + HID) * 4, MIA_STREAM, avg, mx, C, HID,
                       W1, b1, W2, b2, ha, hm, cscale);
    MIA_CHECK_LAUNCH("mia_cbam_channel_fwd");
    return 0;
}

// comp [B,HW,2] = (max_c, mean_c) of x*cscale; pre = conv7x7(comp); part [part_rows][2][1] for BatchNorm(1)
extern "C" int mia_cbam_spatial_fwd(int dtype, const void* x, const float* cscale, int B, int H, int W, int C,
                                    const float* w7, float* comp, int* cmax, float* pre, float* part, int part_rows,
                                    void* stream) {
    MIA_CHECK_ARG(x && cscale && w7 && comp && cmax && pre && part, "mia_cbam_spatial_fwd: null");
    MIA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && part_rows > 0 && part_rows <= 65535, "mia_cbam_spatial_fwd: bad shape");
    const long long npix = (long long)B * H * W;
    if (dtype == MIA_BF16)
        MIA_LAUNCH(cbam_spatial_pool_kernel<bf16_t>, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, MIA_STREAM,
                           (const bf16_t*)x, cscale, B, H * W, C, comp, cmax);
    else
        MIA_LAUNCH(cbam_spatial_pool_kernel<float>, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, MIA_STREAM,
                           (const float*)x, cscale, B, H * W, C, comp, cmax);
    MIA_LAUNCH(cbam_spatial_conv_kernel, dim3(part_rows), dim3(256), 0, MIA_STREAM, comp, B, H, W, w7, pre, part);
    MIA_CHECK_LAUNCH("mia_cbam_spatial_fwd");
    return 0;
}

// out = x * cscale * sigmoid(pre*bnst[0] + bnst[1]); sgate [B,HW] stored
extern "C" int mia_cbam_apply(int dtype, const void* x, const float* cscale, const float* pre, const float* bnst, int B,
                              int HW, int C, void* out, float* sgate, void* stream) {
    MIA_CHECK_ARG(x && cscale && pre && bnst && out && sgate && C % 4 == 0, "mia_cbam_apply: bad arguments");
    const unsigned g = grid1d((long long)B * HW * C / 4);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(cbam_apply_kernel<bf16_t>, dim3(g), dim3(256), 0, MIA_STREAM, (const bf16_t*)x, cscale, pre, bnst,
                           B, HW, C, (bf16_t*)out, sgate);
    else
        MIA_LAUNCH(cbam_apply_kernel<float>, dim3(g), dim3(256), 0, MIA_STREAM, (const float*)x, cscale, pre, bnst, B,
                           HW, C, (float*)out, sgate);
    MIA_CHECK_LAUNCH("mia_cbam_apply");
    return 0;
}

// full CBAM backward. ws: fp32 scratch of mia_cbam_ws_floats(B, H, W, C, HID, 1) floats.
extern "C" int mia_cbam_bwd(int dtype, const void* dx2, const void* x, int B, int H, int W, int C, int HID,
                            const float* cscale, const float* sgate, const float* pre, const float* bnst,
                            const float* bn_gamma, int bn_training, const float* comp, const int* cmax,
                            const int* amax, const float* avg, const float* mx, const float* ha, const float* hm,
                            const float* W1, const float* W2, const float* w7, float* ws, void* dx, float* dW1,
                            float* db1, float* dW2, float* db2, float* dw7, float* dbn_gamma, float* dbn_beta,
                            void* stream) {
    MIA_CHECK_ARG(dx2 && x && cscale && sgate && pre && bnst && bn_gamma && comp && cmax && amax && avg && mx && ha &&
                      hm && W1 && W2 && w7 && ws && dx && dW1 && db1 && dW2 && db2 && dw7 && dbn_gamma && dbn_beta,
                  "mia_cbam_bwd: null argument");
    MIA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && HID > 0, "mia_cbam_bwd: bad shape");
    const int HW = H * W;
    const long long npix = (long long)B * HW;
    float* dsg = ws;                 // [npix]
    float* dpre = dsg + npix;        // [npix]
    float* dcomp = dpre + npix;      // [npix][2]
    float* dcs = dcomp + 2 * npix;   // [B][C]
    float* datt = dcs + (size_t)B * C;
    float* davg = datt + (size_t)B * C;
    float* dmax = davg + (size_t)B * C;   // caller sized ws for 4*npix + 4*B*C + 2*B*HID
    float* dpa = dmax + (size_t)B * C;
    float* dpm = dpa + (size_t)B * HID;
    const int splits = cbam_row_splits(B, HW, C), bn_parts = cbam_bn_parts(npix), chunks = cbam_conv_chunks(npix);
    float* dcs_part = dpm + (size_t)B * HID;                                   // [B][splits][C] when splits > 1
    float* bn_part = dcs_part + (splits > 1 ? (size_t)B * splits * C : 0);      // [bn_parts][2]
    float* cw_part = bn_part + 2 * (size_t)bn_parts;                            // [chunks][98] when chunks > 1
    hipStream_t s = MIA_STREAM;
    const unsigned gp = (unsigned)((npix + 3) / 4);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(cbam_bwd_dsgate_kernel<bf16_t>, dim3(gp), dim3(256), 0, s, (const bf16_t*)dx2, (const bf16_t*)x,
                           cscale, B, HW, C, dsg);
    else
        MIA_LAUNCH(cbam_bwd_dsgate_kernel<float>, dim3(gp), dim3(256), 0, s, (const float*)dx2, (const float*)x,
                           cscale, B, HW, C, dsg);
    if (bn_parts > 0) {
        MIA_LAUNCH(cbam_bwd_spatial_bn_part_kernel, dim3(bn_parts), dim3(256), 0, s, dsg, sgate, pre, bnst, npix,
                           bn_part);
        MIA_LAUNCH(cbam_bwd_spatial_bn_apply_kernel, dim3(grid1d(npix, 1024)), dim3(256), 0, s, dsg, sgate, pre,
                           bnst, bn_gamma, bn_training, npix, bn_part, bn_parts, dpre, dbn_gamma, dbn_beta);
    } else {
        MIA_LAUNCH(cbam_bwd_spatial_bn_kernel, dim3(1), dim3(1024), 0, s, dsg, sgate, pre, bnst, bn_gamma,
                           bn_training, npix, dpre, dbn_gamma, dbn_beta);
    }
    MIA_LAUNCH(cbam_bwd_conv_weight_kernel, dim3(98, chunks), dim3(256), 0, s, dpre, comp, B, H, W,
                       chunks > 1 ? cw_part : dw7);
    if (chunks > 1) MIA_LAUNCH(cbam_bwd_conv_weight_fold_kernel, dim3(1), dim3(128), 0, s, cw_part, chunks, dw7);
    MIA_LAUNCH(cbam_bwd_conv_data_kernel, dim3(grid1d(npix)), dim3(256), 0, s, dpre, B, H, W, w7, dcomp);
    if (splits > 1) {
        if (dtype == MIA_BF16)
            MIA_LAUNCH(cbam_bwd_dcs_split_kernel<bf16_t>, dim3(B * splits), dim3(256), 0, s, (const bf16_t*)dx2,
                               (const bf16_t*)x, sgate, dcomp, cmax, HW, C, splits, dcs_part);
        else
            MIA_LAUNCH(cbam_bwd_dcs_split_kernel<float>, dim3(B * splits), dim3(256), 0, s, (const float*)dx2,
                               (const float*)x, sgate, dcomp, cmax, HW, C, splits, dcs_part);
    } else {
        dim3 gc(B, cdiv(C, 256));
        if (dtype == MIA_BF16)
            MIA_LAUNCH(cbam_bwd_dcs_kernel<bf16_t>, gc, dim3(256), 0, s, (const bf16_t*)dx2, (const bf16_t*)x, sgate,
                               dcomp, cmax, HW, C, dcs);
        else
            MIA_LAUNCH(cbam_bwd_dcs_kernel<float>, gc, dim3(256), 0, s, (const float*)dx2, (const float*)x, sgate,
                               dcomp, cmax, HW, C, dcs);
    }
    MIA_LAUNCH(cbam_bwd_mlp_kernel, dim3(B), dim3(256), (size_t)(C + 2 * HID) * 4, s, splits > 1 ? dcs_part : dcs,
                       splits > 1 ? splits : 1, cscale, ha, hm, C, HID, W1, W2, datt, dpa, dpm, davg, dmax);
    MIA_LAUNCH(cbam_bwd_mlp_param_kernel, dim3(cdiv((long long)C * HID, 256)), dim3(256), 0, s, B, C, HID, datt, dpa,
                       dpm, ha, hm, avg, mx, dW1, db1, dW2, db2);
    const unsigned ge = grid1d((long long)B * HW * C);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(cbam_bwd_dx_kernel<bf16_t>, dim3(ge), dim3(256), 0, s, (const bf16_t*)dx2, sgate, dcomp, cmax,
                           cscale, davg, dmax, amax, B, HW, C, (bf16_t*)dx);
    else
        MIA_LAUNCH(cbam_bwd_dx_kernel<float>, dim3(ge), dim3(256), 0, s, (const float*)dx2, sgate, dcomp, cmax, cscale,
                           davg, dmax, amax, B, HW, C, (float*)dx);
    MIA_CHECK_LAUNCH("mia_cbam_bwd");
    return 0;
}

template <typename T>
static int launch_attention(bool fwd, const void* qkv, const float* P, const void* dout, const void* o, int B, int HW, int C, int heads,
                            float pdrop, unsigned long long seed, const unsigned long long* seed_off, float* Pout, void* out,
                            hipStream_t s) {
    const int HD = C / heads;
    const float scale = 1.0f / sqrtf((float)HD);
    if (attention_use_mfma(sizeof(T) == 2 ? MIA_BF16 : MIA_F32, HW, HD)) {
        switch (HD) {
            case 16: launch_attention_mfma_<16>(fwd, qkv, P, dout, o, B, HW, C, heads, scale, pdrop, seed, seed_off, Pout, out, s); break;
            case 32: launch_attention_mfma_<32>(fwd, qkv, P, dout, o, B, HW, C, heads, scale, pdrop, seed, seed_off, Pout, out, s); break;
            case 48: launch_attention_mfma_<48>(fwd, qkv, P, dout, o, B, HW, C, heads, scale, pdrop, seed, seed_off, Pout, out, s); break;
            default: launch_attention_mfma_<64>(fwd, qkv, P, dout, o, B, HW, C, heads, scale, pdrop, seed, seed_off, Pout, out, s); break;
        }
        return 0;
    }
    const size_t lds = ((size_t)2 * HW * HD + (fwd ? 0 : HW + (size_t)ATT_JC * HW)) * 4;
    dim3 grid(B * heads);
#define MIA_ATT(HD_)                                                                                                  \
    do {                                                                                                              \
        if (fwd) {                                                                                                    \
            auto k = attention_fwd_kernel<T, HD_>;                                                                    \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            MIA_LAUNCH(k, grid, dim3(256), lds, s, (const T*)qkv, HW, C, heads, scale, pdrop, seed, seed_off, Pout, (T*)out); \
        } else {                                                                                                      \
            auto k = attention_bwd_kernel<T, HD_>;                                                                    \
            if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            MIA_LAUNCH(k, grid, dim3(256), lds, s, (const T*)qkv, P, (const T*)dout, (const T*)o, HW, C, heads, scale, pdrop, seed, seed_off, (T*)out); \
        }                                                                                                             \
    } while (0)
    if (HD == 48) MIA_ATT(48);
    else if (HD == 8) MIA_ATT(8);
    else if (HD == 16) MIA_ATT(16);
    else if (HD == 24) MIA_ATT(24);
    else if (HD == 32) MIA_ATT(32);
    else if (HD == 64) MIA_ATT(64);
    else if (HD == 96) MIA_ATT(96);
    else if (HD == 128) MIA_ATT(128);
    else return -1;
#undef MIA_ATT
    return 0;
}

static int attention_checks(const char* name, int B, int HW, int C, int heads) {
    MIA_CHECK_ARG(B > 0 && HW > 0 && HW <= 256 && C > 0 && heads > 0 && C % heads == 0,
                  "%s: bad shape B=%d tokens=%d C=%d heads=%d (tokens <= 256)", name, B, HW, C, heads);
    const int hd = C / heads;
    MIA_CHECK_ARG(hd == 8 || hd == 16 || hd == 24 || hd == 32 || hd == 48 || hd == 64 || hd == 96 || hd == 128,
                  "%s: head_dim %d not built (8,16,24,32,48,64,96,128)", name, hd);
    return 0;
}

// qkv [B*HW, 3C] -> out [B*HW, C]; P [B,heads,HW,HW] fp32 softmax weights kept for the backward pass
extern "C" int mia_attention_fwd(int dtype, const void* qkv, int B, int HW, int C, int heads, float pdrop,
                                 unsigned long long seed, const unsigned long long* seed_offset, float* P, void* out,
                                 void* stream) {
    if (attention_checks("mia_attention_fwd", B, HW, C, heads)) return -1;
    MIA_CHECK_ARG(qkv && P && out, "mia_attention_fwd: null");
    int rc = dtype == MIA_BF16
                 ? launch_attention<bf16_t>(true, qkv, nullptr, nullptr, nullptr, B, HW, C, heads, pdrop, seed, seed_offset, P, out, MIA_STREAM)
                 : launch_attention<float>(true, qkv, nullptr, nullptr, nullptr, B, HW, C, heads, pdrop, seed, seed_offset, P, out, MIA_STREAM);
    MIA_CHECK_ARG(rc == 0, "mia_attention_fwd: dispatch failed");
    MIA_CHECK_LAUNCH("mia_attention_fwd");
    return 0;
}

extern "C" int mia_attention_bwd(int dtype, const void* qkv, const float* P, const void* dout, const void* out, int B,
                                 int HW, int C, int heads, float pdrop, unsigned long long seed,
                                 const unsigned long long* seed_offset, void* dqkv,
                                 void* stream) {
    if (attention_checks("mia_attention_bwd", B, HW, C, heads)) return -1;
    MIA_CHECK_ARG(qkv && P && dout && out && dqkv, "mia_attention_bwd: null");
    int rc = dtype == MIA_BF16
                 ? launch_attention<bf16_t>(false, qkv, P, dout, out, B, HW, C, heads, pdrop, seed, seed_offset, nullptr, dqkv, MIA_STREAM)
                 : launch_attention<float>(false, qkv, P, dout, out, B, HW, C, heads, pdrop, seed, seed_offset, nullptr, dqkv, MIA_STREAM);
    MIA_CHECK_ARG(rc == 0, "mia_attention_bwd: dispatch failed");
    MIA_CHECK_LAUNCH("mia_attention_bwd");
    return 0;
}

extern "C" int mia_pool_ln_dropout_fwd(const float* in_part, int B, int S, int HW, int C, const float* gamma,
                                       const float* beta, float eps, float pdrop, unsigned long long seed,
                                       const unsigned long long* seed_offset, float* out,
                                       float* xhat, float* rstd, void* stream) {
    MIA_CHECK_ARG(in_part && out && xhat && rstd && B > 0 && S > 0 && HW > 0 && C > 0, "mia_pool_ln_dropout_fwd: bad arguments");
    MIA_CHECK_ARG(!(gamma == nullptr) == !(beta == nullptr), "mia_pool_ln_dropout_fwd: gamma/beta come together");
    MIA_LAUNCH(pool_ln_dropout_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, MIA_STREAM, in_part, B, S, C,
                       1.0f / (float)HW, gamma, beta, eps, pdrop, seed, seed_offset, out, xhat, rstd);
    MIA_CHECK_LAUNCH("mia_pool_ln_dropout_fwd");
    return 0;
}

extern "C" int mia_pool_ln_dropout_bwd(const float* dout, int B, int C, const float* gamma, const float* xhat,
                                       const float* rstd, float pdrop, unsigned long long seed,
                                       const unsigned long long* seed_offset, float* dpooled,
                                       float* dgamma_rows, float* dbeta_rows, void* stream) {
    MIA_CHECK_ARG(dout && dpooled && B > 0 && C > 0, "mia_pool_ln_dropout_bwd: bad arguments");
    MIA_CHECK_ARG(!gamma || (xhat && rstd && dgamma_rows && dbeta_rows), "mia_pool_ln_dropout_bwd: LayerNorm needs its state");
    MIA_LAUNCH(pool_ln_dropout_bwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, MIA_STREAM, dout, B, C, gamma, xhat, rstd,
                       pdrop, seed, seed_offset, dpooled, dgamma_rows, dbeta_rows);
    MIA_CHECK_LAUNCH("mia_pool_ln_dropout_bwd");
    return 0;
}

extern "C" int mia_broadcast_rows(int dtype, const float* v, int B, int HW, int C, float mult, const void* res,
                                  void* out, void* stream) {
    MIA_CHECK_ARG(v && out && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "mia_broadcast_rows: bad arguments");
    const unsigned g = grid1d((long long)B * HW * C / 4);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(broadcast_rows_kernel<bf16_t>, dim3(g), dim3(256), 0, MIA_STREAM, v, B, HW, C, mult,
                           (const bf16_t*)res, (bf16_t*)out);
    else
        MIA_LAUNCH(broadcast_rows_kernel<float>, dim3(g), dim3(256), 0, MIA_STREAM, v, B, HW, C, mult,
                           (const float*)res, (float*)out);
    MIA_CHECK_LAUNCH("mia_broadcast_rows");
    return 0;
}

extern "C" int mia_dropout_f32(float* x, const float* relu_ref, long long n, float pdrop, unsigned long long seed,
                               const unsigned long long* seed_offset,
                               void* stream) {
    MIA_CHECK_ARG(x && n > 0, "mia_dropout_f32: bad arguments");
    MIA_LAUNCH(dropout_f32_kernel, dim3(grid1d(n)), dim3(256), 0, MIA_STREAM, x, relu_ref, n, pdrop, seed, seed_offset);
    MIA_CHECK_LAUNCH("mia_dropout_f32");
    return 0;
}

extern "C" int mia_colsum(int dtype, const void* x, long long R, int C, int ld, float* out, float* ws,
                          long long ws_floats, void* stream) {
    MIA_CHECK_ARG(x && out && R > 0 && C > 0 && ld >= C, "mia_colsum: bad arguments");
    // as many row chunks (of at least 64 rows, at most 128 chunks) as the workspace has rows for
    long long chunks = R / 64;
    if (chunks > 128) chunks = 128;
    if (ws == nullptr || chunks > ws_floats / C) chunks = ws == nullptr ? 1 : ws_floats / C;
    if (chunks < 1) chunks = 1;
    float* part = chunks == 1 ? out : ws;
    dim3 grid(cdiv(C, 64), (unsigned)chunks);
    if (dtype == MIA_BF16)
        MIA_LAUNCH(colsum_kernel<bf16_t>, grid, dim3(256), 0, MIA_STREAM, (const bf16_t*)x, R, C, ld, part);
    else
        MIA_LAUNCH(colsum_kernel<float>, grid, dim3(256), 0, MIA_STREAM, (const float*)x, R, C, ld, part);
    if (chunks > 1)
        MIA_LAUNCH(colsum_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, MIA_STREAM, part, (int)chunks, C, out);
    MIA_CHECK_LAUNCH("mia_colsum");
    return 0;
}

extern "C" int mia_softmax_ce(const float* logits, int ld, const long long* labels, int B, int n, float* loss,
                              float* dlogits, int ldd, void* stream) {
    MIA_CHECK_ARG(logits && labels && loss && dlogits && B > 0 && n > 0 && ld >= n && ldd >= n, "mia_softmax_ce: bad arguments");
    MIA_LAUNCH(softmax_ce_kernel, dim3(1), dim3(256), 0, MIA_STREAM, logits, labels, B, n, ld, loss, dlogits, ldd);
    MIA_CHECK_LAUNCH("mia_softmax_ce");
    return 0;
}

// On-device evaluation counters (gfx950): arg-max over classes + confusion-matrix accumulation, so a training or
// validation loop never copies predictions to the host (the reference extends Python lists per batch and calls
// scikit-learn on them at the end of the epoch). Integer counters: results are exact and order-independent.
#include "common.h"

#define CONF_MAXN 64

// element (image b, class k, position p) of the logits sits at b*sb + k*sk + p*sp:
//   [B, n] classification logits: P = 1, sb = n, sk = 1;   [B, n, H, W] segmentation logits: P = HW, sb = n*HW,
//   sk = HW, sp = 1. conf[t*n + pred] += 1 with pred = FIRST maximal class (torch.argmax's rule); targets outside
//   [0, n) are skipped. Optionally loss_acc[0] += loss[0] (the epoch's running loss), done by one thread.
__global__ __launch_bounds__(256) void confusion_kernel(const float* logits, long long sb, long long sk, long long sp,
                                                        const long long* target, long long total, long long P, int n,
                                                        unsigned long long* conf, const float* loss, double* loss_acc) {
    __shared__ unsigned int hist[CONF_MAXN * CONF_MAXN];
    for (int i = threadIdx.x; i < n * n; i += 256) hist[i] = 0u;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / P, p = i % P;
        const float* z = logits + b * sb + p * sp;
        float m = z[0];
        int am = 0;
        for (int k = 1; k < n; ++k) {
            const float v = z[(long long)k * sk];
            if (v > m) { m = v; am = k; }
        }
        const long long t = target[i];
        if (t >= 0 && t < n) atomicAdd(&hist[(int)t * n + am], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * n; i += 256)
        if (hist[i]) atomicAdd(conf + i, (unsigned long long)hist[i]);
    if (loss && loss_acc && blockIdx.x == 0 && threadIdx.x == 0) loss_acc[0] += (double)loss[0];
}

extern "C" int mia_confusion_update(const float* logits, long long stride_image, long long stride_class,
                                    long long stride_pos, const long long* target, long long images,
                                    long long positions, int n, unsigned long long* conf, const float* loss,
                                    double* loss_acc, void* stream) {
    MIA_CHECK_ARG(logits && target && conf && images > 0 && positions > 0, "mia_confusion_update: bad arguments");
    MIA_CHECK_ARG(n > 0 && n <= CONF_MAXN, "mia_confusion_update: classes must be 1..%d (got %d)", CONF_MAXN, n);
    const long long total = images * positions;
    long long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    MIA_LAUNCH(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), logits,
                       stride_image, stride_class, stride_pos, target, total, positions, n, conf, loss, loss_acc);
    MIA_CHECK_LAUNCH("mia_confusion_update");
    return 0;
}

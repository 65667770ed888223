// optim.hip -- the Adam step of the training loop over all parameter tensors in one launch.
//
// The reference steps `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over ten parameter groups every iteration
// (S3Gaussian/scene/gaussian_model.py:188-201, train.py:428).  At 2 M Gaussians that is 504 MB of parameters and as much again
// for each moment; the stock multi-tensor implementation makes ~10 passes over them (2.4 ms on MI355X, more than the whole
// forward + backward of this repository), the stock "fused" one 0.96 ms.  Here every element is read once (param, grad, both
// moments) and written once (param, both moments): 28 B per element, one launch for up to EMD_ADAM_MAX_TENSORS tensors with
// their own learning rates.  HBM-bound; nothing is reused.
//
// Arithmetic, in torch.optim.Adam's order (torch/optim/adam.py, _single_tensor_adam; no weight decay, no amsgrad):
//   m  = m + (1 - beta1) * (g - m)                          exp_avg.lerp_(grad, 1 - beta1)
//   v  = v * beta2 + (1 - beta2) * g * g                    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
//   d  = sqrt(v) / sqrt(1 - beta2^t) + eps                  (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
//   p  = p - (lr / (1 - beta1^t)) * (m / d)                 param.addcdiv_(exp_avg, denom, value=-step_size)
// with the step-dependent scalars computed by the caller in double precision, as torch does.
#include "common.h"

namespace {

#define ADAM_THREADS 256
#define ADAM_PER_THREAD 8
#define ADAM_CHUNK (ADAM_THREADS * ADAM_PER_THREAD)

struct AdamLaunch {
    EmdAdamArgs a;
    uint32_t first_block[EMD_ADAM_MAX_TENSORS + 1];
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const EmdAdamTensor& t) {
    m = fmaf(t.one_minus_beta1, g - m, m);
    v = fmaf(t.one_minus_beta2 * g, g, v * t.beta2);
    const float denom = sqrtf(v) / t.bias_correction2_sqrt + t.eps;
    p = fmaf(-t.step_size, m / denom, p);
}

__global__ void __launch_bounds__(ADAM_THREADS) k_adam(AdamLaunch L) {
    int ti = 0;
#pragma unroll 1
    while (ti + 1 < L.a.num_tensors && blockIdx.x >= L.first_block[ti + 1]) ti++;
    EmdAdamTensor t = L.a.tensors[ti];
    if (t.step_dev) {                      // capturable: the step-dependent scalars from the device-resident step count and learning rate
        const float st = t.step_dev[0], b1 = 1.f - t.one_minus_beta1;
        t.bias_correction2_sqrt = sqrtf(1.f - powf(t.beta2, st));
        t.step_size = t.lr_dev[0] / (1.f - powf(b1, st));
    }
    const int64_t base = (int64_t)(blockIdx.x - L.first_block[ti]) * ADAM_CHUNK;
    const bool vec = ((((uintptr_t)t.param) | ((uintptr_t)t.grad) | ((uintptr_t)t.exp_avg) | ((uintptr_t)t.exp_avg_sq)) & 15) == 0;
#pragma unroll
    for (int r = 0; r < ADAM_PER_THREAD / 4; r++) {
        const int64_t i = base + ((int64_t)r * ADAM_THREADS + threadIdx.x) * 4;
        if (i >= t.numel) continue;
        if (vec && i + 3 < t.numel) {
            float4 p = *reinterpret_cast<float4*>(t.param + i);
            const float4 g = *reinterpret_cast<const float4*>(t.grad + i);
            float4 m = *reinterpret_cast<float4*>(t.exp_avg + i);
            float4 v = *reinterpret_cast<float4*>(t.exp_avg_sq + i);
            adam_one(p.x, g.x, m.x, v.x, t);
            adam_one(p.y, g.y, m.y, v.y, t);
            adam_one(p.z, g.z, m.z, v.z, t);
            adam_one(p.w, g.w, m.w, v.w, t);
            *reinterpret_cast<float4*>(t.param + i) = p;
            *reinterpret_cast<float4*>(t.exp_avg + i) = m;
            *reinterpret_cast<float4*>(t.exp_avg_sq + i) = v;
        } else {
            for (int64_t j = i; j < i + 4 && j < t.numel; j++) {
                float p = t.param[j], m = t.exp_avg[j], v = t.exp_avg_sq[j];
                adam_one(p, t.grad[j], m, v, t);
                t.param[j] = p; t.exp_avg[j] = m; t.exp_avg_sq[j] = v;
            }
        }
    }
}

}  // namespace

extern "C" int emd_adam_step(const EmdAdamArgs* a, void* hip_stream) {
    if (!a) { emd_set_error("adam_step: null args"); return EMD_ERR_INVALID; }
    if (a->num_tensors < 0 || a->num_tensors > EMD_ADAM_MAX_TENSORS) {
        emd_set_error("adam_step: num_tensors %d outside [0, %d]", a->num_tensors, EMD_ADAM_MAX_TENSORS); return EMD_ERR_INVALID;
    }
    AdamLaunch L;
    L.a = *a;
    uint64_t blocks = 0;
    for (int i = 0; i < a->num_tensors; i++) {
        const EmdAdamTensor& t = a->tensors[i];
        if (t.numel < 0 || (t.numel > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) {
            emd_set_error("adam_step: tensor %d has a null pointer or a negative size", i); return EMD_ERR_INVALID;
        }
        if (t.step_dev && !t.lr_dev) { emd_set_error("adam_step: tensor %d: step_dev needs lr_dev", i); return EMD_ERR_INVALID; }
        if (!t.step_dev && !(t.bias_correction2_sqrt > 0.f)) { emd_set_error("adam_step: tensor %d: bias_correction2_sqrt must be > 0", i); return EMD_ERR_INVALID; }
        L.first_block[i] = (uint32_t)blocks;
        blocks += (uint64_t)((t.numel + ADAM_CHUNK - 1) / ADAM_CHUNK);
    }
    L.first_block[a->num_tensors] = (uint32_t)blocks;
    if (blocks == 0) return EMD_OK;
    if (blocks > 0x7fffffffull) { emd_set_error("adam_step: too many elements for one launch"); return EMD_ERR_INVALID; }
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(ADAM_THREADS), 0, (hipStream_t)hip_stream, L);
    EMD_LAUNCH_CHECK();
    return EMD_OK;
}

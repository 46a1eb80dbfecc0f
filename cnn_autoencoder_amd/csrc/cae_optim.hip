// One clip + Adam step over ALL parameters of a training step in two launches.
//
// Reference: the per-module loop of train_cae_ms.py:221-230 -- for every optimiser `clip_grad_norm_(params, 1.0)` then
// `opt.step()` (torch.optim.Adam, train_cae_ms.py:529-655: one optimiser per module plus the `_aux` optimiser of the
// quantiles).  As torch ops that is ~18 small launches per optimiser (norms, clip coefficient, the foreach Adam chain): at
// batch 16 a quarter of the step's launches.  Here a "group" is one optimiser: kernel 1 writes one partial sum of squares
// per 2048-element chunk, kernel 2 sums its group's partials in a fixed order (deterministic, unlike atomics), forms the
// clip coefficient min(1, max_norm / (norm + 1e-6)) and applies torch.optim.Adam's update (no amsgrad) to its chunk.
#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

using namespace cae;

namespace {

constexpr int kChunk = 2048;

struct OptimArgs {
    float *p[CAE_OPTIM_MAX_TENSORS];
    const float *g[CAE_OPTIM_MAX_TENSORS];
    float *m[CAE_OPTIM_MAX_TENSORS];
    float *v[CAE_OPTIM_MAX_TENSORS];
    int n[CAE_OPTIM_MAX_TENSORS];
    int chunk0[CAE_OPTIM_MAX_TENSORS + 1];  // first chunk of tensor t (tensors are ordered by group)
    unsigned char group[CAE_OPTIM_MAX_TENSORS];
    int gchunk0[CAE_OPTIM_MAX_GROUPS + 1];  // first chunk of group g
    float lr[CAE_OPTIM_MAX_GROUPS], beta1[CAE_OPTIM_MAX_GROUPS], beta2[CAE_OPTIM_MAX_GROUPS], eps[CAE_OPTIM_MAX_GROUPS];
    float bc1[CAE_OPTIM_MAX_GROUPS], bc2_sqrt[CAE_OPTIM_MAX_GROUPS], wd[CAE_OPTIM_MAX_GROUPS], max_norm[CAE_OPTIM_MAX_GROUPS];
    int ntensors, ngroups;
};

__device__ __forceinline__ int tensor_of_chunk(const OptimArgs &a, int chunk) {
    int t = 0;
    while (t + 1 < a.ntensors && a.chunk0[t + 1] <= chunk) ++t;
    return t;
}

__device__ __forceinline__ float block_sum(float v, float *red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];  // fixed order
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(256) sqnorm_chunks_kernel(const OptimArgs a, float *partial) {
    __shared__ float red[4];
    const int chunk = blockIdx.x, t = tensor_of_chunk(a, chunk);
    const int base = (chunk - a.chunk0[t]) * kChunk, n = a.n[t];
    const float *g = a.g[t];
    float s = 0.0f;
    for (int i = base + threadIdx.x; i < base + kChunk && i < n; i += 256) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[chunk] = s;
}

__global__ void __launch_bounds__(256) clip_adam_kernel(const OptimArgs a, const float *partial) {
    __shared__ float red[4];
    const int chunk = blockIdx.x, t = tensor_of_chunk(a, chunk);
    const int gr = a.group[t];
    float s = 0.0f;
    for (int c = a.gchunk0[gr] + threadIdx.x; c < a.gchunk0[gr + 1]; c += 256) s += partial[c];
    const float norm = sqrtf(block_sum(s, red));
    // torch.nn.utils.clip_grad_norm_: coefficient max_norm / (total_norm + 1e-6), clamped to 1
    const float coef = a.max_norm[gr] > 0.0f ? fminf(a.max_norm[gr] / (norm + 1e-6f), 1.0f) : 1.0f;
    const float lr = a.lr[gr], b1 = a.beta1[gr], b2 = a.beta2[gr], eps = a.eps[gr], wd = a.wd[gr];
    const float step_size = lr / a.bc1[gr], bc2s = a.bc2_sqrt[gr];
    const int base = (chunk - a.chunk0[t]) * kChunk, n = a.n[t];
    float *p = a.p[t], *m = a.m[t], *v = a.v[t];
    const float *g = a.g[t];
    for (int i = base + threadIdx.x; i < base + kChunk && i < n; i += 256) {
        float gi = g[i] * coef;
        const float pi = p[i];
        if (wd != 0.0f) gi += wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);  // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
}

}  // namespace

extern "C" int cae_t_clip_adam(int ntensors, float *const *params, const float *const *grads, float *const *exp_avg,
                               float *const *exp_avg_sq, const int *numel, const int *group, int ngroups, const float *lr,
                               const float *beta1, const float *beta2, const float *eps, const float *weight_decay,
                               const float *max_norm, const int *step, float *partial_ws, size_t partial_elems,
                               void *stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !group || !lr || !beta1 || !beta2 || !eps || !weight_decay ||
        !max_norm || !step || !partial_ws)
        return fail(CAE_ERR_ARG, "NULL argument");
    if (ntensors < 1 || ntensors > CAE_OPTIM_MAX_TENSORS || ngroups < 1 || ngroups > CAE_OPTIM_MAX_GROUPS)
        return fail(CAE_ERR_UNSUPPORTED, "at most %d tensors in %d groups per call", CAE_OPTIM_MAX_TENSORS, CAE_OPTIM_MAX_GROUPS);
    OptimArgs a{};
    a.ntensors = ntensors;
    a.ngroups = ngroups;
    int chunks = 0, prev = 0;
    for (int g = 0; g <= ngroups; ++g) a.gchunk0[g] = -1;
    for (int t = 0; t < ntensors; ++t) {
        if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || numel[t] < 1) return fail(CAE_ERR_ARG, "bad tensor %d", t);
        if (group[t] < prev || group[t] >= ngroups) return fail(CAE_ERR_ARG, "tensors must be ordered by group");
        prev = group[t];
        a.p[t] = params[t];
        a.g[t] = grads[t];
        a.m[t] = exp_avg[t];
        a.v[t] = exp_avg_sq[t];
        a.n[t] = numel[t];
        a.group[t] = (unsigned char)group[t];
        a.chunk0[t] = chunks;
        if (a.gchunk0[group[t]] < 0) a.gchunk0[group[t]] = chunks;
        chunks += (numel[t] + kChunk - 1) / kChunk;
    }
    a.chunk0[ntensors] = chunks;
    a.gchunk0[ngroups] = chunks;
    for (int g = ngroups - 1; g >= 0; --g)
        if (a.gchunk0[g] < 0) a.gchunk0[g] = a.gchunk0[g + 1];  // (a group without tensors)
    if ((size_t)chunks > partial_elems) return fail(CAE_ERR_ARG, "partial-sum workspace of %zu floats, %d needed", partial_elems, chunks);
    for (int g = 0; g < ngroups; ++g) {
        if (step[g] < 1) return fail(CAE_ERR_ARG, "step counts start at 1");
        a.lr[g] = lr[g];
        a.beta1[g] = beta1[g];
        a.beta2[g] = beta2[g];
        a.eps[g] = eps[g];
        a.wd[g] = weight_decay[g];
        a.max_norm[g] = max_norm[g];
        a.bc1[g] = (float)(1.0 - std::pow((double)beta1[g], step[g]));
        a.bc2_sqrt[g] = (float)std::sqrt(1.0 - std::pow((double)beta2[g], step[g]));
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sqnorm_chunks_kernel, dim3(chunks), dim3(256), 0, st, a, partial_ws);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(chunks), dim3(256), 0, st, a, (const float *)partial_ws);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

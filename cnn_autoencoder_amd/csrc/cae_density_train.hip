// Training-mode density of the factorized entropy bottleneck as ONE forward and ONE backward kernel
// (include/cae_hip.h: cae_t_density_forward / cae_t_density_backward).
//
// Reference: compressai EntropyBottleneck.forward(training=True) as the reference's training loop calls it
// (models/tasks/_taskutils.py:95-108 -> _autoencoders.py:502): y~ = y + U(-1/2, 1/2), p = c(y~ + 1/2) - c(y~ - 1/2) with
// the per-channel cumulative-logit network, lower-bounded at 1e-9 with the LowerBound gradient rule
// (Appendix A.1).  Written with torch element-wise ops (entropy.py) one training step spent ~250 of its ~600 kernel
// launches here, each a few microseconds of work on 192 x B x 256 elements with ~7 us of dispatch gap behind it.
//
// One thread evaluates the network at y~ - 1/2 and y~ + 1/2 for its elements (4 layers of D x D, everything in
// registers), the backward kernel re-evaluates it and back-propagates by hand; the 58 parameter gradients of a channel
// are accumulated per thread, reduced over the block (shuffles, LDS) and added to the output with one atomic per
// parameter and block.  Gradients are returned with respect to the RAW parameters (softplus of the matrices, tanh of the
// factors differentiated here), so no parameter-sized autograd graph remains.
// Built for the reference's filters = (3, 3, 3, 3) (its EntropyBottleneck default); other shapes keep the torch-op path.
#include <hip/hip_runtime.h>

#include "cae_hip.h"
#include "cae_internal.hpp"
#include "cae_launch.hpp"

#include <algorithm>

using namespace cae;

namespace {

template <int D, int K>
struct Lay {  // raw parameter block of one channel: matrices | biases | factors
    static constexpr int NM = D + (K - 1) * D * D + D, NB = K * D + 1, NF = K * D, NP = NM + NB + NF;
    __host__ __device__ static constexpr int din(int i) { return i == 0 ? 1 : D; }
    __host__ __device__ static constexpr int dout(int i) { return i == K ? 1 : D; }
    __host__ __device__ static constexpr int m_off(int i) { return i == 0 ? 0 : D + (i - 1) * D * D; }
    __host__ __device__ static constexpr int b_off(int i) { return NM + i * D; }
    __host__ __device__ static constexpr int f_off(int i) { return NM + NB + i * D; }
};

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // torch's threshold
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// transformed parameters of channel c into LDS: softplus(matrix) | bias | tanh(factor)
template <int D, int K>
__device__ __forceinline__ void load_params(const float *raw, float *T) {
    using L = Lay<D, K>;
    for (int i = threadIdx.x; i < L::NP; i += blockDim.x) {
        const float v = raw[i];
        T[i] = i < L::NM ? softplus_f(v) : (i < L::NM + L::NB ? v : tanhf(v));
    }
}

// logits(v); KEEP: h[i] = input vector of layer i, tz[i] = tanh of layer i's pre-activation
template <int D, int K, bool KEEP>
__device__ __forceinline__ float logits(float v, const float *T, float (&h)[K + 1][D], float (&tz)[K][D]) {
    using L = Lay<D, K>;
    float in[D];
    in[0] = v;
    float z0 = 0.0f;
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        if (KEEP) {
#pragma unroll
            for (int k = 0; k < L::din(i); ++k) h[i][k] = in[k];
        }
        float out[D];
#pragma unroll
        for (int j = 0; j < L::dout(i); ++j) {
            float z = T[L::b_off(i) + j];
#pragma unroll
            for (int k = 0; k < L::din(i); ++k) z = fmaf(T[L::m_off(i) + j * L::din(i) + k], in[k], z);
            if (i < K) {
                const float t = tanhf(z);
                if (KEEP) tz[i][j] = t;
                out[j] = fmaf(T[L::f_off(i) + j], t, z);
            } else {
                z0 = z;
            }
        }
        if (i < K) {
#pragma unroll
            for (int j = 0; j < D; ++j) in[j] = out[j];
        }
    }
    return z0;
}

// back-propagation of one evaluation: gl = d loss / d logit; adds the parameter gradients (w.r.t. the TRANSFORMED
// parameters) into g[NP]; -> d loss / d v
template <int D, int K>
__device__ __forceinline__ float backprop(float gl, const float *T, const float (&h)[K + 1][D], const float (&tz)[K][D],
                                          float (&g)[Lay<D, K>::NP]) {
    using L = Lay<D, K>;
    float dout[D];
    // layer K: z = M h + b
    g[L::b_off(K)] += gl;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        g[L::m_off(K) + k] += gl * h[K][k];
        dout[k] = T[L::m_off(K) + k] * gl;
    }
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        float dz[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float t = tz[i][j];
            g[L::f_off(i) + j] += dout[j] * t;
            dz[j] = dout[j] * fmaf(T[L::f_off(i) + j], 1.0f - t * t, 1.0f);
            g[L::b_off(i) + j] += dz[j];
        }
        float din[D];
#pragma unroll
        for (int k = 0; k < L::din(i); ++k) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                g[L::m_off(i) + j * L::din(i) + k] += dz[j] * h[i][k];
                s = fmaf(T[L::m_off(i) + j * L::din(i) + k], dz[j], s);
            }
            din[k] = s;
        }
#pragma unroll
        for (int k = 0; k < L::din(i); ++k) dout[k] = din[k];
    }
    return dout[0];
}

template <int D, int K>
__global__ void __launch_bounds__(256) density_fwd_kernel(const float *y, const float *noise, const float *raw, int N, int C,
                                                          int HW, int plain, float bound, float *out, float *lik) {
    using L = Lay<D, K>;
    __shared__ float T[L::NP];
    const int c = blockIdx.x;
    load_params<D, K>(raw + (size_t)c * L::NP, T);
    __syncthreads();
    const long total = (long)N * HW;
    float h[K + 1][D], tz[K][D];
    for (long e = (long)blockIdx.y * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.y * blockDim.x) {
        const long n = e / HW, hw = e - n * HW;
        const size_t idx = ((size_t)n * C + c) * HW + hw;
        const float v = y[idx] + (noise ? noise[idx] : 0.0f);
        const float lo = logits<D, K, false>(v - 0.5f, T, h, tz), up = logits<D, K, false>(v + 0.5f, T, h, tz);
        float p;
        if (plain) {
            p = sigmoid_f(up) - sigmoid_f(lo);
        } else {
            const float sum = lo + up, s = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
            p = fabsf(sigmoid_f(s * up) - sigmoid_f(s * lo));
        }
        out[idx] = v;
        lik[idx] = fmaxf(p, bound);
    }
}

template <int D, int K>
__global__ void __launch_bounds__(256) density_bwd_kernel(const float *vout, const float *g_lik, const float *g_out,
                                                          const float *raw, int N, int C, int HW, int plain, float bound,
                                                          float *g_y, float *g_raw) {
    using L = Lay<D, K>;
    __shared__ float T[L::NP];
    __shared__ float red[4][L::NP];
    const int c = blockIdx.x;
    load_params<D, K>(raw + (size_t)c * L::NP, T);
    __syncthreads();
    const long total = (long)N * HW;
    float g[L::NP];
#pragma unroll
    for (int i = 0; i < L::NP; ++i) g[i] = 0.0f;
    float hl[K + 1][D], tl[K][D], hu[K + 1][D], tu[K][D];
    for (long e = (long)blockIdx.y * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.y * blockDim.x) {
        const long n = e / HW, hw = e - n * HW;
        const size_t idx = ((size_t)n * C + c) * HW + hw;
        const float v = vout[idx];
        const float lo = logits<D, K, true>(v - 0.5f, T, hl, tl), up = logits<D, K, true>(v + 0.5f, T, hu, tu);
        float p, dpu, dpl;  // p and its derivatives with respect to the two logits
        if (plain) {
            const float su = sigmoid_f(up), sl = sigmoid_f(lo);
            p = su - sl;
            dpu = su * (1.0f - su);
            dpl = -sl * (1.0f - sl);
        } else {
            const float sum = lo + up, s = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
            const float su = sigmoid_f(s * up), sl = sigmoid_f(s * lo), q = su - sl;
            const float sq = q > 0.0f ? 1.0f : (q < 0.0f ? -1.0f : 0.0f);
            p = fabsf(q);
            dpu = sq * s * su * (1.0f - su);
            dpl = -sq * s * sl * (1.0f - sl);
        }
        float gp = g_lik[idx];
        if (!(p >= bound || gp < 0.0f)) gp = 0.0f;  // LowerBound: gradient passes where p >= bound or it would raise p
        float gv = 0.0f;
        gv += backprop<D, K>(gp * dpu, T, hu, tu, g);
        gv += backprop<D, K>(gp * dpl, T, hl, tl, g);
        g_y[idx] = gv + (g_out ? g_out[idx] : 0.0f);
    }
    // block reduction of the parameter gradients, chain rule to the raw parameters, one atomic per parameter
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < L::NP; ++i) {
        float s = g[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L::NP; i += blockDim.x) {
        float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        const float r = raw[(size_t)c * L::NP + i];
        if (i < L::NM)
            s *= r > 20.0f ? 1.0f : sigmoid_f(r);  // d softplus
        else if (i >= L::NM + L::NB)
            s *= 1.0f - T[i] * T[i];  // d tanh
        atomicAdd(g_raw + (size_t)c * L::NP + i, s);
    }
}

// NonNegativeParametrizer of the GDN parameters (compressai.ops.parametrizers) as one kernel each way:
// out = max(x, bound)^2 - pedestal;  g_x = g 2 max(x, bound) where x >= bound or that product is negative (LowerBound rule)
__global__ void reparam_fwd_kernel(const float *x, long n, float bound, float pedestal, float *out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float c = fmaxf(x[i], bound);
        out[i] = c * c - pedestal;
    }
}

__global__ void reparam_bwd_kernel(const float *x, const float *g, long n, float bound, float *gx) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float xv = x[i], c = fmaxf(xv, bound);
        const float gi = g[i] * (2.0f * c);  // gradient arriving at the LowerBound
        gx[i] = (xv >= bound || gi < 0.0f) ? gi : 0.0f;
    }
}

unsigned blocks_per_channel(long elems) {
    return (unsigned)std::min<long>(std::max<long>((elems + 2047) / 2048, 1), 32);
}

}  // namespace

extern "C" {

int cae_t_density_params(int filters_d, int n_filters) {
    if (filters_d == 3 && n_filters == 4) return Lay<3, 4>::NP;
    return 0;  // shape not built: the caller keeps its element-wise path
}

int cae_t_density_forward(const float *y, const float *noise, const float *raw_params, int n, int channels, int hw, int plain,
                          float bound, float *out, float *lik, void *stream) {
    if (!y || !raw_params || !out || !lik) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || channels < 1 || hw < 1) return fail(CAE_ERR_ARG, "bad tensor shape");
    hipLaunchKernelGGL((density_fwd_kernel<3, 4>), dim3(channels, blocks_per_channel((long)n * hw)), dim3(256), 0,
                       (hipStream_t)stream, y, noise, raw_params, n, channels, hw, plain, bound, out, lik);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_density_backward(const float *out, const float *g_lik, const float *g_out, const float *raw_params, int n,
                           int channels, int hw, int plain, float bound, float *g_y, float *g_raw_params, void *stream) {
    if (!out || !g_lik || !raw_params || !g_y || !g_raw_params) return fail(CAE_ERR_ARG, "NULL argument");
    if (n < 1 || channels < 1 || hw < 1) return fail(CAE_ERR_ARG, "bad tensor shape");
    HIP_TRY(hipMemsetAsync(g_raw_params, 0, (size_t)channels * Lay<3, 4>::NP * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL((density_bwd_kernel<3, 4>), dim3(channels, blocks_per_channel((long)n * hw)), dim3(256), 0,
                       (hipStream_t)stream, out, g_lik, g_out, raw_params, n, channels, hw, plain, bound, g_y, g_raw_params);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_reparam_forward(const float *x, long n, float bound, float pedestal, float *out, void *stream) {
    if (!x || !out || n < 1) return fail(CAE_ERR_ARG, "bad argument");
    const unsigned grid = (unsigned)std::min<long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, bound, pedestal, out);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

int cae_t_reparam_backward(const float *x, const float *g, long n, float bound, float *gx, void *stream) {
    if (!x || !g || !gx || n < 1) return fail(CAE_ERR_ARG, "bad argument");
    const unsigned grid = (unsigned)std::min<long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, g, n, bound, gx);
    HIP_TRY(hipGetLastError());
    return CAE_OK;
}

}  // extern "C"

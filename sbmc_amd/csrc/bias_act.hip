// Fused bias + activation for the per-sample 1x1 ConvChains of the kernel-predicting CNN.
//
// The reference builds those chains from nn.Conv2d(1x1) + ReLU / LeakyReLU (sbmc/modules.py:
// 154-175, used at sbmc/models.py:79-102).  On MI355X the 1x1 convolution is a plain batched
// GEMM on the planar activations (y[b] = W @ x[b], rocBLAS / hipBLASLt); what is left around it
// is memory-bound glue that PyTorch runs as separate passes over the [B, C, H*W] tensor (bias
// broadcast, activation, activation backward, bias-gradient reduction).  These two kernels do
// that glue in ONE pass per direction, in place:
//   forward : y = act(y + bias[c])
//   backward: gx = gy * act'(y);  partial[b, c, chunk] = sum of gx over the workgroup's pixels
//             (the caller adds the partials up: no atomics, deterministic; y is the OUTPUT: for
//             relu and leaky_relu the sign of the output equals the sign of the input)
// HBM-bound: 8 bytes/element forward, 12 bytes/element backward; float4 accesses.
#include "common.hpp"
#include "../../include/sbmc_hip.h"

namespace sbmc {

__device__ __forceinline__ float act_fwd(float v, float slope) { return v > 0.f ? v : v * slope; }

// grid: (chunks, C, B); each workgroup streams a contiguous chunk of one (b, c) plane
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                          size_t hw, int C, float slope, int linear) {
    const int c = blockIdx.y, b = blockIdx.z;
    float* plane = y + ((size_t)b * C + c) * hw;
    const float bv = bias[c];
    const size_t n4 = hw / 4;
    float4* p4 = reinterpret_cast<float4*>(plane);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p4[i];
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        if (!linear) {
            v.x = act_fwd(v.x, slope); v.y = act_fwd(v.y, slope);
            v.z = act_fwd(v.z, slope); v.w = act_fwd(v.w, slope);
        }
        p4[i] = v;
    }
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < hw; i += blockDim.x) {
            const float v = plane[i] + bv;
            plane[i] = linear ? v : act_fwd(v, slope);
        }
    }
}

__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                          float* __restrict__ gx, float* __restrict__ partial,
                                                          size_t hw, int C, float slope, int linear) {
    __shared__ float red[4];
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t off = ((size_t)b * C + c) * hw;
    const size_t n4 = hw / 4;
    const float4* g4 = reinterpret_cast<const float4*>(gy + off);
    const float4* y4 = reinterpret_cast<const float4*>(y + off);
    float4* o4 = reinterpret_cast<float4*>(gx + off);
    const bool store = !(linear && gx == gy);     // linear and in place: gx IS gy, only the sums are needed
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 g = g4[i];
        if (!linear) {
            const float4 v = y4[i];
            g.x = v.x > 0.f ? g.x : g.x * slope; g.y = v.y > 0.f ? g.y : g.y * slope;
            g.z = v.z > 0.f ? g.z : g.z * slope; g.w = v.w > 0.f ? g.w : g.w * slope;
        }
        if (store) o4[i] = g;
        acc += (g.x + g.y) + (g.z + g.w);
    }
    if (blockIdx.x == 0) {
        for (size_t i = n4 * 4 + threadIdx.x; i < hw; i += blockDim.x) {
            float g = gy[off + i];
            if (!linear) g = y[off + i] > 0.f ? g : g * slope;
            if (store) gx[off + i] = g;
            acc += g;
        }
    }
    // wave reduction, then one partial sum per workgroup (no atomics: deterministic)
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_down(acc, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((size_t)b * C + c) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Per-sample chains whose first layer sees [sample features ; pixel context] (reference
// sbmc/models.py:147-153,171-177,196-199: th.cat([f, propagated], 1) for every sample): the
// layer is linear, so W [f ; ctx] = W_f f + W_c ctx, and the context term W_c ctx + bias is the
// same for all S samples of a pixel.  It is computed ONCE per pixel (t[b, c, p]) and added here,
// together with the activation, in one in-place pass over the per-sample product:
//   forward : y[b,s,c,p] = act(y[b,s,c,p] + t[b,c,p*tp] + bias[c])        (tp = 0: t is per image)
//   backward: gx = gy * act'(y);  gt[b,c,p] = sum_s gx;  partial[b,c,chunk] = sum_{s,p in chunk} gx
__global__ __launch_bounds__(256) void ctx_act_fwd_kernel(float* __restrict__ y, const float* __restrict__ t,
                                                         const float* __restrict__ bias, size_t hw, int C,
                                                         int S, int t_per_pixel, float slope, int linear) {
    const int c = blockIdx.y, b = blockIdx.z;
    const float bv = bias[c];
    const size_t n4 = hw / 4;
    const float* tplane = t + ((size_t)b * C + c) * (t_per_pixel ? hw : 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 tv;
        if (t_per_pixel) tv = reinterpret_cast<const float4*>(tplane)[i];
        else tv = make_float4(tplane[0], tplane[0], tplane[0], tplane[0]);
        tv.x += bv; tv.y += bv; tv.z += bv; tv.w += bv;
        for (int s = 0; s < S; ++s) {
            float4* p4 = reinterpret_cast<float4*>(y + (((size_t)b * S + s) * C + c) * hw) + i;
            float4 v = *p4;
            v.x += tv.x; v.y += tv.y; v.z += tv.z; v.w += tv.w;
            if (!linear) {
                v.x = act_fwd(v.x, slope); v.y = act_fwd(v.y, slope);
                v.z = act_fwd(v.z, slope); v.w = act_fwd(v.w, slope);
            }
            *p4 = v;
        }
    }
}

__global__ __launch_bounds__(256) void ctx_act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                         float* __restrict__ gx, float* __restrict__ gt,
                                                         float* __restrict__ partial, size_t hw, int C, int S,
                                                         int t_per_pixel, float slope, int linear) {
    __shared__ float red[4];
    const int c = blockIdx.y, b = blockIdx.z;
    const size_t n4 = hw / 4;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < S; ++s) {
            const size_t off = (((size_t)b * S + s) * C + c) * hw;
            float4 g = reinterpret_cast<const float4*>(gy + off)[i];
            if (!linear) {
                const float4 v = reinterpret_cast<const float4*>(y + off)[i];
                g.x = v.x > 0.f ? g.x : g.x * slope; g.y = v.y > 0.f ? g.y : g.y * slope;
                g.z = v.z > 0.f ? g.z : g.z * slope; g.w = v.w > 0.f ? g.w : g.w * slope;
            }
            reinterpret_cast<float4*>(gx + off)[i] = g;
            sum.x += g.x; sum.y += g.y; sum.z += g.z; sum.w += g.w;
        }
        if (t_per_pixel) reinterpret_cast<float4*>(gt + ((size_t)b * C + c) * hw)[i] = sum;
        acc += (sum.x + sum.y) + (sum.z + sum.w);
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_down(acc, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((size_t)b * C + c) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace sbmc

using namespace sbmc;

static unsigned chunks_for(size_t hw, int planes) {
    // enough workgroups to fill 256 CUs several times over, at least ~16 KB per workgroup
    size_t per = (hw / 4 + 255) / 256;          // float4 iterations if one workgroup took the plane
    size_t want = per / 4 ? per / 4 : 1;        // >= 4 iterations per thread
    size_t cap = (size_t)(8192 / (planes > 0 ? planes : 1)) + 1;
    if (want > cap) want = cap;
    if (want > 65535) want = 65535;
    return (unsigned)want;
}

extern "C" int sbmc_bias_act_fwd_f32(float* y, const float* bias, int b, int c, long hw, int act,
                                     float slope, void* stream) {
    if (b < 0 || c < 0 || hw < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (b == 0 || c == 0 || hw == 0) return 0;
    // every (b, c) plane must start 16-byte aligned for the float4 path
    if (!y || !bias || c > 65535 || b > 65535 || hw % 4 || (uintptr_t)y % 16) return SBMC_HIP_EINVAL;
    const float s = act == 1 ? 0.f : slope;
    hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(chunks_for((size_t)hw, b * c), c, b), dim3(256), 0,
                       (hipStream_t)stream, y, bias, (size_t)hw, c, s, act == 0);
    return (int)hipGetLastError();
}

extern "C" int sbmc_bias_act_chunks(int b, int c, long hw) {
    if (b <= 0 || c <= 0 || hw <= 0) return 1;
    return (int)chunks_for((size_t)hw, b * c);
}

extern "C" int sbmc_bias_act_bwd_f32(const float* gy, const float* y, float* gx, float* partial,
                                     int b, int c, long hw, int act, float slope, void* stream) {
    if (b < 0 || c < 0 || hw < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (b == 0 || c == 0 || hw == 0) return 0;
    if (!gy || !y || !gx || !partial || c > 65535 || b > 65535 || hw % 4) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)y % 16 || (uintptr_t)gx % 16) return SBMC_HIP_EINVAL;
    const float s = act == 1 ? 0.f : slope;
    hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(chunks_for((size_t)hw, b * c), c, b), dim3(256), 0,
                       (hipStream_t)stream, gy, y, gx, partial, (size_t)hw, c, s, act == 0);
    return (int)hipGetLastError();
}

extern "C" int sbmc_ctx_act_fwd_f32(float* y, const float* t, const float* bias, int b, int s, int c, long hw,
                                    int t_per_pixel, int act, float slope, void* stream) {
    if (b < 0 || s < 1 || c < 0 || hw < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (b == 0 || c == 0 || hw == 0) return 0;
    if (!y || !t || !bias || c > 65535 || b > 65535 || hw % 4 || (uintptr_t)y % 16 || (uintptr_t)t % 16)
        return SBMC_HIP_EINVAL;
    const float sl = act == 1 ? 0.f : slope;
    hipLaunchKernelGGL(ctx_act_fwd_kernel, dim3(chunks_for((size_t)hw, b * c), c, b), dim3(256), 0,
                       (hipStream_t)stream, y, t, bias, (size_t)hw, c, s, t_per_pixel != 0, sl, act == 0);
    return (int)hipGetLastError();
}

extern "C" int sbmc_ctx_act_bwd_f32(const float* gy, const float* y, float* gx, float* gt, float* partial,
                                    int b, int s, int c, long hw, int t_per_pixel, int act, float slope,
                                    void* stream) {
    if (b < 0 || s < 1 || c < 0 || hw < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (b == 0 || c == 0 || hw == 0) return 0;
    if (!gy || !y || !gx || !partial || (t_per_pixel && !gt) || c > 65535 || b > 65535 || hw % 4)
        return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)y % 16 || (uintptr_t)gx % 16 || (t_per_pixel && (uintptr_t)gt % 16))
        return SBMC_HIP_EINVAL;
    const float sl = act == 1 ? 0.f : slope;
    hipLaunchKernelGGL(ctx_act_bwd_kernel, dim3(chunks_for((size_t)hw, b * c), c, b), dim3(256), 0,
                       (hipStream_t)stream, gy, y, gx, gt, partial, (size_t)hw, c, s, t_per_pixel != 0, sl,
                       act == 0);
    return (int)hipGetLastError();
}

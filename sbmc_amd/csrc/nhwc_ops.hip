// Channels-last (NHWC) forms of the memory-bound glue around the U-nets' 3x3 convolutions.
//
// MIOpen's fastest fp32 3x3 solvers on gfx950 (ConvAsmImplicitGemmGTCDynamic{Fwd,Bwd,Wrw}XdlopsNHWC) are
// NHWC-native: fed NCHW tensors they transpose every input and output (36 ms of a 527 ms training step at
// 1280x720, `batched_transpose_*`).  With the U-net's activations kept channels-last between its
// convolutions those transposes disappear -- provided the glue between the convolutions speaks NHWC too:
//   bias + ReLU / LeakyReLU behind every convolution (reference sbmc/modules.py:154-175)   -> bias_act_nhwc_*
//   bilinear x2 upsampling + concatenation with the skip connection (modules.py:300-320)   -> upcat_nhwc_*
// Same arithmetic as the planar kernels of bias_act.hip / resample.hip; a pixel's channels are contiguous,
// so a thread owns one float4 of channels of one pixel and neighbouring lanes neighbouring channels.
// HBM-bound, one pass per direction, no atomics (bias gradient: per-workgroup partial sums, added up by
// the caller).
#include "common.hpp"
#include "../../include/sbmc_hip.h"

namespace sbmc {

__device__ __forceinline__ float nact(float v, float slope) { return v > 0.f ? v : v * slope; }

// four consecutive channels of one pixel: 16 bytes of float or 8 bytes of _Float16 (fp16 activations under
// torch.autocast: half storage, fp32 interpolation arithmetic)
template <typename T> struct Quad;
template <> struct Quad<float> {
    static __device__ __forceinline__ float4 load(const float* p, size_t i) { return reinterpret_cast<const float4*>(p)[i]; }
    static __device__ __forceinline__ void store(float* p, size_t i, float4 v) { reinterpret_cast<float4*>(p)[i] = v; }
};
template <> struct Quad<_Float16> {
    using h4 = __attribute__((ext_vector_type(4))) _Float16;
    static __device__ __forceinline__ float4 load(const _Float16* p, size_t i) {
        const h4 h = reinterpret_cast<const h4*>(p)[i];
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    static __device__ __forceinline__ void store(_Float16* p, size_t i, float4 v) {
        h4 h;
        h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
        reinterpret_cast<h4*>(p)[i] = h;
    }
};


// y [pixels, C] in place; one thread = one float4 of channels of one pixel
__global__ __launch_bounds__(256) void bias_act_nhwc_fwd_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                               size_t total4, int c4n, float slope, int linear,
                                                               unsigned* __restrict__ amax) {
    float4* y4 = reinterpret_cast<float4*>(y);
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 bv = b4[i % (size_t)c4n];
        float4 v = y4[i];
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (!linear) { v.x = nact(v.x, slope); v.y = nact(v.y, slope); v.z = nact(v.z, slope); v.w = nact(v.w, slope); }
        y4[i] = v;
        m = amax4(m, v);
    }
    if (amax) amax_publish(m, amax);
}

// The same forward that also records one SIGN BIT per element (pre-activation > 0): thread i owns float4 number i,
// its 4 bits are nibble i % 8 of word i / 8 (eight neighbouring lanes combine theirs).  The backward then reads
// 1 bit instead of 32 per element for the activation adjoint: two streams instead of three.
__global__ __launch_bounds__(256) void bias_act_nhwc_fwd_signs_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                                     unsigned* __restrict__ signs, size_t total4,
                                                                     int c4n, float slope, unsigned* __restrict__ amax) {
    float4* y4 = reinterpret_cast<float4*>(y);
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    const size_t n8 = (total4 + 7) & ~(size_t)7;      // (the 8 lanes of a word run the same iterations)
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bool ok = i < total4;
        unsigned bits = 0;
        if (ok) {
            const float4 bv = b4[i % (size_t)c4n];
            float4 v = y4[i];
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            bits = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            v.x = nact(v.x, slope); v.y = nact(v.y, slope); v.z = nact(v.z, slope); v.w = nact(v.w, slope);
            y4[i] = v;
            m = amax4(m, v);
        }
        unsigned word = bits << (4 * (threadIdx.x & 7));
        word |= __shfl_xor(word, 1);
        word |= __shfl_xor(word, 2);
        word |= __shfl_xor(word, 4);
        if ((threadIdx.x & 7) == 0) signs[i >> 3] = word;
    }
    if (amax) amax_publish(m, amax);
}

// gx = gy * act'(y); partial[chunk, C] = this workgroup's sums of gx per channel.
// 256 threads = (256 / c4n) pixel lanes x c4n channel quads; c4n must divide 256.
// SG: `y` holds the forward's sign bits (bias_act_nhwc_fwd_signs_kernel) instead of its output.
// T: storage type of gy / gx (float, or _Float16 for half activations: fp32 arithmetic and sums, one rounding)
template <bool SG, typename T = float>
__global__ __launch_bounds__(256) void bias_act_nhwc_bwd_kernel(const T* __restrict__ gy, const float* __restrict__ y,
                                                               T* __restrict__ gx, float* __restrict__ partial,
                                                               size_t pixels, int c4n, float slope, int linear,
                                                               unsigned* __restrict__ amax) {
    __shared__ float4 red[256];
    unsigned m = 0;
    const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n, npl = 256 / c4n;
    const size_t per = (pixels + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per;
    const size_t p1 = p0 + per < pixels ? p0 + per : pixels;
    const bool store = !(linear && gx == gy);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t px = p0 + pl; px < p1; px += npl) {
        const size_t i = px * c4n + cq;
        float4 g = Quad<T>::load(gy, i);
        if (!linear) {
            if constexpr (SG) {
                const unsigned bits = reinterpret_cast<const unsigned*>(y)[i >> 3] >> (4 * (unsigned)(i & 7));
                g.x = (bits & 1u) ? g.x : g.x * slope; g.y = (bits & 2u) ? g.y : g.y * slope;
                g.z = (bits & 4u) ? g.z : g.z * slope; g.w = (bits & 8u) ? g.w : g.w * slope;
            } else {
                const float4 v = reinterpret_cast<const float4*>(y)[i];
                g.x = v.x > 0.f ? g.x : g.x * slope; g.y = v.y > 0.f ? g.y : g.y * slope;
                g.z = v.z > 0.f ? g.z : g.z * slope; g.w = v.w > 0.f ? g.w : g.w * slope;
            }
        }
        if (store) Quad<T>::store(gx, i, g);
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        m = amax4(m, g);
    }
    if (amax) amax_publish(m, amax);
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0) {
        for (int j = 1; j < npl; ++j) {
            const float4 o = red[j * c4n + cq];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * c4n + cq] = acc;
    }
}

// out[b, y, x, :cu] = bilinear x2 of coarse[b, :, :, :cu] (align_corners = False), out[b, y, x, cu:] = left
// Row-slab form (sbmc_amd/dist.py): coarse holds hc = top + h + bot rows, the first `top` / last `bot` (0 or 1)
// being the neighbouring slabs' edge rows; out / left are the 2h fine rows of this slab (see resample.hip).
template <typename T>
__global__ __launch_bounds__(256) void upcat_nhwc_fwd_kernel(const T* __restrict__ coarse, const T* __restrict__ left,
                                                            T* __restrict__ out, int cu4, int cl4, int hc, int w,
                                                            int top, int bot, size_t total4) {
    const int h = hc - top - bot;
    const int W = 2 * w, H = 2 * h, ct4 = cu4 + cl4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % ct4);
        size_t rest = idx / ct4;
        const int x = (int)(rest % W);
        rest /= W;
        const int y = (int)(rest % H);
        const size_t b = rest / H;
        float4 v;
        if (q >= cu4) {
            v = Quad<T>::load(left, ((b * H + y) * W + x) * (size_t)cl4 + (q - cu4));
        } else {
            // source coordinate 0.5 * dst - 0.25 clamped at 0: even dst -> (i-1: .25, i: .75), odd -> (i: .75, i+1: .25)
            int r0, r1, c0, c1;
            float ly, lx;                                     // weights of r1 / c1
            const int yf = y + 2 * top;                       // row of the upsampled (padded) coarse map
            const int i = yf >> 1, j = x >> 1;
            if (yf == 0) { r0 = r1 = 0; ly = 0.f; } else if (yf & 1) { r0 = i; r1 = i + 1 < hc ? i + 1 : i; ly = 0.25f; } else { r0 = i - 1; r1 = i; ly = 0.75f; }
            if (x == 0) { c0 = c1 = 0; lx = 0.f; } else if (x & 1) { c0 = j; c1 = j + 1 < w ? j + 1 : j; lx = 0.25f; } else { c0 = j - 1; c1 = j; lx = 0.75f; }
            const size_t cz = (b * hc) * (size_t)w * cu4 + q;
            const float4 a = Quad<T>::load(coarse, cz + ((size_t)r0 * w + c0) * cu4);
            const float4 bq = Quad<T>::load(coarse, cz + ((size_t)r0 * w + c1) * cu4);
            const float4 c = Quad<T>::load(coarse, cz + ((size_t)r1 * w + c0) * cu4);
            const float4 d = Quad<T>::load(coarse, cz + ((size_t)r1 * w + c1) * cu4);
            const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
            v.x = w00 * a.x + w01 * bq.x + w10 * c.x + w11 * d.x;
            v.y = w00 * a.y + w01 * bq.y + w10 * c.y + w11 * d.y;
            v.z = w00 * a.z + w01 * bq.z + w10 * c.z + w11 * d.z;
            v.w = w00 * a.w + w01 * bq.w + w10 * c.w + w11 * d.w;
        }
        Quad<T>::store(out, idx, v);
    }
}

// gcoarse[b, i, j, :] = sum over the <= 4x4 fine neighbours, separable weights {.25, .75, .75, .25}, a partner
// that fell off the image was clamped onto the border row / column in the forward: its weight comes back there.
// gleft (may be null) = gout[..., cu:] as a contiguous tensor.
template <typename T>
__global__ __launch_bounds__(256) void upcat_nhwc_bwd_kernel(const T* __restrict__ gout, T* __restrict__ gcoarse,
                                                            int cu4, int cl4, int hc, int w, int top, int bot,
                                                            size_t total4) {
    const int h = hc - top - bot;
    const int W = 2 * w, H = 2 * h, ct4 = cu4 + cl4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cu4);
        size_t rest = idx / cu4;
        const int j = (int)(rest % w);
        rest /= w;
        const int i = (int)(rest % hc);                      // coarse row (of the hc padded rows)
        const size_t b = rest / hc;
        float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        if (i == 0 && top == 0) { wy[0] = 0.f; wy[1] = 1.f; }            // true image borders only
        if (i == hc - 1 && bot == 0) { wy[3] = 0.f; wy[2] = 1.f; }
        if (j == 0) { wx[0] = 0.f; wx[1] = 1.f; }
        if (j == w - 1) { wx[3] = 0.f; wx[2] = 1.f; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t g = (b * H) * (size_t)W * ct4 + q;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            if (wy[dy] == 0.f) continue;
            const int yy = 2 * i - 1 + dy - 2 * top;           // fine row of this slab; outside: another slab's share
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                if (wx[dx] == 0.f) continue;
                const int xx = 2 * j - 1 + dx;
                const float4 v = Quad<T>::load(gout, g + ((size_t)yy * W + xx) * ct4);
                const float wt = wy[dy] * wx[dx];
                acc.x += wt * v.x; acc.y += wt * v.y; acc.z += wt * v.z; acc.w += wt * v.w;
            }
        }
        Quad<T>::store(gcoarse, idx, acc);
    }
}

// The coarse half of the same adjoint (whole frames: no slab rows) with the adjoint of the bias + activation pass that
// PRODUCED the coarse map behind it (a convolution chain's activated output that only the upsampling reads): gcoarse =
// gather * act'(z) from the producer's sign words, the bias gradient's per-workgroup partial sums, max |gcoarse|.
// 256 threads = (256 / cu4) lanes of coarse pixels x cu4 channel quads; cu4 divides 256.
__global__ __launch_bounds__(256) void upcat_nhwc_bwd_adj_kernel(const float* __restrict__ gout, float* __restrict__ gcoarse,
                                                                const unsigned* __restrict__ signs, float slope,
                                                                float* __restrict__ partial, unsigned* __restrict__ amax,
                                                                int cu4, int cl4, int hc, int w, size_t pixels) {
    __shared__ float4 red[256];
    unsigned m = 0;
    const int W = 2 * w, H = 2 * hc, ct4 = cu4 + cl4;
    const int q = threadIdx.x % cu4, pl = threadIdx.x / cu4, npl = 256 / cu4;
    const size_t per = (pixels + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per;
    const size_t p1 = p0 + per < pixels ? p0 + per : pixels;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t px = p0 + pl; px < p1; px += npl) {
        const int j = (int)(px % w);
        const size_t rest = px / w;
        const int i = (int)(rest % hc);
        const size_t b = rest / hc;
        float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f}, wx[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        if (i == 0) { wy[0] = 0.f; wy[1] = 1.f; }
        if (i == hc - 1) { wy[3] = 0.f; wy[2] = 1.f; }
        if (j == 0) { wx[0] = 0.f; wx[1] = 1.f; }
        if (j == w - 1) { wx[3] = 0.f; wx[2] = 1.f; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t g = (b * H) * (size_t)W * ct4 + q;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            if (wy[dy] == 0.f) continue;
            const int yy = 2 * i - 1 + dy;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                if (wx[dx] == 0.f) continue;
                const int xx = 2 * j - 1 + dx;
                const float4 v = Quad<float>::load(gout, g + ((size_t)yy * W + xx) * ct4);
                const float wt = wy[dy] * wx[dx];
                acc.x += wt * v.x; acc.y += wt * v.y; acc.z += wt * v.z; acc.w += wt * v.w;
            }
        }
        const size_t e4 = px * cu4 + q;
        const unsigned bits = signs[e4 >> 3] >> (4 * (unsigned)(e4 & 7));
        acc.x = (bits & 1u) ? acc.x : acc.x * slope; acc.y = (bits & 2u) ? acc.y : acc.y * slope;
        acc.z = (bits & 4u) ? acc.z : acc.z * slope; acc.w = (bits & 8u) ? acc.w : acc.w * slope;
        Quad<float>::store(gcoarse, e4, acc);
        sum.x += acc.x; sum.y += acc.y; sum.z += acc.z; sum.w += acc.w;
        m = amax4(m, acc);
    }
    amax_publish(m, amax);
    red[threadIdx.x] = sum;
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < npl; ++k) {
            const float4 o = red[k * cu4 + q];
            sum.x += o.x; sum.y += o.y; sum.z += o.z; sum.w += o.w;
        }
        reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * cu4 + q] = sum;
    }
}

// 2 x 2 / stride 2 max-pooling of a channels-last map (the U-nets' `pooling="max"`, reference sbmc/modules.py:262-263,
// 300-302) and its adjoint FUSED with the addition of the skip connection's gradient: the pooled map's input also feeds
// the up path's concatenation, so autograd ran max_pool_backward, wrote a full-size gradient, and added the skip's to it
// in a third pass.  Here gx = gskip + (this pixel is the FIRST maximum of its window, the element torch's kernel records
// ? gpool : 0) in one pass; the arg-max is recomputed from the saved input instead of an int64 index tensor.
__device__ __forceinline__ float pool_max4(float a, float b, float c, float d) {     // NaN-propagating, as torch
    float m = a;
    m = (b > m || b != b) ? b : m;
    m = (c > m || c != c) ? c : m;
    m = (d > m || d != d) ? d : m;
    return m;
}
// One thread = one float4 of channels of one COARSE pixel (its four fine pixels).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_nhwc_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int c4n, int hc,
                                                               int wc, size_t total4) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % c4n);
        size_t rest = idx / c4n;
        const int j = (int)(rest % wc);
        rest /= wc;
        const int i = (int)(rest % hc);
        const size_t b = rest / hc;
        const size_t f = ((b * 2 * hc + 2 * i) * (size_t)(2 * wc) + 2 * j) * c4n + q;     // fine pixel (2i, 2j)
        const size_t row = (size_t)(2 * wc) * c4n;
        const float4 a = Quad<T>::load(x, f), bq = Quad<T>::load(x, f + c4n), c = Quad<T>::load(x, f + row),
                     d = Quad<T>::load(x, f + row + c4n);
        float4 m;
        m.x = pool_max4(a.x, bq.x, c.x, d.x); m.y = pool_max4(a.y, bq.y, c.y, d.y);
        m.z = pool_max4(a.z, bq.z, c.z, d.z); m.w = pool_max4(a.w, bq.w, c.w, d.w);
        Quad<T>::store(y, idx, m);
    }
}
// The window's winner as torch's max_pool2d picks it: the FIRST maximum, and a NaN beats everything (the last NaN of
// the window takes the gradient) -- so that a non-finite activation reaches the loss's non-finite guard through the
// pooled branch as well.
__device__ __forceinline__ int first_max4(float a, float b, float c, float d) {
    int k = 0;
    float m = a;
    if (b > m || b != b) { m = b; k = 1; }
    if (c > m || c != c) { m = c; k = 2; }
    if (d > m || d != d) { k = 3; }
    return k;
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_nhwc_bwd_add_kernel(const T* __restrict__ x, const T* __restrict__ gpool,
                                                                   const T* __restrict__ gskip, T* __restrict__ gx, int c4n,
                                                                   int hc, int wc, size_t total4) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % c4n);
        size_t rest = idx / c4n;
        const int j = (int)(rest % wc);
        rest /= wc;
        const int i = (int)(rest % hc);
        const size_t b = rest / hc;
        const size_t f = ((b * 2 * hc + 2 * i) * (size_t)(2 * wc) + 2 * j) * c4n + q;
        const size_t row = (size_t)(2 * wc) * c4n;
        const size_t at[4] = {f, f + c4n, f + row, f + row + c4n};
        const float4 a = Quad<T>::load(x, at[0]), bq = Quad<T>::load(x, at[1]), c = Quad<T>::load(x, at[2]),
                     d = Quad<T>::load(x, at[3]);
        const float4 g = Quad<T>::load(gpool, idx);
        const int kx = first_max4(a.x, bq.x, c.x, d.x), ky = first_max4(a.y, bq.y, c.y, d.y);
        const int kz = first_max4(a.z, bq.z, c.z, d.z), kw = first_max4(a.w, bq.w, c.w, d.w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o = gskip ? Quad<T>::load(gskip, at[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
            o.x += kx == k ? g.x : 0.f; o.y += ky == k ? g.y : 0.f; o.z += kz == k ? g.z : 0.f; o.w += kw == k ? g.w : 0.f;
            Quad<T>::store(gx, at[k], o);
        }
    }
}

// The same routing with the adjoint of the bias + activation pass that PRODUCED x behind it (x is a convolution chain's
// activated output and this node its only reader: functions.PoolSkip with an `_AdjLink`): gx = (gskip + routed gpool) *
// act'(z) from the producer's sign words, per-workgroup partial sums of gx per channel (the bias gradient) and max |gx| --
// what bias_act_nhwc_bwd_kernel<true> would do in a pass of its own over the map just written.
// 256 threads = (256 / c4n) lanes of POOLED pixels x c4n channel quads; c4n divides 256.
__global__ __launch_bounds__(256) void maxpool2_nhwc_bwd_add_adj_kernel(const float* __restrict__ x, const float* __restrict__ gpool,
                                                                       const float* __restrict__ gskip, float* __restrict__ gx,
                                                                       const unsigned* __restrict__ signs, float slope,
                                                                       float* __restrict__ partial, unsigned* __restrict__ amax,
                                                                       int c4n, int hc, int wc, size_t ppixels) {
    __shared__ float4 red[256];
    unsigned m = 0;
    const int cq = threadIdx.x % c4n, pl = threadIdx.x / c4n, npl = 256 / c4n;
    const size_t per = (ppixels + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per;
    const size_t p1 = p0 + per < ppixels ? p0 + per : ppixels;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t pp = p0 + pl; pp < p1; pp += npl) {
        const int j = (int)(pp % wc);
        size_t rest = pp / wc;
        const int i = (int)(rest % hc);
        const size_t b = rest / hc;
        const size_t f = ((b * 2 * hc + 2 * i) * (size_t)(2 * wc) + 2 * j) * c4n + cq;
        const size_t row = (size_t)(2 * wc) * c4n;
        const size_t at[4] = {f, f + c4n, f + row, f + row + c4n};
        const float4 a = Quad<float>::load(x, at[0]), bq = Quad<float>::load(x, at[1]), c = Quad<float>::load(x, at[2]),
                     d = Quad<float>::load(x, at[3]);
        const float4 g = Quad<float>::load(gpool, pp * c4n + cq);
        const int kx = first_max4(a.x, bq.x, c.x, d.x), ky = first_max4(a.y, bq.y, c.y, d.y);
        const int kz = first_max4(a.z, bq.z, c.z, d.z), kw = first_max4(a.w, bq.w, c.w, d.w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 o = gskip ? Quad<float>::load(gskip, at[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
            o.x += kx == k ? g.x : 0.f; o.y += ky == k ? g.y : 0.f; o.z += kz == k ? g.z : 0.f; o.w += kw == k ? g.w : 0.f;
            const unsigned bits = signs[at[k] >> 3] >> (4 * (unsigned)(at[k] & 7));
            o.x = (bits & 1u) ? o.x : o.x * slope; o.y = (bits & 2u) ? o.y : o.y * slope;
            o.z = (bits & 4u) ? o.z : o.z * slope; o.w = (bits & 8u) ? o.w : o.w * slope;
            Quad<float>::store(gx, at[k], o);
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            m = amax4(m, o);
        }
    }
    amax_publish(m, amax);
    red[threadIdx.x] = acc;
    __syncthreads();
    if (pl == 0) {
        for (int j = 1; j < npl; ++j) {
            const float4 o = red[j * c4n + cq];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        reinterpret_cast<float4*>(partial)[(size_t)blockIdx.x * c4n + cq] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void slice_channels_nhwc_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                 int ct4, int c0_4, int cn4, size_t total4) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t px = idx / cn4;
        const int q = (int)(idx % cn4);
        Quad<T>::store(dst, idx, Quad<T>::load(src, px * ct4 + c0_4 + q));
    }
}

// Batched 2-d transpose src[b][R][Cn] -> dst[b][Cn][R] through a 64 x 64 LDS tile (both sides move float4s):
// planar -> channels-last is (R, Cn) = (c, h*w), channels-last -> planar (h*w, c).  torch's generic strided
// copy does the same conversion at 0.5-1.5 TB/s (1.9 ms for a [128, 720, 1280] map); this one runs at the
// copy rate.  R and Cn multiples of 4.
constexpr int TT = 64;
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int R, int Cn, int tiles_r, int tiles_c,
                                                         unsigned* __restrict__ amax) {
    __shared__ float tile[TT][TT + 1];
    unsigned m = 0;
    unsigned blk = blockIdx.x;
    const int tc = blk % tiles_c; blk /= tiles_c;
    const int tr = blk % tiles_r;
    const size_t b = blk / tiles_r;
    const int r0 = tr * TT, c0 = tc * TT;
    const float* s = src + b * (size_t)R * Cn;
    float* d = dst + b * (size_t)R * Cn;
    const int q = threadIdx.x % 16, line = threadIdx.x / 16;          // float4 column, row within a pass of 16
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + line + 16 * i, c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < Cn) v = *reinterpret_cast<const float4*>(s + (size_t)r * Cn + c);
        m = amax4(m, v);
        float* t = &tile[line + 16 * i][4 * q];
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + line + 16 * i, r = r0 + 4 * q;               // output row c, columns r .. r + 3
        if (c < Cn && r < R) {
            float4 v;
            v.x = tile[4 * q + 0][line + 16 * i]; v.y = tile[4 * q + 1][line + 16 * i];
            v.z = tile[4 * q + 2][line + 16 * i]; v.w = tile[4 * q + 3][line + 16 * i];
            *reinterpret_cast<float4*>(d + (size_t)c * R + r) = v;
        }
    }
    if (amax) amax_publish(m, amax);
}

static inline unsigned grid_for(size_t n) {
    const size_t blocks = (n + 255) / 256;
    return (unsigned)(blocks < 65536 * 8 ? (blocks ? blocks : 1) : 65536 * 8);
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_bias_act_nhwc_supported(int c) {
    return (c >= 4 && c % 4 == 0 && c / 4 <= 256 && 256 % (c / 4) == 0) ? 1 : 0;
}

extern "C" int sbmc_bias_act_nhwc_chunks(long pixels, int c) {
    if (pixels <= 0 || c <= 0) return 1;
    const long rows = 256 / (c / 4 > 0 ? c / 4 : 1);          // pixels one pass of a workgroup covers
    long want = pixels / (rows * 16);                          // >= 16 passes per workgroup
    if (want < 1) want = 1;
    if (want > 2048) want = 2048;
    return (int)want;
}

extern "C" int sbmc_bias_act_nhwc_fwd_f32(float* y, const float* bias, long pixels, int c, int act, float slope,
                                          void* stream) {
    if (pixels < 0 || c < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!y || !bias || !sbmc_bias_act_nhwc_supported(c) || (uintptr_t)y % 16 || (uintptr_t)bias % 16) return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)pixels * (c / 4);
    hipLaunchKernelGGL(bias_act_nhwc_fwd_kernel, dim3(grid_for(total4 / 4 + 1)), dim3(256), 0, (hipStream_t)stream, y,
                       bias, total4, c / 4, act == 1 ? 0.f : slope, act == 0, (unsigned*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int sbmc_bias_act_nhwc_bwd_f32(const float* gy, const float* y, float* gx, float* partial, long pixels,
                                          int c, int act, float slope, void* stream) {
    if (pixels < 0 || c < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!gy || !y || !gx || !partial || !sbmc_bias_act_nhwc_supported(c)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)y % 16 || (uintptr_t)gx % 16 || (uintptr_t)partial % 16) return SBMC_HIP_EINVAL;
    hipLaunchKernelGGL(bias_act_nhwc_bwd_kernel<false>, dim3((unsigned)sbmc_bias_act_nhwc_chunks(pixels, c)), dim3(256), 0,
                       (hipStream_t)stream, gy, y, gx, partial, (size_t)pixels, c / 4, act == 1 ? 0.f : slope, act == 0,
                       (unsigned*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int sbmc_bias_act_nhwc_fwd_signs_f32(float* y, const float* bias, unsigned* signs, long pixels, int c, int act,
                                                float slope, void* stream) {
    if (pixels < 0 || c < 0 || act < 1 || act > 2) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!y || !bias || !signs || !sbmc_bias_act_nhwc_supported(c) || (uintptr_t)y % 16 || (uintptr_t)bias % 16 ||
        (uintptr_t)signs % 4) return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)pixels * (c / 4);
    hipLaunchKernelGGL(bias_act_nhwc_fwd_signs_kernel, dim3(grid_for(total4 / 4 + 1)), dim3(256), 0, (hipStream_t)stream,
                       y, bias, signs, total4, c / 4, act == 1 ? 0.f : slope, (unsigned*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int sbmc_bias_act_nhwc_bwd_signs_f32(const float* gy, const unsigned* signs, float* gx, float* partial,
                                                long pixels, int c, int act, float slope, void* stream) {
    if (pixels < 0 || c < 0 || act < 1 || act > 2) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!gy || !signs || !gx || !partial || !sbmc_bias_act_nhwc_supported(c)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)signs % 4 || (uintptr_t)gx % 16 || (uintptr_t)partial % 16) return SBMC_HIP_EINVAL;
    hipLaunchKernelGGL(bias_act_nhwc_bwd_kernel<true>, dim3((unsigned)sbmc_bias_act_nhwc_chunks(pixels, c)), dim3(256), 0,
                       (hipStream_t)stream, gy, reinterpret_cast<const float*>(signs), gx, partial, (size_t)pixels, c / 4,
                       act == 1 ? 0.f : slope, 0, (unsigned*)nullptr);
    return (int)hipGetLastError();
}

// The same two passes, also RAISING *amax to the bit pattern of the result's largest magnitude (see amax_publish;
// the caller hands in a zeroed word: sbmc_amd hands them out of one zero-filled block per step):
// fwd with or without sign bits (signs == nullptr: act may be 0), bwd reading sign bits (act 1 / 2) or nothing
// (act 0: signs == nullptr).
extern "C" int sbmc_bias_act_nhwc_fwd_amax_f32(float* y, const float* bias, unsigned* signs, unsigned* amax, long pixels,
                                               int c, int act, float slope, void* stream) {
    if (pixels < 0 || c < 0 || act < 0 || act > 2 || !amax || (signs && act == 0)) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!y || !bias || !sbmc_bias_act_nhwc_supported(c) || (uintptr_t)y % 16 || (uintptr_t)bias % 16 ||
        (uintptr_t)signs % 4) return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)pixels * (c / 4);
    if (signs)
        hipLaunchKernelGGL(bias_act_nhwc_fwd_signs_kernel, dim3(grid_for(total4 / 4 + 1)), dim3(256), 0, (hipStream_t)stream,
                           y, bias, signs, total4, c / 4, act == 1 ? 0.f : slope, amax);
    else
        hipLaunchKernelGGL(bias_act_nhwc_fwd_kernel, dim3(grid_for(total4 / 4 + 1)), dim3(256), 0, (hipStream_t)stream, y,
                           bias, total4, c / 4, act == 1 ? 0.f : slope, act == 0, amax);
    return (int)hipGetLastError();
}

// half activations: gy, gx _Float16 (gx may alias gy), sign bits as written by the convolution's epilogue, fp32 partial sums
extern "C" int sbmc_bias_act_nhwc_bwd_signs_f16(const void* gy, const unsigned* signs, void* gx, float* partial,
                                                long pixels, int c, int act, float slope, void* stream) {
    if (pixels < 0 || c < 0 || act < 0 || act > 2 || ((signs == nullptr) != (act == 0))) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!gy || !gx || !partial || !sbmc_bias_act_nhwc_supported(c)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 8 || (uintptr_t)signs % 4 || (uintptr_t)gx % 8 || (uintptr_t)partial % 16) return SBMC_HIP_EINVAL;
    const unsigned grid = (unsigned)sbmc_bias_act_nhwc_chunks(pixels, c);
    if (signs)
        hipLaunchKernelGGL((bias_act_nhwc_bwd_kernel<true, _Float16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const _Float16*>(gy), reinterpret_cast<const float*>(signs), static_cast<_Float16*>(gx),
                           partial, (size_t)pixels, c / 4, act == 1 ? 0.f : slope, 0, (unsigned*)nullptr);
    else
        hipLaunchKernelGGL((bias_act_nhwc_bwd_kernel<false, _Float16>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const _Float16*>(gy), (const float*)nullptr, static_cast<_Float16*>(gx), partial,
                           (size_t)pixels, c / 4, 0.f, 1, (unsigned*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int sbmc_bias_act_nhwc_bwd_amax_f32(const float* gy, const unsigned* signs, float* gx, float* partial,
                                               unsigned* amax, long pixels, int c, int act, float slope, void* stream) {
    if (pixels < 0 || c < 0 || act < 0 || act > 2 || !amax || ((signs == nullptr) != (act == 0))) return SBMC_HIP_EINVAL;
    if (pixels == 0 || c == 0) return 0;
    if (!gy || !gx || !partial || !sbmc_bias_act_nhwc_supported(c)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)signs % 4 || (uintptr_t)gx % 16 || (uintptr_t)partial % 16) return SBMC_HIP_EINVAL;
    const unsigned grid = (unsigned)sbmc_bias_act_nhwc_chunks(pixels, c);
    if (signs)
        hipLaunchKernelGGL(bias_act_nhwc_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gy,
                           reinterpret_cast<const float*>(signs), gx, partial, (size_t)pixels, c / 4, act == 1 ? 0.f : slope, 0,
                           amax);
    else
        hipLaunchKernelGGL(bias_act_nhwc_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, gy, gy, gx, partial,
                           (size_t)pixels, c / 4, slope, 1, amax);
    return (int)hipGetLastError();
}

extern "C" int sbmc_upsample2x_cat_nhwc_supported(int cu, int cl, int h, int w) {
    return (cu >= 4 && cu % 4 == 0 && cl >= 0 && cl % 4 == 0 && h >= 1 && w >= 1) ? 1 : 0;
}

template <typename T>
static int upcat_nhwc_fwd_impl(const T* coarse, const T* left, T* out, int b, int cu, int cl, int hc,
                               int w, int top, int bot, void* stream) {
    const int h = hc - top - bot;
    if (b < 0 || top < 0 || top > 1 || bot < 0 || bot > 1 || !sbmc_upsample2x_cat_nhwc_supported(cu, cl, h, w))
        return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!coarse || !out || (cl > 0 && !left) || (uintptr_t)coarse % 16 || (uintptr_t)out % 16 || (uintptr_t)left % 16)
        return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)b * (2 * (size_t)h) * (2 * (size_t)w) * ((cu + cl) / 4);
    hipLaunchKernelGGL((upcat_nhwc_fwd_kernel<T>), dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, coarse,
                       left, out, cu / 4, cl / 4, hc, w, top, bot, total4);
    return (int)hipGetLastError();
}

template <typename T>
static int upcat_nhwc_bwd_impl(const T* gout, T* gcoarse, T* gleft, int b, int cu, int cl, int hc, int w,
                               int top, int bot, void* stream) {
    const int h = hc - top - bot;
    if (b < 0 || top < 0 || top > 1 || bot < 0 || bot > 1 || !sbmc_upsample2x_cat_nhwc_supported(cu, cl, h, w))
        return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!gout || (!gcoarse && !gleft) || (uintptr_t)gout % 16 || (uintptr_t)gcoarse % 16 || (uintptr_t)gleft % 16)
        return SBMC_HIP_EINVAL;
    if (gcoarse) {
        const size_t total4 = (size_t)b * hc * w * (cu / 4);
        hipLaunchKernelGGL((upcat_nhwc_bwd_kernel<T>), dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, gout,
                           gcoarse, cu / 4, cl / 4, hc, w, top, bot, total4);
        const int err = (int)hipGetLastError();
        if (err) return err;
    }
    if (gleft && cl > 0) {
        const size_t total4 = (size_t)b * (2 * (size_t)h) * (2 * (size_t)w) * (cl / 4);
        hipLaunchKernelGGL((slice_channels_nhwc_kernel<T>), dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           gout, gleft, (cu + cl) / 4, cu / 4, cl / 4, total4);
    }
    return (int)hipGetLastError();
}

// (ABI 7) the whole frame's adjoint with the activation adjoint of the layer that produced the coarse map in the same
// pass over gcoarse: gcoarse = gather(gout) * (z > 0 ? 1 : slope), partial [sbmc_bias_act_nhwc_chunks(b h w, cu)][cu], *amax
// raised to max |gcoarse|; gleft (or NULL) as sbmc_upsample2x_cat_nhwc_bwd_f32.  cu as sbmc_bias_act_nhwc_supported.
extern "C" int sbmc_upsample2x_cat_nhwc_bwd_adj_f32(const float* gout, float* gcoarse, float* gleft, const unsigned* signs,
                                                    float slope, float* partial, unsigned* amax, int b, int cu, int cl,
                                                    int h, int w, void* stream) {
    if (b < 0 || !sbmc_upsample2x_cat_nhwc_supported(cu, cl, h, w) || !sbmc_bias_act_nhwc_supported(cu)) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!gout || !gcoarse || !signs || !partial || !amax || (uintptr_t)gout % 16 || (uintptr_t)gcoarse % 16 ||
        (uintptr_t)gleft % 16 || (uintptr_t)signs % 4 || (uintptr_t)partial % 16) return SBMC_HIP_EINVAL;
    const size_t pixels = (size_t)b * h * w;
    hipLaunchKernelGGL(upcat_nhwc_bwd_adj_kernel, dim3((unsigned)sbmc_bias_act_nhwc_chunks((long)pixels, cu)), dim3(256), 0,
                       (hipStream_t)stream, gout, gcoarse, signs, slope, partial, amax, cu / 4, cl / 4, h, w, pixels);
    const int err = (int)hipGetLastError();
    if (err || !gleft || cl == 0) return err;
    const size_t total4 = (size_t)b * (2 * (size_t)h) * (2 * (size_t)w) * (cl / 4);
    hipLaunchKernelGGL((slice_channels_nhwc_kernel<float>), dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, gout,
                       gleft, (cu + cl) / 4, cu / 4, cl / 4, total4);
    return (int)hipGetLastError();
}

extern "C" int sbmc_upsample2x_cat_nhwc_fwd_f32(const float* coarse, const float* left, float* out, int b, int cu,
                                                int cl, int h, int w, void* stream) {
    return upcat_nhwc_fwd_impl(coarse, left, out, b, cu, cl, h, w, 0, 0, stream);
}
extern "C" int sbmc_upsample2x_cat_nhwc_bwd_f32(const float* gout, float* gcoarse, float* gleft, int b, int cu, int cl,
                                                int h, int w, void* stream) {
    return upcat_nhwc_bwd_impl(gout, gcoarse, gleft, b, cu, cl, h, w, 0, 0, stream);
}
extern "C" int sbmc_upsample2x_cat_nhwc_slab_fwd_f32(const float* coarse, const float* left, float* out, int b, int cu,
                                                     int cl, int hc, int w, int top, int bot, void* stream) {
    return upcat_nhwc_fwd_impl(coarse, left, out, b, cu, cl, hc, w, top, bot, stream);
}
extern "C" int sbmc_upsample2x_cat_nhwc_slab_bwd_f32(const float* gout, float* gcoarse, float* gleft, int b, int cu,
                                                     int cl, int hc, int w, int top, int bot, void* stream) {
    return upcat_nhwc_bwd_impl(gout, gcoarse, gleft, b, cu, cl, hc, w, top, bot, stream);
}
extern "C" int sbmc_upsample2x_cat_nhwc_slab_fwd_f16(const void* coarse, const void* left, void* out, int b, int cu,
                                                     int cl, int hc, int w, int top, int bot, void* stream) {
    return upcat_nhwc_fwd_impl(static_cast<const _Float16*>(coarse), static_cast<const _Float16*>(left),
                               static_cast<_Float16*>(out), b, cu, cl, hc, w, top, bot, stream);
}
extern "C" int sbmc_upsample2x_cat_nhwc_slab_bwd_f16(const void* gout, void* gcoarse, void* gleft, int b, int cu,
                                                     int cl, int hc, int w, int top, int bot, void* stream) {
    return upcat_nhwc_bwd_impl(static_cast<const _Float16*>(gout), static_cast<_Float16*>(gcoarse),
                               static_cast<_Float16*>(gleft), b, cu, cl, hc, w, top, bot, stream);
}

// the same transpose of _Float16 tensors (the U-nets' entry / exit layout change under torch.autocast(float16): torch's
// strided copy takes 0.96 ms for a [128, 720, 1280] half map, ~10 per training step)
__global__ __launch_bounds__(256) void transpose2d_h_kernel(const _Float16* __restrict__ src, _Float16* __restrict__ dst,
                                                           int R, int Cn, int tiles_r, int tiles_c) {
    using h4 = __attribute__((ext_vector_type(4))) _Float16;
    __shared__ _Float16 tile[TT][TT + 2];
    unsigned blk = blockIdx.x;
    const int tc = blk % tiles_c; blk /= tiles_c;
    const int tr = blk % tiles_r;
    const size_t b = blk / tiles_r;
    const int r0 = tr * TT, c0 = tc * TT;
    const _Float16* s = src + b * (size_t)R * Cn;
    _Float16* d = dst + b * (size_t)R * Cn;
    const int q = threadIdx.x % 16, line = threadIdx.x / 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + line + 16 * i, c = c0 + 4 * q;
        h4 v = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (r < R && c < Cn) v = *reinterpret_cast<const h4*>(s + (size_t)r * Cn + c);
        _Float16* t = &tile[line + 16 * i][4 * q];
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + line + 16 * i, r = r0 + 4 * q;
        if (c < Cn && r < R) {
            h4 v;
            v[0] = tile[4 * q + 0][line + 16 * i]; v[1] = tile[4 * q + 1][line + 16 * i];
            v[2] = tile[4 * q + 2][line + 16 * i]; v[3] = tile[4 * q + 3][line + 16 * i];
            *reinterpret_cast<h4*>(d + (size_t)c * R + r) = v;
        }
    }
}
// x [b, 2 hc, 2 wc, c] -> y [b, hc, wc, c]; elem: 4 (float) or 2 (_Float16); c % 4 == 0
extern "C" int sbmc_maxpool2_nhwc_fwd(const void* x, void* y, int b, int hc, int wc, int c, int elem, void* stream) {
    if (b < 0 || hc < 0 || wc < 0 || c < 0 || (elem != 2 && elem != 4)) return SBMC_HIP_EINVAL;
    if (b == 0 || hc == 0 || wc == 0 || c == 0) return 0;
    if (!x || !y || c % 4 || (uintptr_t)x % (4 * elem) || (uintptr_t)y % (4 * elem)) return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)b * hc * wc * (c / 4);
    if (elem == 4)
        hipLaunchKernelGGL(maxpool2_nhwc_fwd_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const float*>(x), static_cast<float*>(y), c / 4, hc, wc, total4);
    else
        hipLaunchKernelGGL(maxpool2_nhwc_fwd_kernel<_Float16>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const _Float16*>(x), static_cast<_Float16*>(y), c / 4, hc, wc, total4);
    return (int)hipGetLastError();
}
// gx [b, 2 hc, 2 wc, c] = gskip (or 0 if NULL) + the pooled map's gradient gpool [b, hc, wc, c] routed to the first maximum
// of each window of x
extern "C" int sbmc_maxpool2_nhwc_bwd_add(const void* x, const void* gpool, const void* gskip, void* gx, int b, int hc,
                                          int wc, int c, int elem, void* stream) {
    if (b < 0 || hc < 0 || wc < 0 || c < 0 || (elem != 2 && elem != 4)) return SBMC_HIP_EINVAL;
    if (b == 0 || hc == 0 || wc == 0 || c == 0) return 0;
    if (!x || !gpool || !gx || c % 4 || (uintptr_t)x % (4 * elem) || (uintptr_t)gpool % (4 * elem) ||
        (uintptr_t)gskip % (4 * elem) || (uintptr_t)gx % (4 * elem)) return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)b * hc * wc * (c / 4);
    if (elem == 4)
        hipLaunchKernelGGL(maxpool2_nhwc_bwd_add_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const float*>(x), static_cast<const float*>(gpool), static_cast<const float*>(gskip),
                           static_cast<float*>(gx), c / 4, hc, wc, total4);
    else
        hipLaunchKernelGGL(maxpool2_nhwc_bwd_add_kernel<_Float16>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const _Float16*>(x), static_cast<const _Float16*>(gpool),
                           static_cast<const _Float16*>(gskip), static_cast<_Float16*>(gx), c / 4, hc, wc, total4);
    return (int)hipGetLastError();
}

// ... fp32, with the adjoint of the bias + activation pass that produced x in the same pass (ABI 7): gx = (gskip + routed
// gpool) * (z > 0 ? 1 : slope) from the producer's sign words (one bit per element of x, as sbmc_conv3x3_bias_act_nhwc_f32
// leaves them), partial [sbmc_bias_act_nhwc_chunks(b hc wc, c)][c]: the bias gradient's partial sums, *amax raised to max |gx|
extern "C" int sbmc_maxpool2_nhwc_bwd_add_adj_f32(const float* x, const float* gpool, const float* gskip, float* gx,
                                                  const unsigned* signs, float slope, float* partial, unsigned* amax, int b,
                                                  int hc, int wc, int c, void* stream) {
    if (b < 0 || hc < 0 || wc < 0 || c < 0) return SBMC_HIP_EINVAL;
    if (b == 0 || hc == 0 || wc == 0 || c == 0) return 0;
    if (!x || !gpool || !gx || !signs || !partial || !amax || !sbmc_bias_act_nhwc_supported(c) || (uintptr_t)x % 16 ||
        (uintptr_t)gpool % 16 || (uintptr_t)gskip % 16 || (uintptr_t)gx % 16 || (uintptr_t)signs % 4 || (uintptr_t)partial % 16)
        return SBMC_HIP_EINVAL;
    const size_t ppixels = (size_t)b * hc * wc;
    hipLaunchKernelGGL(maxpool2_nhwc_bwd_add_adj_kernel, dim3((unsigned)sbmc_bias_act_nhwc_chunks((long)ppixels, c)), dim3(256), 0,
                       (hipStream_t)stream, x, gpool, gskip, gx, signs, slope, partial, amax, c / 4, hc, wc, ppixels);
    return (int)hipGetLastError();
}

extern "C" int sbmc_transpose2d_f16(const void* src, void* dst, int b, int rows, int cols, void* stream) {
    if (b < 0 || rows < 0 || cols < 0) return SBMC_HIP_EINVAL;
    if (b == 0 || rows == 0 || cols == 0) return 0;
    if (!src || !dst || rows % 4 || cols % 4 || (uintptr_t)src % 8 || (uintptr_t)dst % 8) return SBMC_HIP_EINVAL;
    const int tr = (rows + TT - 1) / TT, tc = (cols + TT - 1) / TT;
    const unsigned long long blocks = (unsigned long long)b * tr * tc;
    if (blocks > 0x7fffffffull) return SBMC_HIP_EINVAL;
    hipLaunchKernelGGL(transpose2d_h_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const _Float16*>(src), static_cast<_Float16*>(dst), rows, cols, tr, tc);
    return (int)hipGetLastError();
}

extern "C" int sbmc_transpose2d_f32(const float* src, float* dst, int b, int rows, int cols, void* stream) {
    if (b < 0 || rows < 0 || cols < 0) return SBMC_HIP_EINVAL;
    if (b == 0 || rows == 0 || cols == 0) return 0;
    if (!src || !dst || rows % 4 || cols % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16) return SBMC_HIP_EINVAL;
    const int tr = (rows + TT - 1) / TT, tc = (cols + TT - 1) / TT;
    const unsigned long long blocks = (unsigned long long)b * tr * tc;
    if (blocks > 0x7fffffffull) return SBMC_HIP_EINVAL;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, rows,
                       cols, tr, tc, (unsigned*)nullptr);
    return (int)hipGetLastError();
}

// The same transpose, also raising *amax (zeroed by the caller) to the bit pattern of the tensor's largest magnitude
// (the U-net's first convolution scales by it: csrc/conv3x3.hip).
extern "C" int sbmc_transpose2d_amax_f32(const float* src, float* dst, unsigned* amax, int b, int rows, int cols,
                                         void* stream) {
    if (b < 0 || rows < 0 || cols < 0 || !amax) return SBMC_HIP_EINVAL;
    if (b == 0 || rows == 0 || cols == 0) return 0;
    if (!src || !dst || rows % 4 || cols % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16) return SBMC_HIP_EINVAL;
    const int tr = (rows + TT - 1) / TT, tc = (cols + TT - 1) / TT;
    const unsigned long long blocks = (unsigned long long)b * tr * tc;
    if (blocks > 0x7fffffffull) return SBMC_HIP_EINVAL;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, rows,
                       cols, tr, tc, amax);
    return (int)hipGetLastError();
}

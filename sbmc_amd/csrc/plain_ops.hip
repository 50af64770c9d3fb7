// gfx950 kernels for the three boundary-level operators of sbmc.halide_ops:
//   scatter2gather          (reference semantics: src/scatter2gather.cpp:29-52)
//   kernel_weighting        (src/kernel_weighting.cpp:28-64)
//   kernel_weighting_grad   (src/kernel_weighting.cpp:68-124)
// Written from the operator definitions, not from the Halide schedules: the
// reference GPU schedule launches 2*(C+1)*bs + 2 kernels for the forward and
// C*bs + kh*kw*bs kernels for the backward; here each operator output is ONE
// launch, each [kh*kw, H, W] tensor is read or written exactly once, and the
// small [C, H, W] operand lives in an LDS halo tile.
//
// All four kernels are HBM-bound streams over the [kh*kw, H, W] tensor
// (algorithmic bytes per pixel: S2G 8*k^2, KW fwd 4*k^2 + 4*(2C+1),
// KW bwd 8*k^2 + ...).
#include "common.hpp"
#include "../../include/sbmc_hip.h"

namespace sbmc {

constexpr int PLAIN_TY = 4;  // 4 waves = 256 threads per workgroup

// T: storage type of EVERY tensor of a call -- float (the reference's `*_float32` operators) or
// _Float16 (`*_float16`: half storage, fp32 arithmetic; the naming scheme of the reference's
// extension, setup.py:65-84).  The LDS halo tiles and all accumulators are fp32 either way.
struct PlainParams {
    const void* data;       // [bs, ctot, h, w]  (KW) -- already offset to the channel group
    const void* weights;    // [bs, kh*kw, h, w]
    const void* d_output;   // [bs, ctot, h, w]
    const void* d_sum_w;    // [bs, h, w]
    void* out0;             // output / d_data / d_weights / s2g output
    void* out1;             // sum_w
    int bs, ctot, h, w, kh, kw;
    int ntx, nty;
    int write_sum_w;        // KW fwd: this channel group writes sum_w
    int accumulate;         // KW bwd d_weights: add to the existing value (channel groups > 0)
};

// ---------------------------------------------------------------- scatter2gather
// One wave per 64-pixel row strip (4 x-adjacent strips per workgroup), taps looped.  Writes
// are aligned 256-B segments of plane (dy,dx); reads are the same segment shifted by (dx-pw)
// floats in the mirrored plane (kh-1-dy, kw-1-dx) of row y+dy-ph.  Raw buffer addressing:
// per-tap offsets are scalar, lanes whose source column is outside the image carry an
// out-of-range voffset and read 0 (no branches).
template <int K, typename T>
__global__ __launch_bounds__(PLAIN_TY * TX) void s2g_kernel(PlainParams p) {
    const int kh = K > 0 ? K : p.kh, kw = K > 0 ? K : p.kw;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const long item = (long)logical_block_id() * PLAIN_TY + wv;
    const long per_img = (long)p.h * p.ntx;
    if (item >= per_img * p.bs) return;
    const int n = __builtin_amdgcn_readfirstlane((int)(item / per_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)(item % per_img));
    const int y = __builtin_amdgcn_readfirstlane(rem / p.ntx);
    const int x0 = __builtin_amdgcn_readfirstlane((rem % p.ntx) * TX);
    const int x = x0 + lane;
    const size_t hw = (size_t)p.h * p.w;
    const T* src = static_cast<const T*>(p.weights) + (size_t)n * kh * kw * hw;
    T* dst = static_cast<T*>(p.out0) + (size_t)n * kh * kw * hw + (size_t)y * p.w + x0;
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(T);
    const unsigned wvoff = x < p.w ? voff : BUF_OOB;          // stores of lanes past the edge are dropped
    const unsigned plane_stride = (unsigned)hw * (unsigned)sizeof(T);
    const unsigned tap_stride = (unsigned)(hw - 1) * (unsigned)sizeof(T);   // source: one plane back, one column on
    const int dx_lo = pw - x, dx_hi = p.w + pw - x;           // source column inside the image
    for (int dy = 0; dy < kh; ++dy) {
        const int ys = y + dy - ph;
        const bool yin = (ys >= 0) && (ys < p.h);             // wave-uniform
        const rsrc_t ws = make_rsrc(dst + (size_t)(dy * kw) * hw);
        if (!yin) {
#pragma unroll 7
            for (int dx = 0; dx < kw; ++dx) logit_store<T>(0.f, ws, wvoff, (unsigned)dx * plane_stride);
            continue;
        }
        // source of tap dx: plane (kh-1-dy)*kw + (kw-1-dx), row ys, column x0-pw+dx+lane
        //   = rowmin + (kw-1-dx)*(hw-1) + lane with rowmin the address of tap kw-1, lane 0
        const rsrc_t rs = make_rsrc(src + ((long)((kh - 1 - dy) * kw) * (long)hw + (long)ys * p.w +
                                           (long)(x0 - pw + kw - 1)));
#pragma unroll 7
        for (int dx = 0; dx < kw; ++dx) {
            const unsigned vo = (dx >= dx_lo && dx < dx_hi) ? voff : BUF_OOB;
            const float v = logit_load<T>(rs, vo, (unsigned)(kw - 1 - dx) * tap_stride);   // half <-> float: exact
            logit_store<T>(v, ws, wvoff, (unsigned)dx * plane_stride);
        }
    }
}

// ---------------------------------------------------------------- kernel_weighting fwd
template <int K, int C, typename T>
__global__ __launch_bounds__(PLAIN_TY * TX) void kw_fwd_kernel(PlainParams p) {
    extern __shared__ float lds[];  // [C][th][tw] data halo tile
    const int kh = K > 0 ? K : p.kh, kw = K > 0 ? K : p.kw;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    const int th = PLAIN_TY + kh - 1, tw = TX + kw - 1;
    const TileCoord t = decode_tile(p.ntx, p.nty, PLAIN_TY);
    const size_t hw = (size_t)p.h * p.w;
    const T* data = static_cast<const T*>(p.data) + (size_t)t.n * p.ctot * hw;
#pragma unroll
    for (int c = 0; c < C; ++c)
        stage_plane(lds + c * th * tw, data + c * hw, p.h, p.w, t.y0 - ph, t.x0 - pw, th, tw, 0.f);
    __syncthreads();

    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const int y = t.y0 + wv, x = t.x0 + lane;
    if (y >= p.h) return;
    const bool xact = x < p.w;
    // weights of this row strip: plane (ry, rx) sits (ry*kw + rx) * hw floats further; raw buffer
    // access with one descriptor per kernel row keeps every per-tap offset scalar and < 4 GiB
    const T* wgt = static_cast<const T*>(p.weights) + (size_t)t.n * kh * kw * hw + (size_t)y * p.w + t.x0;
    const unsigned voff = xact ? (unsigned)lane * (unsigned)sizeof(T) : BUF_OOB;   // lanes past the edge read 0
    const unsigned plane_stride = (unsigned)hw * (unsigned)sizeof(T);

    float acc[C];
    float accw = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;

    for (int ry = 0; ry < kh; ++ry) {
        const float* trow = lds + (wv + ry) * tw + lane;
        const rsrc_t rs = make_rsrc(wgt + (size_t)(ry * kw) * hw);
#pragma unroll 7
        for (int rx = 0; rx < kw; ++rx) {
            const float wt = logit_load<T>(rs, voff, (unsigned)rx * plane_stride);
            accw += wt;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fmaf(wt, trow[c * th * tw + rx], acc[c]);
        }
    }
    if (xact) {
        T* out = static_cast<T*>(p.out0) + (size_t)t.n * p.ctot * hw + (size_t)y * p.w + x;
#pragma unroll
        for (int c = 0; c < C; ++c) out[c * hw] = (T)acc[c];
        if (p.write_sum_w) static_cast<T*>(p.out1)[(size_t)t.n * hw + (size_t)y * p.w + x] = (T)accw;
    }
}

// ---------------------------------------------------------------- kernel_weighting bwd: d_weights
// d_weights[n,dy,dx,y,x] = d_sum_w[n,y,x] + sum_c Dz[n,c,y+dy-ph,x+dx-pw] * dO[n,c,y,x]
template <int K, int C, typename T>
__global__ __launch_bounds__(PLAIN_TY * TX) void kw_bwd_dweights_kernel(PlainParams p) {
    extern __shared__ float lds[];  // [C][th][tw] data halo tile
    const int kh = K > 0 ? K : p.kh, kw = K > 0 ? K : p.kw;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    const int th = PLAIN_TY + kh - 1, tw = TX + kw - 1;
    const TileCoord t = decode_tile(p.ntx, p.nty, PLAIN_TY);
    const size_t hw = (size_t)p.h * p.w;
    const T* data = static_cast<const T*>(p.data) + (size_t)t.n * p.ctot * hw;
#pragma unroll
    for (int c = 0; c < C; ++c)
        stage_plane(lds + c * th * tw, data + c * hw, p.h, p.w, t.y0 - ph, t.x0 - pw, th, tw, 0.f);
    __syncthreads();

    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const int y = t.y0 + wv, x = t.x0 + lane;
    if (y >= p.h || x >= p.w) return;
    const size_t pix = (size_t)y * p.w + x;
    float go[C];
#pragma unroll
    for (int c = 0; c < C; ++c) go[c] = (float)static_cast<const T*>(p.d_output)[((size_t)t.n * p.ctot + c) * hw + pix];
    const float dsw = p.accumulate ? 0.f : (float)static_cast<const T*>(p.d_sum_w)[(size_t)t.n * hw + pix];
    T* dw = static_cast<T*>(p.out0) + (size_t)t.n * kh * kw * hw + (size_t)y * p.w + t.x0;
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(T);
    const unsigned plane_stride = (unsigned)hw * (unsigned)sizeof(T);

    for (int dy = 0; dy < kh; ++dy) {
        const float* trow = lds + (wv + dy) * tw + lane;
        const rsrc_t ws = make_rsrc(dw + (size_t)(dy * kw) * hw);
#pragma unroll 7
        for (int dx = 0; dx < kw; ++dx) {
            float v = dsw;
#pragma unroll
            for (int c = 0; c < C; ++c) v = fmaf(trow[c * th * tw + dx], go[c], v);
            if (p.accumulate) v += logit_load<T>(ws, voff, (unsigned)dx * plane_stride);
            logit_store<T>(v, ws, voff, (unsigned)dx * plane_stride);
        }
    }
}

// ---------------------------------------------------------------- kernel_weighting bwd: d_data
// d_data[n,c,y,x] = sum_{ry,rx} Wz[n,kh-1-ry,kw-1-rx,y+ry-ph,x+rx-pw] * dOz[n,c,y+ry-ph,x+rx-pw]
template <int K, int C, typename T>
__global__ __launch_bounds__(PLAIN_TY * TX) void kw_bwd_ddata_kernel(PlainParams p) {
    extern __shared__ float lds[];  // [C][th][tw] d_output halo tile
    const int kh = K > 0 ? K : p.kh, kw = K > 0 ? K : p.kw;
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
    const int th = PLAIN_TY + kh - 1, tw = TX + kw - 1;
    const TileCoord t = decode_tile(p.ntx, p.nty, PLAIN_TY);
    const size_t hw = (size_t)p.h * p.w;
    const T* dout = static_cast<const T*>(p.d_output) + (size_t)t.n * p.ctot * hw;
#pragma unroll
    for (int c = 0; c < C; ++c)
        stage_plane(lds + c * th * tw, dout + c * hw, p.h, p.w, t.y0 - ph, t.x0 - pw, th, tw, 0.f);
    __syncthreads();

    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const int y = t.y0 + wv, x = t.x0 + lane;
    if (y >= p.h) return;
    const T* wgt = static_cast<const T*>(p.weights) + (size_t)t.n * kh * kw * hw;
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(T);
    const unsigned tap_stride = (unsigned)(hw - 1) * (unsigned)sizeof(T);   // one plane back, one column on
    const int rx_lo = pw - x, rx_hi = p.w + pw - x;        // source column inside the image

    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;

    for (int ry = 0; ry < kh; ++ry) {
        const int ys = y + ry - ph;
        if (ys < 0 || ys >= p.h) continue;  // wave-uniform
        const float* trow = lds + (wv + ry) * tw + lane;
        // tap rx: plane (kh-1-ry)*kw + (kw-1-rx), row ys, column x0-pw+rx+lane
        //   = rowmin + (kw-1-rx)*(hw-1) + lane, rowmin = address of tap kw-1, lane 0
        const rsrc_t rs = make_rsrc(wgt + ((long)((kh - 1 - ry) * kw) * (long)hw + (long)ys * p.w +
                                           (long)(t.x0 - pw + kw - 1)));
#pragma unroll 7
        for (int rx = 0; rx < kw; ++rx) {
            const unsigned vo = (rx >= rx_lo && rx < rx_hi) ? voff : BUF_OOB;   // outside: weight 0
            const float wt = logit_load<T>(rs, vo, (unsigned)(kw - 1 - rx) * tap_stride);
            {
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(wt, trow[c * th * tw + rx], acc[c]);
            }
        }
    }
    if (x < p.w) {
        T* out = static_cast<T*>(p.out0) + (size_t)t.n * p.ctot * hw + (size_t)y * p.w + x;
#pragma unroll
        for (int c = 0; c < C; ++c) out[c * hw] = (T)acc[c];
    }
}

static inline size_t plain_lds_bytes(int c, int kh, int kw) {
    return (size_t)c * (PLAIN_TY + kh - 1) * (TX + kw - 1) * sizeof(float);
}

// largest channel group whose halo tile fits the default 64 KiB dynamic-LDS limit
static inline int channel_group(int c, int kh, int kw) {
    int g = c < SBMC_HIP_MAX_CHANNELS ? c : SBMC_HIP_MAX_CHANNELS;
    while (g > 1 && plain_lds_bytes(g, kh, kw) > 64 * 1024) --g;
    return g;
}

}  // namespace sbmc

using namespace sbmc;

static bool bad_dims(int bs, int c, int h, int w, int kh, int kw) {
    return bs < 0 || c < 0 || h < 0 || w < 0 || kh <= 0 || kw <= 0;
}

template <typename T>
static int s2g_impl(const void* weights, void* output, int bs, int h, int w, int kh, int kw, void* stream) {
    if (bad_dims(bs, 0, h, w, kh, kw)) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!weights || !output) return SBMC_HIP_EINVAL;
    // per-row buffer offsets span up to kw planes and must stay in the 2 GiB voffset range
    if ((size_t)h * w * sizeof(T) * (size_t)(kw + 1) >= 0x7ff00000ull) return SBMC_HIP_EINVAL;
    PlainParams p{};
    p.weights = weights; p.out0 = output;
    p.bs = bs; p.h = h; p.w = w; p.kh = kh; p.kw = kw;
    p.ntx = tiles_x(w); p.nty = h;
    const long items = (long)bs * h * p.ntx;
    const unsigned grid = (unsigned)((items + PLAIN_TY - 1) / PLAIN_TY);
    hipStream_t s = (hipStream_t)stream;
    if (kh == 21 && kw == 21)
        hipLaunchKernelGGL((s2g_kernel<21, T>), dim3(grid), dim3(PLAIN_TY * TX), 0, s, p);
    else
        hipLaunchKernelGGL((s2g_kernel<0, T>), dim3(grid), dim3(PLAIN_TY * TX), 0, s, p);
    return (int)hipGetLastError();
}

template <typename T>
static int kw_fwd_impl(const void* data, const void* weights, void* output, void* sum_w,
                       int bs, int c, int h, int w, int kh, int kw, void* stream) {
    if (bad_dims(bs, c, h, w, kh, kw) || c == 0) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!weights || !sum_w || !data || !output) return SBMC_HIP_EINVAL;
    if (plain_lds_bytes(1, kh, kw) > 64 * 1024) return SBMC_HIP_EINVAL;
    if ((size_t)h * w * sizeof(T) * (size_t)(kw + 1) >= 0x7ff00000ull) return SBMC_HIP_EINVAL;  // buffer offset range
    hipStream_t s = (hipStream_t)stream;
    const size_t hw = (size_t)h * w;
    PlainParams p{};
    p.weights = weights; p.out1 = sum_w;
    p.bs = bs; p.ctot = c; p.h = h; p.w = w; p.kh = kh; p.kw = kw;
    p.ntx = tiles_x(w); p.nty = tiles_y(h, PLAIN_TY);
    const unsigned grid = (unsigned)bs * p.ntx * p.nty;
    const int group = channel_group(c, kh, kw);
    const bool k21 = (kh == 21 && kw == 21);
    for (int c0 = 0; c0 < c; c0 += group) {
        const int cg = c - c0 < group ? c - c0 : group;
        const size_t lds = plain_lds_bytes(cg, kh, kw);
        p.data = static_cast<const T*>(data) + c0 * hw;
        p.out0 = static_cast<T*>(output) + c0 * hw;
        p.write_sum_w = (c0 == 0);
        if (k21) {
            SBMC_DISPATCH_C(cg, hipLaunchKernelGGL((kw_fwd_kernel<21, C, T>), dim3(grid),
                                                   dim3(PLAIN_TY * TX), lds, s, p));
        } else {
            SBMC_DISPATCH_C(cg, hipLaunchKernelGGL((kw_fwd_kernel<0, C, T>), dim3(grid),
                                                   dim3(PLAIN_TY * TX), lds, s, p));
        }
        const int err = (int)hipGetLastError();
        if (err) return err;
    }
    return 0;
}

template <typename T>
static int kw_bwd_impl(const void* data, const void* weights, const void* sum_w, const void* d_output,
                       const void* d_sum_w, void* d_data, void* d_weights,
                       int bs, int c, int h, int w, int kh, int kw, void* stream) {
    (void)sum_w;  // unused by the reference algorithm as well (kernel_weighting.cpp:68-124)
    if (bad_dims(bs, c, h, w, kh, kw) || c == 0) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !weights || !d_output || !d_sum_w || !d_data || !d_weights) return SBMC_HIP_EINVAL;
    if (plain_lds_bytes(1, kh, kw) > 64 * 1024) return SBMC_HIP_EINVAL;
    if ((size_t)h * w * sizeof(T) * (size_t)(kw + 1) >= 0x7ff00000ull) return SBMC_HIP_EINVAL;  // buffer offset range
    hipStream_t s = (hipStream_t)stream;
    const size_t hw = (size_t)h * w;
    PlainParams p{};
    p.weights = weights; p.d_sum_w = d_sum_w;
    p.bs = bs; p.ctot = c; p.h = h; p.w = w; p.kh = kh; p.kw = kw;
    p.ntx = tiles_x(w); p.nty = tiles_y(h, PLAIN_TY);
    const unsigned grid = (unsigned)bs * p.ntx * p.nty;
    const int group = channel_group(c, kh, kw);
    const bool k21 = (kh == 21 && kw == 21);
    for (int c0 = 0; c0 < c; c0 += group) {
        const int cg = c - c0 < group ? c - c0 : group;
        const size_t lds = plain_lds_bytes(cg, kh, kw);
        // d_data for this channel group
        p.d_output = static_cast<const T*>(d_output) + c0 * hw;
        p.out0 = static_cast<T*>(d_data) + c0 * hw;
        if (k21) {
            SBMC_DISPATCH_C(cg, hipLaunchKernelGGL((kw_bwd_ddata_kernel<21, C, T>), dim3(grid),
                                                   dim3(PLAIN_TY * TX), lds, s, p));
        } else {
            SBMC_DISPATCH_C(cg, hipLaunchKernelGGL((kw_bwd_ddata_kernel<0, C, T>), dim3(grid),
                                                   dim3(PLAIN_TY * TX), lds, s, p));
        }
        int err = (int)hipGetLastError();
        if (err) return err;
        // d_weights: first group writes d_sum_w + sum_c, later groups accumulate
        p.data = static_cast<const T*>(data) + c0 * hw;
        p.out0 = d_weights;
        p.accumulate = (c0 > 0);
        if (k21) {
            SBMC_DISPATCH_C(cg, hipLaunchKernelGGL((kw_bwd_dweights_kernel<21, C, T>), dim3(grid),
                                                   dim3(PLAIN_TY * TX), lds, s, p));
        } else {
            SBMC_DISPATCH_C(cg, hipLaunchKernelGGL((kw_bwd_dweights_kernel<0, C, T>), dim3(grid),
                                                   dim3(PLAIN_TY * TX), lds, s, p));
        }
        err = (int)hipGetLastError();
        if (err) return err;
    }
    return 0;
}

extern "C" int sbmc_scatter2gather_f32(const float* weights, float* output, int bs, int h, int w, int kh, int kw,
                                       void* stream) {
    return s2g_impl<float>(weights, output, bs, h, w, kh, kw, stream);
}
extern "C" int sbmc_scatter2gather_f16(const void* weights, void* output, int bs, int h, int w, int kh, int kw,
                                       void* stream) {
    return s2g_impl<_Float16>(weights, output, bs, h, w, kh, kw, stream);
}
extern "C" int sbmc_kernel_weighting_fwd_f32(const float* data, const float* weights, float* output, float* sum_w,
                                             int bs, int c, int h, int w, int kh, int kw, void* stream) {
    return kw_fwd_impl<float>(data, weights, output, sum_w, bs, c, h, w, kh, kw, stream);
}
extern "C" int sbmc_kernel_weighting_fwd_f16(const void* data, const void* weights, void* output, void* sum_w,
                                             int bs, int c, int h, int w, int kh, int kw, void* stream) {
    return kw_fwd_impl<_Float16>(data, weights, output, sum_w, bs, c, h, w, kh, kw, stream);
}
extern "C" int sbmc_kernel_weighting_bwd_f32(const float* data, const float* weights, const float* sum_w,
                                             const float* d_output, const float* d_sum_w, float* d_data,
                                             float* d_weights, int bs, int c, int h, int w, int kh, int kw,
                                             void* stream) {
    return kw_bwd_impl<float>(data, weights, sum_w, d_output, d_sum_w, d_data, d_weights, bs, c, h, w, kh, kw, stream);
}
extern "C" int sbmc_kernel_weighting_bwd_f16(const void* data, const void* weights, const void* sum_w,
                                             const void* d_output, const void* d_sum_w, void* d_data,
                                             void* d_weights, int bs, int c, int h, int w, int kh, int kw,
                                             void* stream) {
    return kw_bwd_impl<_Float16>(data, weights, sum_w, d_output, d_sum_w, d_data, d_weights, bs, c, h, w, kh, kw,
                                 stream);
}

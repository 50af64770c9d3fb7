// Fused progressive splat update for gfx950 (MI355X) -- forward and backward.
//
// Replaces, in one pass over the [k*k, H, W] logits per direction, the body of
// ProgressiveKernelApply.forward for splat=True (reference sbmc/modules.py:422-471:
// Scatter2Gather -> max -> running-max merge -> sub_ -> exp_ -> KernelWeighting ->
// running sums) and the autograd graph behind it (sbmc/functions.py:62-71,102-115
// plus torch's max / sub / exp / mul / add backward).  The reference streams the
// 1.6 GB (720p, k=21) logit tensor ~9 times forward and ~12 times backward; here
// it is read once forward, and read once + written once backward.
//
// Forward (destination-centred, "gather" view of the splat):
//   lane = destination pixel q=(Y,X).  For gather tap (dy,dx) the contributing
//   sample sits at (Y+dy-p, X+dx-p) and its logit is the sample's splat tap
//   (2p-dy, 2p-dx):  g = S[(2p-dy)*k + (2p-dx)][Y+dy-p][X+dx-p]   (0 if outside).
//   A wave therefore reads, per tap, one contiguous 256-byte row segment of one
//   tap plane, shifted by (dx-p) floats: coalesced, merely misaligned.  Radiance
//   of the contributing sample comes from an LDS halo tile.  The softmax over the
//   441 taps (and the merge with the running state of earlier samples) is an
//   online softmax with one rescale per kernel row (21 taps), all in registers.
//
// Backward (sample-centred, "scatter" view):
//   lane = sample pixel (ys,xs).  Tap (ky,kx) lands on destination
//   (ys+ky-p, xs+kx-p).  Reads of S and writes of dS are *aligned* 256-byte
//   segments of plane (ky,kx); the per-destination quantities (final max M,
//   upstream dR[c], dW) come from an LDS halo tile:
//       e  = exp(S - M[q]);  dS = e * (dW[q] + sum_c dR[q][c] * D[c]);  dD[c] += e * dR[q][c]
//   Destinations outside the image carry M = +1e30 => e = 0 (Scatter2Gather's
//   zero fill and its adjoint).  The dependence of the outputs on the running
//   max itself (torch routes it to the arg-max tap, modules.py:429,450) is
//   handled exactly by two small per-pixel kernels around the main one.
#include "common.hpp"
#include "../../include/sbmc_hip.h"
#include <math.h>

namespace sbmc {

constexpr int FWD_TY = 4;   // forward : 4 waves, LDS tile [C][TY+k-1][64+k-1]
constexpr int BWD_TY = 8;   // backward: 8 waves, LDS tile [C+2][TY+k-1][64+k-1]

struct SplatFwdParams {
    const float* data;       // [bs, c, h, w]
    const float* kernels;    // [bs, k*k, h, w]
    const float* sum_r_in;   // [bs, c, h, w] or null
    const float* sum_w_in;   // [bs, h, w]    or null
    const float* max_w_in;   // [bs, h, w]    or null
    float* sum_r_out;
    float* sum_w_out;
    float* max_w_out;
    float* kmax_out;
    int32_t* arow_out;
    int bs, h, w, k;
    int ntx, nty;
};

// One kernel row (K taps) of the online softmax for one destination pixel.
//   v[dx]    : the K gather logits of this row
//   drow     : LDS pointer to Dtile[0][wave+dy][lane]; channel stride cstride
template <int K, int C>
__device__ __forceinline__ void fwd_row_update(const float (&v)[K], const float* drow, int cstride,
                                               int dy, float& m, float& kmax, int& arow,
                                               float (&acc)[C], float& accw) {
    float rmax = v[0];
#pragma unroll
    for (int dx = 1; dx < K; ++dx) rmax = fmaxf(rmax, v[dx]);
    if (rmax > kmax) { kmax = rmax; arow = dy; }
    const float mn = fmaxf(m, rmax);
    const float sc = fast_exp2((m - mn) * LOG2E);  // m == -inf on the first row of an init call -> 0
    m = mn;
    accw *= sc;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] *= sc;
#pragma unroll
    for (int dx = 0; dx < K; ++dx) {
        const float e = fast_exp2((v[dx] - mn) * LOG2E);
        accw += e;
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = fmaf(e, drow[c * cstride + dx], acc[c]);
    }
}

// K > 0: compile-time kernel size (row-buffered online softmax, unrolled taps).
// K == 0: any odd runtime k (tap-at-a-time online softmax; correctness path).
template <int K, int C>
__global__ __launch_bounds__(FWD_TY * TX) void splat_fwd_kernel(SplatFwdParams p) {
    extern __shared__ float lds[];  // [C][th][tw] radiance halo tile, zero outside the image
    const int k = K > 0 ? K : p.k;
    const int pad = (k - 1) / 2;
    const int th = FWD_TY + k - 1, tw = TX + k - 1;
    const TileCoord t = decode_tile(p.ntx, p.nty, FWD_TY);
    const size_t hw = (size_t)p.h * p.w;
    const float* data = p.data + (size_t)t.n * C * hw;
#pragma unroll
    for (int c = 0; c < C; ++c)
        stage_plane(lds + c * th * tw, data + c * hw, p.h, p.w, t.y0 - pad, t.x0 - pad, th, tw, 0.f);
    __syncthreads();

    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const int Y = t.y0 + wv;
    if (Y >= p.h) return;  // whole wave
    const int X = t.x0 + lane;
    const bool xact = X < p.w;
    const size_t pix = (size_t)Y * p.w + (xact ? X : p.w - 1);
    const bool first = (p.sum_r_in == nullptr);

    float m = -INFINITY, accw = 0.f, acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    if (!first) {
        m = p.max_w_in[(size_t)t.n * hw + pix];
        accw = p.sum_w_in[(size_t)t.n * hw + pix];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = p.sum_r_in[((size_t)t.n * C + c) * hw + pix];
    }
    float kmax = -INFINITY;
    int arow = 0;

    const float* S = p.kernels + (size_t)t.n * k * k * hw;
    // wave-uniform: every lane's source column is inside the image for every dx
    const bool interior_x = (t.x0 - pad >= 0) && (t.x0 + TX - 1 + pad < p.w);
    const int cstride = th * tw;

    for (int dy = 0; dy < k; ++dy) {
        const int ys = Y + dy - pad;
        const bool yin = (ys >= 0) && (ys < p.h);  // wave-uniform
        const float* drow = lds + (wv + dy) * tw + lane;
        // plane (2p-dy, 2p-dx), row ys, column X+dx-p  ==  rowbase - dx*(hw-1) + lane
        const float* rowbase = S + ((long)((2 * pad - dy) * k + 2 * pad) * (long)hw +
                                    (long)(yin ? ys : 0) * p.w + (long)(t.x0 - pad));
        if constexpr (K > 0) {
            float v[K];
            if (!yin) {
#pragma unroll
                for (int dx = 0; dx < K; ++dx) v[dx] = 0.f;
            } else if (interior_x) {
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const float* q = rowbase - (long)dx * (long)(hw - 1);  // uniform base
                    v[dx] = q[lane];
                }
            } else {
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    const int xs = X + dx - pad;
                    const float* q = rowbase - (long)dx * (long)(hw - 1);
                    v[dx] = (xs >= 0 && xs < p.w) ? q[lane] : 0.f;
                }
            }
            fwd_row_update<K, C>(v, drow, cstride, dy, m, kmax, arow, acc, accw);
        } else {
            for (int dx = 0; dx < k; ++dx) {
                const int xs = X + dx - pad;
                const float* q = rowbase - (long)dx * (long)(hw - 1);
                const float v = (yin && xs >= 0 && xs < p.w) ? q[lane] : 0.f;
                if (v > kmax) { kmax = v; arow = dy; }
                if (v > m) {
                    const float sc = fast_exp2((m - v) * LOG2E);
                    accw *= sc;
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] *= sc;
                    m = v;
                }
                const float e = fast_exp2((v - m) * LOG2E);
                accw += e;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(e, drow[c * cstride + dx], acc[c]);
            }
        }
    }

    if (xact) {
        const size_t o = (size_t)t.n * hw + pix;
        p.sum_w_out[o] = accw;
        p.max_w_out[o] = m;
        p.kmax_out[o] = kmax;
        p.arow_out[o] = arow;
#pragma unroll
        for (int c = 0; c < C; ++c) p.sum_r_out[((size_t)t.n * C + c) * hw + pix] = acc[c];
    }
}

// ------------------------------------------------------------------ backward
struct SplatBwdParams {
    const float* data;         // [bs, c, h, w]
    const float* kernels;      // [bs, k*k, h, w]
    const float* sum_r_in;     // or null (initialisation call)
    const float* sum_w_in;
    const float* max_w_in;
    const float* sum_r_out;
    const float* sum_w_out;
    const float* max_w_out;
    const float* kmax;
    const int32_t* arow;
    const float* d_sum_r_out;
    const float* d_sum_w_out;
    const float* d_max_w_out;
    float* d_data;
    float* d_kernels;
    float* d_sum_r_in;         // or null
    float* d_sum_w_in;
    float* d_max_w_in;
    float* d_kmax;             // scratch [bs, h, w]
    int bs, c, h, w, k;
    int ntx, nty;
};

// Per-pixel state adjoint (everything in modules.py:450-457,470-471 that is not a
// per-tap quantity).  With sigma = exp(max_in - M), M = max(kmax, max_in):
//   d_sum_r_in = dR * sigma,  d_sum_w_in = dW * sigma
//   dM_total   = dM_out - (dR . sum_r_out + dW * sum_w_out)      (d/dM of every exp(. - M))
//   d_max_in   = sigma * (dR . sum_r_in + dW * sum_w_in) + dM_total * [max_in >  kmax] (1/2 on ties)
//   d_kmax     =                                           dM_total * [kmax   >  max_in] (1/2 on ties)
// (torch.max(a, b) splits the gradient evenly on ties.)
template <int C>
__global__ __launch_bounds__(256) void splat_bwd_state_kernel(SplatBwdParams p) {
    const size_t hw = (size_t)p.h * p.w;
    const size_t total = (size_t)p.bs * hw;
    const bool first = (p.sum_r_in == nullptr);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / hw, pix = i % hw;
        const float dW = p.d_sum_w_out[i];
        float dR[C];
        float dot_out = dW * p.sum_w_out[i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dR[c] = p.d_sum_r_out[(n * C + c) * hw + pix];
            dot_out = fmaf(dR[c], p.sum_r_out[(n * C + c) * hw + pix], dot_out);
        }
        const float dMtot = p.d_max_w_out[i] - dot_out;
        if (first) {
            p.d_kmax[i] = dMtot;
        } else {
            const float M = p.max_w_out[i], Mp = p.max_w_in[i], km = p.kmax[i];
            const float sigma = expf(Mp - M);
            float dot_in = dW * p.sum_w_in[i];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dot_in = fmaf(dR[c], p.sum_r_in[(n * C + c) * hw + pix], dot_in);
                p.d_sum_r_in[(n * C + c) * hw + pix] = dR[c] * sigma;
            }
            p.d_sum_w_in[i] = dW * sigma;
            const float sel_prev = Mp > km ? 1.f : (Mp == km ? 0.5f : 0.f);
            p.d_max_w_in[i] = sigma * dot_in + dMtot * sel_prev;
            p.d_kmax[i] = dMtot * (1.f - sel_prev);
        }
    }
}

template <int K, int C>
__global__ __launch_bounds__(BWD_TY * TX) void splat_bwd_main_kernel(SplatBwdParams p) {
    extern __shared__ float lds[];  // [C+2][th][tw]: M (1e30 outside), dR[0..C) (0 outside), dW (0 outside)
    const int k = K > 0 ? K : p.k;
    const int pad = (k - 1) / 2;
    const int th = BWD_TY + k - 1, tw = TX + k - 1;
    const int fstride = th * tw;
    const TileCoord t = decode_tile(p.ntx, p.nty, BWD_TY);
    const size_t hw = (size_t)p.h * p.w;
    stage_plane(lds, p.max_w_out + (size_t)t.n * hw, p.h, p.w, t.y0 - pad, t.x0 - pad, th, tw, OUTSIDE_MAX);
#pragma unroll
    for (int c = 0; c < C; ++c)
        stage_plane(lds + (1 + c) * fstride, p.d_sum_r_out + ((size_t)t.n * C + c) * hw, p.h, p.w,
                    t.y0 - pad, t.x0 - pad, th, tw, 0.f);
    stage_plane(lds + (1 + C) * fstride, p.d_sum_w_out + (size_t)t.n * hw, p.h, p.w,
                t.y0 - pad, t.x0 - pad, th, tw, 0.f);
    __syncthreads();

    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const int ys = t.y0 + wv;
    if (ys >= p.h) return;  // whole wave
    const int xs = t.x0 + lane;
    if (xs >= p.w) return;  // lanes past the right edge have no sample
    const size_t pix = (size_t)ys * p.w + xs;

    float D[C], dD[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        D[c] = p.data[((size_t)t.n * C + c) * hw + pix];
        dD[c] = 0.f;
    }
    const float* S = p.kernels + (size_t)t.n * k * k * hw + (size_t)ys * p.w + t.x0;
    float* dS = p.d_kernels + (size_t)t.n * k * k * hw + (size_t)ys * p.w + t.x0;

    for (int ky = 0; ky < k; ++ky) {
        const float* trow = lds + (wv + ky) * tw + lane;
        if constexpr (K > 0) {
            float s[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) s[kx] = (S + (size_t)(ky * K + kx) * hw)[lane];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float e = fast_exp2((s[kx] - trow[kx]) * LOG2E);
                float g = trow[(1 + C) * fstride + kx];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float a = trow[(1 + c) * fstride + kx];
                    g = fmaf(a, D[c], g);
                    dD[c] = fmaf(e, a, dD[c]);
                }
                (dS + (size_t)(ky * K + kx) * hw)[lane] = e * g;
            }
        } else {
            for (int kx = 0; kx < k; ++kx) {
                const float s = (S + (size_t)(ky * k + kx) * hw)[lane];
                const float e = fast_exp2((s - trow[kx]) * LOG2E);
                float g = trow[(1 + C) * fstride + kx];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float a = trow[(1 + c) * fstride + kx];
                    g = fmaf(a, D[c], g);
                    dD[c] = fmaf(e, a, dD[c]);
                }
                (dS + (size_t)(ky * k + kx) * hw)[lane] = e * g;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) p.d_data[((size_t)t.n * C + c) * hw + pix] = dD[c];
}

// Routes d_kmax to the arg-max tap (torch: kernels_view.max(1) backward scatters to
// one index, modules.py:429).  One lane per destination pixel; only the K taps of the
// recorded arg-max row are scanned, and only where d_kmax != 0.  Distinct
// destinations map to distinct (tap, sample) elements of d_kernels, so plain
// read-modify-write is race free.  Must run after splat_bwd_main_kernel.
__global__ __launch_bounds__(256) void splat_bwd_route_kernel(SplatBwdParams p) {
    const int k = p.k, pad = (k - 1) / 2;
    const size_t hw = (size_t)p.h * p.w;
    const size_t total = (size_t)p.bs * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const float dk = p.d_kmax[i];
        if (dk == 0.f) continue;
        const size_t n = i / hw, pix = i % hw;
        const int Y = (int)(pix / p.w), X = (int)(pix % p.w);
        const int dy = p.arow[i];
        const int ys = Y + dy - pad;
        if (ys < 0 || ys >= p.h) continue;  // arg-max is a zero-filled tap: gradient dropped
        const float km = p.kmax[i];
        const size_t base = n * (size_t)k * k * hw;
        for (int dx = 0; dx < k; ++dx) {
            const int xs = X + dx - pad;
            const bool in = (xs >= 0) && (xs < p.w);
            const size_t idx = base + ((size_t)((2 * pad - dy) * k + (2 * pad - dx))) * hw +
                               (size_t)ys * p.w + (in ? xs : 0);
            const float v = in ? p.kernels[idx] : 0.f;
            if (v == km) {
                if (in) p.d_kernels[idx] += dk;
                break;
            }
        }
    }
}

static inline size_t fwd_lds_bytes(int c, int k) {
    return (size_t)c * (FWD_TY + k - 1) * (TX + k - 1) * sizeof(float);
}
static inline size_t bwd_lds_bytes(int c, int k) {
    return (size_t)(c + 2) * (BWD_TY + k - 1) * (TX + k - 1) * sizeof(float);
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_splat_update_fwd_f32(const float* data, const float* kernels,
                                         const float* sum_r_in, const float* sum_w_in,
                                         const float* max_w_in,
                                         float* sum_r_out, float* sum_w_out,
                                         float* max_w_out, float* kmax_out,
                                         int32_t* arow_out,
                                         int bs, int c, int h, int w, int k,
                                         void* stream) {
    if (bs < 0 || h < 0 || w < 0 || c < 1 || c > SBMC_HIP_MAX_CHANNELS || k < 1 || (k % 2) == 0)
        return SBMC_HIP_EINVAL;
    const int nin = (sum_r_in != nullptr) + (sum_w_in != nullptr) + (max_w_in != nullptr);
    if (nin != 0 && nin != 3) return SBMC_HIP_EINVAL;  // modules.py:431-435
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !sum_r_out || !sum_w_out || !max_w_out || !kmax_out || !arow_out)
        return SBMC_HIP_EINVAL;
    const size_t lds = fwd_lds_bytes(c, k);
    if (lds > 64 * 1024) return SBMC_HIP_EINVAL;
    SplatFwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out,
                     max_w_out, kmax_out, arow_out, bs, h, w, k, tiles_x(w), tiles_y(h, FWD_TY)};
    const unsigned grid = (unsigned)bs * p.ntx * p.nty;
    hipStream_t s = (hipStream_t)stream;
    if (k == 21) {
        SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_fwd_kernel<21, C>), dim3(grid),
                                              dim3(FWD_TY * TX), lds, s, p));
    } else {
        SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_fwd_kernel<0, C>), dim3(grid),
                                              dim3(FWD_TY * TX), lds, s, p));
    }
    return (int)hipGetLastError();
}

extern "C" int sbmc_splat_update_bwd_f32(const float* data, const float* kernels,
                                         const float* sum_r_in, const float* sum_w_in,
                                         const float* max_w_in,
                                         const float* sum_r_out, const float* sum_w_out,
                                         const float* max_w_out, const float* kmax,
                                         const int32_t* arow,
                                         const float* d_sum_r_out, const float* d_sum_w_out,
                                         const float* d_max_w_out,
                                         float* d_data, float* d_kernels,
                                         float* d_sum_r_in, float* d_sum_w_in,
                                         float* d_max_w_in, float* d_kmax_scratch,
                                         int bs, int c, int h, int w, int k,
                                         void* stream) {
    if (bs < 0 || h < 0 || w < 0 || c < 1 || c > SBMC_HIP_MAX_CHANNELS || k < 1 || (k % 2) == 0)
        return SBMC_HIP_EINVAL;
    const int nin = (sum_r_in != nullptr) + (sum_w_in != nullptr) + (max_w_in != nullptr);
    const int ndin = (d_sum_r_in != nullptr) + (d_sum_w_in != nullptr) + (d_max_w_in != nullptr);
    if ((nin != 0 && nin != 3) || ndin != nin) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !sum_r_out || !sum_w_out || !max_w_out || !kmax || !arow ||
        !d_sum_r_out || !d_sum_w_out || !d_max_w_out || !d_data || !d_kernels || !d_kmax_scratch)
        return SBMC_HIP_EINVAL;
    const size_t lds = bwd_lds_bytes(c, k);
    if (lds > 64 * 1024) return SBMC_HIP_EINVAL;
    SplatBwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out, max_w_out,
                     kmax, arow, d_sum_r_out, d_sum_w_out, d_max_w_out, d_data, d_kernels,
                     d_sum_r_in, d_sum_w_in, d_max_w_in, d_kmax_scratch,
                     bs, c, h, w, k, tiles_x(w), tiles_y(h, BWD_TY)};
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)bs * h * w;
    unsigned egrid = (unsigned)((total + 255) / 256);
    if (egrid > 4096) egrid = 4096;

    SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_bwd_state_kernel<C>), dim3(egrid), dim3(256), 0, s, p));
    int err = (int)hipGetLastError();
    if (err) return err;

    const unsigned grid = (unsigned)bs * p.ntx * p.nty;
    if (k == 21) {
        SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_bwd_main_kernel<21, C>), dim3(grid),
                                              dim3(BWD_TY * TX), lds, s, p));
    } else {
        SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_bwd_main_kernel<0, C>), dim3(grid),
                                              dim3(BWD_TY * TX), lds, s, p));
    }
    err = (int)hipGetLastError();
    if (err) return err;

    hipLaunchKernelGGL(splat_bwd_route_kernel, dim3(egrid), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

extern "C" int sbmc_splat_update_supported(int c, int k) {
    if (c < 1 || c > SBMC_HIP_MAX_CHANNELS || k < 1 || (k % 2) == 0) return 0;
    return fwd_lds_bytes(c, k) <= 64 * 1024 && bwd_lds_bytes(c, k) <= 64 * 1024;
}

extern "C" int sbmc_hip_abi_version(void) { return SBMC_HIP_ABI_VERSION; }

extern "C" const char* sbmc_hip_strerror(int code) {
    if (code == 0) return "success";
    if (code == SBMC_HIP_EINVAL) return "sbmc_hip: invalid argument (shape, null pointer, channel count or LDS budget)";
    return hipGetErrorString((hipError_t)code);
}

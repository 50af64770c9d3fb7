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
//   of the contributing sample comes from LDS.  The softmax over the k*k taps (and
//   the merge with the running state of earlier samples) is an online softmax in
//   registers with one rescale per kernel row.
//
// Backward (sample-centred, "scatter" view):
//   lane = sample pixel (ys,xs).  Tap (ky,kx) lands on destination
//   (ys+ky-p, xs+kx-p).  Reads of S and writes of dS are *aligned* 256-byte
//   segments of plane (ky,kx); the per-destination quantities (final max M,
//   upstream dR[c], dW, and the gradient of the max with its arg-max tap) come
//   from LDS:
//       e  = exp(S - M[q]);  dS = e * (dW[q] + sum_c dR[q][c] * D[c]);  dD[c] += e * dR[q][c]
//       dS += d_kmax[q]  where this tap is q's arg-max  (torch routes max() that way)
//   Destinations outside the image carry M = +1e30 => e = 0 (Scatter2Gather's
//   zero fill and its adjoint).
//
// Kernels in this file:
//   "tile" kernels (splat_{fwd,bwd}_tile_kernel, splat_bwd_route_kernel): a workgroup owns a
//       64 x TY pixel tile and stages a haloed tile of the small operand in LDS.  Any odd k,
//       up to 8 channels, fp32.  The generic path.
//   "strip" kernels (splat_{fwd,bwd}_strip_kernel; k = 21 with <= 4 channels in fp32 / fp16
//       logits, and every odd k in 3..19 with 3 channels): a wavefront owns one 64-pixel row
//       strip; the 4 waves of a workgroup are x-adjacent so that the cache lines their
//       misaligned segments share are requested from one CU at nearly the same time; the small
//       operand is staged per wave, one kernel row at a time, in a 1-2 KB LDS row buffer (no
//       s_barrier anywhere); the big streams go through raw buffer loads/stores whose per-tap
//       offsets are scalar, so a strip needs one VGPR of addressing in total.
//   gather-kernel variant (GATHER flag of the forward strip kernel, gather_bwd_{dg,ddata}_kernel):
//       ProgressiveKernelApply(splat=False).
//   per-pixel state kernels: splat_bwd_state_kernel (state adjoint + destination records),
//       splat_merge_fwd_kernel / splat_chain_bwd_kernel (all samples of a frame per launch).
#include "common.hpp"
#include "../../include/sbmc_hip.h"
#include <math.h>
#include <stdlib.h>

namespace sbmc {

constexpr int FWD_TY = 4;   // tile forward : 4 waves, LDS tile [C][TY+k-1][64+k-1]
constexpr int BWD_TY = 8;   // tile backward: 8 waves, LDS tile [C+2][TY+k-1][64+k-1]
constexpr int V2_WAVES = 4; // strip kernels: 4 x-adjacent strips per workgroup
constexpr int V2_ROW = 96;  // strip kernels: staged positions per strip (>= 64 + k - 1)
constexpr int REC = 8;      // strip backward: floats per destination record
#ifndef FWD_MIN_WAVES
#define FWD_MIN_WAVES 7
#endif
#ifndef FWD_TAP_GROUP
#define FWD_TAP_GROUP 7
#endif
#ifndef BWD_TAP_GROUP
#define BWD_TAP_GROUP 3
#endif
// cache policy of the big streams (A/B knobs; see common.hpp buf_load)
#ifndef AUX_FWD_LD
#define AUX_FWD_LD 0
#endif
// backward: nt on both the logit loads and the d_logit stores measured -1.2% (three interleaved
// A/B rounds); forward: nt loads measured +12% (they defeat the L1/L2 sharing of straddled lines)
#ifndef AUX_BWD_LD
#define AUX_BWD_LD 2
#endif
#ifndef AUX_BWD_ST
#define AUX_BWD_ST 2
#endif

struct SplatFwdParams {
    const float* data;       // [bs, c, h, w]
    const void* kernels;     // [bs, k*k, h, w] float (or _Float16 for the strip kernels)
    const float* sum_r_in;   // [bs, c, h, w] or null
    const float* sum_w_in;   // [bs, h, w]    or null
    const float* max_w_in;   // [bs, h, w]    or null
    float* sum_r_out;
    float* sum_w_out;
    float* max_w_out;
    float* kmax_out;
    int32_t* atap_out;
    int bs, h, w, k;
    int ntx, nty;
    // row-slab form (strip kernels only; hd == h, top == 0, zero_top == zero_bot == 1 is the whole
    // frame): data / kernels hold the slab's own h source rows, the outputs hd = top + h + bottom
    // destination rows; destination row r is source row r - top.  A slab edge that is the image
    // edge (zero_*) sees Scatter2Gather's zero-filled taps beyond it; beyond an inner edge the
    // taps belong to the neighbouring slab and contribute nothing here.
    int hd, top, zero_top, zero_bot;
};

// ------------------------------------------------------------------ tile forward
template <int C>
__global__ __launch_bounds__(FWD_TY * TX) void splat_fwd_tile_kernel(SplatFwdParams p) {
    extern __shared__ float lds[];  // [C][th][tw] radiance halo tile, zero outside the image
    const int k = p.k;
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

    // This sample is reduced on its own (m starts at -inf, sums at 0) and merged with the
    // incoming state at the very end, like the reference (modules.py:450-471): folding 441 small
    // terms one by one into an already large running sum would cost ~1e-4 of relative accuracy
    // over 32 samples.
    float m = -INFINITY, accw = 0.f, acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    float kmax = -INFINITY;
    int atap = 0;

    const float* S = static_cast<const float*>(p.kernels) + (size_t)t.n * k * k * hw;
    const int cstride = th * tw;

    for (int dy = 0; dy < k; ++dy) {
        const int ys = Y + dy - pad;
        const bool yin = (ys >= 0) && (ys < p.h);  // wave-uniform
        const float* drow = lds + (wv + dy) * tw + lane;
        // plane (2p-dy, 2p-dx), row ys, column X+dx-p  ==  rowbase - dx*(hw-1) + lane
        const float* rowbase = S + ((long)((2 * pad - dy) * k + 2 * pad) * (long)hw +
                                    (long)(yin ? ys : 0) * p.w + (long)(t.x0 - pad));
        for (int dx = 0; dx < k; ++dx) {
            const int xs = X + dx - pad;
            const float* q = rowbase - (long)dx * (long)(hw - 1);
            const float v = (yin && xs >= 0 && xs < p.w) ? q[lane] : 0.f;
            if (v > kmax) { kmax = v; atap = dy * k + dx; }
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

    if (xact) {
        const size_t o = (size_t)t.n * hw + pix;
        float M = m;  // == kmax here
        if (!first) {
            const float Mp = p.max_w_in[o];
            M = fmaxf(Mp, m);
            const float sigma = expf(Mp - M), tau = expf(m - M);
            accw = p.sum_w_in[o] * sigma + accw * tau;
#pragma unroll
            for (int c = 0; c < C; ++c)
                acc[c] = p.sum_r_in[((size_t)t.n * C + c) * hw + pix] * sigma + acc[c] * tau;
        }
        p.sum_w_out[o] = accw;
        p.max_w_out[o] = M;
        p.kmax_out[o] = kmax;
        p.atap_out[o] = atap;
#pragma unroll
        for (int c = 0; c < C; ++c) p.sum_r_out[((size_t)t.n * C + c) * hw + pix] = acc[c];
    }
}

// ------------------------------------------------------------------ strip forward
// One kernel row (K taps) of the online softmax for one destination pixel.
//   v[dx] : the K gather logits of this row;  srow : LDS, staged radiance positions
//   VT: float, or _Float16 for half logits.  The half kernel is VALU-bound (98 % busy, profiles/r02_half_splat_pmc.txt),
//   so its logits stay half through the row maximum (exact in half: v_max_f16, no conversion) and enter fp32
//   only inside the exponent's multiply-add (v_fma_mix_f32 converts its half operand on the fly).
//   (v: float, or the raw half in the low 16 bits of an unsigned -- RawLogit)
template <int K, int C, typename VT>
__device__ __forceinline__ void fwd_row_update(const VT (&v)[K], const float* srow, int dy,
                                               float& m, float& kmax, int& atap,
                                               float (&acc)[C], float& accw) {
    constexpr bool HALF = !__is_same(VT, float);
    float rmax;
    if constexpr (HALF) {
        _Float16 r = raw_half(v[0]);
#pragma unroll
        for (int dx = 1; dx < K; ++dx) r = __builtin_fmaxf16(r, raw_half(v[dx]));
        rmax = (float)r;
    } else {
        rmax = v[0];
#pragma unroll
        for (int dx = 1; dx < K; ++dx) rmax = fmaxf(rmax, v[dx]);
    }
    if (rmax > kmax) {  // strict: the first row attaining the max wins
        kmax = rmax;
        int idx = K - 1;
#pragma unroll
        for (int dx = K - 2; dx >= 0; --dx) {           // first tap in the row
            bool eq;
            if constexpr (HALF) eq = (float)raw_half(v[dx]) == rmax;
            else eq = v[dx] == rmax;
            idx = eq ? dx : idx;
        }
        atap = dy * K + idx;
    }
    const float mn = fmaxf(m, rmax);
    const float sc = fast_exp2((m - mn) * LOG2E);  // m == -inf on the first row of an init call -> 0
    m = mn;
    accw *= sc;
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] *= sc;
    const float nmn = -mn * LOG2E;
    // taps in groups of G: the scheduling barrier keeps the compiler from hoisting all
    // K*C LDS reads of the row to the top (which costs >100 VGPRs and the occupancy)
    constexpr int G = FWD_TAP_GROUP;
#pragma unroll
    for (int g = 0; g < K; g += G) {
#pragma unroll
        for (int dx = g; dx < (g + G < K ? g + G : K); ++dx) {
            float e;
            if constexpr (HALF) e = fast_exp2(fmaf((float)raw_half(v[dx]), LOG2E, nmn));
            else e = fast_exp2((v[dx] - mn) * LOG2E);
            accw += e;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fmaf(e, srow[c * V2_ROW + dx], acc[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// GATHER = true: the kernels are pixel-centred (ProgressiveKernelApply(splat=False), the
// reference's `--gather` ablation): tap (dy,dx) of destination (Y,X) is plane dy*K+dx at (Y,X)
// itself -- aligned loads, no Scatter2Gather index algebra, and every tap counts in the softmax
// (only the *radiance* outside the image is zero).
template <int K, int C, typename LT, bool GATHER = false>
__global__ __launch_bounds__(V2_WAVES * TX, FWD_MIN_WAVES) void splat_fwd_strip_kernel(SplatFwdParams p) {
    static_assert(TX + K - 1 <= V2_ROW, "staged row too short");
    constexpr int P = (K - 1) / 2;
    __shared__ float lds[V2_WAVES * C * V2_ROW];  // per wave: [C][V2_ROW] radiance of one source row
    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const long item = (long)logical_block_id() * V2_WAVES + wv;
    const int nseg = p.ntx;
    const long per_img = (long)p.hd * nseg;   // one item per 64-pixel strip of a DESTINATION row
    if (item >= per_img * p.bs) return;  // whole wave; no block-level barrier is used below
    // readfirstlane: the divisions run on the VALU; force the (uniform) results back to
    // SGPRs so that every address below is "SGPR base + lane" (global_load saddr form)
    const int n = __builtin_amdgcn_readfirstlane((int)(item / per_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)(item % per_img));
    const int Y = __builtin_amdgcn_readfirstlane(rem / nseg);
    const int X0 = __builtin_amdgcn_readfirstlane((rem % nseg) * TX);
    const int X = X0 + lane;
    const bool xact = X < p.w;
    const size_t hw = (size_t)p.h * p.w;      // source planes
    const size_t hwd = (size_t)p.hd * p.w;    // destination planes (== hw unless this is a row slab)
    const size_t pix = (size_t)Y * p.w + (xact ? X : p.w - 1);
    const bool first = (p.sum_r_in == nullptr);
    float* buf = lds + wv * (C * V2_ROW);

    // the sample is reduced on its own and merged with the incoming state at the end (see the
    // tile kernel above for why)
    float m = -INFINITY, accw = 0.f, acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.f;
    float kmax = -INFINITY;
    int atap = 0;

    const LT* S = static_cast<const LT*>(p.kernels) + (size_t)n * K * K * hw;
    const float* data = p.data + (size_t)n * C * hw;
    const bool interior_x = (X0 - P >= 0) && (X0 + TX - 1 + P < p.w);  // wave-uniform
    // staged source columns: positions 0..63 by every lane, 64..64+K-2 by the first K-1 lanes
    const int colA = X0 - P + lane, colB = colA + TX;
    const bool inA = (colA >= 0) && (colA < p.w);
    const bool inB = (lane < K - 1) && (colB < p.w);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(LT);
    const unsigned tap_stride = (unsigned)(hw - 1) * (unsigned)sizeof(LT);  // bytes between taps dx+1 -> dx
    // border strips: lane's source column X+dx-P is inside the image for dx in [dx_lo, dx_hi)
    const int dx_lo = P - X, dx_hi = p.w + P - X;

    using RL = typename RawLogit<LT>::type;
    auto load_row = [&](int dy, RL (&v)[K], float (&s)[2 * C]) {
        const int ys = Y - p.top + dy - P;
        const bool yin = (ys >= 0) && (ys < p.h);  // wave-uniform
        if constexpr (GATHER) {
            const rsrc_t rs = make_rsrc(S + ((size_t)(dy * K) * hw + (size_t)Y * p.w + (size_t)X0));
            const unsigned vo = xact ? voff : BUF_OOB;
#pragma unroll
            for (int dx = 0; dx < K; ++dx)
                v[dx] = logit_load_raw<LT, AUX_FWD_LD>(rs, vo, (unsigned)dx * (unsigned)(hw * sizeof(LT)));
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float* dr = data + c * hw + (size_t)(yin ? ys : 0) * p.w;
                s[2 * c] = (yin && inA) ? dr[colA] : 0.f;
                s[2 * c + 1] = (yin && inB) ? dr[colB] : 0.f;
            }
            return;
        }
        if (!yin) {   // beyond an IMAGE edge (rows beyond an inner slab edge are never visited, see below)
#pragma unroll
            for (int dx = 0; dx < K; ++dx) v[dx] = (RL)0;      // (+0.0 in either representation)
#pragma unroll
            for (int j = 0; j < 2 * C; ++j) s[j] = 0.f;
            return;
        }
        // tap dx of this row: plane (2P-dy)*K + (K-1-dx), row ys, column X0-P+dx+lane
        //   = rowmin + (K-1-dx) * (hw-1) + lane,  rowmin = address of tap K-1, lane 0
        const rsrc_t rs = make_rsrc(S + ((long)((2 * P - dy) * K) * (long)hw + (long)ys * p.w + (long)(X0 + P)));
        if (interior_x) {
#pragma unroll
            for (int dx = 0; dx < K; ++dx) v[dx] = logit_load_raw<LT, AUX_FWD_LD>(rs, voff, (unsigned)(K - 1 - dx) * tap_stride);
        } else {
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                const unsigned vo = (dx >= dx_lo && dx < dx_hi) ? voff : BUF_OOB;  // OOB lanes read 0
                v[dx] = logit_load_raw<LT, AUX_FWD_LD>(rs, vo, (unsigned)(K - 1 - dx) * tap_stride);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* dr = data + c * hw + (size_t)ys * p.w;
            s[2 * c] = inA ? dr[colA] : 0.f;
            s[2 * c + 1] = inB ? dr[colB] : 0.f;
        }
    };
    auto step = [&](int dy, const RL (&v)[K], const float (&s)[2 * C]) {
        wave_lds_sync();  // previous row's reads are done before its slots are overwritten
#pragma unroll
        for (int c = 0; c < C; ++c) {
            buf[c * V2_ROW + lane] = s[2 * c];
            if (lane < K - 1) buf[c * V2_ROW + TX + lane] = s[2 * c + 1];
        }
        wave_lds_sync();
        fwd_row_update<K, C, RL>(v, buf + lane, dy, m, kmax, atap, acc, accw);
    };

    // (A register double-buffer that prefetches row dy+1 while row dy is reduced was tried:
    // it costs ~40 VGPRs, i.e. 2-3 waves/SIMD of occupancy, and measured slower.)
    RL v[K];
    float s[2 * C];
    // kernel rows whose source row lies beyond an INNER slab edge belong to the neighbouring slab and are
    // not visited at all (scalar loop bounds; whole frame: 0 .. K)
    const int dy_lo = p.zero_top ? 0 : max(0, P + p.top - Y);
    const int dy_hi = p.zero_bot ? K : min(K, p.h + P + p.top - Y);
#pragma unroll 1
    for (int dy = dy_lo; dy < dy_hi; ++dy) {
        load_row(dy, v, s);
        step(dy, v, s);
    }

    if (xact) {
        const size_t o = (size_t)n * hwd + pix;
        float M = m;  // == kmax here
        if (!first) {
            const float Mp = p.max_w_in[o];
            M = fmaxf(Mp, m);
            const float sigma = expf(Mp - M), tau = expf(m - M);
            accw = p.sum_w_in[o] * sigma + accw * tau;
#pragma unroll
            for (int c = 0; c < C; ++c)
                acc[c] = p.sum_r_in[((size_t)n * C + c) * hwd + pix] * sigma + acc[c] * tau;
        }
        p.sum_w_out[o] = accw;
        p.max_w_out[o] = M;
        if (p.kmax_out) p.kmax_out[o] = kmax;
        p.atap_out[o] = atap;
#pragma unroll
        for (int c = 0; c < C; ++c) p.sum_r_out[((size_t)n * C + c) * hwd + pix] = acc[c];
    }
}

// ------------------------------------------------------------------ backward
struct SplatBwdParams {
    const float* data;         // [bs, c, h, w]
    const void* kernels;       // [bs, k*k, h, w] float (or _Float16 for the strip kernels)
    const float* sum_r_in;     // or null (initialisation call)
    const float* sum_w_in;
    const float* max_w_in;
    const float* sum_r_out;
    const float* sum_w_out;
    const float* max_w_out;
    const float* kmax;
    const int32_t* atap;
    const float* d_sum_r_out;
    const float* d_sum_w_out;
    const float* d_max_w_out;
    float* d_data;
    void* d_kernels;           // same storage type as kernels
    float* d_sum_r_in;         // or null
    float* d_sum_w_in;
    float* d_max_w_in;
    float* scratch;            // tile: d_kmax [bs, h, w];  strip: destination records [bs, hd, w, REC]
    int bs, c, h, w, k;
    int ntx, nty;
    int hd, top;               // row-slab form (splat_bwd_strip_kernel only), see SplatFwdParams
};

// Per-pixel state adjoint (everything in modules.py:450-457,470-471 that is not a
// per-tap quantity).  With sigma = exp(max_in - M), M = max(kmax, max_in):
//   d_sum_r_in = dR * sigma,  d_sum_w_in = dW * sigma
//   dM_total   = dM_out - (dR . sum_r_out + dW * sum_w_out)      (d/dM of every exp(. - M))
//   d_max_in   = sigma * (dR . sum_r_in + dW * sum_w_in) + dM_total * [max_in >  kmax] (1/2 on ties)
//   d_kmax     =                                           dM_total * [kmax   >  max_in] (1/2 on ties)
// (torch.max(a, b) splits the gradient evenly on ties.)
// RECORDS = false: writes d_kmax to scratch[bs,h,w]                          (tile kernels)
// RECORDS = true : writes {M, dW, d_kmax, atap | dR0..dR3} to scratch[bs,h,w,8] (strip kernels)
template <int C, bool RECORDS>
__global__ __launch_bounds__(256) void splat_bwd_state_kernel(SplatBwdParams p) {
    const size_t hw = (size_t)p.h * p.w;
    const size_t total = (size_t)p.bs * hw;
    const bool first = (p.sum_r_in == nullptr);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / hw, pix = i % hw;
        // (the two dot products and their difference in double: they cancel wherever the loss does not depend on
        // the running max, and this kernel is per pixel, not per tap -- see splat_chain_bwd_kernel)
        const float dW = p.d_sum_w_out[i];
        float dR[C];
        double dot_out = (double)dW * (double)p.sum_w_out[i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            dR[c] = p.d_sum_r_out[(n * C + c) * hw + pix];
            dot_out += (double)dR[c] * (double)p.sum_r_out[(n * C + c) * hw + pix];
        }
        const double dMtot = (double)p.d_max_w_out[i] - dot_out;
        const float M = p.max_w_out[i];
        float dkmax = (float)dMtot;
        if (!first) {
            const float Mp = p.max_w_in[i], km = p.kmax[i];
            const float sigma = expf(Mp - M);
            double dot_in = (double)dW * (double)p.sum_w_in[i];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dot_in += (double)dR[c] * (double)p.sum_r_in[(n * C + c) * hw + pix];
                p.d_sum_r_in[(n * C + c) * hw + pix] = dR[c] * sigma;
            }
            p.d_sum_w_in[i] = dW * sigma;
            const double sel_prev = Mp > km ? 1. : (Mp == km ? 0.5 : 0.);
            p.d_max_w_in[i] = (float)((double)sigma * dot_in + dMtot * sel_prev);
            dkmax = (float)(dMtot * (1. - sel_prev));
        }
        if constexpr (RECORDS) {
            static_assert(C <= 4, "records hold up to 4 channels");
            float4 r0, r1;
            r0.x = M; r0.y = dW; r0.z = dkmax; r0.w = __int_as_float(p.atap[i]);
            r1.x = dR[0];
            r1.y = C > 1 ? dR[C > 1 ? 1 : 0] : 0.f;
            r1.z = C > 2 ? dR[C > 2 ? 2 : 0] : 0.f;
            r1.w = C > 3 ? dR[C > 3 ? 3 : 0] : 0.f;
            float4* rec = reinterpret_cast<float4*>(p.scratch) + i * 2;
            rec[0] = r0;
            rec[1] = r1;
        } else {
            p.scratch[i] = dkmax;
        }
    }
}

// tile backward main kernel, any odd k, up to 8 channels.
template <int C>
__global__ __launch_bounds__(BWD_TY * TX) void splat_bwd_tile_kernel(SplatBwdParams p) {
    extern __shared__ float lds[];  // [C+2][th][tw]: M (1e30 outside), dR[0..C) (0 outside), dW (0 outside)
    const int k = p.k;
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
    const float* S = static_cast<const float*>(p.kernels) + (size_t)t.n * k * k * hw + (size_t)ys * p.w + t.x0;
    float* dS = static_cast<float*>(p.d_kernels) + (size_t)t.n * k * k * hw + (size_t)ys * p.w + t.x0;

    for (int ky = 0; ky < k; ++ky) {
        const float* trow = lds + (wv + ky) * tw + lane;
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
#pragma unroll
    for (int c = 0; c < C; ++c) p.d_data[((size_t)t.n * C + c) * hw + pix] = dD[c];
}

// tile backward routing: adds d_kmax to the arg-max tap recorded by the forward (torch:
// kernels_view.max(1) backward scatters to one index, modules.py:429).  Distinct
// destinations map to distinct (tap, sample) elements of d_kernels, so plain
// read-modify-write is race free.  Must run after splat_bwd_tile_kernel.
__global__ __launch_bounds__(256) void splat_bwd_route_kernel(SplatBwdParams p) {
    const int k = p.k, pad = (k - 1) / 2;
    const size_t hw = (size_t)p.h * p.w;
    const size_t total = (size_t)p.bs * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const float dk = p.scratch[i];
        if (dk == 0.f) continue;
        const size_t n = i / hw, pix = i % hw;
        const int Y = (int)(pix / p.w), X = (int)(pix % p.w);
        const int t = p.atap[i];
        const int dy = t / k, dx = t % k;
        const int ys = Y + dy - pad, xs = X + dx - pad;
        if (ys < 0 || ys >= p.h || xs < 0 || xs >= p.w) continue;  // arg-max is a zero-filled tap
        const size_t idx = n * (size_t)k * k * hw + ((size_t)((2 * pad - dy) * k + (2 * pad - dx))) * hw +
                           (size_t)ys * p.w + xs;
        static_cast<float*>(p.d_kernels)[idx] += dk;
    }
}

// strip backward main kernel (one wave = one 64-sample row strip).  Destination records
// {M, dW, d_kmax, atap | dR0..3} of one destination row at a time are staged per wave
// in LDS as two float4 arrays, so each tap costs two conflict-free ds_read_b128.
template <int K, int C, typename LT>
__global__ __launch_bounds__(V2_WAVES * TX, 8) void splat_bwd_strip_kernel(SplatBwdParams p) {
    static_assert(TX + K - 1 <= V2_ROW, "staged row too short");
    static_assert(C <= 4, "records hold up to 4 channels");
    constexpr int P = (K - 1) / 2;
    __shared__ float4 lds[V2_WAVES * 2 * V2_ROW];  // per wave: [2][V2_ROW] record halves
    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const long item = (long)logical_block_id() * V2_WAVES + wv;
    const int nseg = p.ntx;
    const long per_img = (long)p.h * nseg;
    if (item >= per_img * p.bs) return;  // whole wave
    const int n = __builtin_amdgcn_readfirstlane((int)(item / per_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)(item % per_img));
    const int ys = __builtin_amdgcn_readfirstlane(rem / nseg);
    const int X0 = __builtin_amdgcn_readfirstlane((rem % nseg) * TX);
    const int xs = X0 + lane;
    const bool active = xs < p.w;
    const size_t hw = (size_t)p.h * p.w;
    const size_t pix = (size_t)ys * p.w + (active ? xs : p.w - 1);
    float4* h0 = lds + wv * (2 * V2_ROW);
    float4* h1 = h0 + V2_ROW;

    float D[C], dD[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        D[c] = p.data[((size_t)n * C + c) * hw + pix];
        dD[c] = 0.f;
    }
    const LT* S = static_cast<const LT*>(p.kernels) + (size_t)n * K * K * hw + (size_t)ys * p.w + X0;
    LT* dS = static_cast<LT*>(p.d_kernels) + (size_t)n * K * K * hw + (size_t)ys * p.w + X0;
    const float4* rec = reinterpret_cast<const float4*>(p.scratch) + (size_t)n * p.hd * p.w * 2;
    const unsigned voff = active ? (unsigned)lane * (unsigned)sizeof(LT) : BUF_OOB;  // sample-less lanes: loads 0, stores dropped
    const unsigned plane_stride = (unsigned)hw * (unsigned)sizeof(LT);

    // staged destination columns: positions 0..63 by every lane, 64..64+K-2 by the first K-1 lanes
    const int colA = X0 - P + lane, colB = colA + TX;
    const bool inA = (colA >= 0) && (colA < p.w);
    const bool inB = (lane < K - 1) && (colB < p.w);
    const float4 fill0 = make_float4(OUTSIDE_MAX, 0.f, 0.f, __int_as_float(-1));
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_logits = [&](int ky, float (&s)[K]) {
        const rsrc_t rs = make_rsrc(S + (size_t)(ky * K) * hw);  // tap (ky, 0); tap kx is kx planes further
#pragma unroll
        for (int kx = 0; kx < K; ++kx) s[kx] = logit_load<LT, AUX_BWD_LD>(rs, voff, (unsigned)kx * plane_stride);
    };
    auto step = [&](int ky, const float (&s)[K]) {
        const int yd = ys + p.top + ky - P;         // destination row (of the slab's hd rows)
        const bool yin = (yd >= 0) && (yd < p.hd);  // wave-uniform
        float4 a0 = fill0, a1 = zero4, b0 = fill0, b1 = zero4;
        if (yin) {
            const float4* rrow = rec + (size_t)yd * p.w * 2;
            if (inA) { a0 = rrow[(size_t)colA * 2]; a1 = rrow[(size_t)colA * 2 + 1]; }
            if (inB) { b0 = rrow[(size_t)colB * 2]; b1 = rrow[(size_t)colB * 2 + 1]; }
        }
        wave_lds_sync();  // previous row's reads are done before its slots are overwritten
        h0[lane] = a0;
        h1[lane] = a1;
        if (lane < K - 1) { h0[TX + lane] = b0; h1[TX + lane] = b1; }
        wave_lds_sync();
        const rsrc_t ws = make_rsrc(dS + (size_t)(ky * K) * hw);
        const int tg0 = K * K - 1 - ky * K;  // gather tap index of (ky, kx) is tg0 - kx
        constexpr int G = BWD_TAP_GROUP;  // taps per scheduling group (bounds the live LDS values)
#pragma unroll
        for (int g = 0; g < K; g += G) {
#pragma unroll
        for (int kx = g; kx < (g + G < K ? g + G : K); ++kx) {
            const float4 q0 = h0[lane + kx];  // M, dW, d_kmax, atap
            const float4 q1 = h1[lane + kx];  // dR0..3
            const float e = fast_exp2((s[kx] - q0.x) * LOG2E);
            float g = q0.y;
            g = fmaf(q1.x, D[0], g);
            dD[0] = fmaf(e, q1.x, dD[0]);
            if constexpr (C > 1) { g = fmaf(q1.y, D[C > 1 ? 1 : 0], g); dD[C > 1 ? 1 : 0] = fmaf(e, q1.y, dD[C > 1 ? 1 : 0]); }
            if constexpr (C > 2) { g = fmaf(q1.z, D[C > 2 ? 2 : 0], g); dD[C > 2 ? 2 : 0] = fmaf(e, q1.z, dD[C > 2 ? 2 : 0]); }
            if constexpr (C > 3) { g = fmaf(q1.w, D[C > 3 ? 3 : 0], g); dD[C > 3 ? 3 : 0] = fmaf(e, q1.w, dD[C > 3 ? 3 : 0]); }
            float ds = e * g;
            ds += (__float_as_int(q0.w) == tg0 - kx) ? q0.z : 0.f;
            logit_store<LT, AUX_BWD_ST>(ds, ws, voff, (unsigned)kx * plane_stride);
        }
        __builtin_amdgcn_sched_barrier(0);
        }
    };

    float s[K];
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        load_logits(ky, s);
        step(ky, s);
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < C; ++c) p.d_data[((size_t)n * C + c) * hw + pix] = dD[c];
    }
}

// ------------------------------------------------------------------ gather-mode backward
// Adjoint of the GATHER forward.  The per-pixel state pre-pass (records) is the same as for the
// splat.  d_kernels is destination-centred (aligned read + aligned write of the logits, the
// destination's record in registers, radiance of the tap's source from the LDS row buffer);
// d_data is source-centred (second, read-only sweep of the logits, destination records staged
// per row exactly as in splat_bwd_strip_kernel).
template <int K, int C, typename LT>
__global__ __launch_bounds__(V2_WAVES * TX, 7) void gather_bwd_dg_kernel(SplatBwdParams p) {
    static_assert(TX + K - 1 <= V2_ROW, "staged row too short");
    static_assert(C <= 4, "records hold up to 4 channels");
    constexpr int P = (K - 1) / 2;
    __shared__ float lds[V2_WAVES * C * V2_ROW];
    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const long item = (long)logical_block_id() * V2_WAVES + wv;
    const int nseg = p.ntx;
    const long per_img = (long)p.h * nseg;
    if (item >= per_img * p.bs) return;
    const int n = __builtin_amdgcn_readfirstlane((int)(item / per_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)(item % per_img));
    const int Y = __builtin_amdgcn_readfirstlane(rem / nseg);
    const int X0 = __builtin_amdgcn_readfirstlane((rem % nseg) * TX);
    const int X = X0 + lane;
    const bool xact = X < p.w;
    const size_t hw = (size_t)p.h * p.w;
    float* buf = lds + wv * (C * V2_ROW);
    const float4* rec = reinterpret_cast<const float4*>(p.scratch) + ((size_t)n * hw + (size_t)Y * p.w) * 2;
    float4 q0 = make_float4(OUTSIDE_MAX, 0.f, 0.f, __int_as_float(-1)), q1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xact) { q0 = rec[(size_t)X * 2]; q1 = rec[(size_t)X * 2 + 1]; }
    const float dRv[4] = {q1.x, q1.y, q1.z, q1.w};
    const int atap = __float_as_int(q0.w);
    const LT* S = static_cast<const LT*>(p.kernels) + (size_t)n * K * K * hw + (size_t)Y * p.w + X0;
    LT* dS = static_cast<LT*>(p.d_kernels) + (size_t)n * K * K * hw + (size_t)Y * p.w + X0;
    const float* data = p.data + (size_t)n * C * hw;
    const unsigned voff = xact ? (unsigned)lane * (unsigned)sizeof(LT) : BUF_OOB;
    const unsigned plane_stride = (unsigned)hw * (unsigned)sizeof(LT);
    const int colA = X0 - P + lane, colB = colA + TX;
    const bool inA = (colA >= 0) && (colA < p.w);
    const bool inB = (lane < K - 1) && (colB < p.w);
#pragma unroll 1
    for (int dy = 0; dy < K; ++dy) {
        const int ys = Y + dy - P;
        const bool yin = (ys >= 0) && (ys < p.h);
        const rsrc_t rs = make_rsrc(S + (size_t)(dy * K) * hw);
        const rsrc_t ws = make_rsrc(dS + (size_t)(dy * K) * hw);
        float g[K];
#pragma unroll
        for (int dx = 0; dx < K; ++dx) g[dx] = logit_load<LT, AUX_BWD_LD>(rs, voff, (unsigned)dx * plane_stride);
        float sa[C], sb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float* dr = data + c * hw + (size_t)(yin ? ys : 0) * p.w;
            sa[c] = (yin && inA) ? dr[colA] : 0.f;
            sb[c] = (yin && inB) ? dr[colB] : 0.f;
        }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < C; ++c) {
            buf[c * V2_ROW + lane] = sa[c];
            if (lane < K - 1) buf[c * V2_ROW + TX + lane] = sb[c];
        }
        wave_lds_sync();
        constexpr int G = 7;
#pragma unroll
        for (int g0 = 0; g0 < K; g0 += G) {
#pragma unroll
            for (int dx = g0; dx < (g0 + G < K ? g0 + G : K); ++dx) {
                const float e = fast_exp2((g[dx] - q0.x) * LOG2E);
                float val = q0.y;
#pragma unroll
                for (int c = 0; c < C; ++c) val = fmaf(dRv[c], buf[c * V2_ROW + lane + dx], val);
                float dg = e * val;
                dg += (atap == dy * K + dx) ? q0.z : 0.f;
                logit_store<LT, AUX_BWD_ST>(dg, ws, voff, (unsigned)dx * plane_stride);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// (7 / 3 waves per SIMD instead of 8: at 64 registers this kernel spilled 2-4 VGPRs with up to 3 channels and 87
// with 4 -- tools/kernel_resources.py; a read-only sweep, nowhere near needing the last wave of occupancy)
template <int K, int C, typename LT>
__global__ __launch_bounds__(V2_WAVES * TX, C > 3 ? 3 : 7) void gather_bwd_ddata_kernel(SplatBwdParams p) {
    static_assert(TX + K - 1 <= V2_ROW, "staged row too short");
    static_assert(C <= 4, "records hold up to 4 channels");
    constexpr int P = (K - 1) / 2;
    __shared__ float4 lds[V2_WAVES * 2 * V2_ROW];
    const int wv = wave_id();
    const int lane = threadIdx.x & 63;
    const long item = (long)logical_block_id() * V2_WAVES + wv;
    const int nseg = p.ntx;
    const long per_img = (long)p.h * nseg;
    if (item >= per_img * p.bs) return;
    const int n = __builtin_amdgcn_readfirstlane((int)(item / per_img));
    const int rem = __builtin_amdgcn_readfirstlane((int)(item % per_img));
    const int ys = __builtin_amdgcn_readfirstlane(rem / nseg);
    const int X0 = __builtin_amdgcn_readfirstlane((rem % nseg) * TX);
    const int xs = X0 + lane;
    const bool active = xs < p.w;
    const size_t hw = (size_t)p.h * p.w;
    float4* h0 = lds + wv * (2 * V2_ROW);
    float4* h1 = h0 + V2_ROW;
    float dD[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dD[c] = 0.f;
    const LT* S = static_cast<const LT*>(p.kernels) + (size_t)n * K * K * hw;
    const float4* rec = reinterpret_cast<const float4*>(p.scratch) + (size_t)n * hw * 2;
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(LT);
    const unsigned tap_stride = (unsigned)(hw - 1) * (unsigned)sizeof(LT);   // one plane on, one column back
    const int colA = X0 - P + lane, colB = colA + TX;   // staged destination columns
    const bool inA = (colA >= 0) && (colA < p.w);
    const bool inB = (lane < K - 1) && (colB < p.w);
    const float4 fill0 = make_float4(OUTSIDE_MAX, 0.f, 0.f, __int_as_float(-1));
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // destination column xs - dx + P is inside the image for dx in (xs + P - w, xs + P]
    const int dx_lo = xs + P - p.w + 1, dx_hi = xs + P + 1;
#pragma unroll 1
    for (int dy = 0; dy < K; ++dy) {
        const int yd = ys - dy + P;
        if (yd < 0 || yd >= p.h) continue;   // wave-uniform: no such destination
        // tap dx: plane dy*K + dx, row yd, column X0 + P - dx + lane = base + dx*(hw-1) + lane
        const rsrc_t rs = make_rsrc(S + ((long)(dy * K) * (long)hw + (long)yd * p.w + (long)(X0 + P)));
        float g[K];
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
            const unsigned vo = (active && dx >= dx_lo && dx < dx_hi) ? voff : BUF_OOB;
            g[dx] = logit_load<LT, AUX_FWD_LD>(rs, vo, (unsigned)dx * tap_stride);
        }
        float4 a0 = fill0, a1 = zero4, b0 = fill0, b1 = zero4;
        const float4* rrow = rec + (size_t)yd * p.w * 2;
        if (inA) { a0 = rrow[(size_t)colA * 2]; a1 = rrow[(size_t)colA * 2 + 1]; }
        if (inB) { b0 = rrow[(size_t)colB * 2]; b1 = rrow[(size_t)colB * 2 + 1]; }
        wave_lds_sync();
        h0[lane] = a0;
        h1[lane] = a1;
        if (lane < K - 1) { h0[TX + lane] = b0; h1[TX + lane] = b1; }
        wave_lds_sync();
        constexpr int G = 3;
#pragma unroll
        for (int g0 = 0; g0 < K; g0 += G) {
#pragma unroll
            for (int dx = g0; dx < (g0 + G < K ? g0 + G : K); ++dx) {
                const float4 q0 = h0[lane + (K - 1 - dx)];
                const float4 q1 = h1[lane + (K - 1 - dx)];
                const float e = fast_exp2((g[dx] - q0.x) * LOG2E);
                dD[0] = fmaf(e, q1.x, dD[0]);
                if constexpr (C > 1) dD[C > 1 ? 1 : 0] = fmaf(e, q1.y, dD[C > 1 ? 1 : 0]);
                if constexpr (C > 2) dD[C > 2 ? 2 : 0] = fmaf(e, q1.z, dD[C > 2 ? 2 : 0]);
                if constexpr (C > 3) dD[C > 3 ? 3 : 0] = fmaf(e, q1.w, dD[C > 3 ? 3 : 0]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (active) {
        const size_t pix = (size_t)ys * p.w + xs;
#pragma unroll
        for (int c = 0; c < C; ++c) p.d_data[((size_t)n * C + c) * hw + pix] = dD[c];
    }
}

// ------------------------------------------------------------------ all samples at once
// The running-softmax state is an associative (log-sum-exp) monoid, so the S per-sample
// splats of one frame need not be chained through S launches: every sample is reduced on its
// own (the forward strip kernel with batch = bs*S and no incoming state), and a per-pixel
// kernel folds the S partial states in sample order -- same values as S progressive updates
// up to rounding.  The backward mirrors it: a per-pixel chain kernel walks the samples in
// reverse, applying exactly the per-step state adjoint of splat_bwd_state_kernel, and emits
// the destination records of every sample; ONE launch of splat_bwd_strip_kernel (batch =
// bs*S) then produces all d_kernels / d_data.
struct SplatMergeParams {
    const float* part_r;   // [bs, S, c, h, w]  per-sample sum_r with the sample's own max
    const float* part_w;   // [bs, S, h, w]
    const float* part_m;   // [bs, S, h, w]     per-sample max (kmax)
    float* sum_r;          // [bs, c, h, w]     final state
    float* sum_w;          // [bs, h, w]
    float* max_w;          // [bs, h, w]
    float* run_r;          // [bs, S, c, h, w]  running state after each sample (saved for backward)
    float* run_w;          // [bs, S, h, w]
    float* run_m;          // [bs, S, h, w]
    int bs, s, h, w;
};

template <int C>
__global__ __launch_bounds__(256) void splat_merge_fwd_kernel(SplatMergeParams p) {
    const size_t hw = (size_t)p.h * p.w;
    const size_t total = (size_t)p.bs * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / hw, pix = i % hw;
        float M = -INFINITY, W = 0.f, R[C];
#pragma unroll
        for (int c = 0; c < C; ++c) R[c] = 0.f;
        for (int s = 0; s < p.s; ++s) {
            const size_t o = (n * p.s + s) * hw + pix;
            const float km = p.part_m[o];
            const float Mn = fmaxf(M, km);
            const float sigma = expf(M - Mn);   // 0 for the first sample (M = -inf)
            const float tau = expf(km - Mn);
            W = W * sigma + p.part_w[o] * tau;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const size_t oc = ((n * p.s + s) * C + c) * hw + pix;
                R[c] = R[c] * sigma + p.part_r[oc] * tau;
                p.run_r[oc] = R[c];
            }
            M = Mn;
            p.run_w[o] = W;
            p.run_m[o] = M;
        }
        p.sum_w[i] = W;
        p.max_w[i] = M;
#pragma unroll
        for (int c = 0; c < C; ++c) p.sum_r[(n * C + c) * hw + pix] = R[c];
    }
}

struct SplatChainParams {
    const float* part_m;       // [bs, S, h, w] per-sample max
    const int32_t* atap;       // [bs, S, h, w]
    const float* run_r;        // [bs, S, c, h, w]
    const float* run_w;        // [bs, S, h, w]
    const float* run_m;        // [bs, S, h, w]
    const float* d_sum_r;      // [bs, c, h, w] upstream gradients of the final state
    const float* d_sum_w;      // [bs, h, w]
    const float* d_max_w;      // [bs, h, w]
    float* records;            // [bs, S, h, w, 8]
    int bs, s, h, w;
    // An upper bound of |d_kernels| without a pass over it (or both nullptr): with e = exp(S - M) <= 1 the strip kernel's
    // dS = e (dW + sum_c dR_c D_c) + [arg-max tap] d_kmax is at most |dW| + |d_kmax| + max|D| sum_c |dR_c| -- per-pixel
    // quantities this kernel holds.  `bound` (a word the caller zeroed) is raised to the bit pattern of the largest
    // over all records; `dmax`: the word of max |data|.  The wide 1x1 layer that consumes d_kernels takes its
    // power-of-two scale from it (csrc/pointwise.hip pw_wide_bwd2_kernel).
    const unsigned* dmax;
    unsigned* bound;
};

template <int C>
__global__ __launch_bounds__(256) void splat_chain_bwd_kernel(SplatChainParams p) {
    static_assert(C <= 4, "records hold up to 4 channels");
    const size_t hw = (size_t)p.h * p.w;
    const size_t total = (size_t)p.bs * hw;
    const float dmax = p.dmax ? __builtin_bit_cast(float, *p.dmax) : 0.f;
    float bnd = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / hw, pix = i % hw;
        // The chain is carried in DOUBLE from sample to sample (this kernel is per pixel, not per tap: free):
        // dM - (dW sum_w + dR . sum_r) is a difference of equal sums wherever the loss does not depend on the
        // running max (out = sum_r / sum_w: always), so what reaches d_kernels through the arg-max tap is pure
        // rounding residue; in fp32 it grew with every step of the chain (tools/fuzz_slab.py: up to 4e-5 of the
        // tensor's scale after 5 samples, several times the CPU restatement's in unlucky cases).  Now the only roundings are those of
        // the fp32 inputs and of the records written below.
        double dW = p.d_sum_w[i], dM = p.d_max_w[i], dR[4] = {0., 0., 0., 0.};
#pragma unroll
        for (int c = 0; c < C; ++c) dR[c] = p.d_sum_r[(n * C + c) * hw + pix];
        for (int s = p.s - 1; s >= 0; --s) {
            const size_t o = (n * p.s + s) * hw + pix;
            const float M = p.run_m[o];
            double dot_out = dW * (double)p.run_w[o];
#pragma unroll
            for (int c = 0; c < C; ++c)
                dot_out += dR[c] * (double)p.run_r[((n * p.s + s) * C + c) * hw + pix];
            const double dMtot = dM - dot_out;
            double sel_prev = 0., sigma = 0., dot_in = 0.;
            if (s > 0) {
                const size_t op = o - hw;
                const float Mp = p.run_m[op], km = p.part_m[o];
                sel_prev = Mp > km ? 1. : (Mp == km ? 0.5 : 0.);
                sigma = exp((double)Mp - (double)M);
                dot_in = dW * (double)p.run_w[op];
#pragma unroll
                for (int c = 0; c < C; ++c)
                    dot_in += dR[c] * (double)p.run_r[((n * p.s + s - 1) * C + c) * hw + pix];
            }
            float4 r0, r1;
            r0.x = M; r0.y = (float)dW; r0.z = (float)(dMtot * (1. - sel_prev)); r0.w = __int_as_float(p.atap[o]);
            r1.x = (float)dR[0]; r1.y = (float)dR[1]; r1.z = (float)dR[2]; r1.w = (float)dR[3];
            float4* rec = reinterpret_cast<float4*>(p.records) + o * 2;
            rec[0] = r0;
            rec[1] = r1;
            {
                const float b = fabsf(r0.y) + fabsf(r0.z) + dmax * ((fabsf(r1.x) + fabsf(r1.y)) + (fabsf(r1.z) + fabsf(r1.w)));
                bnd = b > bnd ? b : bnd;               // (a NaN leaves it: the word then says nothing about NaN inputs, nor need it)
            }
            // adjoint of the incoming state of this step = outgoing state of the previous one
            dM = sigma * dot_in + dMtot * sel_prev;
            dW *= sigma;
#pragma unroll
            for (int c = 0; c < C; ++c) dR[c] *= sigma;
        }
    }
    if (p.bound != nullptr) amax_publish(__builtin_bit_cast(unsigned, bnd), p.bound);
}

static inline size_t fwd_tile_lds_bytes(int c, int k) {
    return (size_t)c * (FWD_TY + k - 1) * (TX + k - 1) * sizeof(float);
}
static inline size_t bwd_tile_lds_bytes(int c, int k) {
    return (size_t)(c + 2) * (BWD_TY + k - 1) * (TX + k - 1) * sizeof(float);
}
// Strip kernels exist for k = 21 with 1..4 channels in fp32 and fp16 (the SBMC configuration),
// and for the other odd kernel sizes up to 19 with the 3 radiance channels in fp32 (--ksize of
// the reference's train.py); everything else runs on the generic tile kernels.  All per-row
// buffer offsets (up to k planes) must stay below the 2 GiB voffset range of the descriptors.
static inline bool strip_ok(int c, int k, int h, int w, bool half = false) {
    if ((size_t)h * w * 4 * (size_t)(k + 1) >= 0x7ff00000ull) return false;
    if (k == 21) return c >= 1 && c <= 4;
    return !half && c == 3 && k >= 3 && k <= 19 && (k % 2) == 1;
}

// launches KERNEL<K, C, LT> for the (k, c) combinations strip_ok admits
#define SBMC_LAUNCH_STRIP(KERNEL, grid, stream, params, ...)                                                 \
    do {                                                                                                  \
        if (k == 21) {                                                                                    \
            SBMC_DISPATCH_C4(c, hipLaunchKernelGGL((KERNEL<21, C, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX),  \
                                                   0, stream, params));                                   \
        } else if constexpr (sizeof(LT) == 4) {                                                           \
            switch (k) {                                                                                  \
                case 3: hipLaunchKernelGGL((KERNEL<3, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break;   \
                case 5: hipLaunchKernelGGL((KERNEL<5, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break;   \
                case 7: hipLaunchKernelGGL((KERNEL<7, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break;   \
                case 9: hipLaunchKernelGGL((KERNEL<9, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break;   \
                case 11: hipLaunchKernelGGL((KERNEL<11, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break; \
                case 13: hipLaunchKernelGGL((KERNEL<13, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break; \
                case 15: hipLaunchKernelGGL((KERNEL<15, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break; \
                case 17: hipLaunchKernelGGL((KERNEL<17, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break; \
                case 19: hipLaunchKernelGGL((KERNEL<19, 3, LT, ##__VA_ARGS__>), dim3(grid), dim3(V2_WAVES * TX), 0, stream, params); break; \
                default: return SBMC_HIP_EINVAL;                                                          \
            }                                                                                             \
        } else {                                                                                          \
            return SBMC_HIP_EINVAL;                                                                       \
        }                                                                                                 \
    } while (0)

// Development knob (not part of the ABI): SBMC_HIP_SPLAT_VARIANT=0 forces the generic tile
// kernels even where the strip kernels apply (used by the tests to cover both at k = 21).
static int splat_variant() {
    return env_knob("SBMC_HIP_SPLAT_VARIANT", 1);
}

}  // namespace sbmc

using namespace sbmc;

static bool bad_splat_dims(int bs, int c, int h, int w, int k) {
    return bs < 0 || h < 0 || w < 0 || c < 1 || c > SBMC_HIP_MAX_CHANNELS || k < 1 || (k % 2) == 0;
}

extern "C" int sbmc_splat_update_supported(int c, int k) {
    if (c < 1 || c > SBMC_HIP_MAX_CHANNELS || k < 1 || (k % 2) == 0) return 0;
    if (strip_ok(c, k, 1, 1)) return 1;
    return fwd_tile_lds_bytes(c, k) <= 64 * 1024 && bwd_tile_lds_bytes(c, k) <= 64 * 1024;
}

extern "C" size_t sbmc_splat_update_bwd_scratch_bytes(int bs, int c, int h, int w, int k) {
    if (bs < 0 || h < 0 || w < 0) return 0;
    (void)c; (void)k;
    // sized for the widest layout (destination records) so that the variant can be
    // chosen per call without reallocating
    return (size_t)bs * h * w * REC * sizeof(float);
}

template <typename LT>
static int splat_update_fwd_impl(const float* data, const void* kernels,
                                 const float* sum_r_in, const float* sum_w_in,
                                 const float* max_w_in,
                                 float* sum_r_out, float* sum_w_out,
                                 float* max_w_out, float* kmax_out,
                                 int32_t* atap_out,
                                 int bs, int c, int h, int w, int k,
                                 void* stream, int top = 0, int bot = 0, int zero_top = 1, int zero_bot = 1) {
    if (bad_splat_dims(bs, c, h, w, k)) return SBMC_HIP_EINVAL;
    const int nin = (sum_r_in != nullptr) + (sum_w_in != nullptr) + (max_w_in != nullptr);
    if (nin != 0 && nin != 3) return SBMC_HIP_EINVAL;  // modules.py:431-435
    const bool slab = top != 0 || bot != 0 || !zero_top || !zero_bot;
    if (top < 0 || bot < 0 || top > (k - 1) / 2 || bot > (k - 1) / 2) return SBMC_HIP_EINVAL;
    const int hd = top + h + bot;
    if (slab && !strip_ok(c, k, hd, w, sizeof(LT) != 4)) return SBMC_HIP_EINVAL;   // strip kernels only
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !sum_r_out || !sum_w_out || !max_w_out || !atap_out || (!kmax_out && !slab))
        return SBMC_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int variant = slab ? 1 : splat_variant();
    if (variant > 0 && strip_ok(c, k, hd, w, sizeof(LT) != 4)) {
        SplatFwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out,
                         max_w_out, kmax_out, atap_out, bs, h, w, k, tiles_x(w), h, hd, top, zero_top, zero_bot};
        const long items = (long)bs * hd * p.ntx;
        const unsigned grid = (unsigned)((items + V2_WAVES - 1) / V2_WAVES);
        SBMC_LAUNCH_STRIP(splat_fwd_strip_kernel, grid, s, p);
        return (int)hipGetLastError();
    }
    if (sizeof(LT) != 4) return SBMC_HIP_EINVAL;  // the generic tile kernels are fp32 only
    const size_t lds = fwd_tile_lds_bytes(c, k);
    if (lds > 64 * 1024) return SBMC_HIP_EINVAL;
    SplatFwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out,
                     max_w_out, kmax_out, atap_out, bs, h, w, k, tiles_x(w), tiles_y(h, FWD_TY), h, 0, 1, 1};
    const unsigned grid = (unsigned)bs * p.ntx * p.nty;
    SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_fwd_tile_kernel<C>), dim3(grid),
                                          dim3(FWD_TY * TX), lds, s, p));
    return (int)hipGetLastError();
}

template <typename LT>
static int splat_update_bwd_impl(const float* data, const void* kernels,
                                 const float* sum_r_in, const float* sum_w_in,
                                 const float* max_w_in,
                                 const float* sum_r_out, const float* sum_w_out,
                                 const float* max_w_out, const float* kmax,
                                 const int32_t* atap,
                                 const float* d_sum_r_out, const float* d_sum_w_out,
                                 const float* d_max_w_out,
                                 float* d_data, void* d_kernels,
                                 float* d_sum_r_in, float* d_sum_w_in,
                                 float* d_max_w_in, float* scratch,
                                 int bs, int c, int h, int w, int k,
                                 void* stream) {
    if (bad_splat_dims(bs, c, h, w, k)) return SBMC_HIP_EINVAL;
    const int nin = (sum_r_in != nullptr) + (sum_w_in != nullptr) + (max_w_in != nullptr);
    const int ndin = (d_sum_r_in != nullptr) + (d_sum_w_in != nullptr) + (d_max_w_in != nullptr);
    if ((nin != 0 && nin != 3) || ndin != nin) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !sum_r_out || !sum_w_out || !max_w_out || !kmax || !atap ||
        !d_sum_r_out || !d_sum_w_out || !d_max_w_out || !d_data || !d_kernels || !scratch)
        return SBMC_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)bs * h * w;
    unsigned egrid = (unsigned)((total + 255) / 256);
    if (egrid > 8192) egrid = 8192;
    const int variant = splat_variant();

    if (variant > 0 && strip_ok(c, k, h, w, sizeof(LT) != 4)) {
        SplatBwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out, max_w_out,
                         kmax, atap, d_sum_r_out, d_sum_w_out, d_max_w_out, d_data, d_kernels,
                         d_sum_r_in, d_sum_w_in, d_max_w_in, scratch, bs, c, h, w, k, tiles_x(w), h, h, 0};
        SBMC_DISPATCH_C4(c, hipLaunchKernelGGL((splat_bwd_state_kernel<C, true>), dim3(egrid), dim3(256), 0, s, p));
        int err = (int)hipGetLastError();
        if (err) return err;
        const long items = (long)bs * h * p.ntx;
        const unsigned grid = (unsigned)((items + V2_WAVES - 1) / V2_WAVES);
        SBMC_LAUNCH_STRIP(splat_bwd_strip_kernel, grid, s, p);
        return (int)hipGetLastError();
    }

    if (sizeof(LT) != 4) return SBMC_HIP_EINVAL;  // the generic tile kernels are fp32 only
    const size_t lds = bwd_tile_lds_bytes(c, k);
    if (lds > 64 * 1024) return SBMC_HIP_EINVAL;
    SplatBwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out, max_w_out,
                     kmax, atap, d_sum_r_out, d_sum_w_out, d_max_w_out, d_data, d_kernels,
                     d_sum_r_in, d_sum_w_in, d_max_w_in, scratch,
                     bs, c, h, w, k, tiles_x(w), tiles_y(h, BWD_TY), h, 0};
    SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_bwd_state_kernel<C, false>), dim3(egrid), dim3(256), 0, s, p));
    int err = (int)hipGetLastError();
    if (err) return err;
    const unsigned grid = (unsigned)bs * p.ntx * p.nty;
    SBMC_DISPATCH_C(c, hipLaunchKernelGGL((splat_bwd_tile_kernel<C>), dim3(grid),
                                          dim3(BWD_TY * TX), lds, s, p));
    err = (int)hipGetLastError();
    if (err) return err;
    hipLaunchKernelGGL(splat_bwd_route_kernel, dim3(egrid), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

extern "C" int sbmc_splat_all_supported(int c, int k, int h, int w) { return strip_ok(c, k, h, w) ? 1 : 0; }
extern "C" int sbmc_splat_f16_supported(int c, int k, int h, int w) {
    return splat_variant() > 0 && strip_ok(c, k, h, w, true) ? 1 : 0;
}

extern "C" int sbmc_splat_merge_fwd_f32(const float* part_r, const float* part_w, const float* part_m,
                                        float* sum_r, float* sum_w, float* max_w,
                                        float* run_r, float* run_w, float* run_m,
                                        int bs, int s, int c, int h, int w, void* stream) {
    if (bs < 0 || s < 1 || h < 0 || w < 0 || c < 1 || c > 4) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!part_r || !part_w || !part_m || !sum_r || !sum_w || !max_w || !run_r || !run_w || !run_m)
        return SBMC_HIP_EINVAL;
    SplatMergeParams p{part_r, part_w, part_m, sum_r, sum_w, max_w, run_r, run_w, run_m, bs, s, h, w};
    const size_t total = (size_t)bs * h * w;
    unsigned egrid = (unsigned)((total + 255) / 256);
    if (egrid > 8192) egrid = 8192;
    SBMC_DISPATCH_C4(c, hipLaunchKernelGGL((splat_merge_fwd_kernel<C>), dim3(egrid), dim3(256), 0,
                                           (hipStream_t)stream, p));
    return (int)hipGetLastError();
}

template <typename LT>
static int splat_all_bwd_impl(const float* data, const void* kernels,
                              const float* part_m, const int32_t* atap,
                              const float* run_r, const float* run_w, const float* run_m,
                              const float* d_sum_r, const float* d_sum_w, const float* d_max_w,
                              float* d_data, void* d_kernels, float* scratch,
                              int bs, int s, int c, int h, int w, int k, void* stream, int top = 0, int bot = 0,
                              const unsigned* dmax = nullptr, unsigned* bound = nullptr) {
    if (top < 0 || bot < 0 || h < 0 || top > (k - 1) / 2 || bot > (k - 1) / 2) return SBMC_HIP_EINVAL;
    if ((dmax == nullptr) != (bound == nullptr)) return SBMC_HIP_EINVAL;
    const int hd = top + h + bot;   // the per-pixel quantities (state, records) live on the destination rows
    if (bs < 0 || s < 1 || w < 0 || c < 1 || !strip_ok(c, k, hd, w, sizeof(LT) != 4)) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !part_m || !atap || !run_r || !run_w || !run_m || !d_sum_r || !d_sum_w ||
        !d_max_w || !d_data || !d_kernels || !scratch)
        return SBMC_HIP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    SplatChainParams cp{part_m, atap, run_r, run_w, run_m, d_sum_r, d_sum_w, d_max_w, scratch, bs, s, hd, w, dmax, bound};
    const size_t total = (size_t)bs * hd * w;
    unsigned egrid = (unsigned)((total + 255) / 256);
    if (egrid > 8192) egrid = 8192;
    SBMC_DISPATCH_C4(c, hipLaunchKernelGGL((splat_chain_bwd_kernel<C>), dim3(egrid), dim3(256), 0, st, cp));
    int err = (int)hipGetLastError();
    if (err) return err;
    // one strip launch over all bs*S samples; it only reads data, kernels and the records
    SplatBwdParams p{};
    p.data = data; p.kernels = kernels; p.d_data = d_data; p.d_kernels = d_kernels; p.scratch = scratch;
    p.bs = bs * s; p.c = c; p.h = h; p.w = w; p.k = k; p.ntx = tiles_x(w); p.nty = h;
    p.hd = hd; p.top = top;
    const long items = (long)p.bs * h * p.ntx;
    const unsigned grid = (unsigned)((items + V2_WAVES - 1) / V2_WAVES);
    SBMC_LAUNCH_STRIP(splat_bwd_strip_kernel, grid, st, p);
    return (int)hipGetLastError();
}

// ---- gather-kernel (splat=False) progressive update: strip kernels only, fp32
extern "C" int sbmc_gather_update_supported(int c, int k, int h, int w) {
    return splat_variant() > 0 && strip_ok(c, k, h, w) ? 1 : 0;
}

extern "C" int sbmc_gather_update_fwd_f32(const float* data, const float* kernels,
                                          const float* sum_r_in, const float* sum_w_in, const float* max_w_in,
                                          float* sum_r_out, float* sum_w_out, float* max_w_out,
                                          float* kmax_out, int32_t* atap_out,
                                          int bs, int c, int h, int w, int k, void* stream) {
    using LT = float;
    if (bad_splat_dims(bs, c, h, w, k) || !strip_ok(c, k, h, w)) return SBMC_HIP_EINVAL;
    const int nin = (sum_r_in != nullptr) + (sum_w_in != nullptr) + (max_w_in != nullptr);
    if (nin != 0 && nin != 3) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !sum_r_out || !sum_w_out || !max_w_out || !kmax_out || !atap_out)
        return SBMC_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    SplatFwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out,
                     max_w_out, kmax_out, atap_out, bs, h, w, k, tiles_x(w), h, h, 0, 1, 1};
    const long items = (long)bs * h * p.ntx;
    const unsigned grid = (unsigned)((items + V2_WAVES - 1) / V2_WAVES);
    SBMC_LAUNCH_STRIP(splat_fwd_strip_kernel, grid, s, p, true);
    return (int)hipGetLastError();
}

extern "C" int sbmc_gather_update_bwd_f32(const float* data, const float* kernels,
                                          const float* sum_r_in, const float* sum_w_in, const float* max_w_in,
                                          const float* sum_r_out, const float* sum_w_out, const float* max_w_out,
                                          const float* kmax, const int32_t* atap,
                                          const float* d_sum_r_out, const float* d_sum_w_out,
                                          const float* d_max_w_out,
                                          float* d_data, float* d_kernels,
                                          float* d_sum_r_in, float* d_sum_w_in, float* d_max_w_in,
                                          float* scratch, int bs, int c, int h, int w, int k, void* stream) {
    using LT = float;
    if (bad_splat_dims(bs, c, h, w, k) || !strip_ok(c, k, h, w)) return SBMC_HIP_EINVAL;
    const int nin = (sum_r_in != nullptr) + (sum_w_in != nullptr) + (max_w_in != nullptr);
    const int ndin = (d_sum_r_in != nullptr) + (d_sum_w_in != nullptr) + (d_max_w_in != nullptr);
    if ((nin != 0 && nin != 3) || ndin != nin) return SBMC_HIP_EINVAL;
    if (bs == 0 || h == 0 || w == 0) return 0;
    if (!data || !kernels || !sum_r_out || !sum_w_out || !max_w_out || !kmax || !atap ||
        !d_sum_r_out || !d_sum_w_out || !d_max_w_out || !d_data || !d_kernels || !scratch)
        return SBMC_HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    SplatBwdParams p{data, kernels, sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out, max_w_out,
                     kmax, atap, d_sum_r_out, d_sum_w_out, d_max_w_out, d_data, d_kernels,
                     d_sum_r_in, d_sum_w_in, d_max_w_in, scratch, bs, c, h, w, k, tiles_x(w), h, h, 0};
    const size_t total = (size_t)bs * h * w;
    unsigned egrid = (unsigned)((total + 255) / 256);
    if (egrid > 8192) egrid = 8192;
    SBMC_DISPATCH_C4(c, hipLaunchKernelGGL((splat_bwd_state_kernel<C, true>), dim3(egrid), dim3(256), 0, s, p));
    int err = (int)hipGetLastError();
    if (err) return err;
    const long items = (long)bs * h * p.ntx;
    const unsigned grid = (unsigned)((items + V2_WAVES - 1) / V2_WAVES);
    SBMC_LAUNCH_STRIP(gather_bwd_dg_kernel, grid, s, p);
    err = (int)hipGetLastError();
    if (err) return err;
    SBMC_LAUNCH_STRIP(gather_bwd_ddata_kernel, grid, s, p);
    return (int)hipGetLastError();
}

#define SPLAT_FWD_ARGS const float *sum_r_in, const float *sum_w_in, const float *max_w_in, float *sum_r_out,  \
                       float *sum_w_out, float *max_w_out, float *kmax_out, int32_t *atap_out, int bs, int c,  \
                       int h, int w, int k, void *stream
#define SPLAT_FWD_PASS sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out, max_w_out, kmax_out, atap_out, bs, c, h, w, k, stream
extern "C" int sbmc_splat_update_fwd_f32(const float* data, const float* kernels, SPLAT_FWD_ARGS) {
    return splat_update_fwd_impl<float>(data, kernels, SPLAT_FWD_PASS);
}
extern "C" int sbmc_splat_update_fwd_f16(const float* data, const void* kernels, SPLAT_FWD_ARGS) {
    return splat_update_fwd_impl<_Float16>(data, kernels, SPLAT_FWD_PASS);
}

#define SPLAT_BWD_ARGS const float *sum_r_in, const float *sum_w_in, const float *max_w_in,                    \
                       const float *sum_r_out, const float *sum_w_out, const float *max_w_out,                 \
                       const float *kmax, const int32_t *atap, const float *d_sum_r_out,                       \
                       const float *d_sum_w_out, const float *d_max_w_out
#define SPLAT_BWD_TAIL float *d_sum_r_in, float *d_sum_w_in, float *d_max_w_in, float *scratch, int bs, int c, \
                       int h, int w, int k, void *stream
#define SPLAT_BWD_PASS1 sum_r_in, sum_w_in, max_w_in, sum_r_out, sum_w_out, max_w_out, kmax, atap, d_sum_r_out, d_sum_w_out, d_max_w_out
#define SPLAT_BWD_PASS2 d_sum_r_in, d_sum_w_in, d_max_w_in, scratch, bs, c, h, w, k, stream
extern "C" int sbmc_splat_update_bwd_f32(const float* data, const float* kernels, SPLAT_BWD_ARGS,
                                         float* d_data, float* d_kernels, SPLAT_BWD_TAIL) {
    return splat_update_bwd_impl<float>(data, kernels, SPLAT_BWD_PASS1, d_data, d_kernels, SPLAT_BWD_PASS2);
}
extern "C" int sbmc_splat_update_bwd_f16(const float* data, const void* kernels, SPLAT_BWD_ARGS,
                                         float* d_data, void* d_kernels, SPLAT_BWD_TAIL) {
    return splat_update_bwd_impl<_Float16>(data, kernels, SPLAT_BWD_PASS1, d_data, d_kernels, SPLAT_BWD_PASS2);
}

#define SPLAT_ALL_ARGS const float *part_m, const int32_t *atap, const float *run_r, const float *run_w,       \
                       const float *run_m, const float *d_sum_r, const float *d_sum_w, const float *d_max_w
#define SPLAT_ALL_PASS part_m, atap, run_r, run_w, run_m, d_sum_r, d_sum_w, d_max_w
extern "C" int sbmc_splat_all_bwd_f32(const float* data, const float* kernels, SPLAT_ALL_ARGS, float* d_data,
                                      float* d_kernels, float* scratch, int bs, int s, int c, int h, int w,
                                      int k, void* stream) {
    return splat_all_bwd_impl<float>(data, kernels, SPLAT_ALL_PASS, d_data, d_kernels, scratch, bs, s, c, h, w, k, stream);
}
extern "C" int sbmc_splat_all_bwd_f16(const float* data, const void* kernels, SPLAT_ALL_ARGS, float* d_data,
                                      void* d_kernels, float* scratch, int bs, int s, int c, int h, int w,
                                      int k, void* stream) {
    return splat_all_bwd_impl<_Float16>(data, kernels, SPLAT_ALL_PASS, d_data, d_kernels, scratch, bs, s, c, h, w, k, stream);
}

// ABI 6: the same backward that also leaves an upper bound of |d_kernels| in a device word (SplatChainParams::bound);
// top = bot = 0: the whole frame (sbmc_splat_all_bwd_f32), else the row-slab form (sbmc_splat_slab_bwd_f32).
extern "C" int sbmc_splat_all_bwd_bound_f32(const float* data, const float* kernels, SPLAT_ALL_ARGS, float* d_data,
                                            float* d_kernels, float* scratch, const unsigned* dmax, unsigned* bound,
                                            int bs, int s, int c, int h, int w, int k, int top, int bot, void* stream) {
    if (!dmax || !bound) return SBMC_HIP_EINVAL;
    return splat_all_bwd_impl<float>(data, kernels, SPLAT_ALL_PASS, d_data, d_kernels, scratch, bs, s, c, h, w, k, stream,
                                     top, bot, dmax, bound);
}

// ---- row-slab form (one frame sharded along H over several GPUs)
extern "C" int sbmc_splat_slab_supported(int c, int k, int h, int w, int top, int bot) {
    if (h < 1 || w < 1 || top < 0 || bot < 0 || k < 1 || (k % 2) == 0) return 0;
    if (top > (k - 1) / 2 || bot > (k - 1) / 2) return 0;
    return strip_ok(c, k, top + h + bot, w) ? 1 : 0;
}
#define SPLAT_SLAB_FWD_ARGS float *part_r, float *part_w, float *part_m, int32_t *atap, int n, int c, int h, int w, \
                            int k, int top, int bot, int zero_top, int zero_bot, void *stream
extern "C" int sbmc_splat_slab_fwd_f32(const float* data, const float* kernels, SPLAT_SLAB_FWD_ARGS) {
    return splat_update_fwd_impl<float>(data, kernels, nullptr, nullptr, nullptr, part_r, part_w, part_m, nullptr,
                                        atap, n, c, h, w, k, stream, top, bot, zero_top != 0, zero_bot != 0);
}
extern "C" int sbmc_splat_slab_fwd_f16(const float* data, const void* kernels, SPLAT_SLAB_FWD_ARGS) {
    return splat_update_fwd_impl<_Float16>(data, kernels, nullptr, nullptr, nullptr, part_r, part_w, part_m, nullptr,
                                           atap, n, c, h, w, k, stream, top, bot, zero_top != 0, zero_bot != 0);
}
extern "C" int sbmc_splat_slab_bwd_f32(const float* data, const float* kernels, SPLAT_ALL_ARGS, float* d_data,
                                       float* d_kernels, float* scratch, int bs, int s, int c, int h, int w,
                                       int k, int top, int bot, void* stream) {
    return splat_all_bwd_impl<float>(data, kernels, SPLAT_ALL_PASS, d_data, d_kernels, scratch, bs, s, c, h, w, k,
                                     stream, top, bot);
}
extern "C" int sbmc_splat_slab_bwd_f16(const float* data, const void* kernels, SPLAT_ALL_ARGS, float* d_data,
                                       void* d_kernels, float* scratch, int bs, int s, int c, int h, int w,
                                       int k, int top, int bot, void* stream) {
    return splat_all_bwd_impl<_Float16>(data, kernels, SPLAT_ALL_PASS, d_data, d_kernels, scratch, bs, s, c, h, w, k,
                                        stream, top, bot);
}

extern "C" int sbmc_hip_abi_version(void) { return SBMC_HIP_ABI_VERSION; }

extern "C" const char* sbmc_hip_strerror(int code) {
    if (code == 0) return "success";
    if (code == SBMC_HIP_EINVAL) return "sbmc_hip: invalid argument (shape, null pointer, channel count or LDS budget)";
    return hipGetErrorString((hipError_t)code);
}

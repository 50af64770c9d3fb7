// The up-path of the U-net (reference sbmc/modules.py:300-320: F.interpolate(scale 2, bilinear,
// align_corners=False) followed by th.cat([up, left], 1)) in one pass per direction.
//   forward : out[:, :cu] = bilinear x2 of coarse, out[:, cu:] = left       (one write of `out`,
//             instead of an upsampled tensor that is written, read again and copied by cat)
//   backward: gcoarse = adjoint of the x2 upsampling applied to gout[:, :cu], as a GATHER (each
//             coarse pixel sums its <= 4x4 fine neighbours with the separable weights
//             {0.25, 0.75, 0.75, 0.25}, edge-clamped): no atomics, deterministic -- PyTorch's
//             upsample_bilinear2d_backward scatters with atomicAdd (0.82 ms per call at 720p)
// Source coordinates as PyTorch computes them for scale 0.5: src = 0.5 * dst - 0.25, clamped at
// 0; i0 = floor(src), i1 = min(i0 + 1, n - 1), lambda = src - i0.
// HBM-bound: every element is read once and written once.
#include "common.hpp"
#include "../../include/sbmc_hip.h"

namespace sbmc {

// one thread: 4 consecutive output pixels of one row of one (b, c) plane
// Row-slab form (one frame sharded along H, sbmc_amd/dist.py): `coarse` holds hc = top + h + bot rows,
// the first `top` / last `bot` (0 or 1) of which are the neighbouring slabs' edge rows; the output is the
// 2h fine rows of this slab, i.e. rows 2 top .. 2 top + 2h of the upsampled padded map.  The edge clamp
// of the interpolation therefore only ever acts at a true image border (top == 0 / bot == 0).
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const float* __restrict__ coarse, const float* __restrict__ left,
                                                       float* __restrict__ out, int cu, int cl, int hc, int w,
                                                       int top, int bot, size_t total4) {
    const int h = hc - top - bot;
    const int W = 2 * w, H = 2 * h, W4 = W / 4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % W4);
        size_t rest = idx / W4;
        const int y = (int)(rest % H);
        rest /= H;
        const int c = (int)(rest % (cu + cl));
        const size_t b = rest / (cu + cl);
        float4 v;
        if (c >= cu) {
            v = reinterpret_cast<const float4*>(left + ((b * cl + (c - cu)) * H + y) * (size_t)W)[q];
        } else {
            // rows: y even -> (i-1: .25, i: .75), y odd -> (i: .75, i+1: .25), clamped
            const int yf = y + 2 * top;             // row of the upsampled (padded) coarse map
            const int i = yf >> 1;
            int r0, r1;
            float l1;                               // weight of r1
            if (yf == 0) { r0 = 0; r1 = 0; l1 = 0.f; }
            else if (yf & 1) { r0 = i; r1 = i + 1 < hc ? i + 1 : i; l1 = 0.25f; }
            else { r0 = i - 1; r1 = i; l1 = 0.75f; }
            const float l0 = 1.f - l1;
            const float* p0 = coarse + ((b * cu + c) * hc + r0) * (size_t)w;
            const float* p1 = coarse + ((b * cu + c) * hc + r1) * (size_t)w;
            // columns 4q .. 4q+3 come from coarse columns 2q-1 .. 2q+2
            const int j = 2 * q;
            const int jm = j > 0 ? j - 1 : 0, jp = j + 2 < w ? j + 2 : w - 1;
            const float a0 = p0[jm], a1 = p0[j], a2 = p0[j + 1], a3 = p0[jp];
            const float b0 = p1[jm], b1 = p1[j], b2 = p1[j + 1], b3 = p1[jp];
            auto hx = [&](float m, float c0, float c1, float pl) {
                // the four horizontal interpolants of one coarse row
                float4 r;
                r.x = q == 0 ? c0 : 0.25f * m + 0.75f * c0;       // x = 4q     : (2q-1, 2q)
                r.y = 0.75f * c0 + 0.25f * c1;                    // x = 4q + 1 : (2q, 2q+1)
                r.z = 0.25f * c0 + 0.75f * c1;                    // x = 4q + 2 : (2q, 2q+1)
                r.w = 0.75f * c1 + 0.25f * pl;                    // x = 4q + 3 : (2q+1, 2q+2 clamped)
                return r;
            };
            const float4 ra = hx(a0, a1, a2, a3), rb = hx(b0, b1, b2, b3);
            v.x = l0 * ra.x + l1 * rb.x;
            v.y = l0 * ra.y + l1 * rb.y;
            v.z = l0 * ra.z + l1 * rb.z;
            v.w = l0 * ra.w + l1 * rb.w;
        }
        reinterpret_cast<float4*>(out + ((b * (cu + cl) + c) * H + y) * (size_t)W)[q] = v;
    }
}

// one thread: two adjacent coarse pixels (i, 2q), (i, 2q + 1) of one (b, c) plane: their 4 x 6 fine
// neighbourhood is one aligned float4 plus one pixel on each side per row
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gcoarse,
                                                       int cu, int cl, int hc, int w, int top, int bot,
                                                       size_t total2) {
    const int h = hc - top - bot;
    const int W = 2 * w, H = 2 * h, w2 = w / 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total2;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % w2);
        size_t rest = idx / w2;
        const int i = (int)(rest % hc);             // coarse row (of the hc padded rows)
        rest /= hc;
        const int c = (int)(rest % cu);
        const size_t b = rest / cu;
        const float* g = gout + ((b * (cu + cl) + c) * H) * (size_t)W;
        // fine rows 2i-1 .. 2i+2 with weights .25 .75 .75 .25; a partner that falls off the image
        // was clamped to this row / column in the forward, so its weight comes back here
        float wy[4] = {0.25f, 0.75f, 0.75f, 0.25f};
        if (i == 0 && top == 0) { wy[0] = 0.f; wy[1] = 1.f; }            // true image borders only
        if (i == hc - 1 && bot == 0) { wy[3] = 0.f; wy[2] = 1.f; }
        const bool first = q == 0, last = q == w2 - 1;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            if (wy[dy] == 0.f) continue;
            const int y = 2 * i - 1 + dy - 2 * top;   // fine row of this slab; outside: another slab's share
            if (y < 0 || y >= H) continue;
            const float* row = g + (size_t)y * W + 4 * q;
            const float4 m = *reinterpret_cast<const float4*>(row);       // fine columns 4q .. 4q+3
            const float lft = first ? 0.f : row[-1], rgt = last ? 0.f : row[4];
            // coarse column 2q   <- fine 4q-1 (.25), 4q (.75, or 1 at the left edge), 4q+1 (.75), 4q+2 (.25)
            // coarse column 2q+1 <- fine 4q+1 (.25), 4q+2 (.75), 4q+3 (.75, or 1 at the right edge), 4q+4 (.25)
            const float s0 = 0.25f * lft + (first ? 1.f : 0.75f) * m.x + 0.75f * m.y + 0.25f * m.z;
            const float s1 = 0.25f * m.y + 0.75f * m.z + (last ? 1.f : 0.75f) * m.w + 0.25f * rgt;
            a0 += wy[dy] * s0;
            a1 += wy[dy] * s1;
        }
        *reinterpret_cast<float2*>(gcoarse + ((b * cu + c) * hc + i) * (size_t)w + 2 * q) = make_float2(a0, a1);
    }
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_upsample2x_cat_supported(int h, int w) { return (h >= 1 && w >= 2 && w % 2 == 0) ? 1 : 0; }

static int upcat_fwd_impl(const float* coarse, const float* left, float* out, int b, int cu, int cl,
                          int hc, int w, int top, int bot, void* stream) {
    const int h = hc - top - bot;
    if (b < 0 || cu < 1 || cl < 0 || top < 0 || top > 1 || bot < 0 || bot > 1 || !sbmc_upsample2x_cat_supported(h, w))
        return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!coarse || !out || (cl > 0 && !left) || (uintptr_t)out % 16 || (uintptr_t)left % 16) return SBMC_HIP_EINVAL;
    const size_t total4 = (size_t)b * (cu + cl) * (2 * (size_t)h) * (2 * (size_t)w / 4);
    const size_t blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(upcat_fwd_kernel, dim3((unsigned)(blocks < 65536 * 16 ? blocks : 65536 * 16)), dim3(256), 0,
                       (hipStream_t)stream, coarse, left, out, cu, cl, hc, w, top, bot, total4);
    return (int)hipGetLastError();
}

static int upcat_bwd_impl(const float* gout, float* gcoarse, int b, int cu, int cl, int hc, int w, int top, int bot,
                          void* stream) {
    const int h = hc - top - bot;
    if (b < 0 || cu < 1 || cl < 0 || top < 0 || top > 1 || bot < 0 || bot > 1 || !sbmc_upsample2x_cat_supported(h, w))
        return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!gout || !gcoarse || (uintptr_t)gout % 16 || (uintptr_t)gcoarse % 8) return SBMC_HIP_EINVAL;
    const size_t total2 = (size_t)b * cu * hc * (w / 2);
    const size_t blocks = (total2 + 255) / 256;
    hipLaunchKernelGGL(upcat_bwd_kernel, dim3((unsigned)(blocks < 65536 * 16 ? blocks : 65536 * 16)), dim3(256), 0,
                       (hipStream_t)stream, gout, gcoarse, cu, cl, hc, w, top, bot, total2);
    return (int)hipGetLastError();
}

extern "C" int sbmc_upsample2x_cat_fwd_f32(const float* coarse, const float* left, float* out, int b, int cu, int cl,
                                           int h, int w, void* stream) {
    return upcat_fwd_impl(coarse, left, out, b, cu, cl, h, w, 0, 0, stream);
}
extern "C" int sbmc_upsample2x_cat_bwd_f32(const float* gout, float* gcoarse, int b, int cu, int cl, int h, int w,
                                           void* stream) {
    return upcat_bwd_impl(gout, gcoarse, b, cu, cl, h, w, 0, 0, stream);
}
extern "C" int sbmc_upsample2x_cat_slab_fwd_f32(const float* coarse, const float* left, float* out, int b, int cu,
                                                int cl, int hc, int w, int top, int bot, void* stream) {
    return upcat_fwd_impl(coarse, left, out, b, cu, cl, hc, w, top, bot, stream);
}
extern "C" int sbmc_upsample2x_cat_slab_bwd_f32(const float* gout, float* gcoarse, int b, int cu, int cl, int hc,
                                                int w, int top, int bot, void* stream) {
    return upcat_bwd_impl(gout, gcoarse, b, cu, cl, hc, w, top, bot, stream);
}

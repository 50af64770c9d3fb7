// Per-sample 1x1 convolutions of the kernel-predicting CNN as fused fp32 MFMA kernels.
//
// The reference builds its per-sample embeddings and its kernel regressor from chains of
// nn.Conv2d(1x1) + ReLU / LeakyReLU (sbmc/modules.py:154-175, used at sbmc/models.py:79-102,
// 147-153, 171-177, 196-199).  On the planar activations x[B, Cin, H*W] such a layer is
//     y[b] = act(W @ x[b] + bias (+ t))            W: [Cout, Cin],  Cin <= 128
// with 2*Cin flop per 4-byte output element and per 4*Cin/Cout-byte input: at Cin = Cout = 128
// exactly the machine balance of MI355X (157 TFLOP/s fp32 MFMA : ~5 TB/s), so every extra pass
// over the activations (bias, activation, their adjoints) costs as much as the GEMM itself.
// These kernels do the whole layer in one pass per direction:
//   forward : one persistent workgroup per CU walks tiles of 128 pixels; the [Cin, 128] input
//             tile is staged in LDS (double buffered, prefetched through registers), the weight
//             rows live in VGPRs as MFMA A-operands, each wave owns a 32-row x 64-pixel block of
//             the output (2 accumulators of v_mfma_f32_32x32x2_f32), bias / context term /
//             activation are applied to the accumulators and the result is stored once.
// fp32 MFMA is exact fp32 (an fmaf chain over k): results differ from a library GEMM by
// summation order only.
#include "common.hpp"
#include "../../include/sbmc_hip.h"

namespace sbmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int PW_NT = 128;          // pixels per tile
constexpr int PW_THREADS = 512;     // 8 waves: 4 row blocks of 32 output channels x 2 pixel halves of 64
constexpr unsigned PW_OOB = 0xFFFFFFF0u;
#ifndef PW_KGROUP_V
#define PW_KGROUP_V 8
#endif
constexpr int PW_KGROUP = PW_KGROUP_V;   // k-steps per scheduling group

// Descriptor over [base, base + bytes).  Both are wave-uniform by construction; saying so keeps the
// descriptor in SGPRs (the compiler otherwise guards every access with a waterfall loop).
__device__ __forceinline__ rsrc_t make_rsrc_n(const void* base, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* u = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

struct PwFwdParams {
    const float* x;      // [B, K, hw]
    const float* w;      // [Cout, K]
    const float* bias;   // [Cout]
    const float* t;      // context term: nullptr, [B/S, Cout] (t_mode 1) or [B/S, Cout, hw] (t_mode 2)
    float* y;            // [B, Cout, hw]
    int B, S, K, Cout;
    unsigned hw, tiles_per_plane, ntiles;
    int nrt;             // row tiles of 128 output channels
    int t_mode;
    float slope;         // 1: linear, 0: relu, else leaky relu
};

// KP: Cin rounded up to a multiple of 32 (<= 128); TMODE: the context term (0 none, 1 per image, 2 per pixel).
//
// Persistent grid, one workgroup (8 waves, 2 per SIMD) per CU.  Software pipeline per workgroup:
//   * the input tile of step i+1 is fetched into registers while step i computes, then written to
//     the other LDS buffer (one s_barrier per tile);
//   * the results of step i-1 (already activated, held in registers) are stored BETWEEN the MFMAs
//     of step i, so neither loads nor stores ever stall the matrix pipe.
// Work assignment: workgroup g works on row tile (g / 8) % nrt of the pixel tiles
// ((g / 8) / nrt) * 8 + g % 8 + i * (gridDim.x / nrt): the nrt workgroups that read the same
// input tile sit on the same XCD (g % 8) and walk in step, so the tile comes from HBM once.
template <int KP, int TMODE>
__global__ __launch_bounds__(PW_THREADS) void pw_fwd_kernel(PwFwdParams p) {
    extern __shared__ float4 pw_lds[];
    float* xs = reinterpret_cast<float*>(pw_lds);      // [2][KP][PW_NT]
    constexpr int KS = KP / 2;                         // MFMA k-steps
    constexpr int NG = KS / PW_KGROUP;                 // scheduling groups per tile
    constexpr int NLD = KP / 16;                       // float4 loads per thread per tile
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;

    const unsigned g = blockIdx.x, slot = g / NUM_XCD;
    const int rt = (int)(slot % (unsigned)p.nrt);
    const unsigned first = (slot / (unsigned)p.nrt) * NUM_XCD + g % NUM_XCD;
    const unsigned stride = gridDim.x / (unsigned)p.nrt;
    const int r0 = rt * 128 + rb * 32;
    const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;

    // staging role of this thread: float4 column c4 of rows (threadIdx.x >> 5) + 16 i
    const unsigned c4 = (threadIdx.x & 31) * 4, srow = threadIdx.x >> 5;

    auto tile_coords = [&](unsigned tile, unsigned& b, unsigned& bq, unsigned& p0) {
        // samples of one pixel tile are adjacent in the walk: their context tile stays in L2
        const unsigned s = tile % (unsigned)p.S, rest = tile / (unsigned)p.S;
        const unsigned pt = rest % p.tiles_per_plane;
        bq = rest / p.tiles_per_plane;
        b = bq * (unsigned)p.S + s;
        p0 = pt * PW_NT;
    };
    auto issue_loads = [&](unsigned tile, u32x4 (&regs)[NLD]) {
        unsigned b, bq, p0;
        tile_coords(tile, b, bq, p0);
        const rsrc_t rx = make_rsrc_n(p.x + (size_t)b * p.K * hw, (unsigned)p.K * hw * 4u);
        const bool colok = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const unsigned k = srow + 16u * i;
            const unsigned off = (colok && k < (unsigned)p.K) ? (k * hw + p0 + c4) * 4u : PW_OOB;
            regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
    };
    auto commit = [&](int buf, const u32x4 (&regs)[NLD]) {
        float* dst = xs + buf * (KP * PW_NT);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            *reinterpret_cast<u32x4*>(dst + (srow + 16 * i) * PW_NT + c4) = regs[i];
    };

    // weight rows of this wave as MFMA A-operands: a[kk] = W[r0 + lane % 32][2 kk + lane / 32]
    float a[KS];
    {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int row = r0 + l31;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const bool ok = row < p.Cout && 2 * kk + lhi < p.K;
            a[kk] = buf_load(rw, ok ? (unsigned)(row * p.K + 2 * kk + lhi) * 4u : PW_OOB, 0);
        }
    }
    // accumulator row of register j: r0 + (j & 3) + 8 (j >> 2) + 4 (lane / 32)
    float bias[16];
    {
        const rsrc_t rbias = make_rsrc_n(p.bias, (unsigned)p.Cout * 4u);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            bias[j] = buf_load(rbias, (unsigned)(r0 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * 4u, 0);
    }

    u32x4 pre[NLD];
    unsigned tile = first;
    if (tile < p.ntiles) {
        issue_loads(tile, pre);
        commit(0, pre);
    }
    __syncthreads();

    // results of the previous step, waiting to be stored
    f32x16 out0, out1;
#pragma unroll
    for (int j = 0; j < 16; ++j) out0[j] = out1[j] = 0.f;
    unsigned b_prev = 0;                               // (uniform) batch element they belong to
    unsigned o_prev0 = PW_OOB, o_prev1 = PW_OOB;       // byte offset of register 0's element, or switched off
    auto store_prev = [&](int j) {
        const rsrc_t ry_prev = make_rsrc_n(p.y + ((size_t)b_prev * p.Cout + r0) * hw, (unsigned)nrows * hw * 4u);
        const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
        buf_store(out0[j], ry_prev, o_prev0 != PW_OOB ? o_prev0 + ro : PW_OOB, 0);
        buf_store(out1[j], ry_prev, o_prev1 != PW_OOB ? o_prev1 + ro : PW_OOB, 0);
    };

    int buf = 0;
    for (; tile < p.ntiles; tile += stride, buf ^= 1) {
        const unsigned next = tile + stride;
#ifndef PW_EXP_NOLOAD
        if (next < p.ntiles) issue_loads(next, pre);
#endif

        unsigned b, bq, p0;
        tile_coords(tile, b, bq, p0);
        const unsigned col = p0 + ph * 64 + l31;
        const unsigned o0 = (col < hw && nrows > 0) ? (4u * lhi * hw + col) * 4u : PW_OOB;
        const unsigned o1 = (col + 32 < hw && nrows > 0) ? (4u * lhi * hw + col + 32) * 4u : PW_OOB;

        f32x16 acc0, acc1;
        {
            const rsrc_t rt1 = make_rsrc_n(TMODE == 1 ? p.t + (size_t)bq * p.Cout : p.bias, (unsigned)p.Cout * 4u);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float v = bias[j];
                if (TMODE == 1) v += buf_load(rt1, (unsigned)(r0 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * 4u, 0);
                acc0[j] = v;
                acc1[j] = v;
            }
        }
        // per-pixel context term of this tile: fetched now, added after the accumulation
        f32x16 t0, t1;
        if (TMODE == 2) {
            const rsrc_t rt2 = make_rsrc_n(p.t + ((size_t)bq * p.Cout + r0) * hw, (unsigned)nrows * hw * 4u);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
                t0[j] = buf_load(rt2, o0 != PW_OOB ? o0 + ro : PW_OOB, 0);
                t1[j] = buf_load(rt2, o1 != PW_OOB ? o1 + ro : PW_OOB, 0);
            }
        }

        const float* xb = xs + buf * (KP * PW_NT) + lhi * PW_NT + ph * 64 + l31;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
            for (int kk = grp * PW_KGROUP; kk < (grp + 1) * PW_KGROUP; ++kk) {
#ifdef PW_EXP_NOLDS
                const float b0 = a[(kk + 1) % KS], b1 = a[(kk + 2) % KS];
#else
                const float b0 = xb[(2 * kk) * PW_NT];
                const float b1 = xb[(2 * kk) * PW_NT + 32];
#endif
#ifndef PW_EXP_NOMFMA
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b1, acc1, 0, 0, 0);
#else
                acc0[kk & 15] += a[kk] * b0;
                acc1[kk & 15] += a[kk] * b1;
#endif
            }
#ifndef PW_EXP_NOSTORE
#pragma unroll
            for (int j = grp * 16 / NG; j < (grp + 1) * 16 / NG; ++j) store_prev(j);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }

        // context term + activation; the stores happen during the next step
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v0 = acc0[j], v1 = acc1[j];
            if (TMODE == 2) {
                v0 += t0[j];
                v1 += t1[j];
            }
            out0[j] = v0 > 0.f ? v0 : v0 * p.slope;
            out1[j] = v1 > 0.f ? v1 : v1 * p.slope;
        }
        b_prev = __builtin_amdgcn_readfirstlane(b);
        o_prev0 = o0;
        o_prev1 = o1;

        if (next < p.ntiles) commit(buf ^ 1, pre);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) store_prev(j);
}

static bool pw_dims_ok(int cin, int cout, long hw) {
    if (cin < 1 || cin > 128 || cout < 1 || hw < 4 || hw % 4) return false;
    const int kp = (cin + 31) / 32 * 32;
    return (double)kp * (double)hw * 4.0 < 4294967000.0 && hw < (1L << 27);
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_pointwise_supported(int cin, int cout, long hw) { return pw_dims_ok(cin, cout, hw) ? 1 : 0; }

extern "C" int sbmc_pointwise_fwd_f32(const float* x, const float* w, const float* bias, const float* t, float* y,
                                      int b, int s, int cin, int cout, long hw, int t_mode, int act, float slope,
                                      void* stream) {
    if (b < 0 || s < 1 || act < 0 || act > 2 || t_mode < 0 || t_mode > 2) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!pw_dims_ok(cin, cout, hw) || b % s || !x || !w || !bias || !y || (t_mode && !t)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)x % 16) return SBMC_HIP_EINVAL;
    PwFwdParams p;
    p.x = x; p.w = w; p.bias = bias; p.t = t; p.y = y;
    p.B = b; p.S = t_mode ? s : 1; p.K = cin; p.Cout = cout;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + PW_NT - 1) / PW_NT);
    const unsigned long long nt = (unsigned long long)p.tiles_per_plane * (unsigned)b;
    if (nt > 0xFFFFFFFFull - 4096) return SBMC_HIP_EINVAL;
    p.ntiles = (unsigned)nt;
    p.t_mode = t_mode;
    p.slope = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
    p.nrt = (cout + 127) / 128;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int kp = (cin + 31) / 32 * 32;
    const size_t lds = (size_t)2 * kp * PW_NT * sizeof(float);
    // a multiple of 8 * nrt workgroups (see the kernel's work assignment), no more than there is work
    unsigned unit = (unsigned)(NUM_XCD * p.nrt);
    unsigned grid = (unsigned)cus / unit * unit;
    const unsigned long long need = ((unsigned long long)p.ntiles + NUM_XCD - 1) / NUM_XCD * unit;
    if (grid > need) grid = (unsigned)need;
    if (grid < unit) grid = unit;
    hipError_t e = hipSuccess;
#define SBMC_PW_LAUNCH(KPV)                                                                              \
    do {                                                                                                 \
        auto kern = t_mode == 2 ? pw_fwd_kernel<KPV, 2> : (t_mode == 1 ? pw_fwd_kernel<KPV, 1> : pw_fwd_kernel<KPV, 0>); \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        if (e == hipSuccess)                                                                             \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(PW_THREADS), lds, (hipStream_t)stream, p);         \
    } while (0)
    switch (kp) {
        case 32: SBMC_PW_LAUNCH(32); break;
        case 64: SBMC_PW_LAUNCH(64); break;
        case 96: SBMC_PW_LAUNCH(96); break;
        default: SBMC_PW_LAUNCH(128); break;
    }
#undef SBMC_PW_LAUNCH
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

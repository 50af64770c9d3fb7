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
#include <cstring>
#include <type_traits>
#include <utility>
#include "../../include/sbmc_hip.h"
#include <stdlib.h>

namespace sbmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int PW_FWD_PH = 2;        // pixel halves per workgroup: 2 = one 8-wave workgroup per CU, 128-pixel tiles
                                    // (1 = two 4-wave workgroups per CU on 64-pixel tiles: measured equal)
constexpr int PW_THREADS = 512;     // 8 waves: 4 row blocks of 32 output channels x 2 pixel halves of 64
constexpr unsigned PW_OOB = 0xFFFFFFF0u;
constexpr int PW_KGROUP = 8;        // k-steps per scheduling group (2 / 4 / 16 measured: no difference)

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
    const void* x;       // [B, K, hw], float or _Float16 (template TI)
    const float* w;      // [Cout, K]
    const float* bias;   // [Cout]
    const float* t;      // context term: nullptr, [B/S, Cout] (t_mode 1) or [B/S, Cout, hw] (t_mode 2)
    void* y;             // [B, Cout, hw], float or _Float16 (template TO)
    int B, S, K, Cout;
    unsigned hw, tiles_per_plane, ntiles;
    int nrt;             // row tiles of 128 output channels
    int t_mode;
    float slope;         // 1: linear, 0: relu, else leaky relu
    unsigned* signs;     // [B, Cout, ceil(hw / 32)] one bit per output: pre-activation > 0 (nullptr: not wanted)
    float* ymean;        // [B / S, Cout, hw] mean of y over the S samples of a pixel (pw_fwd_s_kernel; nullptr: not wanted)
    const unsigned* xmax;   // pw_fwd_s_kernel<.., F2>: device word, bit pattern of a float >= max |x| (the tensor's scale)
    unsigned* amax;         // pw_fwd_s_kernel: raised to the bit pattern of max |y| (a word zeroed by the caller); or nullptr
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
//
// TI / TO: storage type of x / y in HBM (float, or _Float16 for "fp16 activations": BASELINE
// configs[4]); the LDS tile, the MFMAs and the epilogue are fp32 either way.
template <int KP, int TMODE, int PH, typename TI, typename TO>
__global__ __launch_bounds__(256 * PH) void pw_fwd_kernel(PwFwdParams p) {
    constexpr int NT = 64 * PH;                        // pixels per tile
    constexpr bool HI = sizeof(TI) == 2, HO = sizeof(TO) == 2;
    // staging: every thread moves 16 bytes of one row per pass = 4 float / 8 half pixels
    constexpr int PXT = HI ? 8 : 4;                    // pixels per thread and pass
    constexpr int RPP = 256 * PH / (NT / PXT);         // rows per pass
    const TI* xg = static_cast<const TI*>(p.x);
    TO* yg = static_cast<TO*>(p.y);
    extern __shared__ float4 pw_lds[];
    float* xs = reinterpret_cast<float*>(pw_lds);      // [2][KP][NT]
    constexpr int KS = KP / 2;                         // MFMA k-steps
    constexpr int NG = KS / PW_KGROUP;                 // scheduling groups per tile
    constexpr int NLD = KP / RPP;                      // 16-byte loads per thread per tile
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

    // staging role of this thread: pixels c4 .. c4 + PXT - 1 of rows srow + RPP i
    const unsigned c4 = (threadIdx.x % (NT / PXT)) * PXT, srow = threadIdx.x / (NT / PXT);

    auto tile_coords = [&](unsigned tile, unsigned& b, unsigned& bq, unsigned& p0) {
        // samples of one pixel tile are adjacent in the walk: their context tile stays in L2
        const unsigned s = tile % (unsigned)p.S, rest = tile / (unsigned)p.S;
        const unsigned pt = rest % p.tiles_per_plane;
        bq = rest / p.tiles_per_plane;
        b = bq * (unsigned)p.S + s;
        p0 = pt * NT;
    };
    auto issue_loads = [&](unsigned tile, u32x4 (&regs)[NLD]) {
        unsigned b, bq, p0;
        tile_coords(tile, b, bq, p0);
        const rsrc_t rx = make_rsrc_n(xg + (size_t)b * p.K * hw, (unsigned)p.K * hw * (unsigned)sizeof(TI));
        const bool colok = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const unsigned k = srow + (unsigned)RPP * i;
            const unsigned off = (colok && k < (unsigned)p.K) ? (k * hw + p0 + c4) * (unsigned)sizeof(TI) : PW_OOB;
            regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
    };
    auto commit = [&](int buf, const u32x4 (&regs)[NLD]) {
        float* dst = xs + buf * (KP * NT);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float* d = dst + (srow + RPP * i) * NT + c4;
            if constexpr (HI) {
                using h8 = __attribute__((ext_vector_type(8))) _Float16;
                const h8 v = __builtin_bit_cast(h8, regs[i]);
                *reinterpret_cast<float4*>(d) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
                *reinterpret_cast<float4*>(d + 4) = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
            } else {
                *reinterpret_cast<u32x4*>(d) = regs[i];
            }
        }
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
    // (o_prev*: byte offsets for 4-byte elements; half outputs sit at half of them)
    auto store_prev = [&](int j) {
        const rsrc_t ry_prev = make_rsrc_n(yg + ((size_t)b_prev * p.Cout + r0) * hw,
                                           (unsigned)nrows * hw * (unsigned)sizeof(TO));
        const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
        const unsigned a0 = o_prev0 != PW_OOB ? (o_prev0 + ro) / (HO ? 2u : 1u) : PW_OOB;
        const unsigned a1 = o_prev1 != PW_OOB ? (o_prev1 + ro) / (HO ? 2u : 1u) : PW_OOB;
        logit_store<TO>(out0[j], ry_prev, a0, 0);
        logit_store<TO>(out1[j], ry_prev, a1, 0);
    };

    int buf = 0;
    for (; tile < p.ntiles; tile += stride, buf ^= 1) {
        const unsigned next = tile + stride;
        if (next < p.ntiles) issue_loads(next, pre);

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

        const float* xb = xs + buf * (KP * NT) + lhi * NT + ph * 64 + l31;
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
#pragma unroll
            for (int kk = grp * PW_KGROUP; kk < (grp + 1) * PW_KGROUP; ++kk) {
                const float b0 = xb[(2 * kk) * NT];
                const float b1 = xb[(2 * kk) * NT + 32];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = grp * 16 / NG; j < (grp + 1) * 16 / NG; ++j) store_prev(j);
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

// ---------------------------------------------------------------------------------------------
// The fp32 forward on the bf16 matrix pipe, at fp32 accuracy ("split precision").
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate, and at 128 -> 128 channels the layer needs the
// fp32 matrix pipe and ~5 TB/s of HBM at once (the kernel above: 0.6 of the fp32 MFMA peak, clocks down under
// the load).  An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 mantissa bits, fp32's exponent
// range):  x = xh + xm + xl,  xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)  -- so
//     w x = wh xh + (wh xm + wm xh) + (wm xm + wh xl + wl xh) + [wm xl + wl xm + wl xl: <= 2^-23 |w x|, dropped]
// is six bf16 products with exact fp32 partial products, accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
// 6 / 16 of the fp32 MFMA time.  Error per product term <= ~2^-23 relative, the size of one fp32 rounding (the
// fp32-MFMA kernel rounds once per term as well); measured against float64 both kernels sit at ~1e-7 of the
// output scale.  The layer becomes HBM-bound (the split costs ~6 VALU operations per input value, hidden behind
// the other wave of the SIMD).
//
// Tile: 64 pixels; 8 waves = 4 blocks of 32 output channels x 2 pixel halves of 32 (one 32 x 32 accumulator
// pair per wave: large and small terms apart, so that no MFMA waits for the previous one's result).  The B
// operand of the bf16 MFMA wants 8 CONSECUTIVE input channels of one pixel per lane: a staging thread owns one
// pixel (lane) and one channel octet (wave), loads its 8 values with 8 coalesced row reads (256 B per wave and
// row), splits them in registers and writes three 16-byte entries; the LDS image of a tile is
// [plane][K / 8][64 pixels] entries, an operand fetch one conflict-free ds_read_b128.  The A operand (weights)
// lives in registers in its three planes.  Double-buffered LDS stage, next tile's loads in flight during the
// MFMAs.  Optionally writes one SIGN BIT per output (pre-activation > 0): the backward then needs no y.
using bf8 = __attribute__((ext_vector_type(8))) __bf16;
using bf2 = __attribute__((ext_vector_type(2))) __bf16;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);                 // v_cvt_pk_bf16_f32 (round to nearest even)
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 floats -> their three bf16 planes (each 8 bf16 = 16 bytes, element i in bits 16 (i & 1) of word i / 2)
__device__ __forceinline__ void split3(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned hp = pack_bf16(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - bf16_lo(hp), r1 = v[2 * i + 1] - bf16_hi(hp);       // exact
        const unsigned mp = pack_bf16(r0, r1);
        const float s0 = r0 - bf16_lo(mp), s1 = r1 - bf16_hi(mp);                       // exact
        h[i] = hp;
        m[i] = mp;
        l[i] = pack_bf16(s0, s1);
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL
// access (s_waitcnt vmcnt(0)): loads issued for later tiles would have to land before each barrier, i.e.
// within one tile's time -- exactly what a prefetch across tiles must avoid.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// lanes LANE and LANE + 32 of v = the halves of a ballot (v_writelane_b32: this compiler has the readlane builtin only; the
// lane is an inline constant -- the instruction reads one scalar register).  The s_nop: on gfx940+ a VALU that reads
// a scalar register a VALU (the compare) has just written needs two wait states, and the compiler's hazard recognizer
// does not look into inline assembly (without it the low word of some rows came out stale).
template <int LANE>
__device__ __forceinline__ void write_lanes(unsigned& v, unsigned long long ballot) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
        : "+v"(v) : "s"((unsigned)ballot), "s"((unsigned)(ballot >> 32)), "n"(LANE), "n"(LANE + 32));
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in its body
template <class F, int... J>
__device__ __forceinline__ void unrolled_impl(F&& f, std::integer_sequence<int, J...>) {
    (f(std::integral_constant<int, J>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void unrolled(F&& f) { unrolled_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
using hf8 = __attribute__((ext_vector_type(8))) _Float16;
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a), __builtin_bit_cast(hf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16_f16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, a), __builtin_bit_cast(hf8, b), c, 0, 0, 0);
}
// 8 floats under a (wave-uniform) power-of-two scale -> their two f16 planes (common.hpp f16_split_pair)
__device__ __forceinline__ void split2(const float (&v)[8], float c, u32x4& h, u32x4& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hp, lp;
        f16_split_pair(v[2 * q], v[2 * q + 1], c, hp, lp);
        h[q] = hp;
        l[q] = lp;
    }
}
// the largest magnitude over a wave (bit patterns of magnitudes order like unsigned integers), in every lane
__device__ __forceinline__ unsigned wave_umax(unsigned m) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, s, 64);
        m = m > o ? m : o;
    }
    return m;
}

constexpr int PS_NT = 64;           // pixels per tile of the split-precision kernels

// WAVES: 8 (two waves per SIMD, 256 registers each; a wave owns 32 channels x 32 pixels) or 4 (one wave per
// SIMD with the whole 512-entry register file -- accumulators in AccVGPRs --; a wave owns 32 channels x 64
// pixels): the second form leaves the compiler room to keep LDS reads and the split of the next tile in
// flight between the MFMAs instead of waiting for each operand.
// TO: storage type of y (float; _Float16 for a chain's FIRST layer under fp16 activations -- fp32 network input in, half
// out: until round 4 that layer ran on the fp32-MFMA kernel; no sign bits or mean in that form).
// MEAN: p.ymean is written (compile time: the epilogue has no branch on it).
// F2 (round 5): the 3 x 3 kernels' number format -- TWO f16 planes under a power-of-two scale (x: from the device word
// p.xmax, the producer's largest magnitude; the weights: from the largest of the wave's own 32 rows) and THREE products
// hh + hl + lh on v_mfma_f32_32x32x16_f16 instead of three bf16 planes and six: half the matrix-pipe cycles, two thirds of
// the LDS image, a third of the split's instructions (common.hpp).  Per product ~2 bits coarser than the bf16 form (a
// term <= 2^-22 |x w| dropped instead of <= 2^-23), fp32 accumulation either way.
// Both forms raise *p.amax to max |y| (where the caller asks): the next layer's scale without a pass of its own.
template <int KP, int TMODE, int WAVES, typename TO = float, bool MEAN = false, bool F2 = false>
__global__ __launch_bounds__(64 * WAVES) void pw_fwd_s_kernel(PwFwdParams p) {
    constexpr unsigned SO = (unsigned)sizeof(TO);
    constexpr int NP = F2 ? 2 : 3;                     // planes of an operand
    constexpr int KO = KP / 8;                         // channel octets
    constexpr int KS = KP / 16;                        // MFMA k-steps
    constexpr int NOCT = (KO + WAVES - 1) / WAVES;     // octets a staging thread owns
    constexpr int NPH = 8 / WAVES;                     // 32-pixel blocks a wave owns
    // large and small terms in accumulators of their own (no MFMA waits for the previous one's result) -- except
    // where a per-pixel context term needs the 16 registers at 128 channels
    constexpr bool TWO = F2 || !(TMODE == 2 && KP == 128 && WAVES == 8);      // (F2: 32 weight registers fewer)
    const float* xg = static_cast<const float*>(p.x);
    TO* yg = static_cast<TO*>(p.y);
    extern __shared__ float4 pw_lds[];
    u32x4* xs = reinterpret_cast<u32x4*>(pw_lds);      // [2][NP][KO][PS_NT]
    const float cx = F2 ? pow2_scale_of(*p.xmax) : 1.f;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph0 = (wave >> 2) * NPH;  // (WAVES = 4: ph0 = 0, both pixel blocks)
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;

    // Walk: a workgroup takes UNITS (one 64-pixel tile of one pixel plane) first, first + stride, ... and within
    // a unit the S samples one after the other -- step v is sample v % S of unit first + (v / S) stride.  The
    // samples of a pixel share their per-pixel context tile (it comes from HBM once) and their mean over the
    // samples (p.ymean) is accumulated by the workgroup itself, in LDS.  S = 1: plain tile order.
    const unsigned g = blockIdx.x, slot = g / NUM_XCD;
    const int rt = (int)(slot % (unsigned)p.nrt);
    const unsigned first = (slot / (unsigned)p.nrt) * NUM_XCD + g % NUM_XCD;
    const unsigned stride = gridDim.x / (unsigned)p.nrt;
    const unsigned S = (unsigned)p.S, nunits = p.ntiles / S;
    // The walk is kept as a cursor that ADVANCES (sample, unit, the unit's pixel tile and image group): decoding
    // step numbers by division cost ~450 scalar instructions and 96 SGPR spill moves per tile -- half of the
    // loop's instructions, in a loop whose time is its instruction count.  The divisions left: four, before the loop.
    struct Cur { unsigned s, unit, pt, bq; };
    const unsigned tpp = p.tiles_per_plane;
    const unsigned st_bq = stride / tpp, st_pt = stride % tpp;
    auto advance = [&](Cur c) -> Cur {
        c.s += 1;
        if (c.s == S) {
            c.s = 0;
            c.unit += stride;
            c.pt += st_pt;
            c.bq += st_bq;
            if (c.pt >= tpp) {
                c.pt -= tpp;
                c.bq += 1;
            }
        }
        return c;
    };
    auto live = [&](const Cur& c) -> bool { return c.unit < nunits; };
    const int r0 = rt * 128 + rb * 32;
    const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;

    auto tile_coords = [&](const Cur& c, unsigned& b, unsigned& bq, unsigned& p0) {
        bq = c.bq;
        b = c.bq * S + c.s;
        p0 = c.pt * PS_NT;
    };
    // staging role: pixel `lane` of the tile, channel octets wave + WAVES i.  One lane offset per tile, the row
    // offsets are scalar (the octet is the wave's); rows beyond K (padding up to KP) are zeros.
    auto issue_loads = [&](const Cur& tile, float (&regs)[NOCT][8]) {
        unsigned b, bq, p0;
        tile_coords(tile, b, bq, p0);
        const rsrc_t rx = make_rsrc_n(xg + (size_t)b * p.K * hw, (unsigned)p.K * hw * 4u);
        const unsigned voff = p0 + lane < hw ? (p0 + lane) * 4u : PW_OOB;
#pragma unroll
        for (int i = 0; i < NOCT; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const unsigned k = 8u * (unsigned)(wave + WAVES * i) + r;      // wave-uniform
                const bool krow = k < (unsigned)p.K;                           // (no branch around the load)
                regs[i][r] = buf_load(rx, krow ? voff : PW_OOB, krow ? k * hw * 4u : 0u);
            }
        }
    };
    auto commit = [&](int buf, const float (&regs)[NOCT][8]) {
#pragma unroll
        for (int i = 0; i < NOCT; ++i) {
            const int o = wave + WAVES * i;
            if (o < KO) {
                u32x4* d = xs + ((buf * NP) * KO + o) * PS_NT + lane;
                if constexpr (F2) {
                    u32x4 h, l;
                    split2(regs[i], cx, h, l);
                    d[0] = h;
                    d[KO * PS_NT] = l;
                } else {
                    u32x4 h, m, l;
                    split3(regs[i], h, m, l);
                    d[0] = h;
                    d[KO * PS_NT] = m;
                    d[2 * KO * PS_NT] = l;
                }
            }
        }
    };

    // weight rows of this wave as A operands, three planes: a*[s] = W[r0 + lane % 32][16 s + 8 (lane / 32) + 0..7]
    // (F2: ah / am = the high / low f16 plane under the scale of the wave's own rows -- each accumulator row is scaled
    // back by its own wave: no layer-wide weight maximum needed)
    u32x4 ah[KS], am[KS], al[F2 ? 1 : KS];
    float osc = 1.f;                                   // F2: 1 / (cx cw), applied to the accumulators
    {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int row = r0 + l31;
        if constexpr (F2) {
            float v[KS][8];
            unsigned wm = 0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * s + 8 * lhi + i;
                    v[s][i] = buf_load(rw, (row < p.Cout && k < p.K) ? (unsigned)(row * p.K + k) * 4u : PW_OOB, 0);
                    const unsigned a = abits(v[s][i]);
                    wm = wm > a ? wm : a;
                }
            }
            const float cw = pow2_scale_of((unsigned)__builtin_amdgcn_readfirstlane((int)wave_umax(wm)));
            osc = (1.f / cx) * (1.f / cw);
#pragma unroll
            for (int s = 0; s < KS; ++s) split2(v[s], cw, ah[s], am[s]);
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * s + 8 * lhi + i;
                    v[i] = buf_load(rw, (row < p.Cout && k < p.K) ? (unsigned)(row * p.K + k) * 4u : PW_OOB, 0);
                }
                split3(v, ah[s], am[s], al[s]);
            }
        }
    }
    // (accumulator row of register j: r0 + (j & 3) + 8 (j >> 2) + 4 (lane / 32))
    const unsigned wpr = (hw + 31) / 32;               // sign words per row

    // Two tiles in flight per workgroup (64 KB per CU): at ~2 us of work per tile one tile's bytes do not cover
    // the memory latency at full bandwidth.  The two register sets swap roles from tile to tile (an in-flight
    // load is never copied: a copy would wait for it), and the barrier orders LDS traffic only.  The split of
    // the next tile (registers -> LDS) is dealt out between the MFMAs of the k-steps.
    // DEEP (the two-plane form: 32 weight registers fewer): THREE tiles in flight -- the two-plane kernels wait (s_waitcnt /
    // barrier: 34-39 % of a wave's cycles, PMC) where the three-plane ones were busy with twice the matrix work.
    // 1.78 -> 1.71 ms (128 -> 128), 1.91 -> 1.75 with the mean, 5.65 -> 5.22 at 441 channels.
    constexpr bool DEEP = F2 && WAVES == 8;
    float preA[NOCT][8], preB[NOCT][8], preC[DEEP ? NOCT : 1][8];
    Cur tile;
    tile.s = 0;
    tile.unit = first;
    tile.pt = first % tpp;
    tile.bq = first / tpp;
    Cur next = advance(tile), next2 = advance(next);
    if (live(tile)) {
        issue_loads(tile, preA);
        commit(0, preA);
        if (live(next)) issue_loads(next, preA);
        if constexpr (DEEP) {
            if (live(next2)) issue_loads(next2, preB);
        }
    }
    Cur next3 = advance(next2);                        // (DEEP: the tile whose loads a step issues)
    __syncthreads();
    // the mean over a pixel's samples: every wave accumulates its own 32 x 32 block of outputs in LDS
    float* macc = reinterpret_cast<float*>(xs + 2 * NP * KO * PS_NT);       // [128][PS_NT]
    const float inv_s = 1.f / (float)S;

    // one pair of values of the next tile -> its three bf16 words; a finished octet goes to LDS
    u32x4 ch[NOCT], cm[NOCT], cl[F2 ? 1 : NOCT];
    auto commit_unit = [&](const float (&src)[NOCT][8], int u, int buf) {     // u = 4 (octet slot) + pair
        const int i = u >> 2, q = u & 3;
        if constexpr (F2) {
            unsigned hp, lp;
            f16_split_pair(src[i][2 * q], src[i][2 * q + 1], cx, hp, lp);
            ch[i][q] = hp;
            cm[i][q] = lp;
        } else {
            const unsigned hp = pack_bf16(src[i][2 * q], src[i][2 * q + 1]);
            const float r0 = src[i][2 * q] - bf16_lo(hp), r1 = src[i][2 * q + 1] - bf16_hi(hp);
            const unsigned mp = pack_bf16(r0, r1);
            ch[i][q] = hp;
            cm[i][q] = mp;
            cl[i][q] = pack_bf16(r0 - bf16_lo(mp), r1 - bf16_hi(mp));
        }
        if (q == 3) {
            const int o = wave + WAVES * i;
            if (o < KO) {
                u32x4* d = xs + ((buf * NP) * KO + o) * PS_NT + lane;
                d[0] = ch[i];
                d[KO * PS_NT] = cm[i];
                if constexpr (!F2) d[2 * KO * PS_NT] = cl[i];
            }
        }
    };
    unsigned amax_run = 0;                              // largest |y| this thread has stored
    const bool want_amax = p.amax != nullptr;

    // the bias of this wave's 16 accumulator rows: without a context term it is the same for every tile -- loaded once
    // (16 of a tile's 48 vector-memory instructions per wave)
    float bias_rows[TMODE == 0 ? 16 : 1];
    if constexpr (TMODE == 0) {
        const rsrc_t rbias = make_rsrc_n(p.bias, (unsigned)p.Cout * 4u);
#pragma unroll
        for (int j = 0; j < 16; ++j) bias_rows[j] = buf_load(rbias, (unsigned)(r0 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * 4u, 0);
    }

    // one tile; `cur` holds the NEXT tile's values (split and written to LDS stage buf ^ 1 during this one),
    // `fill` receives the loads of the tile after that
    auto step = [&](const float (&cur)[NOCT][8], float (&fill)[NOCT][8], const int buf) {
        if constexpr (DEEP) {
            if (live(next3)) issue_loads(next3, fill);
        } else {
            if (live(next2)) issue_loads(next2, fill);
        }

        unsigned b, bq, p0;
        tile_coords(tile, b, bq, p0);
        unsigned o0[NPH];
#pragma unroll
        for (int h = 0; h < NPH; ++h) {
            const unsigned col = p0 + (ph0 + h) * 32 + l31;
            o0[h] = (col < hw && nrows > 0) ? (4u * lhi * hw + col) * 4u : PW_OOB;
        }

        f32x16 acc[NPH], small[NPH], t0[NPH];
        float add[16];                                 // bias (+ the per-image context term) of the 16 accumulator rows
        if constexpr (TMODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) add[j] = bias_rows[j];     // (loaded once: they are live through a tile anyway)
        }
#pragma unroll
        for (int h = 0; h < NPH; ++h) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[h][j] = small[h][j] = 0.f;
        }
        if (TMODE == 2) {
            const rsrc_t rt2 = make_rsrc_n(p.t + ((size_t)bq * p.Cout + r0) * hw, (unsigned)nrows * hw * 4u);
#pragma unroll
            for (int h = 0; h < NPH; ++h) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
                    t0[h][j] = buf_load(rt2, o0[h] != PW_OOB ? o0[h] + ro : PW_OOB, 0);
                }
            }
        }

        const u32x4* xb = xs + ((buf * NP) * KO + lhi) * PS_NT + ph0 * 32 + l31;
        // the B operands of the next k-step are fetched while this one's MFMAs run (where 12 registers are to
        // spare: without a context term) -- else each group of MFMAs starts by waiting for its own LDS reads
        constexpr bool AHEAD = (TMODE == 0 && NPH == 1);
        u32x4 nbh, nbm, nbl = {};
        if constexpr (AHEAD) {
            nbh = xb[0];
            nbm = xb[KO * PS_NT];
            if constexpr (!F2) nbl = xb[2 * KO * PS_NT];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int h = 0; h < NPH; ++h) {
                u32x4 bh, bm, bl;
                if constexpr (AHEAD) {
                    bh = nbh; bm = nbm;
                    if constexpr (!F2) bl = nbl;
                    if (s + 1 < KS) {
                        nbh = xb[(2 * s + 2) * PS_NT];
                        nbm = xb[(KO + 2 * s + 2) * PS_NT];
                        if constexpr (!F2) nbl = xb[(2 * KO + 2 * s + 2) * PS_NT];
                    }
                } else {
                    bh = xb[(2 * s) * PS_NT + 32 * h];
                    bm = xb[(KO + 2 * s) * PS_NT + 32 * h];
                    if constexpr (!F2) bl = xb[(2 * KO + 2 * s) * PS_NT + 32 * h];
                }
                if constexpr (F2) {
                    acc[h] = mfma_f16(ah[s], bh, acc[h]);          // (bm / am: the LOW planes)
                    small[h] = mfma_f16(ah[s], bm, small[h]);
                    small[h] = mfma_f16(am[s], bh, small[h]);
                } else if constexpr (TWO) {
                    acc[h] = mfma_bf16(ah[s], bh, acc[h]);
                    small[h] = mfma_bf16(ah[s], bl, small[h]);
                    acc[h] = mfma_bf16(ah[s], bm, acc[h]);
                    small[h] = mfma_bf16(al[s], bh, small[h]);
                    acc[h] = mfma_bf16(am[s], bh, acc[h]);
                    small[h] = mfma_bf16(am[s], bm, small[h]);
                } else {
                    acc[h] = mfma_bf16(ah[s], bl, acc[h]);
                    acc[h] = mfma_bf16(al[s], bh, acc[h]);
                    acc[h] = mfma_bf16(am[s], bm, acc[h]);
                    acc[h] = mfma_bf16(ah[s], bm, acc[h]);
                    acc[h] = mfma_bf16(am[s], bh, acc[h]);
                    acc[h] = mfma_bf16(ah[s], bh, acc[h]);
                }
            }
            // fillers of this k-step
            if (TMODE != 0 && s == ((TMODE == 2 && WAVES == 8 && KS > 2) ? KS - 2 : 0)) {
                // bias and per-image context term (cache hits), behind the first MFMAs; at 256 registers with a
                // per-pixel term as well (16 more live registers) late in the loop, just in time for the epilogue
                const rsrc_t rbias = make_rsrc_n(p.bias, (unsigned)p.Cout * 4u);
                const rsrc_t rt1 = make_rsrc_n(TMODE == 1 ? p.t + (size_t)bq * p.Cout : p.bias, (unsigned)p.Cout * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned ro = (unsigned)(r0 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * 4u;
                    add[j] = buf_load(rbias, ro, 0);
                    if (TMODE == 1) add[j] += buf_load(rt1, ro, 0);
                }
            }
            // (no branch on `more`: behind the last tile this stages stale registers into the stage nobody reads)
#pragma unroll
            for (int u = 4 * NOCT * s / KS; u < 4 * NOCT * (s + 1) / KS; ++u) commit_unit(cur, u, buf ^ 1);   // in order
            __builtin_amdgcn_sched_barrier(0);
        }

        // bias, context term, sign bits, activation, store.  Branch-free: a store that is not wanted gets an offset
        // beyond its descriptor (dropped by the hardware), the mean's running sum is a select.
        const rsrc_t ry = make_rsrc_n(yg + ((size_t)b * p.Cout + r0) * hw, (unsigned)nrows * hw * SO);
        const rsrc_t rm = make_rsrc_n(MEAN ? (const void*)(p.ymean + ((size_t)bq * p.Cout + r0) * hw) : (const void*)yg,
                                      MEAN ? (unsigned)nrows * hw * 4u : 0u);
        const rsrc_t rsg = make_rsrc_n(p.signs != nullptr ? p.signs + ((size_t)b * p.Cout + r0) * wpr : (const unsigned*)yg,
                                       p.signs != nullptr ? (unsigned)nrows * wpr * 4u : 0u);
        const unsigned s_in = tile.s;                  // which of the pixel's samples this tile is
        const bool first_s = s_in == 0, last_s = s_in + 1 == S;
#pragma unroll
        for (int h = 0; h < NPH; ++h) {
            unsigned myword = 0;
            unrolled<16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                float v;
                if constexpr (F2) v = __builtin_fmaf(acc[h][j] + small[h][j], osc, add[j]);
                else v = TWO ? acc[h][j] + (small[h][j] + add[j]) : acc[h][j] + add[j];
                if (TMODE == 2) v += t0[h][j];
                // lane j (j + 32) keeps the sign word of register j's row (+ 4): the ballot's halves, written
                // straight from the scalar registers into those lanes
                const unsigned long long pos = __ballot(v > 0.f);
                write_lanes<j>(myword, pos);
                v = v > 0.f ? v : v * p.slope;
                if (want_amax) {                       // (columns beyond the plane hold bias only: not stored, not counted)
                    const unsigned a = o0[h] != PW_OOB ? abits(v) : 0u;
                    amax_run = amax_run > a ? amax_run : a;
                }
                const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
                logit_store<TO>(v, ry, o0[h] != PW_OOB ? (o0[h] + ro) / (4u / SO) : PW_OOB, 0);
                if constexpr (MEAN) {
                    float* mp = macc + (rb * 32 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * PS_NT + (ph0 + h) * 32 + l31;
                    const float m = (first_s ? 0.f : *mp) + v;
                    *mp = m;
                    buf_store(m * inv_s, rm, (last_s && o0[h] != PW_OOB) ? o0[h] + ro : PW_OOB, 0);
                }
            });
            {
                // lanes 0-15 / 32-47: row (l31 & 3) + 8 (l31 >> 2) + 4 lhi of the wave's 32, word of this 32-pixel block
                const unsigned row = (unsigned)((l31 & 3) + 8 * (l31 >> 2) + 4 * lhi);
                const unsigned c0 = p0 + (ph0 + h) * 32;
                const bool ok = l31 < 16 && (int)row < nrows && c0 < hw;
                __builtin_amdgcn_raw_buffer_store_b32(myword, rsg, ok ? (row * wpr + c0 / 32) * 4u : PW_OOB, 0, 0);
            }
        }
        lds_barrier();
    };
    auto roll = [&]() {
        tile = next;
        next = next2;
        if constexpr (DEEP) {
            next2 = next3;
            next3 = advance(next3);
        } else {
            next2 = advance(next2);
        }
    };
    if constexpr (DEEP) {
        // step i: the next tile's values sit in set (i + 1) % 3, the loads of tile i + 3 go to set i % 3 (whose tile is in
        // LDS); LDS stage i % 2
        while (live(tile)) {
            step(preA, preC, 0); roll(); if (!live(tile)) break;
            step(preB, preA, 1); roll(); if (!live(tile)) break;
            step(preC, preB, 0); roll(); if (!live(tile)) break;
            step(preA, preC, 1); roll(); if (!live(tile)) break;
            step(preB, preA, 0); roll(); if (!live(tile)) break;
            step(preC, preB, 1); roll();
        }
    } else {
        while (live(tile)) {
            step(preA, preB, 0);
            roll();
            if (!live(tile)) break;
            step(preB, preA, 1);
            roll();
        }
    }
    if (want_amax) amax_publish(amax_run, p.amax);
}

// ---------------------------------------------------------------------------------------------
// The forward for half activations on the f16 matrix pipe ("fp16 activations", BASELINE configs[4]):
// x and y are _Float16 in HBM, the weights are rounded to half once (what torch.autocast does to a
// convolution's weight), products are exact and accumulate in fp32 (v_mfma_f32_32x32x8_f16), bias /
// context term / activation in fp32 on the accumulators.  At 8x the fp32 MFMA rate the layer is purely
// HBM-bound at half the bytes of the fp32 layer.
//
// The B operand of the f16 MFMA wants, per lane, 4 CONSECUTIVE k of one pixel -- the planar activations
// are k-major (a row per channel).  The transpose happens on the way into LDS: a staging thread loads a
// 4 (channels) x 4 (pixels) block as four 8-byte row pieces, transposes it in registers and writes four
// 8-byte {k, k+1, k+2, k+3} entries; the LDS image of a tile is [K/4][NT] such entries, so an operand
// fetch is one conflict-free ds_read_b64 per lane (lanes = consecutive pixels).
// Same persistent tile walk and software pipeline as pw_fwd_kernel.
using h4 = __attribute__((ext_vector_type(4))) _Float16;

// MEAN: also the mean of y over the S samples of a pixel (p.ymean, here _Float16: what feeds the U-net, reference
// sbmc/models.py:179) -- the walk is then unit-major (a workgroup takes the S samples of a pixel tile one after the other)
// and every wave keeps the running sum of its own 32 x 64 block in registers: no pass over y for the mean (three
// reductions over 7.5 GB per frame at 32 spp otherwise).
template <int KP, int TMODE, bool MEAN = false>
__global__ __launch_bounds__(512) void pw_fwd_h_kernel(PwFwdParams p) {
    constexpr int NT = 128;                            // pixels per tile
    constexpr int KQ = KP / 4;                         // channel quads
    constexpr int NPASS = (KQ + 15) / 16;              // staging passes: 16 quads x 32 pixel groups per pass
    const _Float16* xg = static_cast<const _Float16*>(p.x);
    _Float16* yg = static_cast<_Float16*>(p.y);
    extern __shared__ float4 pw_lds[];
    u32x2* xs = reinterpret_cast<u32x2*>(pw_lds);      // [2][KQ][NT] entries of 4 halves
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

    // staging role: pixels 4 pg .. 4 pg + 3 of the channel quads sq + 16 i
    const unsigned pg = threadIdx.x & 31, sq = threadIdx.x >> 5;

    auto tile_coords = [&](unsigned tile, unsigned& b, unsigned& bq, unsigned& p0) {
        const unsigned s = tile % (unsigned)p.S, rest = tile / (unsigned)p.S;
        const unsigned pt = rest % p.tiles_per_plane;
        bq = rest / p.tiles_per_plane;
        b = bq * (unsigned)p.S + s;
        p0 = pt * NT;
    };
    auto issue_loads = [&](unsigned tile, u32x2 (&regs)[NPASS][4]) {
        unsigned b, bq, p0;
        tile_coords(tile, b, bq, p0);
        const rsrc_t rx = make_rsrc_n(xg + (size_t)b * p.K * hw, (unsigned)p.K * hw * 2u);
        const bool colok = p0 + 4 * pg < hw;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned k = 4u * (sq + 16u * i) + r;
                const unsigned off = (colok && k < (unsigned)p.K) ? (k * hw + p0 + 4 * pg) * 2u : PW_OOB;
                regs[i][r] = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
            }
        }
    };
    auto commit = [&](int buf, const u32x2 (&regs)[NPASS][4]) {
        u32x2* dst = xs + buf * (KQ * NT);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned q = sq + 16u * i;
            if (q < (unsigned)KQ) {
                const h4 a0 = __builtin_bit_cast(h4, regs[i][0]), a1 = __builtin_bit_cast(h4, regs[i][1]);
                const h4 a2 = __builtin_bit_cast(h4, regs[i][2]), a3 = __builtin_bit_cast(h4, regs[i][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h4 o;
                    o[0] = a0[j]; o[1] = a1[j]; o[2] = a2[j]; o[3] = a3[j];
                    dst[q * NT + 4 * pg + j] = __builtin_bit_cast(u32x2, o);
                }
            }
        }
    };

    // weight rows of this wave as f16 A-operands of v_mfma_f32_32x32x16_f16 (gfx950: twice the K per issue of the
    // 32x32x8 form these kernels were written on): a[s][i] = W[r0 + lane % 32][16 s + 8 (lane / 32) + i]
    constexpr int KS16 = KP / 16;
    hf8 a[KS16];
    {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int row = r0 + l31;
#pragma unroll
        for (int kk = 0; kk < KS16; ++kk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 16 * kk + 8 * lhi + i;
                const bool ok = row < p.Cout && k < p.K;
                a[kk][i] = (_Float16)buf_load(rw, ok ? (unsigned)(row * p.K + k) * 4u : PW_OOB, 0);
            }
        }
    }
    float bias[16];
    {
        const rsrc_t rbias = make_rsrc_n(p.bias, (unsigned)p.Cout * 4u);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            bias[j] = buf_load(rbias, (unsigned)(r0 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * 4u, 0);
    }

    // step v of this workgroup: sample v % S of unit first + (v / S) stride (S = 1 unless MEAN: plain tile order)
    const unsigned S = MEAN ? (unsigned)p.S : 1u;
    const unsigned nunits = p.ntiles / S;
    auto tile_at = [&](unsigned v) -> unsigned {
        const unsigned unit = first + (v / S) * stride;
        return unit < nunits ? unit * S + v % S : 0xFFFFFFFFu;
    };
    u32x2 pre[NPASS][4];
    unsigned v = 0;
    unsigned tile = tile_at(0);
    if (tile < p.ntiles) {
        issue_loads(tile, pre);
        commit(0, pre);
    }
    __syncthreads();

    f32x16 out0, out1, m0, m1;
#pragma unroll
    for (int j = 0; j < 16; ++j) out0[j] = out1[j] = m0[j] = m1[j] = 0.f;
    const float inv_s = 1.f / (float)S;
    unsigned b_prev = 0;
    unsigned o_prev0 = PW_OOB, o_prev1 = PW_OOB;       // byte offsets for 4-byte elements; halves sit at half of them
    auto store_prev = [&](int j) {
        const rsrc_t ry_prev = make_rsrc_n(yg + ((size_t)b_prev * p.Cout + r0) * hw, (unsigned)nrows * hw * 2u);
        const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
        logit_store<_Float16>(out0[j], ry_prev, o_prev0 != PW_OOB ? (o_prev0 + ro) / 2u : PW_OOB, 0);
        logit_store<_Float16>(out1[j], ry_prev, o_prev1 != PW_OOB ? (o_prev1 + ro) / 2u : PW_OOB, 0);
    };

    int buf = 0;
    for (; tile < p.ntiles; tile = tile_at(++v), buf ^= 1) {
        const unsigned next = tile_at(v + 1);
        if (next < p.ntiles) issue_loads(next, pre);

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

        // B operand of a 16-step: this lane's pixel, channels 16 s + 8 lhi .. + 7 = the quads 4 s + 2 lhi and the next
        const u32x2* xb = xs + buf * (KQ * NT) + 2 * lhi * NT + ph * 64 + l31;
#pragma unroll
        for (int kk = 0; kk < KS16; ++kk) {
            const u32x2 q0 = xb[(4 * kk) * NT], q1 = xb[(4 * kk + 1) * NT];
            const u32x2 r0q = xb[(4 * kk) * NT + 32], r1q = xb[(4 * kk + 1) * NT + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], __builtin_bit_cast(hf8, u32x4{q0[0], q0[1], q1[0], q1[1]}), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], __builtin_bit_cast(hf8, u32x4{r0q[0], r0q[1], r1q[0], r1q[1]}), acc1, 0, 0, 0);
            store_prev(2 * kk);
            store_prev(2 * kk + 1);
        }
#pragma unroll
        for (int j = 2 * KS16; j < 16; ++j) store_prev(j);

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
        if constexpr (MEAN) {
            const unsigned s_in = tile % S;
            // (the mean is taken of the values as stored: rounded to half, like torch's mean over the half tensor)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float h0 = (float)(_Float16)out0[j], h1 = (float)(_Float16)out1[j];
                m0[j] = s_in == 0 ? h0 : m0[j] + h0;
                m1[j] = s_in == 0 ? h1 : m1[j] + h1;
            }
            if (s_in + 1 == S) {
                _Float16* ymh = reinterpret_cast<_Float16*>(p.ymean);
                const rsrc_t rm = make_rsrc_n(ymh + ((size_t)bq * p.Cout + r0) * hw, (unsigned)nrows * hw * 2u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
                    logit_store<_Float16>(m0[j] * inv_s, rm, o0 != PW_OOB ? (o0 + ro) / 2u : PW_OOB, 0);
                    logit_store<_Float16>(m1[j] * inv_s, rm, o1 != PW_OOB ? (o1 + ro) / 2u : PW_OOB, 0);
                }
            }
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

// ---------------------------------------------------------------------------------------------
// The all-half forward for WIDE layers without a context term (128 < Cout <= 512: the 441-channel logits, reference
// sbmc/models.py:98-102): pw_fwd_h_kernel gives every 128-row tile of the output a workgroup of its own, so the input
// tile is fetched (through L2), transposed and written to LDS four times over, and the layer ran at 45 % of the rate
// its bytes allow (3.76 ms at 720p x 8 spp, 15 ms at 32 spp).  Here ONE workgroup stages the tile once and walks the
// row tiles: the weights of all of them stay in registers (half: 32 registers per tile), the B operands are re-read
// from LDS per row tile, the outputs of one (tile, row tile) unit are stored between the MFMAs of the next.
template <int KP>
__global__ __launch_bounds__(512) void pw_fwd_hw_kernel(PwFwdParams p) {
    constexpr int NT = 128;
    constexpr int KQ = KP / 4;
    constexpr int NPASS = (KQ + 15) / 16;
    constexpr int NR = 4;                              // row tiles of 128 output channels
    const _Float16* xg = static_cast<const _Float16*>(p.x);
    _Float16* yg = static_cast<_Float16*>(p.y);
    extern __shared__ float4 pw_lds[];
    u32x2* xs = reinterpret_cast<u32x2*>(pw_lds);      // [2][KQ][NT] entries of 4 halves
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;
    const unsigned first = blockIdx.x, stride = gridDim.x;
    const int nr = (p.Cout + 127) / 128;               // row tiles in use
    const unsigned pg = threadIdx.x & 31, sq = threadIdx.x >> 5;

    auto issue_loads = [&](unsigned tile, u32x2 (&regs)[NPASS][4]) {
        const unsigned b = tile / p.tiles_per_plane, p0 = (tile % p.tiles_per_plane) * NT;
        const rsrc_t rx = make_rsrc_n(xg + (size_t)b * p.K * hw, (unsigned)p.K * hw * 2u);
        const bool colok = p0 + 4 * pg < hw;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned k = 4u * (sq + 16u * i) + r;
                regs[i][r] = __builtin_amdgcn_raw_buffer_load_b64(rx, (colok && k < (unsigned)p.K) ? (k * hw + p0 + 4 * pg) * 2u : PW_OOB, 0, 0);
            }
        }
    };
    auto commit = [&](int buf, const u32x2 (&regs)[NPASS][4]) {
        u32x2* dst = xs + buf * (KQ * NT);
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned q = sq + 16u * i;
            if (q < (unsigned)KQ) {
                const h4 a0 = __builtin_bit_cast(h4, regs[i][0]), a1 = __builtin_bit_cast(h4, regs[i][1]);
                const h4 a2 = __builtin_bit_cast(h4, regs[i][2]), a3 = __builtin_bit_cast(h4, regs[i][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h4 o;
                    o[0] = a0[j]; o[1] = a1[j]; o[2] = a2[j]; o[3] = a3[j];
                    dst[q * NT + 4 * pg + j] = __builtin_bit_cast(u32x2, o);
                }
            }
        }
    };

    // weight rows of this wave for every row tile: a[rt][kk][i] = W[128 rt + 32 rb + lane % 32][8 kk + 4 (lane / 32) + i]
    // (kept as packed words: as _Float16 vectors the compiler holds one half per register)
    // (v_mfma_f32_32x32x16_f16 operands, as in pw_fwd_h_kernel: a[rt][s] = W[..][16 s + 8 (lane / 32) + 0..7])
    constexpr int KS16 = KP / 16;
    u32x4 a[NR][KS16];
    {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) {
            const int row = rt * 128 + rb * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < KS16; ++kk) {
                hf8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = 16 * kk + 8 * lhi + i;
                    v[i] = (_Float16)buf_load(rw, (row < p.Cout && k < p.K) ? (unsigned)(row * p.K + k) * 4u : PW_OOB, 0);
                }
                a[rt][kk] = __builtin_bit_cast(u32x4, v);
            }
        }
    }
    const rsrc_t rbias = make_rsrc_n(p.bias, (unsigned)p.Cout * 4u);

    u32x2 pre[NPASS][4];
    unsigned tile = first;
    if (tile < p.ntiles) {
        issue_loads(tile, pre);
        commit(0, pre);
    }
    __syncthreads();

    int buf = 0;
    for (; tile < p.ntiles; tile += stride, buf ^= 1) {
        const unsigned next = tile + stride;
        if (next < p.ntiles) issue_loads(next, pre);
        const unsigned b = tile / p.tiles_per_plane, p0 = (tile % p.tiles_per_plane) * NT;
        const unsigned col = p0 + ph * 64 + l31;
        const u32x2* xb = xs + buf * (KQ * NT) + 2 * lhi * NT + ph * 64 + l31;
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) {
            if (rt < nr) {
                const int r0 = rt * 128 + rb * 32;
                const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;
                const unsigned o0 = (col < hw && nrows > 0) ? (4u * lhi * hw + col) * 4u : PW_OOB;
                const unsigned o1 = (col + 32 < hw && nrows > 0) ? (4u * lhi * hw + col + 32) * 4u : PW_OOB;
                f32x16 acc0, acc1;
                // (an opaque zero in the scalar offset: the bias values are the same for every pixel tile, and hoisted out of
                // the tile loop they would be 64 live registers -- 39 spilled; reloaded per unit they are cache hits)
                unsigned z;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z));
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float v = buf_load(rbias, (unsigned)(r0 + (j & 3) + 8 * (j >> 2) + 4 * lhi) * 4u, z);   // (rows >= Cout: 0)
                    acc0[j] = v;
                    acc1[j] = v;
                }
#pragma unroll
                for (int kk = 0; kk < KS16; ++kk) {
                    const u32x2 q0 = xb[(4 * kk) * NT], q1 = xb[(4 * kk + 1) * NT];
                    const u32x2 r0q = xb[(4 * kk) * NT + 32], r1q = xb[(4 * kk + 1) * NT + 32];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[rt][kk]),
                                                                  __builtin_bit_cast(hf8, u32x4{q0[0], q0[1], q1[0], q1[1]}), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a[rt][kk]),
                                                                  __builtin_bit_cast(hf8, u32x4{r0q[0], r0q[1], r1q[0], r1q[1]}), acc1, 0, 0, 0);
                }
                // (stored right away: the workgroup's other waves and the CU's second wave per SIMD keep the matrix pipe
                // busy meanwhile; holding a unit's outputs back for the next unit's MFMAs costs 32 registers = spills here)
                const rsrc_t ry = make_rsrc_n(yg + ((size_t)b * p.Cout + r0) * hw, (unsigned)nrows * hw * 2u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
                    const float v0 = acc0[j] > 0.f ? acc0[j] : acc0[j] * p.slope;
                    const float v1 = acc1[j] > 0.f ? acc1[j] : acc1[j] * p.slope;
                    logit_store<_Float16>(v0, ry, o0 != PW_OOB ? (o0 + ro) / 2u : PW_OOB, 0);
                    logit_store<_Float16>(v1, ry, o1 != PW_OOB ? (o1 + ro) / 2u : PW_OOB, 0);
                }
            }
        }
        if (next < p.ntiles) commit(buf ^ 1, pre);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of the layer in ONE pass over gy, y and x (cout <= 128):
//   gz = gy * act'(y)                      (never written to HBM)
//   gx[b]  = w^T @ gz[b]                    (MFMA, reduction over cout)
//   gw    += gz[b] @ x[b]^T                 (MFMA, reduction over pixels; per-workgroup partial sums)
//   gbias += row sums of gz, gt = sum over the samples of a pixel of gz (context term)
// Tiles of 64 pixels; gz and x tiles live in LDS with a row pitch of 66 words: the row-wise operand
// reads (gx) are conflict-free, the column-wise ones (gw: lanes walk rows) 2-way (ds_read_b32 sees 32
// banks); pitch 65 is conflict-free for both but needs dword staging stores -- measured 4 % slower.
// Same software pipeline as the forward: next tile's gy / y / x in flight during the MFMAs, the
// previous tile's gx stored between them.
#ifndef PW_BWD_PIPE
#define PW_BWD_PIPE 0     // 1 = keep a tile's gx in registers and store it between the next tile's MFMAs: 16 more
                          // live registers = 10-12 spilled VGPRs at 2 waves/SIMD; measured 4.98 ms vs 4.66 ms without
#endif
#ifndef PW_BWD_RAW
#define PW_BWD_RAW 1      // two-plane split kernels: the gy and x tiles travel L2/HBM -> LDS by LDS-DMA, a whole iteration ahead
#endif
#ifndef PW_BWD_RAW_ASM
#define PW_BWD_RAW_ASM 1  // the RAW requests as inline assembly with the kernel's own waits (0: the builtin + __syncthreads, for A/B builds)
#endif
#ifndef PW_BWD_GXS
#define PW_BWD_GXS 1      // split-gw kernels with a data gradient: gx on the bf16 pipe too (transposing LDS reads)
#endif
constexpr int PB_NT = 64;
constexpr int PB_PITCH = 66;
constexpr int PB_PITCH_N = 82;      // gz tile of pw_bwd_kernel's NARROW form

// 4 consecutive pixels of one row as they sit in HBM: 16 bytes of float or 8 bytes of _Float16
template <typename T> struct Pack4 { using type = u32x4; };
template <> struct Pack4<_Float16> { using type = u32x2; };
template <typename T>
__device__ __forceinline__ typename Pack4<T>::type load4(rsrc_t r, unsigned off) {
    if constexpr (sizeof(T) == 4) return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    else return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
}
template <typename T>
__device__ __forceinline__ float4 unpack4(typename Pack4<T>::type v) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(float4, v);
    } else {
        using h4 = __attribute__((ext_vector_type(4))) _Float16;
        const h4 h = __builtin_bit_cast(h4, v);
        return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
}

// TA: storage type of gy, y, gm (the layer's output side); TXT: of x and gx (its input side) -- float, or
// _Float16 for training under torch.autocast(float16) ("fp16 activations").  w, the
// partial sums gwp / gbp and the context gradient gt are fp32; so is every product and sum.
struct PwBwdParams {
    const void* gy;      // [B, Cout, hw]
    const void* y;       // [B, Cout, hw] (forward output; unused when linear)
    const void* x;       // [B, K, hw]
    const float* w;      // [Cout, K]
    void* gx;            // [B, K, hw] or nullptr
    float* gwp;          // [G, Cout, K] per-workgroup partial sums of gw
    float* gbp;          // [G, Bq, Cout] per-workgroup partial sums of gbias (zeroed by the host)
    float* gt;           // [B/S, Cout, hw] (t_mode 2) or nullptr
    const void* gm;      // [B/Sm, Cout, hw] or nullptr: gradient of the mean of y over groups of Sm
    int Sm;              //   consecutive batch elements; the kernel sees gy + gm[b / Sm] / Sm
    int B, S, K, Cout, Bq;
    unsigned hw, tiles_per_plane, nunits;
    int t_mode;
    float slope;
    // magnitude words (pw_bwd_kernel<.., F2>, pw_gw_wide_kernel<.., F2>): device words holding the bit pattern of a float
    // >= the largest magnitude of gy / gm / x; gxmax (any form; or nullptr) is raised to the largest |gx| stored
    const unsigned* gmax;
    const unsigned* gmmax;
    const unsigned* xmax;
    unsigned* gxmax;
};

// SG: p.y holds the forward's SIGN BITS ([B, Cout, ceil(hw / 32)] words, written by pw_fwd_s_kernel) instead of
// its output: the activation adjoint then reads 1 bit per element instead of 32 (-12 staging registers too).
// GWS (fp32 tensors only): the weight-gradient product gw += gz x^T runs on the bf16 matrix pipe at fp32 accuracy
// (three-way split of both operands, six products, see pw_fwd_s_kernel) -- 1536 instead of 4096 matrix-pipe
// cycles per wave and tile.  Its reduction runs over PIXELS, so both operands want 8 consecutive pixels of a
// row per lane: the natural planar order (LDS images [plane][row][64 pixels] of bf16, row pitch 144 bytes: a
// 16-lane group of a ds_read_b128 covers the 64 banks exactly once).  gx = w^T gz reduces over OUTPUT CHANNELS
// and stays on the fp32 MFMA with the fp32 gz tile (a channel-contiguous bf16 image of gz would be a second,
// transposed copy).  One LDS stage (fp32 gz 34 KB + two bf16 images 55 KB each), two barriers per tile.
constexpr int PBS_PITCH = 72;       // halves per row of the bf16 images (64 pixels + 8)

// 4 floats under a (wave-uniform) power-of-two scale -> their two f16 planes
__device__ __forceinline__ void split2_4(float4 v, float c, u32x2& h, u32x2& l) {
    unsigned h0, l0, h1, l1;
    f16_split_pair(v.x, v.y, c, h0, l0);
    f16_split_pair(v.z, v.w, c, h1, l1);
    h[0] = h0; h[1] = h1;
    l[0] = l0; l[1] = l1;
}
__device__ __forceinline__ void split3_4(float4 v, u32x2& h, u32x2& m, u32x2& l) {
    const unsigned h0 = pack_bf16(v.x, v.y), h1 = pack_bf16(v.z, v.w);
    const float r0 = v.x - bf16_lo(h0), r1 = v.y - bf16_hi(h0), r2 = v.z - bf16_lo(h1), r3 = v.w - bf16_hi(h1);
    const unsigned m0 = pack_bf16(r0, r1), m1 = pack_bf16(r2, r3);
    h[0] = h0; h[1] = h1;
    m[0] = m0; m[1] = m1;
    l[0] = pack_bf16(r0 - bf16_lo(m0), r1 - bf16_hi(m0));
    l[1] = pack_bf16(r2 - bf16_lo(m1), r3 - bf16_hi(m1));
}

// F2 (round 5, with GWS): both products in the 3 x 3 kernels' number format -- two f16 planes under power-of-two scales
// (gz: from the words of gy and gm; x: from its word; w^T: from the largest of the wave's own rows), three products
// instead of six: half the matrix-pipe cycles, two thirds of the LDS images, a third of the split's instructions.
template <int KP, bool DX, bool TPIX, bool GM, typename TA = float, typename TXT = float, bool SG = false,
          bool GWS = false, bool F2 = false>
__global__ __launch_bounds__(PW_THREADS) void pw_bwd_kernel(PwBwdParams p) {
    static_assert(!GWS || (sizeof(TA) == 4 && sizeof(TXT) == 4), "split gw: fp32 tensors");
    static_assert(!F2 || (GWS && (PW_BWD_GXS || !DX)), "two-plane form: the split kernels");
    constexpr int NP = F2 ? 2 : 3;                      // planes of an operand
    const TA* gy_g = static_cast<const TA*>(p.gy);
    const TA* y_g = static_cast<const TA*>(p.y);
    const unsigned* sg_g = static_cast<const unsigned*>(p.y);
    const unsigned wpr = (p.hw + 31) / 32;
    const TA* gm_g = static_cast<const TA*>(p.gm);
    const TXT* x_g = static_cast<const TXT*>(p.x);
    TXT* gx_g = static_cast<TXT*>(p.gx);
    constexpr unsigned SA = (unsigned)sizeof(TA), SX = (unsigned)sizeof(TXT);
    // gx of a tile is stored right away.  (PW_BWD_PIPE = 1 holds it back and stores it between the MFMAs of
    // the next tile, as the forward does with y: here that costs spills, see the knob.)
    constexpr bool PIPE = PW_BWD_PIPE && !TPIX && !GM;
    // NARROW (split gw with a context or mean gradient, which keep 16 more registers alive): the gx product as
    // 16 x 16 x 4 MFMAs, a wave owning 16 rows of K over all 64 pixels -- w^T is then spread over the 8 waves
    // without a copy (32 instead of 64 registers; same matrix-pipe cycles, twice the 4-byte LDS reads).  Its
    // B operand (4 rows x 16 pixels per read) wants the rows 16 or 18 banks apart; 18 keeps the staging's
    // 8-byte stores conflict-free as well.
    // GXS (split gw, with a data gradient): gx = W^T gz on the bf16 pipe as well -- the six products of the three-way
    // split, 16 x 16 x 32 MFMAs, a wave owning 16 rows of K over all 64 pixels (w^T in three planes: 48 registers).  Its
    // reduction runs over OUTPUT CHANNELS, the planar bf16 image of gz has 8 consecutive PIXELS per 16 bytes: the B
    // operand comes through ds_read_b64_tr_b16 (within a 16-lane group lane i points at row i / 4, columns 4 (i % 4) ..
    // of a 4 x 16 block and receives column i of all four rows: tools/dev/tr16_probe.hip), two reads per plane and
    // k-step.  1536 instead of 4096 matrix-pipe cycles per wave and tile, and no fp32 gz tile in LDS.
    constexpr bool GXS = GWS && DX && PW_BWD_GXS;
    constexpr bool NARROW = GWS && DX && (TPIX || GM) && !GXS;
    constexpr bool GZF = !GWS || (DX && !GXS);          // an fp32 gz tile in LDS: the fp32-MFMA products only
    constexpr int GP = GZF ? (NARROW ? PB_PITCH_N : PB_PITCH) : 0;   // floats per row of the fp32 gz tile
    extern __shared__ float4 pw_lds[];
    float* lds = reinterpret_cast<float*>(pw_lds);
    constexpr int BUF = (128 + KP) * PB_PITCH;          // floats per pipeline stage: gz tile, then x tile
    // GWS: one stage -- fp32 gz tile, then the bf16 images of gz and x ([3][128][PBS_PITCH], [3][KP][PBS_PITCH] halves)
    _Float16* gzn0 = reinterpret_cast<_Float16*>(lds + 128 * GP);
    // DB (the two-plane form: its images are two thirds the size): TWO LDS stages -- a wave commits the next tile as soon
    // as it is through with this one, without waiting for the others (one barrier per tile instead of two: -1 %)
    // RAW (the two-plane form, instead of DB): these kernels WAIT for their loads (39-60 % of a wave's cycles, PMC) with one
    // tile of requests in flight, held in registers from the top of an iteration to its end.  The fp32 gy and x tiles
    // now travel by LDS-DMA into a raw LDS slot each (no register in between: pg / px are gone), requested right after
    // the commit that emptied the slot -- a WHOLE iteration before they are needed instead of one product phase.  A
    // thread reads back exactly the 16-byte pieces its own wave requested (same row / column roles as the register
    // path), so the only ordering needed is the wave's own vmcnt.
    constexpr bool RAW = F2 && PW_BWD_RAW;
    constexpr bool DB = F2 && !RAW;
    constexpr int STG = NP * (128 + KP) * PBS_PITCH;    // halves per stage
    float* rawg = reinterpret_cast<float*>(gzn0 + STG);               // RAW: [128][64] gy, [KP][64] x
    float* rawx = rawg + 128 * PB_NT;
    constexpr int NB = KP / 32;                         // 32-column blocks of gw
    constexpr int NX = KP / 32;                         // staging passes of the x tile
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;
    const unsigned g = blockIdx.x, G = gridDim.x;
    const bool masked = p.slope != 1.f;

    // staging role: float4 column c4 of rows (threadIdx.x >> 4) + 32 i
    const unsigned c4 = (threadIdx.x & 15) * 4, srow = threadIdx.x >> 4;

    // w^T rows of this wave as MFMA A-operands for gx: a2[kk] = w[2 kk + lane / 32][32 rb + lane % 32]
    // (NARROW: a2[kk] = w[4 kk + lane / 16][16 wave + lane % 16])
    float a2[(DX && !GXS) ? (NARROW ? 32 : 64) : 1];
    // GXS: aw*[s] = planes of w[32 s + row(lane / 16, 0..7)][16 wave + lane % 16].  Which output channel of a 32-step sits
    // at which reduction index is free as long as both operands agree: row(g, e) = 16 (g / 2) + 8 (g % 2) + 2 (e % 4) + e / 4
    // makes the 32 lanes of half a wave read EIGHT EVEN (first read) or EIGHT ODD (second) rows of 16 -- with rows 36 banks
    // apart those tile the 64 banks exactly; consecutive rows per group (the obvious map) measured 31 % conflict cycles
    u32x4 awh[GXS ? 4 : 1], awm[GXS ? 4 : 1], awl[(GXS && !F2) ? 4 : 1];      // (F2: awm = the LOW plane)
    // F2 scales: gz by the bound max |gy| + max |gm| / Sm, x by its word; the accumulators are scaled back on the way out
    float cg = 1.f, cx = 1.f, osx = 1.f;               // osx: 1 / (cw cg) of this wave's gx rows
    if constexpr (F2) {
        float gb = __builtin_bit_cast(float, *p.gmax);
        if constexpr (GM) gb += __builtin_bit_cast(float, *p.gmmax) / (float)p.Sm;
        cg = pow2_scale_of(__builtin_bit_cast(unsigned, gb));
        cx = pow2_scale_of(*p.xmax);
    }
    if constexpr (GXS) {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int kr = wave * 16 + (lane & 15);
        auto wrow = [&](int st, float (&v)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g4 = lane >> 4;
                const int co = 32 * st + 16 * (g4 >> 1) + 8 * (g4 & 1) + 2 * (e & 3) + (e >> 2);
                v[e] = buf_load(rw, (co < p.Cout && kr < p.K) ? (unsigned)(co * p.K + kr) * 4u : PW_OOB, 0);
            }
        };
        if constexpr (F2) {
            float v[4][8];
            unsigned wm = 0;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                wrow(st, v[st]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a = abits(v[st][e]);
                    wm = wm > a ? wm : a;
                }
            }
            const float cw = pow2_scale_of((unsigned)__builtin_amdgcn_readfirstlane((int)wave_umax(wm)));
            osx = (1.f / cw) * (1.f / cg);
#pragma unroll
            for (int st = 0; st < 4; ++st) split2(v[st], cw, awh[st], awm[st]);
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                float v[8];
                wrow(st, v);
                split3(v, awh[st], awm[st], awl[st]);
            }
        }
    } else if constexpr (NARROW) {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int k = wave * 16 + (lane & 15);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int co = 4 * kk + (lane >> 4);
            a2[kk] = buf_load(rw, (co < p.Cout && k < p.K) ? (unsigned)(co * p.K + k) * 4u : PW_OOB, 0);
        }
    } else if (DX) {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int k = rb * 32 + l31;
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) {
            const int co = 2 * kk + lhi;
            a2[kk] = buf_load(rw, (co < p.Cout && k < p.K) ? (unsigned)(co * p.K + k) * 4u : PW_OOB, 0);
        }
    }

    struct Cursor { unsigned unit, s; };
    auto coords = [&](Cursor c, unsigned& b, unsigned& bq, unsigned& p0) {
        bq = c.unit / p.tiles_per_plane;
        p0 = (c.unit % p.tiles_per_plane) * PB_NT;
        b = bq * (unsigned)p.S + c.s;
    };
    auto advance = [&](Cursor c) {
        Cursor n;
        n.s = c.s + 1;
        n.unit = c.unit;
        if (n.s == (unsigned)p.S) {
            n.s = 0;
            n.unit = c.unit + G;
        }
        return n;
    };

    typename Pack4<TA>::type pg[4], py[SG ? 1 : 4], pm[GM ? 4 : 1];
    unsigned psg[SG ? 4 : 1];
    typename Pack4<TXT>::type px[NX];
    const float inv_sm = GM ? 1.f / (float)p.Sm : 0.f;
    auto issue = [&](Cursor c) {
        unsigned b, bq, p0;
        coords(c, b, bq, p0);
        const rsrc_t rg = make_rsrc_n(gy_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * SA);
        const rsrc_t rm = make_rsrc_n(GM ? gm_g + (size_t)(b / (unsigned)p.Sm) * p.Cout * hw : gy_g,
                                      (unsigned)p.Cout * hw * SA);
        const rsrc_t ry = make_rsrc_n(y_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * SA);
        const rsrc_t rx = make_rsrc_n(x_g + (size_t)b * p.K * hw, (unsigned)p.K * hw * SX);
        const rsrc_t rs = make_rsrc_n(SG ? sg_g + (size_t)b * p.Cout * wpr : sg_g, (unsigned)p.Cout * wpr * 4u);
        const bool colok = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned r = srow + 32u * i;
            const unsigned off = (colok && r < (unsigned)p.Cout) ? (r * hw + p0 + c4) * SA : PW_OOB;
            if constexpr (!RAW) pg[i] = load4<TA>(rg, off);
            if constexpr (SG) {
                // (rows / columns beyond the tensor read 0: their gradient is 0 as well)
                psg[i] = __builtin_amdgcn_raw_buffer_load_b32(
                    rs, (colok && r < (unsigned)p.Cout) ? (r * wpr + (p0 + c4) / 32) * 4u : PW_OOB, 0, 0);
            } else {
                if (masked) py[i] = load4<TA>(ry, off);
            }
            if (GM) pm[i] = load4<TA>(rm, off);
        }
        if constexpr (!GWS) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const unsigned r = srow + 32u * i;
                const unsigned off = (colok && r < (unsigned)p.K) ? (r * hw + p0 + c4) * SX : PW_OOB;
                px[i] = load4<TXT>(rx, off);
            }
        }
    };
    // RAW: the requests of a tile's gy and x rows, straight into the raw LDS slots (this wave's rows: 4 w + lane / 16 + 32 i)
    // The requests are INLINE ASSEMBLY (round 6): behind the builtin the compiler orders every later LDS access -- and
    // __syncthreads() every barrier -- with s_waitcnt vmcnt(0), i.e. the wave sat out the HBM latency of the rows it had just
    // requested, every iteration, and the "whole iteration ahead" was none (48 % of a wave's cycles waiting, PMC).  The
    // kernel's own waits: `vmcnt(16)` before the commit that reads the slots (the 16 gx stores are the only younger
    // operations: the counter is in order), barriers that wait for LDS only.
    auto issue_raw = [&](Cursor c) {
        unsigned b, bq, p0;
        coords(c, b, bq, p0);
        auto desc = [&](const void* base, unsigned bytes) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(base);
            return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                         (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
        };
        const u32x4 dg = desc(gy_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * SA);
        const u32x4 dx = desc(x_g + (size_t)b * p.K * hw, (unsigned)p.K * hw * SX);
        const bool colok = p0 + c4 < hw;
        const unsigned lg = (unsigned)(uintptr_t)rawg, lx = (unsigned)(uintptr_t)rawx;     // LDS byte addresses
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (this wave's reads of the slots have retired)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned r = srow + 32u * i;
            const unsigned vo = (colok && r < (unsigned)p.Cout) ? (r * hw + p0 + c4) * 4u : PW_OOB;
            const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lg + (unsigned)((32 * i + 4 * wave) * PB_NT * 4)));
#if PW_BWD_RAW_ASM
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(vo), "s"(dg), "s"(m0v) : "memory");
#else
            (void)m0v;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc_n(gy_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * SA),
                                                     (__attribute__((address_space(3))) void*)(rawg + (32 * i + 4 * wave) * PB_NT), 16, vo, 0, 0, 0);
#endif
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const unsigned r = srow + 32u * i;
            const unsigned vo = (colok && r < (unsigned)p.K) ? (r * hw + p0 + c4) * 4u : PW_OOB;
            const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lx + (unsigned)((32 * i + 4 * wave) * PB_NT * 4)));
#if PW_BWD_RAW_ASM
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(vo), "s"(dx), "s"(m0v) : "memory");
#else
            (void)m0v;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc_n(x_g + (size_t)b * p.K * hw, (unsigned)p.K * hw * SX),
                                                     (__attribute__((address_space(3))) void*)(rawx + (32 * i + 4 * wave) * PB_NT), 16, vo, 0, 0, 0);
#endif
        }
    };
    // GWS: the x tile is wanted by the weight-gradient product only -- its loads are issued after the gx
    // product (16 registers less in flight through it), with the gw product's time to land
    auto issue_x = [&](Cursor c) {
        unsigned b, bq, p0;
        coords(c, b, bq, p0);
        const rsrc_t rx = make_rsrc_n(x_g + (size_t)b * p.K * hw, (unsigned)p.K * hw * SX);
        const bool colok = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const unsigned r = srow + 32u * i;
            const unsigned off = (colok && r < (unsigned)p.K) ? (r * hw + p0 + c4) * SX : PW_OOB;
            px[i] = load4<TXT>(rx, off);
        }
    };

    float bsum[4] = {0.f, 0.f, 0.f, 0.f};               // row sums of gz (rows srow + 32 i)
    float4 gts[TPIX ? 4 : 1];                           // sum over the samples of a pixel of gz
    if (TPIX) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gts[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto flush_bias = [&](unsigned bq) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 8, 16);
            v += __shfl_xor(v, 4, 16);
            v += __shfl_xor(v, 2, 16);
            v += __shfl_xor(v, 1, 16);
            const unsigned r = srow + 32u * i;
            if ((threadIdx.x & 15) == 0 && r < (unsigned)p.Cout)
                p.gbp[((size_t)g * p.Bq + bq) * p.Cout + r] += v;     // this workgroup's own slice
            bsum[i] = 0.f;
        }
    };
    // registers -> LDS stage `buf`: gz = gy * act'(y) and x; side sums
    auto commit = [&](Cursor c, int buf) {
        unsigned b, bq, p0;
        coords(c, b, bq, p0);
        float* gzs = lds + (GWS ? 0 : buf) * BUF;
        float* xst = gzs + 128 * PB_PITCH;
        _Float16* gzn = gzn0 + (DB ? buf : 0) * STG;
        _Float16* xn = gzn + NP * 128 * PBS_PITCH;
        const bool colok_c = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 gv;
            if constexpr (RAW) {                       // (rows / columns beyond the tensor: zeros, whatever the DMA left there)
                gv = *reinterpret_cast<const float4*>(rawg + (srow + 32 * i) * PB_NT + c4);
                if (!(colok_c && srow + 32u * i < (unsigned)p.Cout)) gv = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                gv = unpack4<TA>(pg[i]);
            }
            if (GM) {
                const float4 m = unpack4<TA>(pm[i]);
                gv.x += m.x * inv_sm; gv.y += m.y * inv_sm; gv.z += m.z * inv_sm; gv.w += m.w * inv_sm;
            }
            if constexpr (SG) {
                if (masked) {
                    const unsigned bits = psg[i] >> ((p0 + c4) & 31u);
                    gv.x = (bits & 1u) ? gv.x : gv.x * p.slope;
                    gv.y = (bits & 2u) ? gv.y : gv.y * p.slope;
                    gv.z = (bits & 4u) ? gv.z : gv.z * p.slope;
                    gv.w = (bits & 8u) ? gv.w : gv.w * p.slope;
                }
            } else if (masked) {
                const float4 v = unpack4<TA>(py[i]);
                gv.x = v.x > 0.f ? gv.x : gv.x * p.slope;
                gv.y = v.y > 0.f ? gv.y : gv.y * p.slope;
                gv.z = v.z > 0.f ? gv.z : gv.z * p.slope;
                gv.w = v.w > 0.f ? gv.w : gv.w * p.slope;
            }
            bsum[i] += (gv.x + gv.y) + (gv.z + gv.w);
            if (TPIX) {
                gts[i].x += gv.x; gts[i].y += gv.y; gts[i].z += gv.z; gts[i].w += gv.w;
            }
            if constexpr (GZF) {
                float2* d = reinterpret_cast<float2*>(gzs + (srow + 32 * i) * GP + c4);
                d[0] = make_float2(gv.x, gv.y);
                d[1] = make_float2(gv.z, gv.w);
            }
            if constexpr (F2) {
                u32x2 h, l;
                split2_4(gv, cg, h, l);
                _Float16* e = gzn + (srow + 32 * i) * PBS_PITCH + c4;
                *reinterpret_cast<u32x2*>(e) = h;
                *reinterpret_cast<u32x2*>(e + 128 * PBS_PITCH) = l;
            } else if constexpr (GWS) {
                u32x2 h, m, l;
                split3_4(gv, h, m, l);
                _Float16* e = gzn + (srow + 32 * i) * PBS_PITCH + c4;
                *reinterpret_cast<u32x2*>(e) = h;
                *reinterpret_cast<u32x2*>(e + 128 * PBS_PITCH) = m;
                *reinterpret_cast<u32x2*>(e + 2 * 128 * PBS_PITCH) = l;
            }
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            float4 xv;
            if constexpr (RAW) {
                xv = *reinterpret_cast<const float4*>(rawx + (srow + 32 * i) * PB_NT + c4);
                if (!(colok_c && srow + 32u * i < (unsigned)p.K)) xv = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                xv = unpack4<TXT>(px[i]);
            }
            if constexpr (F2) {
                u32x2 h, l;
                split2_4(xv, cx, h, l);
                _Float16* e = xn + (srow + 32 * i) * PBS_PITCH + c4;
                *reinterpret_cast<u32x2*>(e) = h;
                *reinterpret_cast<u32x2*>(e + KP * PBS_PITCH) = l;
            } else if constexpr (GWS) {
                u32x2 h, m, l;
                split3_4(xv, h, m, l);
                _Float16* e = xn + (srow + 32 * i) * PBS_PITCH + c4;
                *reinterpret_cast<u32x2*>(e) = h;
                *reinterpret_cast<u32x2*>(e + KP * PBS_PITCH) = m;
                *reinterpret_cast<u32x2*>(e + 2 * KP * PBS_PITCH) = l;
            } else {
                float2* d = reinterpret_cast<float2*>(xst + (srow + 32 * i) * PB_PITCH + c4);
                d[0] = make_float2(xv.x, xv.y);
                d[1] = make_float2(xv.z, xv.w);
            }
        }
        if (c.s + 1 == (unsigned)p.S) {                 // last sample of this pixel tile
            if (TPIX) {
                const rsrc_t rt = make_rsrc_n(p.gt + (size_t)bq * p.Cout * hw, (unsigned)p.Cout * hw * 4u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned r = srow + 32u * i;
                    const unsigned off = (p0 + c4 < hw && r < (unsigned)p.Cout) ? (r * hw + p0 + c4) * 4u : PW_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gts[i]), rt, off, 0, 0);
                    gts[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (p.t_mode == 1) flush_bias(bq);
        }
    };

    Cursor cur;
    cur.unit = g;
    cur.s = 0;
    bool valid = cur.unit < p.nunits;
    if (valid) {
        issue(cur);
        if constexpr (RAW) {
            issue_raw(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (GWS) {
            issue_x(cur);
        }
        commit(cur, 0);
        if constexpr (RAW) {
            const Cursor n1 = advance(cur);
            if (n1.unit < p.nunits) issue_raw(n1);
        }
    }
    if constexpr (RAW && PW_BWD_RAW_ASM) lds_barrier();     // (LDS only: the second tile's rows stay in flight)
    else __syncthreads();

    f32x16 acc_w[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc_w[0][j] = acc_w[1][j] = 0.f;
    unsigned gxmax_run = 0;                             // largest |gx| this thread has stored (split kernels)
    // (the three-plane forms with a context or mean gradient have no register to spare for it: 27-30 spilled registers
    // and 3.3 -> 5.0 ms per launch when they kept the running maximum)
    constexpr bool GXMAX = (NARROW || GXS) && (F2 || !(TPIX || GM));
    const bool want_gxmax = GXMAX && p.gxmax != nullptr;

    // gx of the previous step, waiting to be stored
    f32x16 out;
#pragma unroll
    for (int j = 0; j < 16; ++j) out[j] = 0.f;
    unsigned b_prev = 0, o_prev = PW_OOB;
    const int kr0 = rb * 32;
    const int nkrows = p.K - kr0 < 32 ? (p.K - kr0 > 0 ? p.K - kr0 : 0) : 32;
    // (o_prev: byte offset for 4-byte elements; half outputs sit at half of it)
    auto store_prev = [&](int j) {
        const rsrc_t rgx = make_rsrc_n(gx_g + ((size_t)b_prev * p.K + kr0) * hw, (unsigned)nkrows * hw * SX);
        const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 4u;
        logit_store<TXT>(out[j], rgx, o_prev != PW_OOB ? (o_prev + ro) / (4u / SX) : PW_OOB, 0);
    };

    int buf = 0;
    while (valid) {
        const Cursor nxt = advance(cur);
        const bool nvalid = nxt.unit < p.nunits;
        if (nvalid) issue(nxt);
        // XE (the two-plane form without a context / mean gradient: 21 registers to spare): the x rows requested up front
        // too -- the two-plane kernels WAIT for their loads (60 % of a wave's cycles, PMC) where the three-plane ones
        // were busy
        constexpr bool XE = F2 && !TPIX && !GM && !RAW;
        if constexpr (XE) {
            if (nvalid) issue_x(nxt);
        }

        unsigned b, bq, p0;
        coords(cur, b, bq, p0);
        const float* gzs = lds + (GWS ? 0 : buf) * BUF;
        const float* xst = gzs + 128 * PB_PITCH;
        const _Float16* gzn = gzn0 + (DB ? buf : 0) * STG;
        const _Float16* xn = gzn + NP * 128 * PBS_PITCH;

        // ---- gx tile: rows 32 rb .. of K, pixels 32 ph ..; reduction over cout
        f32x16 acc_x;
        f32x4 acc_n[(NARROW || GXS) ? 4 : 1];           // NARROW, GXS: rows 16 wave .. of K, pixel blocks of 16
        if constexpr (GXS) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc_n[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
            // this lane's address for the first read: row 16 (g / 2) + 8 (g % 2) + 2 j of the 32-step, g = lane / 16,
            // j = (lane % 16) / 4; columns 4 (lane % 4) .. of the pixel block.  The second read: the odd row below.
            const _Float16* tb = gzn + (16 * (lane >> 5) + 8 * ((lane >> 4) & 1) + 2 * ((lane & 15) >> 2)) * PBS_PITCH + 4 * (lane & 3);
            using v4s = short __attribute__((ext_vector_type(4)));
            using v4sp = __attribute__((address_space(3))) v4s*;
            auto tr8 = [&](const _Float16* q) -> u32x4 {          // this lane's 8 rows of its column
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4sp)q);
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4sp)(q + PBS_PITCH));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                return u32x4{l2[0], l2[1], h2[0], h2[1]};
            };
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const _Float16* q = tb + (32 * st) * PBS_PITCH + 16 * pb;
                    if constexpr (F2) {
                        const u32x4 bh = tr8(q), bl = tr8(q + 128 * PBS_PITCH);
                        acc_n[pb] = mfma16_f16(awh[st], bl, acc_n[pb]);
                        acc_n[pb] = mfma16_f16(awm[st], bh, acc_n[pb]);
                        acc_n[pb] = mfma16_f16(awh[st], bh, acc_n[pb]);
                    } else {
                        const u32x4 bh = tr8(q), bm = tr8(q + 128 * PBS_PITCH), bl = tr8(q + 2 * 128 * PBS_PITCH);
                        acc_n[pb] = mfma16_bf16(awh[st], bl, acc_n[pb]);
                        acc_n[pb] = mfma16_bf16(awl[st], bh, acc_n[pb]);
                        acc_n[pb] = mfma16_bf16(awm[st], bm, acc_n[pb]);
                        acc_n[pb] = mfma16_bf16(awh[st], bm, acc_n[pb]);
                        acc_n[pb] = mfma16_bf16(awm[st], bh, acc_n[pb]);
                        acc_n[pb] = mfma16_bf16(awh[st], bh, acc_n[pb]);
                    }
                }
                if (!XE && !RAW && st == 1 && nvalid) issue_x(nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (NARROW) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc_n[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* gb = gzs + (lane >> 4) * GP + (lane & 15);
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
                for (int kk = grp * 4; kk < grp * 4 + 4; ++kk) {
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb)
                        acc_n[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[kk], gb[(4 * kk) * GP + 16 * pb],
                                                                         acc_n[pb], 0, 0, 0);
                }
                if (grp == 3 && nvalid) issue_x(nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (DX) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc_x[j] = 0.f;
            const float* gb = gzs + lhi * PB_PITCH + ph * 32 + l31;
#pragma unroll
            for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
                for (int kk = grp * 8; kk < grp * 8 + 8; ++kk)
                    acc_x = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[kk], gb[(2 * kk) * PB_PITCH], acc_x, 0, 0, 0);
                if constexpr (GWS) {
                    if (grp == 3 && nvalid) issue_x(nxt);     // half of the gx product + the gw product to land
                }
                if (PIPE) {
                    store_prev(2 * grp);
                    store_prev(2 * grp + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        auto store_gx = [&]() {
            if constexpr (NARROW || GXS) {
                const int r0 = wave * 16;
                const int nr = p.K - r0 < 16 ? (p.K - r0 > 0 ? p.K - r0 : 0) : 16;
                const rsrc_t rgx = make_rsrc_n(gx_g + ((size_t)b * p.K + r0) * hw, (unsigned)nr * hw * SX);
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const unsigned col = p0 + 16 * pb + (lane & 15);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned row = 4 * (lane >> 4) + j;
                        const float v = F2 ? acc_n[pb][j] * osx : acc_n[pb][j];
                        if (want_gxmax) {
                            const unsigned a = (col < hw && nr > 0) ? abits(v) : 0u;
                            gxmax_run = gxmax_run > a ? gxmax_run : a;
                        }
                        logit_store<TXT>(v, rgx, (col < hw && nr > 0) ? (row * hw + col) * SX : PW_OOB, 0);
                    }
                }
            } else if (DX) {
                const unsigned col = p0 + ph * 32 + l31;
#pragma unroll
                for (int j = 0; j < 16; ++j) out[j] = acc_x[j];
                b_prev = __builtin_amdgcn_readfirstlane(b);
                o_prev = (col < hw && nkrows > 0) ? (4u * lhi * hw + col) * 4u : PW_OOB;
                if (!PIPE) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) store_prev(j);
                }
            }
        };
        // ---- gw: rows 32 rb .. of cout, column blocks 2 ph, 2 ph + 1 of K; reduction over the 64 pixels
        if constexpr (GWS) {
            if (!XE && !RAW && !DX && nvalid) issue_x(nxt);
            store_gx();                                // (its 16 accumulator registers are free for the gw product)
            const bool two = 2 * ph + 1 < NB;
            if (2 * ph < NB) {
                const _Float16* ga = gzn + (rb * 32 + l31) * PBS_PITCH + 8 * lhi;
                const _Float16* xb0 = xn + ((2 * ph) * 32 + l31) * PBS_PITCH + 8 * lhi;
                const _Float16* xb1 = xn + ((2 * ph + 1) * 32 + l31) * PBS_PITCH + 8 * lhi;
#pragma unroll
                for (int s = 0; s < 4; ++s) {           // 16 pixels per step
                    const u32x4 ah = *reinterpret_cast<const u32x4*>(ga + 16 * s);
                    const u32x4 am = *reinterpret_cast<const u32x4*>(ga + 16 * s + 128 * PBS_PITCH);
                    if constexpr (F2) {                 // (am / bm: the LOW planes)
                        {
                            const u32x4 bh = *reinterpret_cast<const u32x4*>(xb0 + 16 * s);
                            const u32x4 bm = *reinterpret_cast<const u32x4*>(xb0 + 16 * s + KP * PBS_PITCH);
                            acc_w[0] = mfma_f16(ah, bm, acc_w[0]);
                            acc_w[0] = mfma_f16(am, bh, acc_w[0]);
                            acc_w[0] = mfma_f16(ah, bh, acc_w[0]);
                        }
                        if (two) {
                            const u32x4 bh = *reinterpret_cast<const u32x4*>(xb1 + 16 * s);
                            const u32x4 bm = *reinterpret_cast<const u32x4*>(xb1 + 16 * s + KP * PBS_PITCH);
                            acc_w[1] = mfma_f16(ah, bm, acc_w[1]);
                            acc_w[1] = mfma_f16(am, bh, acc_w[1]);
                            acc_w[1] = mfma_f16(ah, bh, acc_w[1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        continue;
                    }
                    const u32x4 al = *reinterpret_cast<const u32x4*>(ga + 16 * s + 2 * 128 * PBS_PITCH);
                    {
                        const u32x4 bh = *reinterpret_cast<const u32x4*>(xb0 + 16 * s);
                        const u32x4 bm = *reinterpret_cast<const u32x4*>(xb0 + 16 * s + KP * PBS_PITCH);
                        const u32x4 bl = *reinterpret_cast<const u32x4*>(xb0 + 16 * s + 2 * KP * PBS_PITCH);
                        acc_w[0] = mfma_bf16(ah, bl, acc_w[0]);
                        acc_w[0] = mfma_bf16(al, bh, acc_w[0]);
                        acc_w[0] = mfma_bf16(am, bm, acc_w[0]);
                        acc_w[0] = mfma_bf16(ah, bm, acc_w[0]);
                        acc_w[0] = mfma_bf16(am, bh, acc_w[0]);
                        acc_w[0] = mfma_bf16(ah, bh, acc_w[0]);
                    }
                    if (two) {
                        const u32x4 bh = *reinterpret_cast<const u32x4*>(xb1 + 16 * s);
                        const u32x4 bm = *reinterpret_cast<const u32x4*>(xb1 + 16 * s + KP * PBS_PITCH);
                        const u32x4 bl = *reinterpret_cast<const u32x4*>(xb1 + 16 * s + 2 * KP * PBS_PITCH);
                        acc_w[1] = mfma_bf16(ah, bl, acc_w[1]);
                        acc_w[1] = mfma_bf16(al, bh, acc_w[1]);
                        acc_w[1] = mfma_bf16(am, bm, acc_w[1]);
                        acc_w[1] = mfma_bf16(ah, bm, acc_w[1]);
                        acc_w[1] = mfma_bf16(am, bh, acc_w[1]);
                        acc_w[1] = mfma_bf16(ah, bh, acc_w[1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            const float* ga = gzs + (rb * 32 + l31) * PB_PITCH + lhi;
            const float* xb0 = xst + ((2 * ph) * 32 + l31) * PB_PITCH + lhi;
            const float* xb1 = xst + ((2 * ph + 1) * 32 + l31) * PB_PITCH + lhi;
            const bool two = 2 * ph + 1 < NB;
            if (2 * ph < NB) {
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
                    for (int kk = grp * 8; kk < grp * 8 + 8; ++kk) {
                        const float av = ga[2 * kk];
                        acc_w[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xb0[2 * kk], acc_w[0], 0, 0, 0);
                        if (two) acc_w[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xb1[2 * kk], acc_w[1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (!GWS) store_gx();

        if constexpr (GWS && !DB) lds_barrier();     // one stage: every wave is through with it before it is refilled
                                                     // (LDS ordering only: the x loads stay in flight)
        if constexpr (RAW) {
            if (nvalid) {
                // the next tile's rows have landed (requested a whole iteration ago): everything but this iteration's 16 gx
                // stores, which are the youngest operations of the in-order counter, is through
                if constexpr (DX && PW_BWD_RAW_ASM) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                commit(nxt, buf ^ 1);
                const Cursor n2 = advance(nxt);
                if (n2.unit < p.nunits) issue_raw(n2);
            }
            if constexpr (PW_BWD_RAW_ASM) lds_barrier();      // (the planes are visible; the rows of the tile after next stay in flight)
            else __syncthreads();
        } else {
            if (nvalid) commit(nxt, buf ^ 1);
            __syncthreads();
        }
        cur = nxt;
        valid = nvalid;
        buf ^= 1;
    }
    if (DX && PIPE) {
#pragma unroll
        for (int j = 0; j < 16; ++j) store_prev(j);
    }
    if (p.t_mode != 1) flush_bias(0);

    // ---- this workgroup's partial gw
    {
        const int r0 = rb * 32;
        const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;
        const rsrc_t rgw = make_rsrc_n(p.gwp + ((size_t)g * p.Cout + r0) * p.K, (unsigned)(nrows * p.K) * 4u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (2 * ph + n) * 32 + l31;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rl = (j & 3) + 8 * (j >> 2) + 4 * lhi;
                const float v = F2 ? acc_w[n][j] * ((1.f / cg) * (1.f / cx)) : acc_w[n][j];
                buf_store(v, rgw, (col < p.K && nrows > 0) ? (unsigned)(rl * p.K + col) * 4u : PW_OOB, 0);
            }
        }
    }
    if constexpr (GXMAX) {
        if (p.gxmax != nullptr) amax_publish(gxmax_run, p.gxmax);
    }
}

// ---------------------------------------------------------------------------------------------
// Weight and bias gradient of a WIDE linear 1x1 layer (128 < Cout <= 512: the 441-channel kernel regressor output,
// reference sbmc/models.py:98-102): gw[co][k] = sum over images and pixels of gz[co][px] x[k][px], gbias[co] = sum gz.
// Until round 4 a library GEMM (8.4 ms per step on the fp32 matrix pipe) behind a read-only pass over the 13 GB logit
// gradient for the bias sums (2.2 ms).  Here: pw_bwd_kernel's split-precision weight-gradient product (three bf16 planes
// of both operands, six products, fp32 accumulation; both operands want 8 consecutive PIXELS per lane -- the natural
// planar order) with the x tile staged ONCE per pixel tile and the gz tile four times, 128 output channels at a time,
// into four sets of accumulators (128 registers); the row sums for the bias are taken while staging.  One pass over
// gz and x: 16.8 GB at 720p x 8 spp.
// T: storage type of gz and x (float, or _Float16 for training under torch.autocast(float16): the same products, the
// values widened on the way into the split -- their low planes are zero).
template <int KP, typename T = float>
__global__ __launch_bounds__(PW_THREADS) void pw_gw_wide_kernel(PwBwdParams p) {
    constexpr unsigned ST = (unsigned)sizeof(T);
    const T* gz_g = static_cast<const T*>(p.gy);
    const T* x_g = static_cast<const T*>(p.x);
    extern __shared__ float4 pw_lds[];
    _Float16* gzn = reinterpret_cast<_Float16*>(pw_lds);              // [3][128][PBS_PITCH]
    _Float16* xn = gzn + 3 * 128 * PBS_PITCH;                         // [3][KP][PBS_PITCH]
    constexpr int NB = KP / 32, NX = KP / 32;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;
    const unsigned g = blockIdx.x, G = gridDim.x;
    const unsigned c4 = (threadIdx.x & 15) * 4, srow = threadIdx.x >> 4;
    const int nct = (p.Cout + 127) / 128;                             // output-channel tiles in use (<= 4)

    using P4 = typename Pack4<T>::type;
    P4 pgA[4], pgB[4], px[NX];        // two sets of gz rows: a step's rows are requested TWO steps ahead
    auto coords = [&](unsigned unit, unsigned& b, unsigned& p0) {
        b = unit / p.tiles_per_plane;
        p0 = (unit % p.tiles_per_plane) * PB_NT;
    };
    auto issue_g = [&](P4 (&pg)[4], unsigned unit, int ct) {
        unsigned b, p0;
        coords(unit, b, p0);
        const rsrc_t rg = make_rsrc_n(gz_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * ST);
        const bool colok = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned r = 128u * ct + srow + 32u * i;
            pg[i] = load4<T>(rg, (colok && r < (unsigned)p.Cout) ? (r * hw + p0 + c4) * ST : PW_OOB);
        }
    };
    auto issue_x = [&](unsigned unit) {
        unsigned b, p0;
        coords(unit, b, p0);
        const rsrc_t rx = make_rsrc_n(x_g + (size_t)b * p.K * hw, (unsigned)p.K * hw * ST);
        const bool colok = p0 + c4 < hw;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const unsigned r = srow + 32u * i;
            px[i] = load4<T>(rx, (colok && r < (unsigned)p.K) ? (r * hw + p0 + c4) * ST : PW_OOB);
        }
    };
    float bsum[4][4];                                                 // row sums of gz: [cout tile][rows srow + 32 i]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) bsum[a][i] = 0.f;
    auto commit_g = [&](const P4 (&pg)[4], auto ctc) __attribute__((always_inline)) {
        constexpr int CT = decltype(ctc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 gv = unpack4<T>(pg[i]);
            bsum[CT][i] += (gv.x + gv.y) + (gv.z + gv.w);
            u32x2 h, m, l;
            split3_4(gv, h, m, l);
            _Float16* e = gzn + (srow + 32 * i) * PBS_PITCH + c4;
            *reinterpret_cast<u32x2*>(e) = h;
            *reinterpret_cast<u32x2*>(e + 128 * PBS_PITCH) = m;
            *reinterpret_cast<u32x2*>(e + 2 * 128 * PBS_PITCH) = l;
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const float4 xv = unpack4<T>(px[i]);
            u32x2 h, m, l;
            split3_4(xv, h, m, l);
            _Float16* e = xn + (srow + 32 * i) * PBS_PITCH + c4;
            *reinterpret_cast<u32x2*>(e) = h;
            *reinterpret_cast<u32x2*>(e + KP * PBS_PITCH) = m;
            *reinterpret_cast<u32x2*>(e + 2 * KP * PBS_PITCH) = l;
        }
    };

    f32x16 acc_w[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc_w[a][n][j] = 0.f;

    // Steps: (pixel tile, output-channel tile ct = 0 .. nct_e - 1), nct_e = nct rounded up to even (an odd last tile is
    // followed by one of zeros: the register set a step's rows arrive in is then a compile-time function of ct).  The gz
    // rows of step k + 2 are requested at the start of step k (one request ahead left the loads 1.5 us to land: the
    // kernel ran at 2.2 TB/s on latency), the next pixel tile's x rows two steps before its first step; one LDS stage, two
    // barriers per step.
    const int nct_e = (nct + 1) & ~1;
    unsigned unit = g;
    bool valid = unit < p.nunits;
    if (valid) {
        issue_g(pgA, unit, 0);
        issue_x(unit);
        commit_g(pgA, std::integral_constant<int, 0>{});
        commit_x();
        issue_g(pgB, unit, 1);
    }
    __syncthreads();
    const bool two = 2 * ph + 1 < NB;
    auto step = [&](auto ctc) __attribute__((always_inline)) {
        constexpr int CT = decltype(ctc)::value;
        const bool last_ct = CT + 1 >= nct_e;
        {
            const bool wrap = CT + 2 >= nct_e;
            const unsigned u2 = wrap ? unit + G : unit;
            const int c2 = wrap ? CT + 2 - nct_e : CT + 2;
            if (u2 < p.nunits) {
                if constexpr (CT % 2 == 0) issue_g(pgA, u2, c2); else issue_g(pgB, u2, c2);
            }
            if (CT + 2 == nct_e && unit + G < p.nunits) issue_x(unit + G);
        }
        if (2 * ph < NB) {
            const _Float16* ga = gzn + (rb * 32 + l31) * PBS_PITCH + 8 * lhi;
            const _Float16* xb0 = xn + ((2 * ph) * 32 + l31) * PBS_PITCH + 8 * lhi;
            const _Float16* xb1 = xn + ((2 * ph + 1) * 32 + l31) * PBS_PITCH + 8 * lhi;
#pragma unroll
            for (int s = 0; s < 4; ++s) {               // 16 pixels per step
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ga + 16 * s);
                const u32x4 am = *reinterpret_cast<const u32x4*>(ga + 16 * s + 128 * PBS_PITCH);
                const u32x4 al = *reinterpret_cast<const u32x4*>(ga + 16 * s + 2 * 128 * PBS_PITCH);
                {
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(xb0 + 16 * s);
                    const u32x4 bm = *reinterpret_cast<const u32x4*>(xb0 + 16 * s + KP * PBS_PITCH);
                    const u32x4 bl = *reinterpret_cast<const u32x4*>(xb0 + 16 * s + 2 * KP * PBS_PITCH);
                    acc_w[CT][0] = mfma_bf16(ah, bl, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_bf16(al, bh, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_bf16(am, bm, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_bf16(ah, bm, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_bf16(am, bh, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_bf16(ah, bh, acc_w[CT][0]);
                }
                if (two) {
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(xb1 + 16 * s);
                    const u32x4 bm = *reinterpret_cast<const u32x4*>(xb1 + 16 * s + KP * PBS_PITCH);
                    const u32x4 bl = *reinterpret_cast<const u32x4*>(xb1 + 16 * s + 2 * KP * PBS_PITCH);
                    acc_w[CT][1] = mfma_bf16(ah, bl, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_bf16(al, bh, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_bf16(am, bm, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_bf16(ah, bm, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_bf16(am, bh, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_bf16(ah, bh, acc_w[CT][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();                                  // every wave is through with the stage before it is refilled
        {
            // the next step's rows: requested during the step before this one, in the OTHER set
            const unsigned u1 = last_ct ? unit + G : unit;
            if (u1 < p.nunits) {
                if (last_ct) {
                    if constexpr (CT % 2 == 0) commit_g(pgB, std::integral_constant<int, 0>{});
                    else commit_g(pgA, std::integral_constant<int, 0>{});
                    commit_x();
                } else {
                    if constexpr (CT % 2 == 0) commit_g(pgB, std::integral_constant<int, (CT + 1) % 4>{});
                    else commit_g(pgA, std::integral_constant<int, (CT + 1) % 4>{});
                }
            }
        }
        __syncthreads();
        if (last_ct) {
            unit += G;
            valid = unit < p.nunits;
        }
    };
    while (valid) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        if (nct_e > 2) {
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        }
    }

    // ---- this workgroup's partial sums: gbias (rows reduced over the 16 threads of a staging row), gw
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = bsum[a][i];
            v += __shfl_xor(v, 8, 16);
            v += __shfl_xor(v, 4, 16);
            v += __shfl_xor(v, 2, 16);
            v += __shfl_xor(v, 1, 16);
            const unsigned r = 128u * a + srow + 32u * i;
            if ((threadIdx.x & 15) == 0 && r < (unsigned)p.Cout) p.gbp[(size_t)g * p.Cout + r] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r0 = a * 128 + rb * 32;
        const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;
        const rsrc_t rgw = make_rsrc_n(p.gwp + ((size_t)g * p.Cout + r0) * p.K, (unsigned)(nrows * p.K) * 4u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (2 * ph + n) * 32 + l31;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rl = (j & 3) + 8 * (j >> 2) + 4 * lhi;
                buf_store(acc_w[a][n][j], rgw, (col < p.K && nrows > 0 && rl < nrows) ? (unsigned)(rl * p.K + col) * 4u : PW_OOB, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The WHOLE backward of the wide linear layer (the 441-channel logits, reference sbmc/models.py:98-102) in ONE pass over the
// logit gradient (round 5): gw, gbias AND gx = w^T gz, everything in the two-f16-plane format.  Until here the data gradient
// was a library GEMM on the fp32 matrix pipe behind a second pass over the 13 GB gradient (7.6 of the layer's 14.3 ms).
//
// pw_gw_wide_kernel's structure -- the x tile staged once per 64-pixel tile, the gz tile 128 output channels at a time
// (a "step") -- plus, per step, the gx product of pw_bwd_kernel's GXS form on the staged gz chunk: a wave owns 16 rows of
// K over all 64 pixels and adds up its four 16 x 16 blocks over the steps of a tile.  Its A operand, this wave's 16 rows of
// w^T for the step's 128 channels in two planes, does not fit the registers next to 128 of gw accumulators for all four
// steps (128 more), nor the LDS for the whole layer (229 KB): the PREPARED planes (pw_wide_prep_kernel: split once per
// launch, laid out per (step, wave, plane, 32-channel sub-step, lane) so that a wave's slice is eight 1 KB runs) stream
// from L2 -- 8 KB per wave and step, requested at the step's start BEFORE the step's gz request (a wait for them must not
// be a wait for the far slower HBM rows behind them: the vector-memory counter is in order), parked in 32 registers
// through the step and written to the wave's PRIVATE LDS slice at the step's commit.  gz rows one step ahead (a step is
// twice the work of pw_gw_wide_kernel's).
struct PwWideParams {
    PwBwdParams b;
    const u32x4* wprep;      // [4 steps][8 waves][2 planes][4 sub-steps][64 lanes] entries of 8 halves
    const float* wscale;     // the weights' power-of-two scale (written by pw_wide_prep_kernel)
};

// w [Cout, K] -> the planes above + the scale.  One workgroup.
__global__ __launch_bounds__(512) void pw_wide_prep_kernel(const float* __restrict__ w, u32x4* __restrict__ wprep,
                                                           float* __restrict__ wscale, int cout, int k) {
    unsigned m = 0;
    for (int i = threadIdx.x; i < cout * k; i += 512) {
        const unsigned a = abits(w[i]);
        m = m > a ? m : a;
    }
    __shared__ unsigned wmax[8];
    m = wave_umax(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    for (int i = 0; i < 8; ++i) m = m > wmax[i] ? m : wmax[i];
    const float cw = pow2_scale_of((unsigned)__builtin_amdgcn_readfirstlane((int)m));
    if (threadIdx.x == 0) *wscale = cw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g4 = lane >> 4, kr = wave * 16 + (lane & 15);
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {              // (the row -> reduction index map of pw_bwd_kernel's GXS form)
                const int co = 128 * ct + 32 * st + 16 * (g4 >> 1) + 8 * (g4 & 1) + 2 * (e & 3) + (e >> 2);
                v[e] = (co < cout && kr < k) ? w[co * k + kr] : 0.f;
            }
            u32x4 h, l;
            split2(v, cw, h, l);
            wprep[(((ct * 8 + wave) * 2 + 0) * 4 + st) * 64 + lane] = h;
            wprep[(((ct * 8 + wave) * 2 + 1) * 4 + st) * 64 + lane] = l;
        }
    }
}

// G2 (round 6; an even number of steps per tile: the 441-channel layer's four): the gz rows are requested TWO steps ahead into
// one of two register sets (the branch-free requests above freed 19 registers), BEHIND the weight slice's request, so that
// "the slice has landed" is `s_waitcnt vmcnt(rows + stores issued since)` and a step's closing barrier leaves the next-but-one
// step's HBM rows in flight -- with one step of lead (1600 cycles) every step sat out the rest of the HBM latency.  The
// slice's requests are inline assembly: the compiler orders LDS reads behind a builtin LDS-DMA with vmcnt(0).
// MEASURED (same box, 720p x 8 spp): correct (tests, fuzz) and SLOWER, 8.63-8.66 ms against 8.26-8.37 -- the step's exposed HBM
// latency is not what bounds this kernel.  Off by default (SBMC_PW_WIDE_G2=1 selects it), kept for the next experiment.
template <int KP, bool G2 = false>
__global__ __launch_bounds__(PW_THREADS) void pw_wide_bwd2_kernel(PwWideParams pp) {
    const PwBwdParams& p = pp.b;
    const float* gz_g = static_cast<const float*>(p.gy);
    const float* x_g = static_cast<const float*>(p.x);
    float* gx_g = static_cast<float*>(p.gx);
    extern __shared__ float4 pw_lds[];
    _Float16* gzn = reinterpret_cast<_Float16*>(pw_lds);              // [2][128][PBS_PITCH]
    _Float16* xn = gzn + 2 * 128 * PBS_PITCH;                         // [2][KP][PBS_PITCH]
    u32x4* wsl = reinterpret_cast<u32x4*>(xn + 2 * KP * PBS_PITCH);   // [8 waves][2][4][64]: a wave's own slice
    float* bacc = reinterpret_cast<float*>(wsl + 8 * 8 * 64);         // [512]: row sums of gz (the bias gradient)
    constexpr int NB = KP / 32, NX = KP / 32;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;
    const unsigned g = blockIdx.x, G = gridDim.x;
    const unsigned c4 = (threadIdx.x & 15) * 4, srow = threadIdx.x >> 4;
    // steps per pixel tile (<= 4).  G2: FOUR, a constant -- a branch around a step is a join at which the compiler waits for
    // every load in flight, the rows of the step after next among them
    const int nct = G2 ? 4 : (p.Cout + 127) / 128;
    const float cg = pow2_scale_of(*p.gmax), cx = pow2_scale_of(*p.xmax);
    const float osx = (1.f / *pp.wscale) * (1.f / cg);

    u32x4 pg[G2 ? 2 : 1][4], px[NX];
    auto coords = [&](unsigned unit, unsigned& b, unsigned& p0) {
        b = unit / p.tiles_per_plane;
        p0 = (unit % p.tiles_per_plane) * PB_NT;
    };
    // this wave's slice of step ct: eight 1 KB runs, L2 -> LDS without touching a register (LDS-DMA: this kernel has none
    // to spare).  The destination is the wave's PRIVATE region, which only its own gx product reads: the request needs
    // this wave's LDS reads retired (lgkmcnt) and nothing else; its completion is the s_waitcnt vmcnt(0) before the
    // step's closing barrier.  (hipcc does not count an asm load: its own waits for the ordinary loads only get more
    // conservative by it, never less -- the counter is in order and these are the youngest requests.)
    const rsrc_t rwp = make_rsrc_n(pp.wprep, 4u * 8u * 8u * 1024u);
    auto issue_w = [&](int ct, bool valid = true) {
        using lptr = __attribute__((address_space(3))) void*;
        char* dst = reinterpret_cast<char*>(wsl + (size_t)wave * 8 * 64);
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((ct * 8 + wave) * 8192);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (G2) {
            const uintptr_t wa = reinterpret_cast<uintptr_t>(pp.wprep);
            const u32x4 d{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wa),
                          (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(wa >> 32) & 0xffffu)),
                          valid ? 4u * 8u * 8u * 1024u : 0u, 0x00020000u};
            const unsigned l0 = (unsigned)(uintptr_t)dst;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(l0 + (unsigned)i * 1024u));
                const unsigned sv = so + (unsigned)i * 1024u;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                             :: "v"((unsigned)lane * 16u), "s"(d), "s"(m0v), "s"(sv) : "memory");
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rwp, (lptr)(dst + i * 1024), 16, (unsigned)lane * 16u, so + (unsigned)i * 1024u, 0, 0);
        }
    };
    // The rows' requests: ONE lane offset (row srow of a block of 32 rows, this thread's four pixels; beyond the plane: out
    // of range) and a descriptor PER BLOCK OF ROWS that begins at the block and ends with the tensor -- rows beyond Cout / K
    // fall outside it.  With a select per row in the lane offset the compiler built branches around the loads, and at their
    // joins waited for every load in flight (`s_waitcnt vmcnt(0)`): two of the four row blocks' HBM latencies were sat
    // out one after the other at the top of every step, before the first MFMA (round 6, found in the listing).
    auto issue_g = [&](u32x4 (&dst)[4], unsigned unit, int ct, bool valid = true) {
        unsigned b, p0;
        coords(valid ? unit : 0u, b, p0);
        const unsigned lo = (p0 + c4 < hw) ? (srow * hw + p0 + c4) * 4u : PW_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = 128 * ct + 32 * i, left = valid ? p.Cout - row0 : 0;
            const rsrc_t rg = make_rsrc_n(gz_g + ((size_t)b * p.Cout + (left > 0 ? row0 : 0)) * hw,
                                          left > 0 ? (unsigned)left * hw * 4u : 0u);
            dst[i] = load4<float>(rg, lo);
        }
    };
    auto issue_x = [&](unsigned unit, bool valid = true) {
        unsigned b, p0;
        coords(valid ? unit : 0u, b, p0);
        const unsigned lo = (p0 + c4 < hw) ? (srow * hw + p0 + c4) * 4u : PW_OOB;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int row0 = 32 * i, left = valid ? p.K - row0 : 0;
            const rsrc_t rx = make_rsrc_n(x_g + ((size_t)b * p.K + (left > 0 ? row0 : 0)) * hw,
                                          left > 0 ? (unsigned)left * hw * 4u : 0u);
            px[i] = load4<float>(rx, lo);
        }
    };
    // (the row sums live in LDS, not in 16 registers of every thread: a row's 16 staging threads add up their 4 pixels
    // each and one of them adds the sum to the row's word -- this kernel has no register to spare)
    bacc[threadIdx.x] = 0.f;
    auto commit_g = [&](const u32x4 (&src)[4], int ct) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 gv = unpack4<float>(src[i]);
            float rs = (gv.x + gv.y) + (gv.z + gv.w);
            rs += __shfl_xor(rs, 8, 16);
            rs += __shfl_xor(rs, 4, 16);
            rs += __shfl_xor(rs, 2, 16);
            rs += __shfl_xor(rs, 1, 16);
            if ((threadIdx.x & 15) == 0) bacc[128 * ct + srow + 32 * i] += rs;      // (the one thread that owns this word)
            u32x2 h, l;
            split2_4(gv, cg, h, l);
            _Float16* e = gzn + (srow + 32 * i) * PBS_PITCH + c4;
            *reinterpret_cast<u32x2*>(e) = h;
            *reinterpret_cast<u32x2*>(e + 128 * PBS_PITCH) = l;
        }
    };
    auto commit_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const float4 xv = unpack4<float>(px[i]);
            u32x2 h, l;
            split2_4(xv, cx, h, l);
            _Float16* e = xn + (srow + 32 * i) * PBS_PITCH + c4;
            *reinterpret_cast<u32x2*>(e) = h;
            *reinterpret_cast<u32x2*>(e + KP * PBS_PITCH) = l;
        }
    };
    f32x16 acc_w[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc_w[a][n][j] = 0.f;
    f32x4 acc_n[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) acc_n[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned gxmax_run = 0;
    const bool want_gxmax = p.gxmax != nullptr;

    unsigned unit = g;
    bool valid = unit < p.nunits;
    if (valid) {
        issue_g(pg[0], unit, 0);
        issue_x(unit);
        issue_w(0);
        commit_g(pg[0], 0);
        commit_x();
    }
    if constexpr (G2) {
        // the second step's rows (the first step commits them at its end)
        const bool s1 = nct > 1;
        const unsigned u1 = s1 ? unit : unit + G;
        issue_g(pg[1], u1, s1 ? 1 : 0, valid && u1 < p.nunits);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // (everything but those four: the slice has landed)
        lds_barrier();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const bool two = 2 * ph + 1 < NB;
    // this lane's B-operand address of the gx product (pw_bwd_kernel, GXS)
    const _Float16* tb = gzn + (16 * (lane >> 5) + 8 * ((lane >> 4) & 1) + 2 * ((lane & 15) >> 2)) * PBS_PITCH + 4 * (lane & 3);
    using v4s = short __attribute__((ext_vector_type(4)));
    using v4sp = __attribute__((address_space(3))) v4s*;
    auto tr8 = [&](const _Float16* q) -> u32x4 {
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4sp)q);
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4sp)(q + PBS_PITCH));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
    };
    auto step = [&](auto ctc) __attribute__((always_inline)) {
        constexpr int CT = decltype(ctc)::value;
        const bool last_ct = CT + 1 >= nct;
        const unsigned un = last_ct ? unit + G : unit;              // the next step's pixel tile and chunk
        const int cn = last_ct ? 0 : CT + 1;
        const bool more = un < p.nunits;
        // requests of the next step: the HBM rows now, the weight slice (L2) behind the gx product -- everything is consumed
        // at the step's commit, so the in-order vector-memory counter costs nothing, and the slice's 32 registers are
        // live through the gw product only
        if constexpr (!G2) {
            if (more) {
                issue_g(pg[0], un, cn);
                if (last_ct) issue_x(un);
            }
        }
        // ---- gx += w^T[:, chunk] gz[chunk]: rows 16 wave .. of K, four blocks of 16 pixels
        {
            const u32x4* wa = wsl + (size_t)wave * 8 * 64 + lane;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 awh = wa[st * 64], awl = wa[(4 + st) * 64];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const _Float16* q = tb + (32 * st) * PBS_PITCH + 16 * pb;
                    const u32x4 bh = tr8(q), bl = tr8(q + 128 * PBS_PITCH);
                    acc_n[pb] = mfma16_f16(awh, bl, acc_n[pb]);
                    acc_n[pb] = mfma16_f16(awl, bh, acc_n[pb]);
                    acc_n[pb] = mfma16_f16(awh, bh, acc_n[pb]);
                    __builtin_amdgcn_sched_barrier(0);              // (one block of operands in flight, not all four: registers)
                }
            }
        }
        if constexpr (G2) {
            // behind the wave's own reads of its slice: the next step's slice (L2), then the HBM rows -- the next tile's x rows
            // (committed at this step's end) and the gz rows of the step after next.  Always the same NUMBER of requests (an
            // empty descriptor where there is nothing to fetch): the waits below count them.
            issue_w(more ? cn : 0, more);
            if (last_ct) issue_x(un, more);
            const bool wrap2 = CT + 2 >= nct;
            const unsigned un2 = wrap2 ? unit + G : unit;
            issue_g(pg[CT & 1], un2, wrap2 ? CT + 2 - nct : CT + 2, un2 < p.nunits);
        } else {
            if (more) issue_w(cn);                      // (behind the wave's own reads of the region)
        }
        if (last_ct) {                                               // the tile's data gradient is complete
            unsigned b, p0;
            coords(unit, b, p0);
            const int r0 = wave * 16;
            const int nr = p.K - r0 < 16 ? (p.K - r0 > 0 ? p.K - r0 : 0) : 16;
            const rsrc_t rgx = make_rsrc_n(gx_g + ((size_t)b * p.K + r0) * hw, (unsigned)nr * hw * 4u);
            // one lane offset per accumulator register (its row: rows beyond K fall outside the descriptor), the pixel
            // block as a scalar offset; columns beyond the plane through the offset's select
            const unsigned c0 = p0 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned vo = ((4u * (lane >> 4) + j) * hw + c0) * 4u;
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const bool in = c0 + 16u * pb < hw && nr > 0;
                    const float v = acc_n[pb][j] * osx;
                    if (want_gxmax) {
                        const unsigned a = (in && 4 * (lane >> 4) + j < nr) ? abits(v) : 0u;
                        gxmax_run = gxmax_run > a ? gxmax_run : a;
                    }
                    buf_store(v, rgx, in ? vo : PW_OOB, 64u * pb);
                }
            }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc_n[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- gw[chunk] += gz[chunk] x^T: rows 32 rb .. of the chunk, column blocks 2 ph, 2 ph + 1 of K
        if (2 * ph < NB) {
            const _Float16* ga = gzn + (rb * 32 + l31) * PBS_PITCH + 8 * lhi;
            const _Float16* xb0 = xn + ((2 * ph) * 32 + l31) * PBS_PITCH + 8 * lhi;
            const _Float16* xb1 = xn + ((2 * ph + 1) * 32 + l31) * PBS_PITCH + 8 * lhi;
#pragma unroll
            for (int s = 0; s < 4; ++s) {               // 16 pixels per step
                const u32x4 ah = *reinterpret_cast<const u32x4*>(ga + 16 * s);
                const u32x4 al = *reinterpret_cast<const u32x4*>(ga + 16 * s + 128 * PBS_PITCH);
                {
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(xb0 + 16 * s);
                    const u32x4 bl = *reinterpret_cast<const u32x4*>(xb0 + 16 * s + KP * PBS_PITCH);
                    acc_w[CT][0] = mfma_f16(ah, bl, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_f16(al, bh, acc_w[CT][0]);
                    acc_w[CT][0] = mfma_f16(ah, bh, acc_w[CT][0]);
                }
                if (two) {
                    const u32x4 bh = *reinterpret_cast<const u32x4*>(xb1 + 16 * s);
                    const u32x4 bl = *reinterpret_cast<const u32x4*>(xb1 + 16 * s + KP * PBS_PITCH);
                    acc_w[CT][1] = mfma_f16(ah, bl, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_f16(al, bh, acc_w[CT][1]);
                    acc_w[CT][1] = mfma_f16(ah, bh, acc_w[CT][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();                                  // every wave is through with the stage before it is refilled
        if (more) {
            commit_g(pg[G2 ? (CT + 1) & 1 : 0], cn);
            if (last_ct) commit_x();
        }
        if constexpr (G2) {
            // the slice has landed: all but the requests issued behind it (x rows, gz rows, the tile's 16 gx stores)
            if (last_ct) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + 4 + 16) : "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the slice has landed)
            __syncthreads();
        }
        if (last_ct) {
            unit += G;
            valid = unit < p.nunits;
        }
    };
    while (valid) {
        step(std::integral_constant<int, 0>{});
        if (nct > 1) step(std::integral_constant<int, 1>{});
        if (nct > 2) step(std::integral_constant<int, 2>{});
        if (nct > 3) step(std::integral_constant<int, 3>{});
    }

    // ---- this workgroup's partial sums: gbias, gw (scaled back)
    if (threadIdx.x < (unsigned)p.Cout) p.gbp[(size_t)g * p.Cout + threadIdx.x] = bacc[threadIdx.x];     // (behind the loop's barrier)
    const float osw = (1.f / cg) * (1.f / cx);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r0 = a * 128 + rb * 32;
        const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;
        const rsrc_t rgw = make_rsrc_n(p.gwp + ((size_t)g * p.Cout + r0) * p.K, (unsigned)(nrows * p.K) * 4u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (2 * ph + n) * 32 + l31;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rl = (j & 3) + 8 * (j >> 2) + 4 * lhi;
                buf_store(acc_w[a][n][j] * osw, rgw, (col < p.K && nrows > 0 && rl < nrows) ? (unsigned)(rl * p.K + col) * 4u : PW_OOB, 0);
            }
        }
    }
    if (want_gxmax) amax_publish(gxmax_run, p.gxmax);
}

// ---------------------------------------------------------------------------------------------
// The backward for half activations on the f16 matrix pipe (training under torch.autocast(float16)): gy, y, x
// and gx are _Float16 in HBM, gz = gy * act'(y) is formed in fp32 and rounded to half once (what autocast's
// activation backward hands to its convolution backward), the weights are rounded to half, every product is
// exact and accumulates in fp32 (v_mfma_f32_32x32x16_f16); gw / gbias partial sums and gt stay fp32.  The fp32-MFMA
// kernel above is matrix-pipe bound on half tensors (4.2 ms per 128 -> 128 layer at 720p x 8 spp, for 7.5 GB of
// traffic); at 8x the MFMA rate this one is HBM-bound.
//
// Operand layouts.  gw[co][k] = sum over pixels of gz[co][px] x[k][px] wants, per lane, 8 consecutive PIXELS of a
// row of gz / x: the natural planar order, read as two 8-byte words (row pitch 68 halves = 34 words, chosen for the
// 32x32x8 form's 8-byte reads; the 16-step form's pairs of them conflict two ways -- nothing these HBM-bound kernels
// notice).  gx[k][px] = sum over co of w[co][k] gz[co][px] wants 8 consecutive ROWS of gz per pixel: two adjacent entries
// of a second, quad-transposed image of the gz tile ([Cout/4][64] entries of 4 halves, as in pw_fwd_h_kernel), written by
// the same staging thread that holds a 4 x 4 block of gz in registers.
constexpr int PBH_PITCH = 68;

template <int KP, bool DX, bool TPIX, bool GM>
__global__ __launch_bounds__(PW_THREADS) void pw_bwd_h_kernel(PwBwdParams p) {
    using H = _Float16;
    const H* gy_g = static_cast<const H*>(p.gy);
    const H* y_g = static_cast<const H*>(p.y);
    const H* gm_g = static_cast<const H*>(p.gm);
    const H* x_g = static_cast<const H*>(p.x);
    H* gx_g = static_cast<H*>(p.gx);
    extern __shared__ float4 pw_lds[];
    constexpr int GZN = 128 * PBH_PITCH;                // halves: gz tile, natural order
    constexpr int XN = KP * PBH_PITCH;                  // halves: x tile, natural order
    constexpr int GZT = 32 * PB_NT * 4;                 // halves: gz tile, quad-transposed
    constexpr int STAGE = GZN + XN + GZT;               // halves per pipeline stage (a multiple of 4)
    H* lds = reinterpret_cast<H*>(pw_lds);
    constexpr int NB = KP / 32;
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const unsigned hw = p.hw;
    const unsigned g = blockIdx.x, G = gridDim.x;
    const bool masked = p.slope != 1.f;

    // staging role: pixels 4 pgp .. 4 pgp + 3 of the rows 4 q .. 4 q + 3 (of gy / y / gm, and of x)
    const unsigned pgp = threadIdx.x & 15, q = threadIdx.x >> 4;

    // w^T rows of this wave as f16 A-operands for gx (v_mfma_f32_32x32x16_f16: 8 values of the reduction index per lane):
    // a2[kk][i] = w[16 kk + 8 (lane / 32) + i][32 rb + lane % 32]
    hf8 a2[DX ? 8 : 1];
    if (DX) {
        const rsrc_t rw = make_rsrc_n(p.w, (unsigned)(p.Cout * p.K) * 4u);
        const int k = rb * 32 + l31;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int co = 16 * kk + 8 * lhi + i;
                a2[kk][i] = (H)buf_load(rw, (co < p.Cout && k < p.K) ? (unsigned)(co * p.K + k) * 4u : PW_OOB, 0);
            }
        }
    }

    struct Cursor { unsigned unit, s; };
    auto coords = [&](Cursor c, unsigned& b, unsigned& bq, unsigned& p0) {
        bq = c.unit / p.tiles_per_plane;
        p0 = (c.unit % p.tiles_per_plane) * PB_NT;
        b = bq * (unsigned)p.S + c.s;
    };
    auto advance = [&](Cursor c) {
        Cursor n;
        n.s = c.s + 1;
        n.unit = c.unit;
        if (n.s == (unsigned)p.S) {
            n.s = 0;
            n.unit = c.unit + G;
        }
        return n;
    };

    u32x2 pg[4], py[4], pm[GM ? 4 : 1], px[4];
    const float inv_sm = GM ? 1.f / (float)p.Sm : 0.f;
    auto issue = [&](Cursor c) {
        unsigned b, bq, p0;
        coords(c, b, bq, p0);
        const rsrc_t rg = make_rsrc_n(gy_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * 2u);
        const rsrc_t rm = make_rsrc_n(GM ? gm_g + (size_t)(b / (unsigned)p.Sm) * p.Cout * hw : gy_g,
                                      (unsigned)p.Cout * hw * 2u);
        const rsrc_t ry = make_rsrc_n(y_g + (size_t)b * p.Cout * hw, (unsigned)p.Cout * hw * 2u);
        const rsrc_t rx = make_rsrc_n(x_g + (size_t)b * p.K * hw, (unsigned)p.K * hw * 2u);
        const bool colok = p0 + 4 * pgp < hw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned row = 4u * q + r;
            const unsigned off = (colok && row < (unsigned)p.Cout) ? (row * hw + p0 + 4 * pgp) * 2u : PW_OOB;
            pg[r] = __builtin_amdgcn_raw_buffer_load_b64(rg, off, 0, 0);
            if (masked) py[r] = __builtin_amdgcn_raw_buffer_load_b64(ry, off, 0, 0);
            if (GM) pm[r] = __builtin_amdgcn_raw_buffer_load_b64(rm, off, 0, 0);
        }
        if (q < (unsigned)(KP / 4)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned row = 4u * q + r;
                const unsigned off = (colok && row < (unsigned)p.K) ? (row * hw + p0 + 4 * pgp) * 2u : PW_OOB;
                px[r] = __builtin_amdgcn_raw_buffer_load_b64(rx, off, 0, 0);
            }
        }
    };

    float bsum[4] = {0.f, 0.f, 0.f, 0.f};               // row sums of gz (rows 4 q + r)
    float4 gts[TPIX ? 4 : 1];                           // sum over the samples of a pixel of gz
    if (TPIX) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gts[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto flush_bias = [&](unsigned bq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = bsum[r];
            v += __shfl_xor(v, 8, 16);
            v += __shfl_xor(v, 4, 16);
            v += __shfl_xor(v, 2, 16);
            v += __shfl_xor(v, 1, 16);
            const unsigned row = 4u * q + r;
            if (pgp == 0 && row < (unsigned)p.Cout)
                p.gbp[((size_t)g * p.Bq + bq) * p.Cout + row] += v;     // this workgroup's own slice
            bsum[r] = 0.f;
        }
    };
    auto commit = [&](Cursor c, int buf) {
        unsigned b, bq, p0;
        coords(c, b, bq, p0);
        H* gzn = lds + buf * STAGE;
        H* xn = gzn + GZN;
        u32x2* gzt = reinterpret_cast<u32x2*>(xn + XN);
        h4 gh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float4 gv = unpack4<H>(pg[r]);
            if (GM) {
                const float4 m = unpack4<H>(pm[r]);
                gv.x += m.x * inv_sm; gv.y += m.y * inv_sm; gv.z += m.z * inv_sm; gv.w += m.w * inv_sm;
            }
            if (masked) {
                const float4 v = unpack4<H>(py[r]);
                gv.x = v.x > 0.f ? gv.x : gv.x * p.slope;
                gv.y = v.y > 0.f ? gv.y : gv.y * p.slope;
                gv.z = v.z > 0.f ? gv.z : gv.z * p.slope;
                gv.w = v.w > 0.f ? gv.w : gv.w * p.slope;
            }
            bsum[r] += (gv.x + gv.y) + (gv.z + gv.w);
            if (TPIX) {
                gts[r].x += gv.x; gts[r].y += gv.y; gts[r].z += gv.z; gts[r].w += gv.w;
            }
            gh[r][0] = (H)gv.x; gh[r][1] = (H)gv.y; gh[r][2] = (H)gv.z; gh[r][3] = (H)gv.w;
            *reinterpret_cast<u32x2*>(gzn + (4 * q + r) * PBH_PITCH + 4 * pgp) = __builtin_bit_cast(u32x2, gh[r]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h4 o;
            o[0] = gh[0][j]; o[1] = gh[1][j]; o[2] = gh[2][j]; o[3] = gh[3][j];
            gzt[q * PB_NT + 4 * pgp + j] = __builtin_bit_cast(u32x2, o);
        }
        if (q < (unsigned)(KP / 4)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<u32x2*>(xn + (4 * q + r) * PBH_PITCH + 4 * pgp) = px[r];
        }
        if (c.s + 1 == (unsigned)p.S) {                 // last sample of this pixel tile
            if (TPIX) {
                const rsrc_t rt = make_rsrc_n(p.gt + (size_t)bq * p.Cout * hw, (unsigned)p.Cout * hw * 4u);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned row = 4u * q + r;
                    const unsigned off = (p0 + 4 * pgp < hw && row < (unsigned)p.Cout) ? (row * hw + p0 + 4 * pgp) * 4u : PW_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gts[r]), rt, off, 0, 0);
                    gts[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (p.t_mode == 1) flush_bias(bq);
        }
    };

    Cursor cur;
    cur.unit = g;
    cur.s = 0;
    bool valid = cur.unit < p.nunits;
    if (valid) {
        issue(cur);
        commit(cur, 0);
    }
    __syncthreads();

    f32x16 acc_w[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc_w[0][j] = acc_w[1][j] = 0.f;

    const int kr0 = rb * 32;
    const int nkrows = p.K - kr0 < 32 ? (p.K - kr0 > 0 ? p.K - kr0 : 0) : 32;

    int buf = 0;
    while (valid) {
        const Cursor nxt = advance(cur);
        const bool nvalid = nxt.unit < p.nunits;
        if (nvalid) issue(nxt);

        unsigned b, bq, p0;
        coords(cur, b, bq, p0);
        const H* gzn = lds + buf * STAGE;
        const H* xn = gzn + GZN;
        const u32x2* gzt = reinterpret_cast<const u32x2*>(xn + XN);

        // ---- gx tile: rows 32 rb .. of K, pixels 32 ph ..; reduction over cout
        if (DX) {
            f32x16 acc_x;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc_x[j] = 0.f;
            // (a lane's 8 output channels 16 kk + 8 lhi ..: two adjacent quad entries of its pixel)
            const u32x2* gb = gzt + 2 * lhi * PB_NT + ph * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const u32x2 q0 = gb[(4 * kk) * PB_NT], q1 = gb[(4 * kk + 1) * PB_NT];
                acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[kk], __builtin_bit_cast(hf8, u32x4{q0[0], q0[1], q1[0], q1[1]}),
                                                               acc_x, 0, 0, 0);
            }
            const unsigned col = p0 + ph * 32 + l31;
            const unsigned o = (col < hw && nkrows > 0) ? (4u * lhi * hw + col) * 2u : PW_OOB;
            const rsrc_t rgx = make_rsrc_n(gx_g + ((size_t)b * p.K + kr0) * hw, (unsigned)nkrows * hw * 2u);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned ro = (unsigned)((j & 3) + 8 * (j >> 2)) * hw * 2u;
                logit_store<H>(acc_x[j], rgx, o != PW_OOB ? o + ro : PW_OOB, 0);
            }
        }
        // ---- gw: rows 32 rb .. of cout, column blocks 2 ph, 2 ph + 1 of K; reduction over the 64 pixels
        if (2 * ph < NB) {
            // (v_mfma_f32_32x32x16_f16: a lane's 8 consecutive pixels 16 kk + 8 lhi .. of its row -- two 8-byte words of
            // the natural image, whose 136-byte row pitch keeps them 8-byte aligned)
            const H* ga = gzn + (rb * 32 + l31) * PBH_PITCH + 8 * lhi;
            const H* xb0 = xn + ((2 * ph) * 32 + l31) * PBH_PITCH + 8 * lhi;
            const H* xb1 = xn + ((2 * ph + 1 < NB ? 2 * ph + 1 : 2 * ph) * 32 + l31) * PBH_PITCH + 8 * lhi;
            const bool two = 2 * ph + 1 < NB;
            auto oct = [](const H* q) -> hf8 {
                const u32x2 lo = *reinterpret_cast<const u32x2*>(q), hi = *reinterpret_cast<const u32x2*>(q + 4);
                return __builtin_bit_cast(hf8, u32x4{lo[0], lo[1], hi[0], hi[1]});
            };
#pragma unroll
            for (int kk = 0; kk < PB_NT / 16; ++kk) {
                const hf8 av = oct(ga + 16 * kk);
                acc_w[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, oct(xb0 + 16 * kk), acc_w[0], 0, 0, 0);
                if (two) acc_w[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, oct(xb1 + 16 * kk), acc_w[1], 0, 0, 0);
            }
        }

        if (nvalid) commit(nxt, buf ^ 1);
        __syncthreads();
        cur = nxt;
        valid = nvalid;
        buf ^= 1;
    }
    if (p.t_mode != 1) flush_bias(0);

    // ---- this workgroup's partial gw
    {
        const int r0 = rb * 32;
        const int nrows = p.Cout - r0 < 32 ? (p.Cout - r0 > 0 ? p.Cout - r0 : 0) : 32;
        const rsrc_t rgw = make_rsrc_n(p.gwp + ((size_t)g * p.Cout + r0) * p.K, (unsigned)(nrows * p.K) * 4u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (2 * ph + n) * 32 + l31;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rl = (j & 3) + 8 * (j >> 2) + 4 * lhi;
                buf_store(acc_w[n][j], rgw, (col < p.K && nrows > 0) ? (unsigned)(rl * p.K + col) * 4u : PW_OOB, 0);
            }
        }
    }
}

static bool pw_dims_ok(int cin, int cout, long hw) {
    if (cin < 1 || cin > 128 || cout < 1 || hw < 4 || hw % 4) return false;
    const int kp = (cin + 31) / 32 * 32;
    return (double)kp * (double)hw * 4.0 < 4294967000.0 && hw < (1L << 27);
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_pointwise_supported(int cin, int cout, long hw) { return pw_dims_ok(cin, cout, hw) ? 1 : 0; }

// (half output: no per-pixel context form -- a chain's first layer with fp32 input has none in Multisteps, and it spills)
template <int KPV, int WV, typename TO>
static auto pws_pick(int t_mode, bool mean, bool f2) -> void (*)(PwFwdParams) {
    if constexpr (sizeof(TO) == 4) {
        if (f2) {
            if (mean)
                return t_mode == 2 ? pw_fwd_s_kernel<KPV, 2, WV, TO, true, true>
                                   : (t_mode == 1 ? pw_fwd_s_kernel<KPV, 1, WV, TO, true, true> : pw_fwd_s_kernel<KPV, 0, WV, TO, true, true>);
            return t_mode == 2 ? pw_fwd_s_kernel<KPV, 2, WV, TO, false, true>
                               : (t_mode == 1 ? pw_fwd_s_kernel<KPV, 1, WV, TO, false, true> : pw_fwd_s_kernel<KPV, 0, WV, TO, false, true>);
        }
        if (mean)
            return t_mode == 2 ? pw_fwd_s_kernel<KPV, 2, WV, TO, true>
                               : (t_mode == 1 ? pw_fwd_s_kernel<KPV, 1, WV, TO, true> : pw_fwd_s_kernel<KPV, 0, WV, TO, true>);
        return t_mode == 2 ? pw_fwd_s_kernel<KPV, 2, WV, TO> : (t_mode == 1 ? pw_fwd_s_kernel<KPV, 1, WV, TO> : pw_fwd_s_kernel<KPV, 0, WV, TO>);
    } else {
        return t_mode == 1 ? pw_fwd_s_kernel<KPV, 1, WV, TO> : pw_fwd_s_kernel<KPV, 0, WV, TO>;
    }
}

template <typename TI, typename TO>
static int pw_fwd_launch(const void* x, const float* w, const float* bias, const float* t, void* y, int b, int s,
                         int cin, int cout, long hw, int t_mode, int act, float slope, void* stream,
                         unsigned* signs = nullptr, float* ymean = nullptr, int s_mean = 1,
                         const unsigned* xmax = nullptr, unsigned* amax = nullptr) {
    if (b < 0 || s < 1 || act < 0 || act > 2 || t_mode < 0 || t_mode > 2) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!pw_dims_ok(cin, cout, hw) || b % s || !x || !w || !bias || !y || (t_mode && !t)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)x % 16) return SBMC_HIP_EINVAL;
    PwFwdParams p;
    p.x = x; p.w = w; p.bias = bias; p.t = t; p.y = y;
    p.signs = signs;
    p.ymean = ymean;
    p.xmax = xmax;
    p.amax = amax;
    // (a scale word / a magnitude output: the fp32 split-precision kernel only)
    if ((xmax != nullptr || amax != nullptr) && !(sizeof(TI) == 4 && sizeof(TO) == 4)) return SBMC_HIP_EINVAL;
    // (the mean: the fp32 split-precision kernel, or the all-half kernel -- there ymean is a _Float16 tensor)
    if (ymean != nullptr && (s_mean < 1 || b % s_mean || cout > 128 || (t_mode && s != s_mean) ||
                             !((sizeof(TI) == 4 && sizeof(TO) == 4) || (sizeof(TI) == 2 && sizeof(TO) == 2))))
        return SBMC_HIP_EINVAL;
    p.B = b; p.S = t_mode ? s : (ymean != nullptr ? s_mean : 1); p.K = cin; p.Cout = cout;
    p.hw = (unsigned)hw;
    if constexpr (sizeof(TI) == 4) {
        // fp32 in (fp32 or half out): the split-precision kernel on the bf16 matrix pipe (SBMC_HIP_PW_SPLIT=0 keeps the
        // fp32-MFMA kernel: development knob, not part of the ABI)
        if (env_knob("SBMC_HIP_PW_SPLIT", 1) != 0 && (sizeof(TO) == 4 || (signs == nullptr && ymean == nullptr && t_mode != 2))) {
            p.tiles_per_plane = (unsigned)((hw + PS_NT - 1) / PS_NT);
            const unsigned long long nts = (unsigned long long)p.tiles_per_plane * (unsigned)b;
            if (nts > 0xFFFFFFFFull - 4096) return SBMC_HIP_EINVAL;
            p.ntiles = (unsigned)nts;
            p.t_mode = t_mode;
            p.slope = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
            p.nrt = (cout + 127) / 128;
            int sdev = 0, scus = 256;
            if (hipGetDevice(&sdev) != hipSuccess ||
                hipDeviceGetAttribute(&scus, hipDeviceAttributeMultiprocessorCount, sdev) != hipSuccess)
                scus = 256;
            const int skp = (cin + 31) / 32 * 32;
            const bool f2 = xmax != nullptr;
            const size_t slds = (size_t)2 * (f2 ? 2 : 3) * (skp / 8) * PS_NT * 16 + (ymean != nullptr ? (size_t)128 * PS_NT * 4 : 0);
            const unsigned sunit = (unsigned)(NUM_XCD * p.nrt);
            unsigned sgrid = (unsigned)scus / sunit * sunit;
            const unsigned long long sneed = ((unsigned long long)(p.ntiles / (unsigned)p.S) + NUM_XCD - 1) / NUM_XCD * sunit;
            if (sgrid > sneed) sgrid = (unsigned)sneed;
            if (sgrid < sunit) sgrid = sunit;
            hipError_t se = hipSuccess;
#define SBMC_PWS_LAUNCH2(KPV, WV)                                                                        \
    do {                                                                                                 \
        auto kern = pws_pick<KPV, WV, TO>(t_mode, ymean != nullptr, f2);                                  \
        se = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);                 \
        if (se == hipSuccess)                                                                            \
            hipLaunchKernelGGL(kern, dim3(sgrid), dim3(64 * WV), slds, (hipStream_t)stream, p);          \
    } while (0)
            // (WAVES = 4 -- one wave per SIMD with 512 registers -- measured 7-10 % slower than two waves per SIMD
            // and is not instantiated)
#define SBMC_PWS_LAUNCH(KPV) SBMC_PWS_LAUNCH2(KPV, 8)
            switch (skp) {
                case 32: SBMC_PWS_LAUNCH(32); break;
                case 64: SBMC_PWS_LAUNCH(64); break;
                case 96: SBMC_PWS_LAUNCH(96); break;
                default: SBMC_PWS_LAUNCH(128); break;
            }
#undef SBMC_PWS_LAUNCH
#undef SBMC_PWS_LAUNCH2
            if (se != hipSuccess) return (int)se;
            return (int)hipGetLastError();
        }
        if (signs != nullptr || ymean != nullptr || xmax != nullptr || amax != nullptr)
            return SBMC_HIP_EINVAL;                                           // (the fp32-MFMA kernel takes / writes none of them)
    }
    const int ph = PW_FWD_PH;
    const int ntile = 64 * ph;
    p.tiles_per_plane = (unsigned)((hw + ntile - 1) / ntile);
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
    const size_t lds = (size_t)2 * kp * ntile * sizeof(float);
    // a multiple of 8 * nrt workgroups (see the kernel's work assignment), no more than there is work
    unsigned unit = (unsigned)(NUM_XCD * p.nrt);
    unsigned grid = (unsigned)(cus * (3 - ph)) / unit * unit;
    const unsigned long long need = ((unsigned long long)p.ntiles + NUM_XCD - 1) / NUM_XCD * unit;
    if (grid > need) grid = (unsigned)need;
    if (grid < unit) grid = unit;
    hipError_t e = hipSuccess;
    if constexpr (sizeof(TI) == 2 && sizeof(TO) == 2) {
        // half in, half out: the f16 matrix pipe (weights rounded to half); SBMC_HIP_PW_F16MFMA=0 keeps
        // the fp32-MFMA kernel (development knob, not part of the ABI)
        const bool f16mfma = env_knob("SBMC_HIP_PW_F16MFMA", 1) != 0;
        if (ymean != nullptr && !f16mfma) return SBMC_HIP_EINVAL;      // (the fp32-MFMA kernel writes no mean)
        if (f16mfma) {
            const size_t hlds = (size_t)2 * (kp / 4) * 128 * 8;
            if (cout > 128 && cout <= 512 && t_mode == 0 && ymean == nullptr && env_knob("SBMC_HIP_PW_FWD_WIDE", 1) != 0) {
                // wide layer (the 441-channel logits): one workgroup per pixel tile walks all row tiles
                unsigned wgrid = (unsigned)cus;
                if ((unsigned long long)wgrid > nt) wgrid = (unsigned)nt;
#define SBMC_PWHW_LAUNCH(KPV)                                                                            \
    do {                                                                                                 \
        auto kern = pw_fwd_hw_kernel<KPV>;                                                               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)hlds);                  \
        if (e == hipSuccess)                                                                             \
            hipLaunchKernelGGL(kern, dim3(wgrid), dim3(512), hlds, (hipStream_t)stream, p);              \
    } while (0)
                switch (kp) {
                    case 32: SBMC_PWHW_LAUNCH(32); break;
                    case 64: SBMC_PWHW_LAUNCH(64); break;
                    case 96: SBMC_PWHW_LAUNCH(96); break;
                    default: SBMC_PWHW_LAUNCH(128); break;
                }
#undef SBMC_PWHW_LAUNCH
                if (e != hipSuccess) return (int)e;
                return (int)hipGetLastError();
            }
            unsigned hgrid = (unsigned)(2 * cus) / unit * unit;      // two workgroups per CU when they fit
            if (hgrid > need) hgrid = (unsigned)need;
            if (hgrid < unit) hgrid = unit;
#define SBMC_PWH_LAUNCH(KPV)                                                                             \
    do {                                                                                                 \
        auto kern = ymean != nullptr                                                                      \
            ? (t_mode == 2 ? pw_fwd_h_kernel<KPV, 2, true> : (t_mode == 1 ? pw_fwd_h_kernel<KPV, 1, true> : pw_fwd_h_kernel<KPV, 0, true>)) \
            : (t_mode == 2 ? pw_fwd_h_kernel<KPV, 2> : (t_mode == 1 ? pw_fwd_h_kernel<KPV, 1> : pw_fwd_h_kernel<KPV, 0>)); \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)hlds);                  \
        if (e == hipSuccess)                                                                             \
            hipLaunchKernelGGL(kern, dim3(hgrid), dim3(512), hlds, (hipStream_t)stream, p);              \
    } while (0)
            switch (kp) {
                case 32: SBMC_PWH_LAUNCH(32); break;
                case 64: SBMC_PWH_LAUNCH(64); break;
                case 96: SBMC_PWH_LAUNCH(96); break;
                default: SBMC_PWH_LAUNCH(128); break;
            }
#undef SBMC_PWH_LAUNCH
            if (e != hipSuccess) return (int)e;
            return (int)hipGetLastError();
        }
    }
#define SBMC_PW_LAUNCH(KPV)                                                                              \
    do {                                                                                                 \
        auto kern = t_mode == 2 ? pw_fwd_kernel<KPV, 2, PW_FWD_PH, TI, TO>                                \
                                : (t_mode == 1 ? pw_fwd_kernel<KPV, 1, PW_FWD_PH, TI, TO>                 \
                                               : pw_fwd_kernel<KPV, 0, PW_FWD_PH, TI, TO>);               \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        if (e == hipSuccess)                                                                             \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * PW_FWD_PH), lds, (hipStream_t)stream, p);    \
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

extern "C" int sbmc_pointwise_fwd_f32(const float* x, const float* w, const float* bias, const float* t, float* y,
                                      int b, int s, int cin, int cout, long hw, int t_mode, int act, float slope,
                                      void* stream) {
    return pw_fwd_launch<float, float>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream);
}

extern "C" int sbmc_pointwise_fwd_signs_f32(const float* x, const float* w, const float* bias, const float* t, float* y,
                                            unsigned* signs, int b, int s, int cin, int cout, long hw, int t_mode,
                                            int act, float slope, void* stream) {
    return pw_fwd_launch<float, float>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream, signs);
}

extern "C" int sbmc_pointwise_fwd_mean_f32(const float* x, const float* w, const float* bias, const float* t, float* y,
                                           unsigned* signs, float* ymean, int s_mean, int b, int s, int cin, int cout,
                                           long hw, int t_mode, int act, float slope, void* stream) {
    if (ymean == nullptr) return SBMC_HIP_EINVAL;
    return pw_fwd_launch<float, float>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream, signs, ymean,
                                       s_mean);
}

extern "C" int sbmc_pointwise_fwd_f16(const void* x, int x_is_half, const float* w, const float* bias, const float* t,
                                      void* y, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                                      float slope, void* stream) {
    if (x_is_half)
        return pw_fwd_launch<_Float16, _Float16>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream);
    return pw_fwd_launch<float, _Float16>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream);
}

// The fp32 layer with magnitude words (ABI 6): xmax != nullptr -- a device word holding the bit pattern of a float >= max |x|
// -- selects the two-f16-plane form of the split-precision kernel (pw_fwd_s_kernel<.., F2>), nullptr the three-bf16-plane
// form; amax != nullptr (a word the caller zeroed, or holding a lower bound) is raised to the bit pattern of max |y|.
// signs / ymean: as sbmc_pointwise_fwd_signs_f32 / _mean_f32, or nullptr.
extern "C" int sbmc_pointwise_fwd_scaled_f32(const float* x, const float* w, const float* bias, const float* t, float* y,
                                             unsigned* signs, float* ymean, int s_mean, const unsigned* xmax,
                                             unsigned* amax, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                                             float slope, void* stream) {
    if (env_knob("SBMC_HIP_PW_SPLIT", 1) == 0) return SBMC_HIP_EINVAL;       // (no fp32-MFMA form with magnitude words)
    return pw_fwd_launch<float, float>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream, signs, ymean,
                                       ymean != nullptr ? s_mean : 1, xmax, amax);
}

// all-half layer that also writes ymean [b / s_mean, cout, hw] (_Float16): the mean of y over groups of s_mean images
extern "C" int sbmc_pointwise_fwd_mean_f16(const void* x, const float* w, const float* bias, const float* t, void* y,
                                           void* ymean, int s_mean, int b, int s, int cin, int cout, long hw, int t_mode,
                                           int act, float slope, void* stream) {
    if (!ymean) return SBMC_HIP_EINVAL;
    return pw_fwd_launch<_Float16, _Float16>(x, w, bias, t, y, b, s, cin, cout, hw, t_mode, act, slope, stream, nullptr,
                                             static_cast<float*>(ymean), s_mean);
}

static unsigned pw_bwd_grid(int b, int s, long hw, unsigned* nunits_out) {
    const unsigned tpp = (unsigned)((hw + PB_NT - 1) / PB_NT);
    const unsigned long long nunits = (unsigned long long)tpp * (unsigned)(b / s);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    if (nunits_out) *nunits_out = (unsigned)nunits;
    return nunits < (unsigned long long)cus ? (unsigned)nunits : (unsigned)cus;
}

extern "C" int sbmc_pointwise_bwd_supported(int cin, int cout, long hw) {
    return (pw_dims_ok(cin, cout, hw) && cout <= 128 && (double)cout * (double)hw * 4.0 < 4294967000.0) ? 1 : 0;
}

extern "C" int sbmc_pointwise_bwd_groups(int b, int s, int t_mode, long hw) {
    if (b <= 0 || s < 1 || hw <= 0) return 1;
    return (int)pw_bwd_grid(b, t_mode ? s : 1, hw, nullptr);
}

template <typename TA, typename TXT>
static int pw_bwd_launch(const void* gy, const void* y, const void* x, const float* w, void* gx,
                         float* gw_partial, float* gb_partial, float* gt, const void* gmean,
                         int s_mean, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                         float slope, void* stream, bool y_is_signs = false, const unsigned* gmax = nullptr,
                         const unsigned* gmmax = nullptr, const unsigned* xmax = nullptr, unsigned* gxmax = nullptr) {
    if (b < 0 || s < 1 || act < 0 || act > 2 || t_mode < 0 || t_mode > 2) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!sbmc_pointwise_bwd_supported(cin, cout, hw) || b % s || !gy || !x || !w || !gw_partial || !gb_partial ||
        (act != 0 && !y) || (t_mode == 2 && !gt) || (gmean && (s_mean < 1 || b % s_mean || t_mode == 2)))
        return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)x % 16 || (act != 0 && (uintptr_t)y % (y_is_signs ? 4 : 16)) ||
        (t_mode == 2 && (uintptr_t)gt % 16) || (uintptr_t)gmean % 16 || (uintptr_t)gx % 16)
        return SBMC_HIP_EINVAL;
    PwBwdParams p;
    p.gy = gy; p.y = act != 0 ? y : gy; p.x = x; p.w = w; p.gx = gx; p.gwp = gw_partial; p.gbp = gb_partial; p.gt = gt;
    p.gm = gmean; p.Sm = gmean ? s_mean : 1;
    p.gmax = gmax; p.gmmax = gmmax; p.xmax = xmax; p.gxmax = gxmax;
    // walk order: the samples of a pixel tile one after the other when something per-pixel is shared
    // between them (context term, mean gradient: its tile then stays in L2), else plain batch order
    p.B = b; p.S = t_mode ? s : (gmean ? s_mean : 1); p.K = cin; p.Cout = cout;
    p.Bq = t_mode == 1 ? b / s : 1;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + PB_NT - 1) / PB_NT);
    p.t_mode = t_mode;
    p.slope = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
    const unsigned grid = pw_bwd_grid(b, p.S, hw, &p.nunits);
    // (argument errors before anything is written: the largest |gx| comes out of the split kernels only -- of the
    // three-plane form only without a context / mean gradient --, magnitude words go with fp32 tensors)
    if constexpr (sizeof(TA) == 4 && sizeof(TXT) == 4) {
        const int gmode0 = env_knob("SBMC_HIP_PW_GWS", 2);
        const bool side0 = t_mode == 2 || gmean;
        const bool gws0 = gmode0 != 0 && (!side0 || gmode0 >= 2);
        const bool f20 = gws0 && gmax != nullptr && xmax != nullptr && (!gmean || gmmax != nullptr) && (y_is_signs || act == 0) &&
                         (PW_BWD_GXS || !gx);
        if (gxmax != nullptr && !(gws0 && PW_BWD_GXS && (f20 || !side0))) return SBMC_HIP_EINVAL;
    } else if (gmax != nullptr || gxmax != nullptr) {
        return SBMC_HIP_EINVAL;
    }
    hipError_t e = hipMemsetAsync(gb_partial, 0, (size_t)grid * p.Bq * cout * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const int kp = (cin + 31) / 32 * 32;
    if constexpr (sizeof(TA) == 2 && sizeof(TXT) == 2) {
        // all-half layer: the f16 matrix pipe (SBMC_HIP_PW_F16MFMA=0 keeps the fp32-MFMA kernel: development knob)
        if (env_knob("SBMC_HIP_PW_F16MFMA", 1) != 0) {
            const size_t hlds = (size_t)2 * ((128 + kp) * PBH_PITCH + 32 * PB_NT * 4) * 2;
#define SBMC_PWBH_LAUNCH2(KPV, DXV, TPV)                                                                 \
    do {                                                                                                 \
        auto kern = (gmean && !TPV) ? pw_bwd_h_kernel<KPV, DXV, false, true>                              \
                                    : pw_bwd_h_kernel<KPV, DXV, TPV, false>;                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)hlds);                  \
        if (e == hipSuccess)                                                                             \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(PW_THREADS), hlds, (hipStream_t)stream, p);        \
    } while (0)
#define SBMC_PWBH_LAUNCH(KPV)                                                                            \
    do {                                                                                                 \
        if (gx) { if (t_mode == 2) SBMC_PWBH_LAUNCH2(KPV, true, true); else SBMC_PWBH_LAUNCH2(KPV, true, false); } \
        else    { if (t_mode == 2) SBMC_PWBH_LAUNCH2(KPV, false, true); else SBMC_PWBH_LAUNCH2(KPV, false, false); } \
    } while (0)
            switch (kp) {
                case 32: SBMC_PWBH_LAUNCH(32); break;
                case 64: SBMC_PWBH_LAUNCH(64); break;
                case 96: SBMC_PWBH_LAUNCH(96); break;
                default: SBMC_PWBH_LAUNCH(128); break;
            }
#undef SBMC_PWBH_LAUNCH
#undef SBMC_PWBH_LAUNCH2
            if (e != hipSuccess) return (int)e;
            return (int)hipGetLastError();
        }
    }
    size_t lds = (size_t)2 * (128 + kp) * PB_PITCH * sizeof(float);
    // fp32 tensors: the weight-gradient product on the bf16 matrix pipe (split precision); SBMC_HIP_PW_GWS=0 keeps
    // the all-fp32-MFMA kernel (development knob)
    // (plain layers: 4.41 -> 3.96 ms at 720p x 8 spp.  With a per-pixel context gradient or a mean gradient -- 16
    // more live registers -- the kernel's NARROW gx product makes the room; SBMC_HIP_PW_GWS=1 keeps those on the
    // all-fp32-MFMA kernel, 2 = default splits them too)
    bool gws = false, f2 = false;
    if constexpr (sizeof(TA) == 4 && sizeof(TXT) == 4) {
        const int gmode = env_knob("SBMC_HIP_PW_GWS", 2);
        const bool side = t_mode == 2 || gmean;
        gws = gmode != 0 && (!side || gmode >= 2);
        // the two-f16-plane form: wherever the caller handed over the words it needs (a layer with an activation: with
        // the forward's sign bits; a linear layer reads no y at all)
        f2 = gws && gmax != nullptr && xmax != nullptr && (!gmean || gmmax != nullptr) && (y_is_signs || act == 0) &&
             (PW_BWD_GXS || !gx);
        if (gws)
            lds = (size_t)128 * ((PW_BWD_GXS || !gx) ? 0 : (side ? PB_PITCH_N : PB_PITCH)) * sizeof(float) +
                  (f2 ? (PW_BWD_RAW ? (size_t)2 * (128 + kp) * PBS_PITCH * 2 + (size_t)(128 + kp) * PB_NT * 4      // + the raw slots
                                    : (size_t)4 * (128 + kp) * PBS_PITCH * 2)                                    // two stages
                      : (size_t)3 * (128 + kp) * PBS_PITCH * 2);
    }
#define SBMC_PWB_PICK(KPV, DXV, TPV, SGV, GWSV)                                                          \
    ((gmean && !TPV) ? pw_bwd_kernel<KPV, DXV, false, true, float, float, SGV, GWSV>                     \
                     : pw_bwd_kernel<KPV, DXV, TPV, false, float, float, SGV, GWSV>)
#define SBMC_PWB_LAUNCH2(KPV, DXV, TPV)                                                                  \
    do {                                                                                                 \
        auto kern = (gmean && !TPV) ? pw_bwd_kernel<KPV, DXV, false, true, TA, TXT>                       \
                                    : pw_bwd_kernel<KPV, DXV, TPV, false, TA, TXT>;                       \
        if constexpr (sizeof(TA) == 4 && sizeof(TXT) == 4) {                                             \
            if (f2) {                                                                                    \
                kern = (gmean && !TPV) ? pw_bwd_kernel<KPV, DXV, false, true, float, float, true, true, true>   \
                                       : pw_bwd_kernel<KPV, DXV, TPV, false, float, float, true, true, true>;   \
            } else if (y_is_signs) {                                                                     \
                if (gws) kern = SBMC_PWB_PICK(KPV, DXV, TPV, true, true);                                \
                else     kern = SBMC_PWB_PICK(KPV, DXV, TPV, true, false);                               \
            } else if (gws) {                                                                            \
                kern = SBMC_PWB_PICK(KPV, DXV, TPV, false, true);                                        \
            }                                                                                            \
        }                                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
        if (e == hipSuccess)                                                                             \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(PW_THREADS), lds, (hipStream_t)stream, p);         \
    } while (0)
#define SBMC_PWB_LAUNCH(KPV)                                                                             \
    do {                                                                                                 \
        if (gx) { if (t_mode == 2) SBMC_PWB_LAUNCH2(KPV, true, true); else SBMC_PWB_LAUNCH2(KPV, true, false); } \
        else    { if (t_mode == 2) SBMC_PWB_LAUNCH2(KPV, false, true); else SBMC_PWB_LAUNCH2(KPV, false, false); } \
    } while (0)
    switch (kp) {
        case 32: SBMC_PWB_LAUNCH(32); break;
        case 64: SBMC_PWB_LAUNCH(64); break;
        case 96: SBMC_PWB_LAUNCH(96); break;
        default: SBMC_PWB_LAUNCH(128); break;
    }
#undef SBMC_PWB_LAUNCH
#undef SBMC_PWB_LAUNCH2
#undef SBMC_PWB_PICK
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}

// gw / gbias partial sums of a wide linear layer (pw_gw_wide_kernel): gw_partial [groups, cout, cin], gb_partial
// [groups, cout], every element written; groups = sbmc_pointwise_gw_wide_groups(b, hw)
extern "C" int sbmc_pointwise_gw_wide_supported(int cin, int cout, long hw) {
    return (cin >= 1 && cin <= 128 && cout > 128 && cout <= 512 && hw > 0 && hw % 4 == 0 &&
            (double)cout * (double)hw * 4.0 < 4294967000.0) ? 1 : 0;
}
extern "C" int sbmc_pointwise_gw_wide_groups(int b, long hw) {
    if (b <= 0 || hw <= 0) return 1;
    return (int)pw_bwd_grid(b, 1, hw, nullptr);
}
template <typename T>
static int pw_gw_wide_launch(const void* gz, const void* x, float* gw_partial, float* gb_partial, int b, int cin, int cout,
                             long hw, void* stream);
extern "C" int sbmc_pointwise_gw_wide_f32(const float* gz, const float* x, float* gw_partial, float* gb_partial, int b,
                                          int cin, int cout, long hw, void* stream) {
    return pw_gw_wide_launch<float>(gz, x, gw_partial, gb_partial, b, cin, cout, hw, stream);
}
// gz and x _Float16 (training under torch.autocast(float16)); the partial sums stay fp32
extern "C" int sbmc_pointwise_gw_wide_f16(const void* gz, const void* x, float* gw_partial, float* gb_partial, int b,
                                          int cin, int cout, long hw, void* stream) {
    return pw_gw_wide_launch<_Float16>(gz, x, gw_partial, gb_partial, b, cin, cout, hw, stream);
}
template <typename T>
static int pw_gw_wide_launch(const void* gz, const void* x, float* gw_partial, float* gb_partial, int b, int cin, int cout,
                             long hw, void* stream) {
    if (b < 0) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!sbmc_pointwise_gw_wide_supported(cin, cout, hw) || !gz || !x || !gw_partial || !gb_partial) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gz % 16 || (uintptr_t)x % 16) return SBMC_HIP_EINVAL;
    PwBwdParams p;
    memset(&p, 0, sizeof(p));
    p.gy = gz; p.x = x; p.gwp = gw_partial; p.gbp = gb_partial;
    p.B = b; p.S = 1; p.K = cin; p.Cout = cout; p.Bq = 1;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + PB_NT - 1) / PB_NT);
    p.slope = 1.f;
    const unsigned grid = pw_bwd_grid(b, 1, hw, &p.nunits);
    const int kp = (cin + 31) / 32 * 32;
    const size_t lds = (size_t)3 * (128 + kp) * PBS_PITCH * 2;
    hipError_t e = hipSuccess;
#define SBMC_GWW(KPV)                                                                                    \
    do {                                                                                                 \
        auto kern = pw_gw_wide_kernel<KPV, T>;                                                           \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(grid), dim3(PW_THREADS), lds, (hipStream_t)stream, p); \
    } while (0)
    switch (kp) {
        case 32: SBMC_GWW(32); break;
        case 64: SBMC_GWW(64); break;
        case 96: SBMC_GWW(96); break;
        default: SBMC_GWW(128); break;
    }
#undef SBMC_GWW
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return (int)hipGetLastError();
}

// The wide linear layer's whole backward in one pass (ABI 6; pw_wide_bwd2_kernel): gx [b, cin, hw], gw / gbias partial sums
// as sbmc_pointwise_gw_wide_f32.  gmax / xmax: device words holding the bit patterns of floats >= max |gz| / max |x|;
// gxmax (or NULL): a zeroed word raised to max |gx|; ws: sbmc_pointwise_wide_bwd_ws_bytes() bytes of scratch (the
// weights' prepared planes; written and read by this call only).
extern "C" size_t sbmc_pointwise_wide_bwd_ws_bytes(void) { return (size_t)4 * 8 * 2 * 4 * 64 * 16 + 256; }
extern "C" int sbmc_pointwise_wide_bwd_f32(const float* gz, const float* x, const float* w, float* gx, float* gw_partial,
                                           float* gb_partial, void* ws, const unsigned* gmax, const unsigned* xmax,
                                           unsigned* gxmax, int b, int cin, int cout, long hw, void* stream) {
    if (b < 0) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!sbmc_pointwise_gw_wide_supported(cin, cout, hw) || !gz || !x || !w || !gx || !gw_partial || !gb_partial || !ws ||
        !gmax || !xmax || (double)cin * (double)hw * 4.0 >= 4294967000.0)
        return SBMC_HIP_EINVAL;
    if ((uintptr_t)gz % 16 || (uintptr_t)x % 16 || (uintptr_t)gx % 16 || (uintptr_t)ws % 16) return SBMC_HIP_EINVAL;
    PwWideParams pp;
    memset(&pp, 0, sizeof(pp));
    PwBwdParams& p = pp.b;
    p.gy = gz; p.x = x; p.w = w; p.gx = gx; p.gwp = gw_partial; p.gbp = gb_partial;
    p.gmax = gmax; p.xmax = xmax; p.gxmax = gxmax;
    p.B = b; p.S = 1; p.K = cin; p.Cout = cout; p.Bq = 1;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + PB_NT - 1) / PB_NT);
    p.slope = 1.f;
    u32x4* wprep = static_cast<u32x4*>(ws);
    float* wscale = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)4 * 8 * 2 * 4 * 64 * 16);
    pp.wprep = wprep; pp.wscale = wscale;
    hipLaunchKernelGGL(pw_wide_prep_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, w, wprep, wscale, cout, cin);
    const unsigned grid = pw_bwd_grid(b, 1, hw, &p.nunits);
    const int kp = (cin + 31) / 32 * 32;
    const size_t lds = (size_t)2 * (128 + kp) * PBS_PITCH * 2 + (size_t)8 * 8 * 64 * 16 + 512 * 4;
    hipError_t e = hipSuccess;
    const bool g2 = kp == 128 && (cout + 127) / 128 == 4 && env_knob("SBMC_PW_WIDE_G2", 0) != 0;
#define SBMC_WB2(KPV)                                                                                    \
    do {                                                                                                 \
        auto kern = (KPV == 128 && g2) ? pw_wide_bwd2_kernel<128, true> : pw_wide_bwd2_kernel<KPV>;      \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(grid), dim3(PW_THREADS), lds, (hipStream_t)stream, pp); \
    } while (0)
    switch (kp) {
        case 32: SBMC_WB2(32); break;
        case 64: SBMC_WB2(64); break;
        case 96: SBMC_WB2(96); break;
        default: SBMC_WB2(128); break;
    }
#undef SBMC_WB2
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return (int)hipGetLastError();
}

extern "C" int sbmc_pointwise_bwd_f32(const float* gy, const float* y, const float* x, const float* w, float* gx,
                                      float* gw_partial, float* gb_partial, float* gt, const float* gmean,
                                      int s_mean, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                                      float slope, void* stream) {
    return pw_bwd_launch<float, float>(gy, y, x, w, gx, gw_partial, gb_partial, gt, gmean, s_mean, b, s, cin, cout,
                                       hw, t_mode, act, slope, stream);
}

// The fp32 backward with magnitude words (ABI 6; see sbmc_pointwise_fwd_scaled_f32).  gmax, xmax (and gmmax with gmean)
// not NULL: both products in the two-f16-plane form; gxmax != NULL: raised to the bit pattern of max |gx|.
// signs: the forward's sign bits (act != 0), unused for a linear layer.
extern "C" int sbmc_pointwise_bwd_scaled_f32(const float* gy, const unsigned* signs, const float* x, const float* w,
                                             float* gx, float* gw_partial, float* gb_partial, float* gt,
                                             const float* gmean, int s_mean, const unsigned* gmax, const unsigned* gmmax,
                                             const unsigned* xmax, unsigned* gxmax, int b, int s, int cin, int cout,
                                             long hw, int t_mode, int act, float slope, void* stream) {
    if ((gmax == nullptr) != (xmax == nullptr) || (gxmax != nullptr && gx == nullptr)) return SBMC_HIP_EINVAL;
    return pw_bwd_launch<float, float>(gy, signs, x, w, gx, gw_partial, gb_partial, gt, gmean, s_mean, b, s, cin, cout, hw,
                                       t_mode, act, slope, stream, act != 0, gmax, gmmax, xmax, gxmax);
}

extern "C" int sbmc_pointwise_bwd_signs_f32(const float* gy, const unsigned* signs, const float* x, const float* w,
                                            float* gx, float* gw_partial, float* gb_partial, float* gt,
                                            const float* gmean, int s_mean, int b, int s, int cin, int cout, long hw,
                                            int t_mode, int act, float slope, void* stream) {
    return pw_bwd_launch<float, float>(gy, signs, x, w, gx, gw_partial, gb_partial, gt, gmean, s_mean, b, s, cin, cout,
                                       hw, t_mode, act, slope, stream, true);
}

extern "C" int sbmc_pointwise_bwd_f16(const void* gy, const void* y, const void* x, int x_is_half, const float* w,
                                      void* gx, float* gw_partial, float* gb_partial, float* gt, const void* gmean,
                                      int s_mean, int b, int s, int cin, int cout, long hw, int t_mode, int act,
                                      float slope, void* stream) {
    if (x_is_half)
        return pw_bwd_launch<_Float16, _Float16>(gy, y, x, w, gx, gw_partial, gb_partial, gt, gmean, s_mean, b, s,
                                                 cin, cout, hw, t_mode, act, slope, stream);
    return pw_bwd_launch<_Float16, float>(gy, y, x, w, gx, gw_partial, gb_partial, gt, gmean, s_mean, b, s, cin,
                                          cout, hw, t_mode, act, slope, stream);
}

// A CHAIN of per-sample 1x1 convolutions in one pass (round 6).
//
// The reference's per-sample embeddings and kernel regressor are chains of nn.Conv2d(1x1) + ReLU / LeakyReLU
// (sbmc/modules.py:154-175; built at sbmc/models.py:79-102, run at :147-153, 171-177, 196-199).  Layer by layer
// (csrc/pointwise.hip) every 3.77 GB activation (720p x 8 spp x 128 channels) is written by one kernel and read back by
// the next, each at its byte floor: six moves per layer pair.  Here two or three layers run on a 64-pixel tile while
// it sits in LDS: a tile's input is read ONCE, the intermediate activations are written only where a backward needs
// them (training) and never read back.
//
// Number format: the 3 x 3 kernels' -- two f16 planes under a power-of-two scale, three of the four partial products,
// fp32 accumulation (common.hpp) -- but the scale of an operand tile is taken PER PIXEL from the tile itself (the
// reduction runs over channels, so every pixel column of the B operand may have its own scale; the accumulator column is
// scaled back by it in the epilogue).  No magnitude word, no bound: the kernel takes network inputs as they are, and an
// intermediate activation is scaled by its own largest channel, not by a bound on the whole tensor.
//
// Work split (8 waves): a wave owns 16 output channels over all 64 pixels of the tile -- v_mfma_f32_16x16x32_f16, the
// weights of ALL layers of the chain in registers (32 per layer: its 16 rows x 128 inputs in two planes), four 16 x 16
// accumulator blocks.  The B operand (the tile's two planes, [plane][channel octet][pixel] entries of 16 bytes) is read
// by every wave: 256 KB of ds_read_b128 per layer and tile, a quarter of the LDS's time.
//
// Data movement: the fp32 input tile travels HBM -> LDS by LDS-DMA (16 bytes per lane, no register in between), a whole
// tile ahead; outputs that go to HBM are assembled as an fp32 tile in LDS and stored 16 bytes per lane, whole rows of
// 256 bytes per 16 lanes.
#include "common.hpp"
#include <cstring>
#include <type_traits>
#include <utility>
#include "../../include/sbmc_hip.h"

namespace sbmc {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using hf8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr unsigned PC_OOB = 0xFFFFFFF0u;
constexpr int PC_NT = 64;           // pixels per tile
constexpr int PC_MAXL = 3;          // layers per chain

__device__ __forceinline__ rsrc_t rsrc_n(const void* base, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* u = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, a), __builtin_bit_cast(hf8, b), c, 0, 0, 0);
}
// workgroup barrier that orders LDS traffic only (global requests stay in flight)
__device__ __forceinline__ void lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ unsigned wave_max_u(unsigned m) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, s, 64);
        m = m > o ? m : o;
    }
    return m;
}
// f16_split_pair with a per-LANE scale (a power of two)
__device__ __forceinline__ void split_pair_v(float a, float b, float c, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(a), "v"(b), "v"(c));
}
// lanes LANE and LANE + 32 of v = the halves of a ballot (csrc/pointwise.hip write_lanes: the s_nop covers the scalar
// register the compare has just written)
template <int LANE>
__device__ __forceinline__ void put_lanes(unsigned& v, unsigned long long ballot) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
        : "+v"(v) : "s"((unsigned)ballot), "s"((unsigned)(ballot >> 32)), "n"(LANE), "n"(LANE + 32));
}
template <class F, int... J>
__device__ __forceinline__ void unrolled_impl(F&& f, std::integer_sequence<int, J...>) {
    (f(std::integral_constant<int, J>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void unrolled(F&& f) { unrolled_impl(f, std::make_integer_sequence<int, N>{}); }

}  // namespace

struct PwChainParams {
    const float* x;              // [B, K0, hw]
    const float* t;              // context term of the FIRST layer: nullptr, [B/S, C0] (t_mode 1) or [B/S, C0, hw] (t_mode 2)
    const unsigned* tmax;        // device word: bit pattern of a float >= max |t| (t_mode != 0)
    const float* w[PC_MAXL];     // [C_l, K_l] (K_0 = K0, K_l = C_{l-1})
    const float* bias[PC_MAXL];  // [C_l]
    float* y[PC_MAXL];           // [B, C_l, hw]; nullptr: not stored (never for the last layer)
    unsigned* signs[PC_MAXL];    // [B, C_l, ceil(hw / 32)] one bit per output: value > 0 (nullptr: not wanted; needs y[l])
    unsigned* amax[PC_MAXL];     // device words raised to the bit pattern of max |y_l| (nullptr: not wanted)
    float* ymean;                // [B/S, C_last, hw]: mean of the last layer's output over the S samples of a pixel, or nullptr
    float slope[PC_MAXL];        // 1: linear, 0: relu, else leaky relu
    int cout[PC_MAXL];
    int B, S, K0, t_mode;
    unsigned hw, tiles_per_plane, nunits;
    int timing;                  // development: wave 0 of workgroup 0 leaves its cycles per phase in y[last][0 ..]
};

// exponent field of the power of two that brings a magnitude with bit pattern `maxbits` into [2^14, 2^15) (pow2_scale_of):
// the scale is e << 23, its inverse (254 - e) << 23 -- no division anywhere (e = 254, an all-zero column: inverse 0)
__device__ __forceinline__ unsigned scale_exp(unsigned maxbits) {
    const int e = 268 - (int)(maxbits >> 23);
    return (unsigned)(e < 1 ? 1 : (e > 254 ? 254 : e));
}
__device__ __forceinline__ float exp_scale(unsigned e) { return __builtin_bit_cast(float, e << 23); }
__device__ __forceinline__ float exp_inverse(unsigned e) { return __builtin_bit_cast(float, (254u - e) << 23); }
// descriptor over [base, base + bytes): base and bytes are uniform by construction (kernel arguments, block and loop
// counters), which the compiler sees -- no readfirstlane
__device__ __forceinline__ rsrc_t rsrc_u(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// lanes I, I + 4, I + 8, I + 12 (and + 32) of `word` <- eight scalar values (sign words assembled by the scalar unit)
template <int J0>
__device__ __forceinline__ void put_words8(unsigned& word, unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned b0,
                                           unsigned b1, unsigned b2, unsigned b3) {
    asm("s_nop 0\n\t"
        "v_writelane_b32 %0, %1, %9\n\t"
        "v_writelane_b32 %0, %2, %10\n\t"
        "v_writelane_b32 %0, %3, %11\n\t"
        "v_writelane_b32 %0, %4, %12\n\t"
        "v_writelane_b32 %0, %5, %13\n\t"
        "v_writelane_b32 %0, %6, %14\n\t"
        "v_writelane_b32 %0, %7, %15\n\t"
        "v_writelane_b32 %0, %8, %16"
        : "+v"(word)
        : "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(b0), "s"(b1), "s"(b2), "s"(b3),
          "n"(J0), "n"(J0 + 4), "n"(J0 + 8), "n"(J0 + 12), "n"(J0 + 32), "n"(J0 + 36), "n"(J0 + 40), "n"(J0 + 44));
}

// KP0: the first layer's input channels rounded up to a multiple of 32; NL: layers (2 or 3).
// LDS: raw [KP0][64] fp32 (DMA target) | xp [2][KP0/8][64] entries (first layer's operand) | pp [NL-1][2][16][64] entries (the
// intermediate activations as operands) | per-pixel magnitude words of x.
//
// One phase per layer, one barrier per phase: a wave reads its B operands, multiplies, and finishes its 16 x 64 block of the
// layer on the accumulators -- bias / context term / activation, the stores to HBM (straight from the accumulators: four
// rows x 64 bytes per instruction, a row's four pieces back to back), the sign words (ballots of the accumulator
// registers, put together by the scalar unit), and the two planes of the block as the next layer's operand.  The scale
// of that operand is a per-pixel BOUND, carried in registers from layer to layer: with m(px) the largest |x| of the pixel's
// input channels (exact: the tile's own maximum, found while it is staged), |y_0| <= R_0 m + max|b_0| + max|t| and
// |y_l| <= R_l bound_{l-1} + max|b_l|, R_l the largest absolute row sum of the layer's weights -- a few bits above the
// true maximum per layer for weights of mixed sign, harmless (common.hpp: the planes keep 22 bits of anything within
// 2^-10 of the scale and an absolute 2^-39 of it below).  So nothing waits for a maximum over the tile's 128 channels.
//
// The loop is an instruction count (two waves per SIMD; the matrix pipe takes a quarter of the time): nothing in it is
// masked per value.  Columns beyond the plane carry whatever the neighbouring row holds -- a pixel column's scale,
// products and sums never meet another column's, and no store of theirs is issued --, rows beyond a layer's width are zero
// weights and zero bias, rows beyond the input's width are never written in the raw tile (zeroed once; a request beyond
// the image's slice of the tensor returns nothing).  Addresses: one lane offset per role, computed before the loop;
// everything that changes with the tile is a scalar offset, the pixel block an immediate.
template <int KP0, int NL>
__global__ __launch_bounds__(512) void pw_chain_fwd_kernel(PwChainParams p) {
    static_assert(NL >= 2 && NL <= PC_MAXL, "two or three layers");
    constexpr int KO0 = KP0 / 8;                        // channel octets of the first layer's input
    constexpr bool FIRST_ALL = KO0 >= 8;                // a staging thread's first octet (wave) exists for every wave
    constexpr bool SECOND = KO0 > 8;                    // its second octet (wave + 8) exists for some waves
    constexpr bool SECOND_ALL = KO0 == 16;              //   ... for all of them
    extern __shared__ float4 pc_lds[];
    float* raw = reinterpret_cast<float*>(pc_lds);
    u32x4* xp = reinterpret_cast<u32x4*>(raw + KP0 * PC_NT);
    u32x4* pp = xp + 2 * KO0 * PC_NT;                                // [NL - 1][2][16][64]
    unsigned* pmx = reinterpret_cast<unsigned*>(pp + (NL - 1) * 2 * 16 * PC_NT);   // [2][64]: per-pixel max |x|, this / the next tile
    float* red = reinterpret_cast<float*>(pmx + 2 * PC_NT);         // [2][NL][8]: the waves' row-sum / bias maxima (prologue)
    float* btab = red + 2 * PC_MAXL * 8;                            // [NL][128]: the layers' biases
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int r = lane & 15, q = lane >> 4;
    const unsigned hw = p.hw;
    const unsigned G = gridDim.x;
    const unsigned S = (unsigned)p.S;
    const unsigned wpr = (hw + 31) / 32;
    const int r0 = 16 * wave;                           // this wave's first output row

    // ---- the weights of every layer: this wave's 16 rows as A operands, two planes under the scale of the wave's own rows
    // a*[l][s] = planes of W_l[16 wave + lane % 16][32 s + 8 (lane / 16) + 0..7]
    u32x4 ah[NL][4], al[NL][4];
    float icw[NL];                                      // 1 / (the rows' power-of-two scale)
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int K = l == 0 ? p.K0 : p.cout[l - 1], C = p.cout[l];
        const rsrc_t rw = rsrc_n(p.w[l], (unsigned)(C * K) * 4u);
        const int row = r0 + r;
        float v[4][8];
        float wm = 0.f, rs = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 32 * s + 8 * q + i;
                v[s][i] = buf_load(rw, (row < C && k < K) ? (unsigned)(row * K + k) * 4u : PC_OOB, 0);
                wm = __builtin_fmaxf(wm, __builtin_fabsf(v[s][i]));
                rs += __builtin_fabsf(v[s][i]);
            }
        }
        const unsigned ew = scale_exp((unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u(__builtin_bit_cast(unsigned, wm))));
        const float cw = exp_scale(ew);
        icw[l] = exp_inverse(ew);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hp, lp;
                f16_split_pair(v[s][2 * j], v[s][2 * j + 1], cw, hp, lp);
                ah[l][s][j] = hp;
                al[l][s][j] = lp;
            }
        }
        const rsrc_t rb = rsrc_n(p.bias[l], (unsigned)C * 4u);
        float bm = 0.f;
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bv = buf_load(rb, (unsigned)(r0 + 4 * q + i) * 4u, 0);   // (rows >= C: 0)
                btab[l * 128 + r0 + 4 * q + i] = bv;
                bm = __builtin_fmaxf(bm, __builtin_fabsf(bv));
            }
        }
        // the wave's largest absolute row sum (a row's sum: over the four lanes that share it) and largest |bias|
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        const unsigned rsm = wave_max_u(__builtin_bit_cast(unsigned, rs)), bmm = wave_max_u(__builtin_bit_cast(unsigned, bm));
        if (lane == 0) {
            red[l * 8 + wave] = __builtin_bit_cast(float, rsm);
            red[(NL + l) * 8 + wave] = __builtin_bit_cast(float, bmm);
        }
    }

    // ---- lane offsets, once
    // staging role (pixel `lane`, octets wave and wave + 8): byte address of raw[8 wave][lane], of entry xp[wave][lane]
    const unsigned sraw = (unsigned)((8 * wave * PC_NT + lane) * 4);
    const unsigned sxp = (unsigned)((wave * PC_NT + lane) * 16);
    // request role: row lane / 16 of a group of four, pixels 4 (lane % 16) ..
    const unsigned xv = ((unsigned)q * hw + 4u * r) * 4u;
    // accumulator role: rows r0 + 4 q + i, pixel 16 k + r
    const unsigned av = ((unsigned)(4 * q) * hw + (unsigned)r) * 4u;          // in a [rows][hw] plane from row r0 on
    // this lane's half entries of the next operand: channels r0 + 4 q .. of octet 2 wave + q / 2; + 256 k (+ 16 KB: low plane)
    const unsigned ppw = (unsigned)(((2 * wave + (q >> 1)) * PC_NT + r) * 16 + 8 * (q & 1));
    // sign-word role: lane j / j + 32 (j < 16) keeps row r0 + j's words of the tile's two
    const unsigned sgv = (lane & 31) < 16 ? ((unsigned)(lane & 31) * wpr + (unsigned)(lane >> 5)) * 4u : PC_OOB;

    // ---- the walk: units (a 64-pixel tile of an image group) g, g + G, ..; within a unit the S samples one after the other.
    // A cursor that advances: no division in the loop.
    struct Cur { unsigned unit, s, bq, pt; };
    const unsigned tpp = p.tiles_per_plane, Gd = G / tpp, Gm = G % tpp;
    auto advance = [&](Cur c) -> Cur {
        c.s += 1;
        if (c.s == S) {
            c.s = 0;
            c.unit += G;
            c.bq += Gd;
            c.pt += Gm;
            if (c.pt >= tpp) {
                c.pt -= tpp;
                c.bq += 1;
            }
        }
        return c;
    };
    const float* const xg = p.x;
    const unsigned xbytes = (unsigned)p.K0 * hw * 4u;
    // this wave's rows of a tile, HBM -> raw: octets wave, wave + 8; four rows (1 KB) per request.  The WHOLE offset sits in
    // the lane register (the range check sees only that one): rows beyond the input's width and pixels beyond the tensor
    // return nothing.
    // (Inline assembly, not the builtin: the compiler orders every LDS access behind a builtin LDS-DMA with s_waitcnt vmcnt(0),
    // and the vector-memory counter is IN ORDER and counts stores -- the wait for these rows would sit out every store
    // issued since.  The waits are this kernel's own: wait_rows().)
    auto dma = [&](const Cur& c) {
        const unsigned b = c.bq * S + c.s, p0 = c.pt * PC_NT;
        // (the descriptor's four words: base, base[47:32] (stride 0), bytes, the raw-buffer flags of make_buffer_rsrc)
        const uintptr_t xa = reinterpret_cast<uintptr_t>(xg + (size_t)b * p.K0 * hw);
        const u32x4 rxw = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(xa >> 32) & 0xffffu)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)xbytes), 0x00020000u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (this wave's reads of its rows have retired)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0 ? (FIRST_ALL || wave < KO0) : (SECOND_ALL || (SECOND && wave + 8 < KO0))) {
                const int o = wave + 8 * i;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned dst = (unsigned)(uintptr_t)(raw + (8 * o + 4 * j) * PC_NT);      // (LDS byte address)
                    const unsigned vo = xv + ((unsigned)(8 * o + 4 * j) * hw + p0) * 4u;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                                 :: "v"(vo), "s"(rxw), "s"(dst) : "memory");
                }
            }
        }
    };
    auto read_raw = [&](float (&v)[2][8]) {
        const char* src = reinterpret_cast<const char*>(raw) + sraw;
        if (FIRST_ALL || wave < KO0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[0][j] = *reinterpret_cast<const float*>(src + j * PC_NT * 4);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[0][j] = 0.f;
        }
        if (SECOND_ALL || (SECOND && wave + 8 < KO0)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[1][j] = *reinterpret_cast<const float*>(src + (64 + j) * PC_NT * 4);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[1][j] = 0.f;
        }
    };
    auto x_max = [&](unsigned* pm) {
        float v[2][8];
        read_raw(v);
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) m = __builtin_fmaxf(m, __builtin_fabsf(v[i][j]));
        atomicMax(pm + lane, __builtin_bit_cast(unsigned, m));
    };
    auto x_split = [&](const unsigned* pm) {
        float v[2][8];
        read_raw(v);
        const float cx = exp_scale(scale_exp(pm[lane]));
        char* dst = reinterpret_cast<char*>(xp) + sxp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0 ? (FIRST_ALL || wave < KO0) : (SECOND_ALL || (SECOND && wave + 8 < KO0))) {
                u32x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hp, lp;
                    split_pair_v(v[i][2 * j], v[i][2 * j + 1], cx, hp, lp);
                    h[j] = hp;
                    l[j] = lp;
                }
                *reinterpret_cast<u32x4*>(dst + (8 * i) * PC_NT * 16) = h;
                *reinterpret_cast<u32x4*>(dst + (KO0 + 8 * i) * PC_NT * 16) = l;
            }
        }
    };

    if (threadIdx.x < 2 * PC_NT) pmx[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < KP0 * PC_NT; i += 512) raw[i] = 0.f;      // (rows beyond the input's width stay zero)
    Cur cur;
    cur.unit = blockIdx.x;
    cur.s = 0;
    cur.bq = cur.unit / tpp;
    cur.pt = cur.unit % tpp;
    bool valid = cur.unit < p.nunits;
    __syncthreads();
    // the layers' bound coefficients: |y_l| <= rsum[l] * (bound of its input) + bmax[l] (+ max |t| for l = 0)
    float rsum[NL], bmax[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            a = __builtin_fmaxf(a, red[l * 8 + w]);
            c = __builtin_fmaxf(c, red[(NL + l) * 8 + w]);
        }
        // (wave-uniform: scalar registers)
        rsum[l] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a)));
        bmax[l] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c)));
    }
    if (p.t_mode != 0) bmax[0] += __builtin_bit_cast(float, *p.tmax);
    const float* const brow = btab + r0 + 4 * q;       // + 128 l + i
    if (valid) dma(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid) x_max(pmx);
    lds_sync();
    Cur nxt = advance(cur);
    if (valid) {
        x_split(pmx);
        if (nxt.unit < p.nunits) dma(nxt);
    }
    lds_sync();

    // The next tile's rows were requested at the start of phase 1 of the tile before this one; since then this wave has issued
    // that phase's stores (16 + the sign words' one, where layer 1 is stored) and -- three layers -- the last phase's 16: the
    // counter is in order, so "no more outstanding than those" means the rows have landed, without sitting out the stores.
    const int nyounger = (p.y[1] != nullptr ? 16 : 0) + (p.signs[1] != nullptr ? 1 : 0) + (NL == 3 ? 16 : 0);
    bool primed = false;                                // (the first tile's successor was requested in the prologue: nothing younger)
    auto wait_rows = [&]() {
        if (!primed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nyounger >= 33) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
        else if (nyounger >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (nyounger >= 17) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
        else if (nyounger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    float amax_run[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) amax_run[l] = 0.f;
    f32x4 msum[4];                                      // the mean over a pixel's samples (last layer), this wave's block
#pragma unroll
    for (int k = 0; k < 4; ++k) msum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float inv_s = 1.f / (float)S;
    const bool want_mean = p.ymean != nullptr;
    int par = 0;                                        // which half of pmx holds this tile's words
    // (development knob SBMC_PC_TIMING: cycles of wave 0 of workgroup 0 per phase, barrier wait included)
    const bool timing = p.timing != 0 && blockIdx.x == 0 && wave == 0;
    unsigned long long tacc[NL], tlast = 0;
    unsigned ttiles = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) tacc[i] = 0;
    if (timing) tlast = __builtin_amdgcn_s_memtime();

    while (valid) {
        const bool nvalid = nxt.unit < p.nunits;
        const unsigned b = cur.bq * S + cur.s, bq = cur.bq, p0 = cur.pt * PC_NT;
        const bool edge = p0 + PC_NT > hw;              // the plane's last tile: some of its columns are beyond the plane
        unsigned avi[4];                                // this lane's offsets in a [16 rows][hw] plane: row 4 q + i, pixel r (+ 64 k)
#pragma unroll
        for (int i = 0; i < 4; ++i) avi[i] = av + (unsigned)i * hw * 4u;
        float bnd[4];                                   // per pixel block: the bound of the operand the next phase reads
        f32x4 acc[4], hold[4];                          // the products; the layer's finished block

        unrolled<NL>([&](auto lc) {
            constexpr int L = decltype(lc)::value;
            constexpr bool LAST = L + 1 == NL;
            constexpr int KS = L == 0 ? KP0 / 32 : 4;
            constexpr int KO = L == 0 ? KO0 : 16;
            const u32x4* in = L == 0 ? xp : pp + (L - 1) * 2 * 16 * PC_NT;
            const int C = p.cout[L];
            const int nrows = C - r0 < 16 ? (C - r0 > 0 ? C - r0 : 0) : 16;

            if (L == 0) {
                // the next tile: its rows have landed (requested a tile ago) -> its per-pixel maxima
                if (nvalid) {
                    wait_rows();
                    x_max(pmx + (par ^ 1) * PC_NT);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) bnd[k] = __builtin_bit_cast(float, pmx[par * PC_NT + 16 * k + r]);
            }
            if (L == 1) {
                if (threadIdx.x < PC_NT) pmx[par * PC_NT + threadIdx.x] = 0u;   // (this tile's words: read in phase 0; the tile after next's)
                if (nvalid) {
                    // ... its planes, and the requests of the tile after it
                    x_split(pmx + (par ^ 1) * PC_NT);
                    const Cur n2 = advance(nxt);
                    if (n2.unit < p.nunits) dma(n2);
                }
            }
            float iosc[4];                              // 1 / (c_pixel c_w) of this lane's four pixel columns
#pragma unroll
            for (int k = 0; k < 4; ++k) iosc[k] = exp_inverse(scale_exp(__builtin_bit_cast(unsigned, bnd[k]))) * icw[L];
            // the first layer's context term
            float tt[4][4];
            float add[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) add[i] = brow[128 * L + i];
            if (L == 0 && p.t_mode == 2) {
                const rsrc_t rt = rsrc_n(p.t + ((size_t)bq * C + r0) * hw, (unsigned)nrows * hw * 4u);
                if (!edge) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i) tt[k][i] = buf_load(rt, avi[i], p0 * 4u + 64u * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in_k = p0 + 16u * k + r < hw;
#pragma unroll
                        for (int i = 0; i < 4; ++i) tt[k][i] = buf_load(rt, in_k ? avi[i] : PC_OOB, p0 * 4u + 64u * k);
                    }
                }
            } else if (L == 0 && p.t_mode == 1) {
                const rsrc_t rt = rsrc_n(p.t + (size_t)bq * C, (unsigned)C * 4u);
#pragma unroll
                for (int i = 0; i < 4; ++i) add[i] += buf_load(rt, (unsigned)(r0 + 4 * q + i) * 4u, 0);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            const u32x4* bp = in + q * PC_NT + r;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                u32x4 bh[4], bl[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    bh[k] = bp[(4 * s) * PC_NT + 16 * k];
                    bl[k] = bp[(KO + 4 * s) * PC_NT + 16 * k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = mfma16(ah[L][s], bl[k], acc[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = mfma16(al[L][s], bh[k], acc[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = mfma16(ah[L][s], bh[k], acc[k]);
            }
            // ---- the block on the accumulators
            const float slope = p.slope[L];
            float pmax[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = __builtin_fmaf(acc[k][i], iosc[k], add[i]);
                    if (L == 0 && p.t_mode == 2) v += tt[k][i];
                    v = __builtin_fmaxf(v, v * slope);               // slope in [0, 1]: relu, leaky relu, linear alike
                    hold[k][i] = v;
                    m = __builtin_fmaxf(m, __builtin_fabsf(v));
                }
                pmax[k] = m;
            }
            if (!edge) {
#pragma unroll
                for (int k = 0; k < 4; ++k) amax_run[L] = __builtin_fmaxf(amax_run[L], pmax[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    amax_run[L] = __builtin_fmaxf(amax_run[L], p0 + 16u * k + r < hw ? pmax[k] : 0.f);
            }
            if (p.y[L] != nullptr) {
                // Straight from the accumulators: four rows x 64 bytes per instruction, a row's four pieces back to back.  (The
                // pixel block sits in the SCALAR offset: a constant added to the lane offset is not folded into the instruction
                // -- the unsigned sum might wrap -- and cost 16 lane registers of offsets.  Tried and measured slower: the same
                // stores dealt out between the NEXT phase's products from held registers -- a store blocks the wave that issues
                // it wherever it sits, 4.46 -> 4.79 ms --, an fp32 tile in LDS stored 16 bytes per lane, 4.46 <- 4.77, and 16 bytes
                // per lane through a 4 x 4 transpose inside the lane quads (DPP; 16 pieces of 64 bytes per instruction), 4.62 -> 4.87:
                // what a store costs is its 64-byte pieces, not the instruction.)
                const rsrc_t ry = rsrc_n(p.y[L] + ((size_t)b * C + r0) * hw, (unsigned)nrows * hw * 4u);
                if (!edge) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) buf_store(hold[k][i], ry, avi[i], p0 * 4u + 64u * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in_k = p0 + 16u * k + r < hw;
#pragma unroll
                        for (int i = 0; i < 4; ++i) buf_store(hold[k][i], ry, in_k ? avi[i] : PC_OOB, p0 * 4u + 64u * k);
                    }
                }
            }
            if (p.signs[L] != nullptr) {
                // bit (16 g + r) of the ballot of register (k, i) = (value > 0) of row 4 g + i, pixel 16 k + r: the word of
                // row 4 g + i and pixel half h packs the g-th 16 bits of the ballots of blocks 2 h and 2 h + 1; lane j / j + 32
                // of `myword` keeps row j's two words
                unsigned myword = 0;
                unrolled<4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const unsigned long long b0 = __ballot(hold[0][i] > 0.f), b1 = __ballot(hold[1][i] > 0.f);
                    const unsigned long long b2 = __ballot(hold[2][i] > 0.f), b3 = __ballot(hold[3][i] > 0.f);
                    const unsigned l0 = (unsigned)b0, h0 = (unsigned)(b0 >> 32), l1 = (unsigned)b1, h1 = (unsigned)(b1 >> 32);
                    const unsigned l2 = (unsigned)b2, h2 = (unsigned)(b2 >> 32), l3 = (unsigned)b3, h3 = (unsigned)(b3 >> 32);
                    // rows i, 4 + i, 8 + i, 12 + i: first word (blocks 0, 1), second word (blocks 2, 3)
                    const unsigned w00 = (l0 & 0xffffu) | (l1 << 16), w01 = (l0 >> 16) | (l1 & 0xffff0000u);
                    const unsigned w02 = (h0 & 0xffffu) | (h1 << 16), w03 = (h0 >> 16) | (h1 & 0xffff0000u);
                    const unsigned w10 = (l2 & 0xffffu) | (l3 << 16), w11 = (l2 >> 16) | (l3 & 0xffff0000u);
                    const unsigned w12 = (h2 & 0xffffu) | (h3 << 16), w13 = (h2 >> 16) | (h3 & 0xffff0000u);
                    put_words8<i>(myword, w00, w01, w02, w03, w10, w11, w12, w13);
                });
                const rsrc_t rs = rsrc_n(p.signs[L] + ((size_t)b * C + r0) * wpr, (unsigned)nrows * wpr * 4u);
                // (row beyond the layer's width: beyond the descriptor; the tile's second word beyond the plane: switched off)
                const unsigned off = (p0 + 32u * (lane >> 5) < hw) ? sgv : PC_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(myword, rs, off, (p0 / 32) * 4u, 0);
            }
            if constexpr (!LAST) {
                // the block's two planes under the per-pixel bound's scale: this lane's 4 consecutive channels of a pixel = half an entry
                char* dst = reinterpret_cast<char*>(pp + L * 2 * 16 * PC_NT) + ppw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    bnd[k] = __builtin_fmaf(rsum[L], bnd[k], bmax[L]);
                    const float cy = exp_scale(scale_exp(__builtin_bit_cast(unsigned, bnd[k])));
                    unsigned h0, l0, h1, l1;
                    split_pair_v(hold[k][0], hold[k][1], cy, h0, l0);
                    split_pair_v(hold[k][2], hold[k][3], cy, h1, l1);
                    *reinterpret_cast<u32x2*>(dst + 16 * k * 16) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(dst + (16 * PC_NT + 16 * k) * 16) = u32x2{l0, l1};
                }
            } else if (want_mean) {
                const bool first_s = cur.s == 0;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) msum[k][i] = (first_s ? 0.f : msum[k][i]) + hold[k][i];
                if (cur.s + 1 == S) {
                    const rsrc_t rm = rsrc_n(p.ymean + ((size_t)bq * C + r0) * hw, (unsigned)nrows * hw * 4u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in_k = !edge || p0 + 16u * k + r < hw;
#pragma unroll
                        for (int i = 0; i < 4; ++i) buf_store(msum[k][i] * inv_s, rm, in_k ? avi[i] : PC_OOB, p0 * 4u + 64u * k);
                    }
                }
            }
            lds_sync();
            if (timing) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                tacc[L] += now - tlast;
                tlast = now;
            }
        });
        ttiles += 1;
        primed = true;

        cur = nxt;
        nxt = advance(nxt);
        valid = nvalid;
        par ^= 1;
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        // (amax_publish gathers the waves' maxima in ONE shared array: the next layer's must not land in it while thread 0
        // still reads this layer's -- without this barrier layer l's word now and then carried a wave's maximum of layer
        // l + 1, larger or SMALLER than its own; tools/fuzz_pointwise_chain.py found it)
        if (l > 0) __syncthreads();
        if (p.amax[l] != nullptr) amax_publish(__builtin_bit_cast(unsigned, amax_run[l]), p.amax[l]);
    }
    if (timing && lane == 0) {
        float* out = p.y[NL - 1];
        out[0] = (float)ttiles;
#pragma unroll
        for (int i = 0; i < NL; ++i) out[1 + i] = (float)tacc[i];
    }
}

// ---------------------------------------------------------------------------------------------
// The WIDE layer's forward (128 < cout <= 512: the 441-channel logits, reference sbmc/models.py:98-102) on the chain
// kernel's machinery: the input tile is staged ONCE per 64 pixels (LDS-DMA a tile ahead, per-pixel scale) for all row
// tiles -- csrc/pointwise.hip pw_fwd_s_kernel gives every 128-row tile a workgroup of its own, each staging the tile again
// through L2 and its own registers --, a wave owns 16 rows of EACH of the four row tiles (their weights in registers: 4 x 32),
// and what is left is the 13 GB of stores: 16 per wave and row tile, issued behind the tile's products.
struct PwWideFwdParams {
    const float* x;              // [B, K0, hw]
    const float* w;              // [C, K0]
    const float* bias;           // [C]
    float* y;                    // [B, C, hw]
    unsigned* amax;              // raised to the bit pattern of max |y|, or nullptr
    float slope;                 // 1: linear, 0: relu, else leaky relu
    int B, K0, C;
    unsigned hw, tiles_per_plane, nunits;
};

template <int KP0>
__global__ __launch_bounds__(512) void pw_wide_fwd_kernel(PwWideFwdParams p) {
    constexpr int NRT = 4;
    constexpr int KO0 = KP0 / 8;
    constexpr bool FIRST_ALL = KO0 >= 8, SECOND = KO0 > 8, SECOND_ALL = KO0 == 16;
    constexpr int KS = KP0 / 32;
    extern __shared__ float4 pc_lds[];
    float* raw = reinterpret_cast<float*>(pc_lds);                            // [KP0][64]
    u32x4* xp = reinterpret_cast<u32x4*>(raw + KP0 * PC_NT);                  // [2 stages][2 planes][KO0][64]
    unsigned* pmx = reinterpret_cast<unsigned*>(xp + 2 * 2 * KO0 * PC_NT);    // [2][64]
    float* btab = reinterpret_cast<float*>(pmx + 2 * PC_NT);                  // [512]
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int r = lane & 15, q = lane >> 4;
    const unsigned hw = p.hw;
    const unsigned G = gridDim.x;
    const int r0 = 16 * wave;

    // a*[t][s] = planes of W[128 t + 16 wave + lane % 16][32 s + 8 (lane / 16) + 0..7], under the scale of the wave's own rows
    u32x4 ah[NRT][4], al[NRT][4];
    float icw[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
        const rsrc_t rw = rsrc_n(p.w, (unsigned)(p.C * p.K0) * 4u);
        const int row = 128 * t + r0 + r;
        float v[4][8];
        float wm = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = 32 * s + 8 * q + i;
                v[s][i] = buf_load(rw, (row < p.C && k < p.K0) ? (unsigned)(row * p.K0 + k) * 4u : PC_OOB, 0);
                wm = __builtin_fmaxf(wm, __builtin_fabsf(v[s][i]));
            }
        }
        const unsigned ew = scale_exp((unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u(__builtin_bit_cast(unsigned, wm))));
        const float cw = exp_scale(ew);
        icw[t] = exp_inverse(ew);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hp, lp;
                f16_split_pair(v[s][2 * j], v[s][2 * j + 1], cw, hp, lp);
                ah[t][s][j] = hp;
                al[t][s][j] = lp;
            }
        }
    }
    {
        const rsrc_t rb = rsrc_n(p.bias, (unsigned)p.C * 4u);
        btab[threadIdx.x] = buf_load(rb, threadIdx.x * 4u, 0);                // (rows >= C: 0)
    }

    const unsigned sraw = (unsigned)((8 * wave * PC_NT + lane) * 4);
    const unsigned sxp = (unsigned)((wave * PC_NT + lane) * 16);
    const unsigned xv = ((unsigned)q * hw + 4u * r) * 4u;
    const unsigned av = ((unsigned)(4 * q) * hw + (unsigned)r) * 4u;

    struct Cur { unsigned unit, b, pt; };
    const unsigned tpp = p.tiles_per_plane, Gd = G / tpp, Gm = G % tpp;
    auto advance = [&](Cur c) -> Cur {
        c.unit += G;
        c.b += Gd;
        c.pt += Gm;
        if (c.pt >= tpp) {
            c.pt -= tpp;
            c.b += 1;
        }
        return c;
    };
    const unsigned xbytes = (unsigned)p.K0 * hw * 4u;
    auto dma = [&](const Cur& c) {
        const unsigned p0 = c.pt * PC_NT;
        const uintptr_t xa = reinterpret_cast<uintptr_t>(p.x + (size_t)c.b * p.K0 * hw);
        const u32x4 rxw = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(xa >> 32) & 0xffffu)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)xbytes), 0x00020000u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0 ? (FIRST_ALL || wave < KO0) : (SECOND_ALL || (SECOND && wave + 8 < KO0))) {
                const int o = wave + 8 * i;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned dst = (unsigned)(uintptr_t)(raw + (8 * o + 4 * j) * PC_NT);
                    const unsigned vo = xv + ((unsigned)(8 * o + 4 * j) * hw + p0) * 4u;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                                 :: "v"(vo), "s"(rxw), "s"(dst) : "memory");
                }
            }
        }
    };
    auto read_raw = [&](float (&v)[2][8]) {
        const char* src = reinterpret_cast<const char*>(raw) + sraw;
        if (FIRST_ALL || wave < KO0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[0][j] = *reinterpret_cast<const float*>(src + j * PC_NT * 4);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[0][j] = 0.f;
        }
        if (SECOND_ALL || (SECOND && wave + 8 < KO0)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[1][j] = *reinterpret_cast<const float*>(src + (64 + j) * PC_NT * 4);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[1][j] = 0.f;
        }
    };
    auto x_max = [&](unsigned* pm) {
        float v[2][8];
        read_raw(v);
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) m = __builtin_fmaxf(m, __builtin_fabsf(v[i][j]));
        atomicMax(pm + lane, __builtin_bit_cast(unsigned, m));
    };
    auto x_split = [&](const unsigned* pm, int stage) {
        float v[2][8];
        read_raw(v);
        const float cx = exp_scale(scale_exp(pm[lane]));
        char* dst = reinterpret_cast<char*>(xp + stage * 2 * KO0 * PC_NT) + sxp;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i == 0 ? (FIRST_ALL || wave < KO0) : (SECOND_ALL || (SECOND && wave + 8 < KO0))) {
                u32x4 h, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hp, lp;
                    split_pair_v(v[i][2 * j], v[i][2 * j + 1], cx, hp, lp);
                    h[j] = hp;
                    l[j] = lp;
                }
                *reinterpret_cast<u32x4*>(dst + (8 * i) * PC_NT * 16) = h;
                *reinterpret_cast<u32x4*>(dst + (KO0 + 8 * i) * PC_NT * 16) = l;
            }
        }
    };

    if (threadIdx.x < 2 * PC_NT) pmx[threadIdx.x] = 0u;
    for (int i = threadIdx.x; i < KP0 * PC_NT; i += 512) raw[i] = 0.f;
    Cur cur;
    cur.unit = blockIdx.x;
    cur.b = cur.unit / tpp;
    cur.pt = cur.unit % tpp;
    bool valid = cur.unit < p.nunits;
    __syncthreads();
    if (valid) dma(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (valid) x_max(pmx);
    lds_sync();
    Cur nxt = advance(cur);
    if (valid) {
        x_split(pmx, 0);
        if (nxt.unit < p.nunits) dma(nxt);
    }
    lds_sync();

    float amax_run = 0.f;
    int par = 0;
    bool primed = false;
    const float* const brow = btab + r0 + 4 * q;
    const float slope = p.slope;
    while (valid) {
        const bool nvalid = nxt.unit < p.nunits;
        const unsigned b = cur.b, p0 = cur.pt * PC_NT;
        const bool edge = p0 + PC_NT > hw;
        unsigned avi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) avi[i] = av + (unsigned)i * hw * 4u;
        float ics[4];                                   // 1 / c_pixel of this lane's four pixel columns
#pragma unroll
        for (int k = 0; k < 4; ++k) ics[k] = exp_inverse(scale_exp(pmx[par * PC_NT + 16 * k + r]));
        const u32x4* bp = xp + par * 2 * KO0 * PC_NT + q * PC_NT + r;

        unrolled<NRT>([&](auto tc) {
            constexpr int T = decltype(tc)::value;
            if (T == 0 && nvalid) {
                // the next tile's rows (requested at row tile 2 of the tile before: the stores of row tiles 2 and 3 have been
                // issued since, 16 each where the layer has those rows)
                if (primed && p.C > 384) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else if (primed && p.C > 256) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                x_max(pmx + (par ^ 1) * PC_NT);
            }
            if (T == 1) lds_sync();                     // (the next tile's words are complete)
            if (T == 2 && nvalid) {
                x_split(pmx + (par ^ 1) * PC_NT, par ^ 1);
                const Cur n2 = advance(nxt);
                if (n2.unit < p.nunits) dma(n2);
            }
            if (128 * T < p.C) {
                f32x4 acc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    u32x4 bh[4], bl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        bh[k] = bp[(4 * s) * PC_NT + 16 * k];
                        bl[k] = bp[(KO0 + 4 * s) * PC_NT + 16 * k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = mfma16(ah[T][s], bl[k], acc[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = mfma16(al[T][s], bh[k], acc[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = mfma16(ah[T][s], bh[k], acc[k]);
                }
                const int rt0 = 128 * T + r0;
                const int nrows = p.C - rt0 < 16 ? (p.C - rt0 > 0 ? p.C - rt0 : 0) : 16;
                const rsrc_t ry = rsrc_n(p.y + ((size_t)b * p.C + rt0) * hw, (unsigned)nrows * hw * 4u);
                float add[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) add[i] = brow[128 * T + i];
                float m = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float io = ics[k] * icw[T];
                    float mk = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = __builtin_fmaf(acc[k][i], io, add[i]);
                        v = __builtin_fmaxf(v, v * slope);
                        acc[k][i] = v;
                        // (rows beyond the layer's width are zero weights and zero bias: 0)
                        mk = __builtin_fmaxf(mk, __builtin_fabsf(v));
                    }
                    m = __builtin_fmaxf(m, (!edge || p0 + 16u * k + r < hw) ? mk : 0.f);
                }
                amax_run = __builtin_fmaxf(amax_run, m);
                if (!edge) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) buf_store(acc[k][i], ry, avi[i], p0 * 4u + 64u * k);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool in_k = p0 + 16u * k + r < hw;
#pragma unroll
                        for (int i = 0; i < 4; ++i) buf_store(acc[k][i], ry, in_k ? avi[i] : PC_OOB, p0 * 4u + 64u * k);
                    }
                }
            }
        });
        if (threadIdx.x < PC_NT) pmx[par * PC_NT + threadIdx.x] = 0u;      // (this tile's words: read above; the tile after next's)
        lds_sync();                                     // (the next tile's planes are complete; this tile's are free)
        primed = true;
        cur = nxt;
        nxt = advance(nxt);
        valid = nvalid;
        par ^= 1;
    }
    if (p.amax != nullptr) amax_publish(__builtin_bit_cast(unsigned, amax_run), p.amax);
}

}  // namespace sbmc

using namespace sbmc;

// Layers the fused chain takes: 2 or 3, every width <= 128, the plane a multiple of 4 pixels (16-byte rows), offsets in 32 bits.
extern "C" int sbmc_pointwise_chain_supported(int cin, int nl, const int* cout, long hw) {
    if (nl < 2 || nl > PC_MAXL || !cout || cin < 1 || cin > 128 || hw < 4 || hw % 4 || hw >= (1L << 27)) return 0;
    for (int l = 0; l < nl; ++l)
        if (cout[l] < 1 || cout[l] > 128) return 0;
    return (double)128 * (double)hw * 4.0 < 4294967000.0 ? 1 : 0;
}

extern "C" int sbmc_pointwise_chain_fwd_f32(const float* x, const float* t, const unsigned* tmax, const float* const* w, const float* const* bias,
                                            float* const* y, unsigned* const* signs, unsigned* const* amax, float* ymean,
                                            int nl, const int* cout, const int* act, const float* slope, int b, int s, int cin,
                                            long hw, int t_mode, void* stream) {
    if (b < 0 || s < 1 || t_mode < 0 || t_mode > 2 || !cout || !act || !slope || !w || !bias || !y) return SBMC_HIP_EINVAL;
    if (!sbmc_pointwise_chain_supported(cin, nl, cout, hw)) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (b % s || !x || (t_mode && (!t || !tmax)) || !y[nl - 1] || (uintptr_t)x % 16) return SBMC_HIP_EINVAL;
    PwChainParams p;
    memset(&p, 0, sizeof(p));
    p.x = x;
    p.t = t;
    p.tmax = tmax;
    p.ymean = ymean;
    for (int l = 0; l < nl; ++l) {
        if (!w[l] || !bias[l] || act[l] < 0 || act[l] > 2 || (uintptr_t)y[l] % 4) return SBMC_HIP_EINVAL;
        p.w[l] = w[l];
        p.bias[l] = bias[l];
        p.y[l] = y[l];
        p.signs[l] = signs ? signs[l] : nullptr;
        p.amax[l] = amax ? amax[l] : nullptr;
        if (p.signs[l] && !p.y[l]) return SBMC_HIP_EINVAL;           // (the sign words are taken from the stored tile)
        p.slope[l] = act[l] == 0 ? 1.f : (act[l] == 1 ? 0.f : slope[l]);
        p.cout[l] = cout[l];
    }
    // (the walk takes a pixel's samples one after the other where something per-pixel is shared between them)
    p.B = b; p.S = (t_mode || ymean) ? s : 1; p.K0 = cin; p.t_mode = t_mode;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + PC_NT - 1) / PC_NT);
    const unsigned long long nunits = (unsigned long long)p.tiles_per_plane * (unsigned)(b / p.S);
    if (nunits > 0xFFFFFFFFull - 65536) return SBMC_HIP_EINVAL;
    p.nunits = (unsigned)nunits;
    p.timing = env_knob("SBMC_PC_TIMING", 0);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const unsigned grid = nunits < (unsigned long long)cus ? (unsigned)nunits : (unsigned)cus;
    const int kp = (cin + 31) / 32 * 32;
    const size_t lds = (size_t)kp * PC_NT * 4 + (size_t)2 * (kp / 8) * PC_NT * 16 + (size_t)(nl - 1) * 2 * 16 * PC_NT * 16 +
                       (size_t)2 * PC_NT * 4 + (size_t)2 * PC_MAXL * 8 * 4 + (size_t)PC_MAXL * 128 * 4;
    hipError_t e = hipSuccess;
#define SBMC_PCH(KPV, NLV)                                                                               \
    do {                                                                                                 \
        auto kern = pw_chain_fwd_kernel<KPV, NLV>;                                                       \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, p); \
    } while (0)
#define SBMC_PCH_K(NLV)                                                                                  \
    switch (kp) {                                                                                        \
        case 32: SBMC_PCH(32, NLV); break;                                                               \
        case 64: SBMC_PCH(64, NLV); break;                                                               \
        case 96: SBMC_PCH(96, NLV); break;                                                               \
        default: SBMC_PCH(128, NLV); break;                                                              \
    }
    if (nl == 2) { SBMC_PCH_K(2) } else { SBMC_PCH_K(3) }
#undef SBMC_PCH_K
#undef SBMC_PCH
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return (int)hipGetLastError();
}

extern "C" int sbmc_pointwise_wide_fwd_supported(int cin, int cout, long hw) {
    return (cin >= 1 && cin <= 128 && cout > 128 && cout <= 512 && hw >= 4 && hw % 4 == 0 && hw < (1L << 27) &&
            (double)128 * (double)hw * 4.0 < 4294967000.0) ? 1 : 0;
}

// y[b] = act(w x[b] + bias), 128 < cout <= 512 (pw_wide_fwd_kernel); amax (or NULL): a zeroed word raised to max |y|.
extern "C" int sbmc_pointwise_wide_fwd_f32(const float* x, const float* w, const float* bias, float* y, unsigned* amax, int b,
                                           int cin, int cout, long hw, int act, float slope, void* stream) {
    if (b < 0 || act < 0 || act > 2) return SBMC_HIP_EINVAL;
    if (!sbmc_pointwise_wide_fwd_supported(cin, cout, hw)) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (!x || !w || !bias || !y || (uintptr_t)x % 16 || (uintptr_t)y % 4) return SBMC_HIP_EINVAL;
    PwWideFwdParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.amax = amax;
    p.slope = act == 0 ? 1.f : (act == 1 ? 0.f : slope);
    p.B = b; p.K0 = cin; p.C = cout;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + PC_NT - 1) / PC_NT);
    const unsigned long long nunits = (unsigned long long)p.tiles_per_plane * (unsigned)b;
    if (nunits > 0xFFFFFFFFull - 65536) return SBMC_HIP_EINVAL;
    p.nunits = (unsigned)nunits;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const unsigned grid = nunits < (unsigned long long)cus ? (unsigned)nunits : (unsigned)cus;
    const int kp = (cin + 31) / 32 * 32;
    const size_t lds = (size_t)kp * PC_NT * 4 + (size_t)2 * 2 * (kp / 8) * PC_NT * 16 + (size_t)2 * PC_NT * 4 + (size_t)512 * 4;
    hipError_t e = hipSuccess;
#define SBMC_PWF(KPV)                                                                                    \
    do {                                                                                                 \
        auto kern = pw_wide_fwd_kernel<KPV>;                                                             \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, p); \
    } while (0)
    switch (kp) {
        case 32: SBMC_PWF(32); break;
        case 64: SBMC_PWF(64); break;
        case 96: SBMC_PWF(96); break;
        default: SBMC_PWF(128); break;
    }
#undef SBMC_PWF
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return (int)hipGetLastError();
}

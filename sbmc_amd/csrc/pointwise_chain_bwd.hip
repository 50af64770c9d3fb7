// The backward of TWO consecutive per-sample 1x1 layers in one pass (round 6).
//
// Forward (reference sbmc/modules.py:154-175 as built at sbmc/models.py:79-102):  y_A = act_A(W_A x_A + b_A + t),
// y_B = act_B(W_B y_A + b_B), both 128 channels wide.  Layer by layer (csrc/pointwise.hip pw_bwd_kernel) the gradient of
// y_A -- 3.77 GB at 720p x 8 spp -- is written by the upper layer's pass and read back by the lower one's, and y_A's sign
// words with it.  Here a 64-pixel tile of that gradient never leaves the chip:
//     gz_B = gy_B * act_B'            (sign words of y_B)          gw_B += gz_B x_B^T,  gb_B += rows of gz_B     (x_B = y_A)
//     gy_A = W_B^T gz_B               (accumulators only)
//     gz_A = gy_A * act_A'            (act_A' from x_B itself: y_A > 0)   gw_A += gz_A x_A^T,  gb_A, gt += gz_A
//     gx_A = W_A^T gz_A               -> HBM
// Reads per tile: gy_B, x_B, x_A (+ 1 KB of sign words); writes: gx_A.  Seven activation moves of the two separate passes
// become four.
//
// Number format: the 3 x 3 kernels' (two f16 planes under a power-of-two scale, three of four partial products, fp32
// accumulation).  The weight-gradient products reduce over PIXELS, so their operands' scales must be the same for every
// tile: gz_B from the word of gy_B, x_B / x_A from their words (the forward passes left them), gz_A from a BOUND -- the
// largest absolute column sum of W_B times the word of gy_B.
//
// Everything a tile needs travels HBM -> LDS by LDS-DMA issued through inline assembly, and the waits are this kernel's own
// s_waitcnt vmcnt(N): the vector-memory counter is in order and counts stores, a wait for "everything" would sit out the 16
// stores of the tile before (csrc/pointwise_chain.hip found that to be the difference between its training and inference
// forms), and the compiler orders every LDS access behind a builtin LDS-DMA with exactly that wait.
//
// LDS: four 32 KB buffers.  A raw fp32 tile [128 rows][64 pixels] is converted IN PLACE into its two f16 planes by the
// wave that requested it (a row's 256 bytes become plane 0 | plane 1, 128 bytes each): with a pitch of 256 bytes every row
// starts in the same bank, so the 16-byte chunk of (plane p, pixels 8 c ..) of row r sits at slot (8 p + c) ^ (r & 15) --
// the operand reads of sixteen consecutive rows (ds_read_b128) and the transposing reads (ds_read_b64_tr_b16) then touch
// every bank once.  G0 / G1: gy_B -> gz_B -> gz_A of alternate tiles; XB: x_B; XA: x_A.
#include "common.hpp"
#include <cstring>
#include <type_traits>
#include <utility>
#include "../../include/sbmc_hip.h"

namespace sbmc {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using hf8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr unsigned CB_OOB = 0xFFFFFFF0u;
constexpr int CB_NT = 64;
constexpr unsigned CB_BUF = 32768;                      // bytes per buffer
constexpr unsigned CB_G0 = 0, CB_XB = 2 * CB_BUF, CB_XA = 3 * CB_BUF;
constexpr unsigned CB_MSK = 4 * CB_BUF;                 // [128 rows][16]: one nibble per (row of x_B, pixel quad): x_B > 0
constexpr unsigned CB_SGN = CB_MSK + 2048;              // [2 stages][8 waves][64] sign words of y_B: the rows a wave commits (32 used)
constexpr unsigned CB_RED = CB_SGN + 4096;              // [16] floats (prologue)
constexpr unsigned CB_LDS = CB_RED + 64;

__device__ __forceinline__ rsrc_t cb_rsrc(const void* base, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* u = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// the four words of a raw-buffer descriptor (for the inline-assembly requests)
__device__ __forceinline__ u32x4 cb_desc(const void* base, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                 (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf8, a), __builtin_bit_cast(hf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, a), __builtin_bit_cast(hf8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void cb_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float wave_max_f(float m) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, s, 64));
    return m;
}
// 16 bytes per lane, HBM -> LDS at `lds_byte` + 16 lane (a wave's 1 KB), no register in between
__device__ __forceinline__ void dma16(unsigned voffset, u32x4 desc, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voffset), "s"(desc), "s"(lds_byte) : "memory");
}
// 4 bytes per lane (a wave's 256 bytes)
__device__ __forceinline__ void dma4(unsigned voffset, u32x4 desc, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds" :: "v"(voffset), "s"(desc), "s"(lds_byte) : "memory");
}
// at most n of this wave's vector-memory operations still outstanding (n: one of the values this kernel uses)
__device__ __forceinline__ void wait_vm(int n) {
    if (n >= 25) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 19) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// 4 floats under a (wave-uniform) power-of-two scale -> their two f16 planes
__device__ __forceinline__ void split4(float4 v, float c, u32x2& h, u32x2& l) {
    unsigned h0, l0, h1, l1;
    f16_split_pair(v.x, v.y, c, h0, l0);
    f16_split_pair(v.z, v.w, c, h1, l1);
    h = u32x2{h0, h1};
    l = u32x2{l0, l1};
}

}  // namespace

struct PwChainBwdParams {
    const float* gy;             // [B, 128, hw] gradient of y_B
    const unsigned* signs_b;     // [B, 128, ceil(hw / 32)] sign words of y_B (nullptr: layer B is linear)
    const float* xb;             // [B, 128, hw] y_A
    const float* wb;             // [128, 128]
    const float* xa;             // [B, K0, hw]
    const float* wa;             // [128, K0]
    float* gxa;                  // [B, K0, hw] or nullptr
    float* gwp_b;                // [G, 128, 128]
    float* gbp_b;                // [G, 128]
    float* gwp_a;                // [G, 128, K0]
    float* gbp_a;                // [G, Bq, 128] (zeroed by the host)
    float* gt;                   // [B / S, 128, hw] (t_mode 2) or nullptr
    const unsigned* gmax;        // device words: bit patterns of floats >= max |gy|, max |x_B|, max |x_A|
    const unsigned* xbmax;
    const unsigned* xamax;
    unsigned* gxmax;             // raised to max |gx_A|, or nullptr
    float slope_b, slope_a;      // 1: linear, 0: relu, else leaky relu
    int B, S, K0, Bq, t_mode;
    unsigned hw, tiles_per_plane, nunits;
};

// KPA: K0 rounded up to a multiple of 32; TPIX: a per-pixel context gradient (t_mode 2) is wanted -- 16 registers.
template <int KPA, bool TPIX>
__global__ __launch_bounds__(512) void pw_chain_bwd_kernel(PwChainBwdParams p) {
    constexpr int NXA = KPA / 32;                       // requests per wave for x_A
    constexpr int NBA = KPA / 32;                       // 32-column blocks of gw_A
    extern __shared__ float4 cb_lds[];
    char* const lds = reinterpret_cast<char*>(cb_lds);
    const unsigned lbase = (unsigned)(uintptr_t)lds;    // LDS byte address of the dynamic allocation
    const int lane = threadIdx.x & 63, wave = wave_id();
    const int rb = wave & 3, ph = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int l15 = lane & 15, q = lane >> 4;
    const unsigned hw = p.hw, wpr = (hw + 31) / 32;
    const unsigned G = gridDim.x, g = blockIdx.x, S = (unsigned)p.S;
    const bool dx = p.gxa != nullptr;
    const bool maskb = p.signs_b != nullptr;

    // ---- scales
    const float gmaxf = __builtin_bit_cast(float, *p.gmax);
    const float cgB = pow2_scale_of(*p.gmax), cxB = pow2_scale_of(*p.xbmax), cxA = pow2_scale_of(*p.xamax);

    // ---- the weights as operands of the two data-gradient products: planes of w[co][kr], kr = 16 wave + lane % 16, eight output
    // channels co per lane and 32-step in the order the transposing reads deliver gz (csrc/pointwise.hip, GXS):
    // co = 32 st + 16 (g / 2) + 8 (g % 2) + 2 (e % 4) + e / 4, g = lane / 16
    u32x4 wBh[4], wBl[4], wAh[4], wAl[4];
    float icwB, icwA = 0.f;
    float colsum = 0.f;                                 // this lane's part of column kr's absolute sum of W_B
    {
        const rsrc_t rw = cb_rsrc(p.wb, 128u * 128u * 4u);
        const int kr = 16 * wave + l15;
        float v[4][8];
        float wm = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = 32 * st + 16 * (q >> 1) + 8 * (q & 1) + 2 * (e & 3) + (e >> 2);
                v[st][e] = buf_load(rw, (unsigned)(co * 128 + kr) * 4u, 0);
                wm = __builtin_fmaxf(wm, __builtin_fabsf(v[st][e]));
                colsum += __builtin_fabsf(v[st][e]);
            }
        const float cw = pow2_scale_of(__builtin_bit_cast(unsigned, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
            __builtin_bit_cast(int, wave_max_f(wm))))));
        icwB = 1.f / cw;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hp, lp;
                f16_split_pair(v[st][2 * j], v[st][2 * j + 1], cw, hp, lp);
                wBh[st][j] = hp;
                wBl[st][j] = lp;
            }
    }
    if (dx) {
        const rsrc_t rw = cb_rsrc(p.wa, (unsigned)(128 * p.K0) * 4u);
        const int kr = 16 * wave + l15;
        float v[4][8];
        float wm = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = 32 * st + 16 * (q >> 1) + 8 * (q & 1) + 2 * (e & 3) + (e >> 2);
                v[st][e] = buf_load(rw, kr < p.K0 ? (unsigned)(co * p.K0 + kr) * 4u : CB_OOB, 0);
                wm = __builtin_fmaxf(wm, __builtin_fabsf(v[st][e]));
            }
        const float cw = pow2_scale_of(__builtin_bit_cast(unsigned, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
            __builtin_bit_cast(int, wave_max_f(wm))))));
        icwA = 1.f / cw;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hp, lp;
                f16_split_pair(v[st][2 * j], v[st][2 * j + 1], cw, hp, lp);
                wAh[st][j] = hp;
                wAl[st][j] = lp;
            }
    }
    // the bound of gy_A: (largest absolute column sum of W_B) x max |gy_B| -- a column's sum: over the four lanes that share it
    colsum += __shfl_xor(colsum, 16, 64);
    colsum += __shfl_xor(colsum, 32, 64);
    {
        float* red = reinterpret_cast<float*>(lds + CB_RED);
        const float m = wave_max_f(colsum);
        if (lane == 0) red[wave] = m;
    }
    // zero everything once: rows of x_A beyond the layer's input width stay zero planes, columns beyond the plane zero
    for (unsigned i = threadIdx.x; i < CB_SGN / 16; i += 512) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    float cgA, osc2, osc4;
    {
        const float* red = reinterpret_cast<const float*>(lds + CB_RED);
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) m = __builtin_fmaxf(m, red[w]);
        const float bound = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m * gmaxf)));
        cgA = pow2_scale_of(__builtin_bit_cast(unsigned, bound));
        osc2 = icwB * (1.f / cgB);
        osc4 = icwA * (1.f / cgA);
    }

    // ---- lane offsets
    // request / commit role: row tid / 16 + 32 i (= 4 wave + lane / 16: the rows this wave requests), pixels 4 (tid % 16) ..
    const unsigned c16 = threadIdx.x & 15, srow = threadIdx.x >> 4;
    const unsigned xv = (unsigned)((q * hw + 4u * l15) * 4u);               // + ((32 i + 4 wave) hw + p0) 4
    const unsigned craw = srow * 256u + c16 * 16u;                           // raw quad of row srow (+ 8192 i)
    // this thread's 8 bytes of plane 0 of row srow + 32 i (the row's swizzle is the same for every i: 32 i % 16 == 0)
    const unsigned cpl = srow * 256u + ((((c16 >> 1)) ^ (srow & 15u)) << 4) + 8u * (c16 & 1u);      // plane 1: ^ 128
    // weight-gradient products: A rows 32 rb + l31, B rows 64 ph + 32 n + l31; chunk 2 s + lhi of plane p: ^ ((8 p + 2 s) << 4)
    const unsigned gwa = (unsigned)((32 * rb + l31) * 256 + ((((32 * rb + l31) & 15) ^ lhi) << 4));
    const unsigned gwb = (unsigned)((64 * ph + l31) * 256 + ((((64 * ph + l31) & 15) ^ lhi) << 4));        // + 8192 n
    // transposing reads (gz as the reduction operand of a data-gradient product): even row 16 (lane / 32) + 8 ((lane / 16) % 2) +
    // 2 ((lane % 16) / 4) of a 32-step, pixels 4 (lane % 4) .. of a 16-pixel block; the odd row below it
    const unsigned trow = (unsigned)(16 * (lane >> 5) + 8 * ((lane >> 4) & 1) + 2 * (l15 >> 2));
    const unsigned tcl = (unsigned)((lane & 3) >> 1), thb = (unsigned)(lane & 1);
    const unsigned tr0 = trow * 256u + ((tcl ^ (trow & 15u)) << 4) + 8u * thb;
    const unsigned tr1 = (trow + 1u) * 256u + ((tcl ^ ((trow + 1u) & 15u)) << 4) + 8u * thb;
    // gz_A as it leaves the upper layer's data gradient (transposed product): channel 16 wave + lane % 16, pixels 16 pb + 4 q ..
    const unsigned kA = (unsigned)(16 * wave + l15);
    const unsigned gzw = kA * 256u + 8u * (unsigned)(q & 1);                 // + (((8 p + 2 pb + q / 2) ^ (kA % 16)) << 4)
    const unsigned mskr = CB_MSK + kA * 16u + (unsigned)q;                   // + 4 pb: this lane's mask nibble
    const unsigned mskw = CB_MSK + srow * 16u + c16;                         // + 512 i: the nibble this thread writes

    // ---- the walk
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
    const unsigned abytes = 128u * hw * 4u, kbytes = (unsigned)p.K0 * hw * 4u;
    auto issue_g = [&](const Cur& c, int stage) {       // gy_B -> G[stage]; this wave's sign words -> SGN[stage][wave]
        const unsigned b = c.bq * S + c.s, p0 = c.pt * CB_NT;
        const u32x4 d = cb_desc(p.gy + (size_t)b * 128 * hw, abytes);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dma16(xv + ((unsigned)(32 * i + 4 * wave) * hw + p0) * 4u, d, lbase + CB_G0 + stage * CB_BUF + (32 * i + 4 * wave) * 256);
        if (maskb) {
            // lane L < 32: row 32 (L / 8) + 4 wave + (L / 2) % 4, word L % 2 of the tile's two
            const u32x4 ds = cb_desc(p.signs_b + (size_t)b * 128 * wpr, 128u * wpr * 4u);
            const unsigned row = 32u * (unsigned)(lane >> 3) + 4u * wave + (unsigned)((lane >> 1) & 3);
            const unsigned word = p0 / 32 + (unsigned)(lane & 1);
            dma4((lane < 32 && word < wpr) ? (row * wpr + word) * 4u : CB_OOB, ds, lbase + CB_SGN + (stage * 8 + wave) * 256);
        } else {
            dma4(CB_OOB, cb_desc(p.gy, 0u), lbase + CB_SGN + (stage * 8 + wave) * 256);     // (the count of requests stays the same)
        }
    };
    auto issue_xb = [&](const Cur& c) {
        const unsigned b = c.bq * S + c.s, p0 = c.pt * CB_NT;
        const u32x4 d = cb_desc(p.xb + (size_t)b * 128 * hw, abytes);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dma16(xv + ((unsigned)(32 * i + 4 * wave) * hw + p0) * 4u, d, lbase + CB_XB + (32 * i + 4 * wave) * 256);
    };
    auto issue_xa = [&](const Cur& c) {
        const unsigned b = c.bq * S + c.s, p0 = c.pt * CB_NT;
        const u32x4 d = cb_desc(p.xa + (size_t)b * p.K0 * hw, kbytes);
#pragma unroll
        for (int i = 0; i < NXA; ++i)
            dma16(xv + ((unsigned)(32 * i + 4 * wave) * hw + p0) * 4u, d, lbase + CB_XA + (32 * i + 4 * wave) * 256);
    };

    Cur cur;
    cur.unit = g;
    cur.s = 0;
    cur.bq = g / tpp;
    cur.pt = g % tpp;
    bool valid = cur.unit < p.nunits;
    if (valid) {
        issue_g(cur, 0);
        issue_xb(cur);
        issue_xa(cur);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Cur nxt = advance(cur);
    bool primed = false;
    int par = 0;

    f32x16 accB[2], accA[2];                            // gw_B / gw_A: rows 32 rb .., column blocks 2 ph, 2 ph + 1
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 16; ++j) accB[n][j] = accA[n][j] = 0.f;
    float bsB[4] = {0.f, 0.f, 0.f, 0.f};                // row sums of gz_B: rows srow + 32 i
    float bsA = 0.f;                                    // channel kA's sum of gz_A over this lane's pixels
    f32x4 gts[TPIX ? 4 : 1];                            // sum over a pixel's samples of gz_A (t_mode 2)
#pragma unroll
    for (int k = 0; k < (TPIX ? 4 : 1); ++k) gts[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gxm = 0.f;
    const float slopeB = p.slope_b, slopeA = p.slope_a;
    using v4s = short __attribute__((ext_vector_type(4)));
    using v4sp = __attribute__((address_space(3))) v4s*;
    auto tr8 = [&](unsigned a0, unsigned a1) -> u32x4 {   // this lane's 8 reduction rows of its pixel column
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4sp)(lds + a0));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4sp)(lds + a1));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        return u32x4{l2[0], l2[1], h2[0], h2[1]};
    };
    auto ldsq = [&](unsigned a) -> u32x4 { return *reinterpret_cast<const u32x4*>(lds + a); };

    while (valid) {
        const bool nvalid = nxt.unit < p.nunits;
        const unsigned b = cur.bq * S + cur.s, bq = cur.bq, p0 = cur.pt * CB_NT;
        const bool edge = p0 + CB_NT > hw;
        const bool colok = p0 + 4u * c16 < hw;          // (this thread's pixel quad is inside the plane: hw % 4 == 0)
        const unsigned GB = CB_G0 + (unsigned)par * CB_BUF;
        // (the lane offsets are "new" every tile as far as the compiler can tell: it would otherwise keep every one of their
        // ~50 XOR variants in a register of its own across the loop -- 99 spilled registers)
        unsigned gwa_ = gwa, gwb_ = gwb, tr0_ = tr0, tr1_ = tr1;
        asm volatile("" : "+v"(gwa_), "+v"(gwb_), "+v"(tr0_), "+v"(tr1_));

        // ================= phase 0: gy_B and x_B raw -> planes, in place (this wave's own rows)
        // (requested a tile ago; the requests behind them: x_A's and the tile before's 16 stores)
        if (primed) wait_vm(NXA + (dx ? 16 : 0));
        {
            const unsigned* sg = reinterpret_cast<const unsigned*>(lds + CB_SGN + (par * 8 + wave) * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 gv = *reinterpret_cast<const float4*>(lds + GB + craw + 8192 * i);
                if (edge && !colok) gv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (maskb) {
                    const unsigned bits = sg[(i * 4 + (int)(srow & 3)) * 2 + (int)(c16 >> 3)] >> ((4u * c16) & 31u);
                    gv.x = (bits & 1u) ? gv.x : gv.x * slopeB;
                    gv.y = (bits & 2u) ? gv.y : gv.y * slopeB;
                    gv.z = (bits & 4u) ? gv.z : gv.z * slopeB;
                    gv.w = (bits & 8u) ? gv.w : gv.w * slopeB;
                }
                bsB[i] += (gv.x + gv.y) + (gv.z + gv.w);
                u32x2 h, l;
                split4(gv, cgB, h, l);
                *reinterpret_cast<u32x2*>(lds + GB + cpl + 8192 * i) = h;
                *reinterpret_cast<u32x2*>(lds + GB + (cpl ^ 128u) + 8192 * i) = l;
                __builtin_amdgcn_sched_barrier(0);       // (one row quad at a time: this kernel has no register to park the others in)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 xq = *reinterpret_cast<const float4*>(lds + CB_XB + craw + 8192 * i);
                if (edge && !colok) xq = make_float4(0.f, 0.f, 0.f, 0.f);
                // layer A's activation decisions: y_A > 0 (relu and leaky relu alike), one nibble per pixel quad
                const unsigned nib = (xq.x > 0.f ? 1u : 0u) | (xq.y > 0.f ? 2u : 0u) | (xq.z > 0.f ? 4u : 0u) | (xq.w > 0.f ? 8u : 0u);
                *reinterpret_cast<unsigned char*>(lds + mskw + 512 * i) = (unsigned char)nib;
                u32x2 h, l;
                split4(xq, cxB, h, l);
                *reinterpret_cast<u32x2*>(lds + CB_XB + cpl + 8192 * i) = h;
                *reinterpret_cast<u32x2*>(lds + CB_XB + (cpl ^ 128u) + 8192 * i) = l;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nvalid) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_g(nxt, par ^ 1);                      // (the other G buffer: free since the tile before's last products)
        }
        cb_sync();

        // ================= phase 1: gw_B += gz_B x_B^T;  gy_A^T = gz_B^T W_B
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4 ah = ldsq(GB + (gwa_ ^ ((2u * s) << 4))), al = ldsq(GB + (gwa_ ^ ((8u + 2u * s) << 4)));
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const u32x4 bh = ldsq(CB_XB + 8192 * n + (gwb_ ^ ((2u * s) << 4)));
                    const u32x4 bl = ldsq(CB_XB + 8192 * n + (gwb_ ^ ((8u + 2u * s) << 4)));
                    accB[n] = mfma32(ah, bl, accB[n]);
                    accB[n] = mfma32(al, bh, accB[n]);
                    accB[n] = mfma32(ah, bh, accB[n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 ga[4];                                    // gy_A: channel kA, pixels 16 pb + 4 q ..
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) ga[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const unsigned kh = (2u * pb) << 4, kl = (8u + 2u * pb) << 4;
                const u32x4 zh = tr8(GB + 8192 * st + (tr0_ ^ kh), GB + 8192 * st + (tr1_ ^ kh));
                const u32x4 zl = tr8(GB + 8192 * st + (tr0_ ^ kl), GB + 8192 * st + (tr1_ ^ kl));
                // (transposed: rows = pixels, columns = this wave's 16 channels of y_A)
                ga[pb] = mfma16(zh, wBl[st], ga[pb]);
                ga[pb] = mfma16(zl, wBh[st], ga[pb]);
                ga[pb] = mfma16(zh, wBh[st], ga[pb]);
                __builtin_amdgcn_sched_barrier(0);       // (one block of operands in flight, not all four: registers)
            }
        }
        cb_sync();                                      // (every wave is through with gz_B and x_B)

        // ================= phase 2: x_B of the next tile on its way; x_A raw -> planes; gz_A = gy_A * act_A' -> planes over gz_B
        if (nvalid) issue_xb(nxt);
        wait_vm((dx && primed ? 16 : 0) + (nvalid ? 9 : 0));
        {
#pragma unroll
            for (int i = 0; i < NXA; ++i) {
                float4 xq = *reinterpret_cast<const float4*>(lds + CB_XA + craw + 8192 * i);
                if ((edge && !colok) || srow + 32u * i >= (unsigned)p.K0) xq = make_float4(0.f, 0.f, 0.f, 0.f);
                u32x2 h, l;
                split4(xq, cxA, h, l);
                *reinterpret_cast<u32x2*>(lds + CB_XA + cpl + 8192 * i) = h;
                *reinterpret_cast<u32x2*>(lds + CB_XA + (cpl ^ 128u) + 8192 * i) = l;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            const bool first_s = cur.s == 0;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const unsigned nib = *reinterpret_cast<const unsigned char*>(lds + mskr + 4 * pb);
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float u = ga[pb][i] * osc2;
                    v[i] = ((nib >> i) & 1u) ? u : u * slopeA;
                }
                bsA += (v[0] + v[1]) + (v[2] + v[3]);
                if constexpr (TPIX) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gts[pb][i] = (first_s ? 0.f : gts[pb][i]) + v[i];
                }
                u32x2 h, l;
                split4(make_float4(v[0], v[1], v[2], v[3]), cgA, h, l);
                const unsigned sl = ((2u * pb + (unsigned)(q >> 1)) ^ (kA & 15u)) << 4;
                *reinterpret_cast<u32x2*>(lds + GB + gzw + sl) = h;
                *reinterpret_cast<u32x2*>(lds + GB + gzw + (sl ^ 128u)) = l;
            }
        }
        cb_sync();

        // ================= phase 3: gw_A += gz_A x_A^T;  gx_A = W_A^T gz_A
        if (2 * ph < NBA) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4 ah = ldsq(GB + (gwa_ ^ ((2u * s) << 4))), al = ldsq(GB + (gwa_ ^ ((8u + 2u * s) << 4)));
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (2 * ph + n < NBA) {
                        const u32x4 bh = ldsq(CB_XA + 8192 * n + (gwb_ ^ ((2u * s) << 4)));
                        const u32x4 bl = ldsq(CB_XA + 8192 * n + (gwb_ ^ ((8u + 2u * s) << 4)));
                        accA[n] = mfma32(ah, bl, accA[n]);
                        accA[n] = mfma32(al, bh, accA[n]);
                        accA[n] = mfma32(ah, bh, accA[n]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 gxo[4];                                   // gx_A: rows 16 wave + 4 q + j of K, pixel 16 pb + lane % 16
        if (dx) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) gxo[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const unsigned kh = (2u * pb) << 4, kl = (8u + 2u * pb) << 4;
                    const u32x4 zh = tr8(GB + 8192 * st + (tr0_ ^ kh), GB + 8192 * st + (tr1_ ^ kh));
                    const u32x4 zl = tr8(GB + 8192 * st + (tr0_ ^ kl), GB + 8192 * st + (tr1_ ^ kl));
                    gxo[pb] = mfma16(wAh[st], zl, gxo[pb]);
                    gxo[pb] = mfma16(wAl[st], zh, gxo[pb]);
                    gxo[pb] = mfma16(wAh[st], zh, gxo[pb]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        cb_sync();                                      // (every wave is through with gz_A and x_A)
        if (nvalid) issue_xa(nxt);
        // ---- stores, behind every request of the tile
        if (dx) {
            const int r0 = 16 * wave;
            const int nr = p.K0 - r0 < 16 ? (p.K0 - r0 > 0 ? p.K0 - r0 : 0) : 16;
            const rsrc_t rgx = cb_rsrc(p.gxa + ((size_t)b * p.K0 + r0) * hw, (unsigned)nr * hw * 4u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned vo = ((unsigned)(4 * q + j) * hw + (unsigned)l15) * 4u;
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const bool in = !edge || p0 + 16u * pb + l15 < hw;
                    const float v = gxo[pb][j] * osc4;
                    gxm = __builtin_fmaxf(gxm, (in && 4 * q + j < nr) ? __builtin_fabsf(v) : 0.f);
                    buf_store(v, rgx, in ? vo : CB_OOB, (p0 + 16u * pb) * 4u);
                }
            }
        }
        if (cur.s + 1 == S) {                            // the pixel tile's last sample
            if constexpr (TPIX) {
                const rsrc_t rt = cb_rsrc(p.gt + ((size_t)bq * 128 + 16 * wave) * hw, 16u * hw * 4u);
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const bool in = p0 + 16u * pb + 4u * q < hw;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, gts[pb]), rt,
                                                           in ? ((unsigned)l15 * hw + 4u * q) * 4u : CB_OOB, (p0 + 16u * pb) * 4u, 0);
                }
            } else if (p.t_mode == 1) {
                // the per-image context gradient = layer A's bias sums per image group: this workgroup's own slice
                float v = bsA;
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (q == 0) p.gbp_a[((size_t)g * p.Bq + bq) * 128 + kA] += v;
                bsA = 0.f;
            }
        }

        primed = true;
        cur = nxt;
        nxt = advance(nxt);
        valid = nvalid;
        par ^= 1;
    }

    // ---- this workgroup's partial sums
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = bsB[i];
        v += __shfl_xor(v, 8, 16);
        v += __shfl_xor(v, 4, 16);
        v += __shfl_xor(v, 2, 16);
        v += __shfl_xor(v, 1, 16);
        if (c16 == 0) p.gbp_b[(size_t)g * 128 + srow + 32 * i] = v;
    }
    if (p.t_mode != 1) {
        float v = bsA;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (q == 0) p.gbp_a[(size_t)g * 128 + kA] += v;
    }
    {
        const float osB = (1.f / cgB) * (1.f / cxB), osA = (1.f / cgA) * (1.f / cxA);
        const rsrc_t rB = cb_rsrc(p.gwp_b + ((size_t)g * 128 + 32 * rb) * 128, 32u * 128u * 4u);
        const rsrc_t rA = cb_rsrc(p.gwp_a + ((size_t)g * 128 + 32 * rb) * p.K0, 32u * (unsigned)p.K0 * 4u);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (2 * ph + n) * 32 + l31;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int rl = (j & 3) + 8 * (j >> 2) + 4 * lhi;
                buf_store(accB[n][j] * osB, rB, (unsigned)(rl * 128 + col) * 4u, 0);
                buf_store(accA[n][j] * osA, rA, col < p.K0 ? (unsigned)(rl * p.K0 + col) * 4u : CB_OOB, 0);
            }
        }
    }
    if (p.gxmax != nullptr) amax_publish(__builtin_bit_cast(unsigned, gxm), p.gxmax);
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_pointwise_chain_bwd_supported(int cin, long hw) {
    return (cin >= 1 && cin <= 128 && hw >= 4 && hw % 4 == 0 && hw < (1L << 27) && (double)128 * (double)hw * 4.0 < 4294967000.0) ? 1 : 0;
}

extern "C" int sbmc_pointwise_chain_bwd_groups(int b, int s, int t_mode, long hw) {
    if (b <= 0 || s < 1 || hw <= 0) return 1;
    const int S = t_mode ? s : 1;
    const unsigned long long nunits = (unsigned long long)((hw + CB_NT - 1) / CB_NT) * (unsigned)(b / S);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    return (int)(nunits < (unsigned long long)cus ? nunits : (unsigned long long)cus);
}

extern "C" int sbmc_pointwise_chain_bwd_f32(const float* gy, const unsigned* signs_b, const float* xb, const float* wb,
                                            const float* xa, const float* wa, float* gxa, float* gw_partial_b,
                                            float* gb_partial_b, float* gw_partial_a, float* gb_partial_a, float* gt,
                                            const unsigned* gmax, const unsigned* xbmax, const unsigned* xamax, unsigned* gxmax,
                                            int b, int s, int cin, long hw, int t_mode, int act_b, float slope_b, int act_a,
                                            float slope_a, void* stream) {
    if (b < 0 || s < 1 || t_mode < 0 || t_mode > 2 || act_a < 0 || act_a > 2 || act_b < 0 || act_b > 2) return SBMC_HIP_EINVAL;
    if (!sbmc_pointwise_chain_bwd_supported(cin, hw)) return SBMC_HIP_EINVAL;
    if (b == 0) return 0;
    if (b % s || !gy || !xb || !wb || !xa || !wa || !gw_partial_b || !gb_partial_b || !gw_partial_a || !gb_partial_a ||
        !gmax || !xbmax || !xamax || (act_b != 0 && !signs_b) || (t_mode == 2 && !gt) || (gxmax && !gxa))
        return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)xb % 16 || (uintptr_t)xa % 16 || (uintptr_t)gt % 16 || (uintptr_t)signs_b % 4)
        return SBMC_HIP_EINVAL;
    PwChainBwdParams p;
    memset(&p, 0, sizeof(p));
    p.gy = gy; p.signs_b = act_b != 0 ? signs_b : nullptr; p.xb = xb; p.wb = wb; p.xa = xa; p.wa = wa; p.gxa = gxa;
    p.gwp_b = gw_partial_b; p.gbp_b = gb_partial_b; p.gwp_a = gw_partial_a; p.gbp_a = gb_partial_a; p.gt = gt;
    p.gmax = gmax; p.xbmax = xbmax; p.xamax = xamax; p.gxmax = gxmax;
    p.slope_b = act_b == 0 ? 1.f : (act_b == 1 ? 0.f : slope_b);
    p.slope_a = act_a == 0 ? 1.f : (act_a == 1 ? 0.f : slope_a);
    p.B = b; p.S = t_mode ? s : 1; p.K0 = cin; p.t_mode = t_mode;
    p.Bq = t_mode == 1 ? b / s : 1;
    p.hw = (unsigned)hw;
    p.tiles_per_plane = (unsigned)((hw + CB_NT - 1) / CB_NT);
    const unsigned long long nunits = (unsigned long long)p.tiles_per_plane * (unsigned)(b / p.S);
    if (nunits > 0xFFFFFFFFull - 65536) return SBMC_HIP_EINVAL;
    p.nunits = (unsigned)nunits;
    const unsigned grid = (unsigned)sbmc_pointwise_chain_bwd_groups(b, s, t_mode, hw);
    hipError_t e = hipMemsetAsync(gb_partial_a, 0, (size_t)grid * p.Bq * 128 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    const int kp = (cin + 31) / 32 * 32;
    const size_t lds = CB_LDS;
#define SBMC_PCB(KPV)                                                                                    \
    do {                                                                                                 \
        auto kern = t_mode == 2 ? pw_chain_bwd_kernel<KPV, true> : pw_chain_bwd_kernel<KPV, false>;                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, p); \
    } while (0)
    switch (kp) {
        case 32: SBMC_PCB(32); break;
        case 64: SBMC_PCB(64); break;
        case 96: SBMC_PCB(96); break;
        default: SBMC_PCB(128); break;
    }
#undef SBMC_PCB
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    return (int)hipGetLastError();
}

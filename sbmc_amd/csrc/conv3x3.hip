// 3 x 3 convolution of a channels-last fp32 image (stride 1, zero padding 1: every convolution of the U-nets),
// at fp32 accuracy on the f16 matrix pipe.
//
// v_mfma_f32_32x32x2_f32 -- what an fp32 implicit GEMM runs on -- has 1/16 of the f16 / bf16 MFMA rate.  An
// fp32 value x, scaled by a power of two c so that the tensor's largest magnitude lands in [2^14, 2^15), is
//     c x = h + l + e,    h = f16(c x),  l = f16(c x - h),   |e| <= max(2^-22 |c x|, 2^-25)
// (c x - h is exact in fp32; l is a subnormal half below 2^-14, hence the absolute floor: 2^-39 of the tensor's
// largest value), so
//     x w = (hx hw + hx lw + lx hw) / (cx cw) + [lx lw: <= 2^-22 |x w|, dropped]
// is THREE f16 products with exact fp32 partial products, accumulated in fp32 by v_mfma_f32_32x32x16_f16:
// 3 / 16 of the fp32 MFMA time.  Every term carries >= 22 significant bits unless it is more than 2^17 below its
// tensor's largest value (then: an absolute error of 2^-39 of that value) -- the sum over 9 Cin terms in fp32
// loses more than that on either pipe.  The scales are powers of two (exact), taken from the tensors' largest
// magnitudes ON THE DEVICE (no host synchronisation): sbmc_conv3x3_absmax_f32 for the image, the weight
// preparation for the weights.
//
// Implicit GEMM, A = pixels x (tap, cin) from the image, B = (tap, cin) x cout from the weights:
//   * a workgroup (4 waves, one per SIMD, the whole 512-entry register file each) owns 16 x 16 pixels x 128
//     output channels; a wave 8 x 16 pixels x 64 channels = 4 x 2 accumulators of 32 x 32;
//   * the 18 x 18 pixel patch of 32 input channels is fetched ONCE for all nine taps (fp32, 128 contiguous
//     bytes per pixel), split in registers and kept in LDS as [plane][octet of channels][pixel] 16-byte entries:
//     the A operand of tap (ky, kx) is the same image shifted by (ky, kx) entries, one conflict-free
//     ds_read_b128 (16 lanes = 16 consecutive pixels of a patch row = 256 contiguous bytes);
//   * the weights are prepared once per step (sbmc_conv3x3_prepare_weights_f32: split, and laid out so that one
//     STAGE -- a kernel row of 3 taps x 16 input channels x 128 output channels x 2 planes, 24 KB -- is one
//     contiguous block that goes to LDS as it is), double-buffered, one barrier per stage (72 MFMAs per wave);
//   * persistent workgroups: the fetches of the next chunk / stage / tile are in flight during the MFMAs of the
//     current one, across tile boundaries.
// LDS traffic is what bounds this shape: a 32 x 32 x 16 MFMA eats 2 KB of operands in 32 cycles while LDS
// delivers 128 B / cycle to the whole CU, so operands must be re-used from registers -- 12 operand reads per 24
// MFMAs here (0.5 of the LDS bandwidth at full matrix rate).
#include <hip/hip_runtime.h>
#include <utility>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/sbmc_hip.h"
#include "common.hpp"

namespace sbmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f2 = __attribute__((ext_vector_type(2))) float;

constexpr int CV_TS = 16;                    // tile side (pixels)
constexpr int CV_PS = CV_TS + 2;             // patch side
constexpr int CV_PP = CV_PS * CV_PS;         // patch pixels (324)
constexpr int CV_PR = CV_PP + 2;             // entries of one (plane, octet) region: 2 CV_PR = 4 (mod 8), so that the two
                                             // octets a pair of staging lanes writes fall into different halves of 128 B
constexpr int CV_ABUF = 2 * 4 * CV_PR;       // 16-byte entries of one patch: [plane 2][octet 4][pixel]
constexpr int CV_WSTAGE = 3 * 2 * 2 * 128;   // 16-byte entries of one weight stage: [kx 3][plane 2][k half 2][cout 128]
constexpr unsigned CV_LDS_BYTES = (2 * CV_ABUF + 3 * CV_WSTAGE) * 16;   // 156672 of the CU's 163840
constexpr int CV_AROUNDS = (2 * CV_PP + 255) / 256;                     // staging rounds: (pixel, 16 channels) units
constexpr unsigned CV_OOB = 0xFFFFFFF0u;
#ifndef CV_WS_PRIO
#define CV_WS_PRIO 0
#endif
#ifndef CV_WS_AHEAD
#define CV_WS_AHEAD 2          // groups (of 6 MFMAs) a patch operand pair is fetched ahead of its use
#endif

__device__ __forceinline__ rsrc_t cv_rsrc(const void* base, unsigned bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    void* u = reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__device__ __forceinline__ void cv_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float cv_scale_of(unsigned maxbits) { return pow2_scale_of(maxbits); }     // (common.hpp)

__device__ __forceinline__ unsigned cv_pack(float a, float b) {
    h2 v;
    v[0] = (_Float16)a;
    v[1] = (_Float16)b;
    return __builtin_bit_cast(unsigned, v);
}

// 8 floats (already scaled) -> their two f16 planes
__device__ __forceinline__ void cv_split(const float (&v)[8], u32x4& h, u32x4& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h2 hp;
        hp[0] = (_Float16)v[2 * i];
        hp[1] = (_Float16)v[2 * i + 1];
        h[i] = __builtin_bit_cast(unsigned, hp);
        l[i] = cv_pack(v[2 * i] - (float)hp[0], v[2 * i + 1] - (float)hp[1]);      // differences exact
    }
}

// (the four-instruction f16 split of a value pair: common.hpp f16_split_pair -- in the staging paths of the 3 x 3 kernels
// the split is most of what is not an MFMA)
__device__ __forceinline__ void cv_split_pair(float a, float b, float c, unsigned& h, unsigned& l) { f16_split_pair(a, b, c, h, l); }

__device__ __forceinline__ f32x16 cv_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

struct Conv3Params {
    const void* x;           // [N, H, W, Cin] float (HF: _Float16)
    const u32x4* wp;         // prepared weights: [cout tile][Cin / 16][ky][kx][plane][k half][128] entries of 8 halves
    void* y;                 // [N, H, W, Cout] float (HF: _Float16)
    const unsigned* xmax;    // bit pattern of max |x| (HF: unused)
    const float* wscale;     // the weights' scale
    int N, H, W, Cin, Cout;
    int tiles_x, tiles_y, ncot;
    unsigned ntiles;         // N * tiles_y * tiles_x * ncot
    // EPI: y = act(conv + bias) -- the bias + ReLU / LeakyReLU pass behind the convolution (reference
    // sbmc/modules.py:154-175) in the epilogue, with what that pass produces besides
    const float* bias;       // [Cout]
    float slope;             // 1: linear, 0: ReLU, else LeakyReLU
    unsigned* signs;         // one bit per output (pre-activation > 0), bit e % 32 of word e / 32 of the NHWC element index e; or nullptr
    unsigned* amax;          // raised to the bit pattern of max |y| (a word zeroed by the caller); or nullptr
    // stream-K (or nullptr: whole tiles dealt round-robin): [4 KB unused][partial slabs: gridDim.x x 32768 floats][parked
    // slabs: the same]
    void* ws;
    // ADJ (the data gradient of a convolution whose INPUT is the activated output of another one, which nothing else
    // reads): y = conv * act'(z), the adjoint of that layer's bias + activation pass in this kernel's epilogue -- sign
    // bits in (the words the producer's forward left), the bias gradient's partial sums and max |y| out (amax above)
    const unsigned* adj_signs;   // [N H W Cout / 32]
    float adj_slope;             // 0: ReLU, else LeakyReLU
    float* adj_partial;          // [rows][Cout], zeroed by the caller: rows 2 g + (wave & 1) of the main kernel, 2 G + 2 g + .. of the fix-up's
};
constexpr int CV_ADJ_MAXC = 512;                                   // output channels of an ADJ launch (LDS: 2 x Cout floats)
constexpr unsigned CV_ADJ_LDS_BYTES = 2u * CV_ADJ_MAXC * 4u;
constexpr int CV_SLAB = 256 * 128;           // floats of one workgroup's accumulators
constexpr int CV_WS_HDR = 4096;

// ---- a tile is complete: scale back (+ bias, activation, sign bits, largest magnitude), store, clear.  One code for the
// main kernel and for the stream-K fix-up kernel (same thread -> (pixel, channel) map).  park (stream-K): the accumulators
// go to this workgroup's slab RAW instead -- same stores, other descriptor and offsets, selected, not branched to: a
// branch around the epilogue (or a second copy of it) costs the main loop 140 spilled registers. ----
struct CvTile { int n, y0, x0, ct; };
__device__ __forceinline__ void cv_slab_store(const f32x16 (&acc)[4][2], float* base);

// lanes LANE and LANE + 32 of v = the halves of a ballot (v_writelane_b32 with an inline-constant lane; the s_nop: a VALU
// reading a scalar register a VALU has just written needs two wait states on gfx940+, and the compiler's hazard
// recognizer does not look into inline assembly)
template <int LANE>
__device__ __forceinline__ void cv_write_lanes(unsigned& v, unsigned long long ballot) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
        : "+v"(v) : "s"((unsigned)ballot), "s"((unsigned)(ballot >> 32)), "n"(LANE), "n"(LANE + 32));
}
template <class F, int... J>
__device__ __forceinline__ void cv_unrolled_impl(F&& f, std::integer_sequence<int, J...>) {
    (f(std::integral_constant<int, J>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in its body
template <int N, class F>
__device__ __forceinline__ void cv_unrolled(F&& f) { cv_unrolled_impl(f, std::make_integer_sequence<int, N>{}); }

// SK: the launch may park (stream-K): only then is the parking store in the code at all.
// The epilogue runs with the matrix pipe idle and one wave per SIMD: its time is its instruction count.  Tiles whose 16
// columns are all inside the image (every tile of the model's power-of-two widths) take the lean form: one lane offset
// (two: registers 8 and 9 wrap around the tile's 16 columns in the upper half-wave), a scalar offset per (row, column)
// and an immediate per channel block; rows beyond the image select an empty descriptor per row pair; the sign words
// of 16 registers are gathered in a register (lane r / r + 32 <- the ballot's halves) and leave in ONE store; the
// largest magnitude is a masked max.  ~12 instructions per accumulator register instead of ~35.
// CLR = false: the accumulators are left as they are (the caller clears them as whole registers tuples: element-wise clears
// between the reads make the compiler copy tuples, which the 256-register waves of the WS form have no room for).
template <bool EPI, bool HF, bool SK, bool ADJ = false, bool CLR = true>
__device__ __forceinline__ void cv_finish(const Conv3Params& p, f32x16 (&acc)[4][2], const CvTile& t, const bool park,
                                          float* park_slab, const float oscale, unsigned& amax_run, float* bacc = nullptr) {
    static_assert(!(EPI && ADJ) && !(HF && ADJ), "ADJ: the fp32 data gradient");
    constexpr unsigned ES = HF ? 2u : 4u;
    int tid_e = threadIdx.x;
    // (ADJ: the lane's offsets are derived HERE, per tile -- hoisted out of the persistent loop they are four registers more
    // than the main loop has, i.e. spills)
    if constexpr (ADJ) asm volatile("" : "+v"(tid_e));
    const int tid = tid_e, lane = tid & 63, wave = wave_id();
    const int l31 = lane & 31, lhi = lane >> 5;
    const int mh = wave & 1, nh = wave >> 1;
    const bool sign_lane = l31 == 0;
    (void)sign_lane;
    constexpr bool parking = false;                     // (a parked tile has left above)
    char* yb = static_cast<char*>(p.y) + ((((long)t.n * p.H + t.y0) * (long)p.W + t.x0) * (long)p.Cout + t.ct * 128) * ES;
    const rsrc_t ry = cv_rsrc(yb, parking ? 0u : 0x7FFFFFF0u);
    float bv[2] = {0.f, 0.f};
    rsrc_t rs = ry;
    if constexpr (EPI) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bv[ni] = p.bias[t.ct * 128 + nh * 64 + ni * 32 + l31];
        // sign words of this tile's first pixel and output-channel tile: word = pixel (Cout / 32) + channel / 32
        rs = cv_rsrc(p.signs + (((long)t.n * p.H + t.y0) * (long)p.W + t.x0) * (long)(p.Cout / 32) + t.ct * 4,
                     (p.signs && !parking) ? 0x7FFFFFF0u : 0u);
    }
    float bsum[2] = {0.f, 0.f};
    if constexpr (ADJ)
        rs = cv_rsrc(p.adj_signs + (((long)t.n * p.H + t.y0) * (long)p.W + t.x0) * (long)(p.Cout / 32) + t.ct * 4, 0x7FFFFFF0u);
    // ADJ: the tile's bias sums (this wave's 64 channels, both half-waves added) join the workgroup's in LDS
    auto adj_commit = [&]() {
        if constexpr (ADJ) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float o = __shfl_xor(bsum[ni], 32, 64);
                float* d = bacc + mh * p.Cout + t.ct * 128 + nh * 64 + ni * 32 + l31;      // (this wave's own entries)
                if (lhi == 0) *d += bsum[ni] + o;
            }
        }
    };
    if constexpr (SK) {
        // stream-K: a parked tile's accumulators go to the slab raw (its epilogue runs in the fix-up kernel)
        if (park) {
            cv_slab_store(acc, park_slab);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            return;
        }
    }
    if (t.x0 + CV_TS <= p.W) {
        // ---- every column inside ----
        const rsrc_t rnone = cv_rsrc(yb, 0u);
        const unsigned lb = (unsigned)((4 * lhi) * p.Cout + nh * 64 + l31) * ES;
        // registers 8, 9: columns 14 | 15 in the lower half-wave, (14 | 15) + 4 - 16 = 2 | 3 in the upper: based at 2 | 3
        // (a lane offset must not go negative: it is range-checked BEFORE the scalar offset is added)
        const unsigned lbw = (unsigned)((lhi ? 0 : 12) * p.Cout + nh * 64 + l31) * ES;
        const int rows_in = p.H - t.y0 - mh * 8;                               // rows of this wave's 8 inside the image
        // sign words: lane j (j < 16) / 32 + j holds register j's word: pixel (j / 8, column of register j)
        unsigned sgl = CV_OOB;
        if constexpr (EPI || ADJ) {
            const int cj = (l31 & 3) + 8 * ((l31 >> 2) & 1) + 14 * ((l31 >> 3) & 1);
            const int colj = (cj + 4 * lhi) & 15;
            sgl = (unsigned)(((((l31 >> 3) & 1) * p.W + colj) * (p.Cout / 32) + nh * 2) * 4);
        }
        const unsigned amask_in = parking ? 0u : 0x7FFFFFFFu;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            unsigned sw[2] = {0u, 0u};
            if constexpr (ADJ) {
                __builtin_amdgcn_sched_barrier(0);       // (one row pair's words at a time: registers)
                // the sign words of this row pair's 16 registers, where the forward's epilogue put them: lane r / r + 32 <-
                // register r's word of the lower / upper half-wave's pixel
                const bool ok = l31 < 16 && mi * 2 + ((l31 >> 3) & 1) < rows_in;
                const unsigned so = (unsigned)(((mh * 8 + mi * 2) * p.W) * (p.Cout / 32)) * 4u;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    sw[ni] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? sgl + (unsigned)ni * 4u : CV_OOB, so, 0);
            }
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const bool gok = mi * 2 + rh < rows_in;                       // wave-uniform
                const rsrc_t rg = gok ? ry : rnone;
                const unsigned amask = gok ? amask_in : 0u;
                const int row = mh * 8 + mi * 2 + rh;
                cv_unrolled<8>([&](auto rc) {
                    constexpr int r8 = decltype(rc)::value;
                    const int r = rh * 8 + r8;
                    const bool wraps = rh == 1 && r8 < 2;
                    const int col0 = (((r8 & 3) + 8 * ((r8 >> 2) & 1) + 14 * rh) & 15) - (wraps ? 12 : 0);
                    const unsigned vb = wraps ? lbw : lb;
                    const unsigned so = (unsigned)((row * p.W + col0) * p.Cout) * ES;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        float v = acc[mi][ni][r] * oscale;
                        if constexpr (EPI) {
                            v += bv[ni];
                            const bool pos = v > 0.f;
                            const unsigned long long bal = __builtin_amdgcn_ballot_w64(pos);
                            if (rh == 0) cv_write_lanes<r8>(sw[ni], bal);
                            else cv_write_lanes<r8 + 8>(sw[ni], bal);
                            v = pos ? v : v * p.slope;
                            if constexpr (!HF) {                               // (half outputs carry no scale word)
                                const unsigned a = __builtin_bit_cast(unsigned, v) & amask;
                                amax_run = amax_run > a ? amax_run : a;
                            }
                        }
                        if constexpr (ADJ) {
                            // the 64 lanes' sign bits are the two words as they stand: a lane mask
                            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)sw[ni], r);
                            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)sw[ni], r + 32);
                            const bool pos = __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
                            v = pos ? v : v * p.adj_slope;
                            const unsigned a = __builtin_bit_cast(unsigned, v) & amask;
                            amax_run = amax_run > a ? amax_run : a;
                            bsum[ni] += gok ? v : 0.f;
                        }
                        if constexpr (HF)
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), rg,
                                                                  vb + (unsigned)(ni * 32) * ES, so, 0);
                        else
                            buf_store(v, rg, vb + (unsigned)(ni * 32) * ES, so);
                        if constexpr (CLR) acc[mi][ni][r] = 0.f;
                    }
                });
            }
            if constexpr (EPI) {
                // lanes 0-15 / 32-47: the words of registers 0-15; row pair mi, row (l31 / 8) of it
                const bool ok = l31 < 16 && mi * 2 + ((l31 >> 3) & 1) < rows_in;
                const unsigned so = (unsigned)(((mh * 8 + mi * 2) * p.W) * (p.Cout / 32)) * 4u;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    __builtin_amdgcn_raw_buffer_store_b32(sw[ni], rs, ok ? sgl + (unsigned)ni * 4u : CV_OOB, so, 0);
            }
        }
        adj_commit();
        return;
    }
    // (an opaque copy of the scale: with the same expression `acc * oscale` in both forms the compiler computes all 128
    // products -- and the ADJ form's 128 more -- ABOVE the branch between them, in registers nobody has)
    float oscale_g = oscale;
    asm volatile("" : "+v"(oscale_g));
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // accumulator register r of a 32 x 32 block: pixel (r & 3) + 8 (r >> 2) + 4 (lane / 32) of the
            // block's 2 rows x 16 columns; output channel lane % 32
            const int row = mh * 8 + mi * 2 + (r >> 3);
            const int col = ((r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi + 14 * (r >> 3)) & 15;     // (the rotation above)
            const bool ok = t.y0 + row < p.H && t.x0 + col < p.W;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const unsigned voff = ok ? (unsigned)((row * p.W + col) * p.Cout + nh * 64 + ni * 32 + l31) * ES : CV_OOB;
                float v = acc[mi][ni][r] * oscale_g;
                if constexpr (EPI) {
                    v += bv[ni];
                    const bool pos = v > 0.f;
                    {
                        // lanes 0-31 are the 32 channels of one sign word (pixel of the lower half), 32-63 the next
                        // pixel's; no branch on `signs`: without them every lane's offset is out of range
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(pos);
                        const unsigned word = lhi ? (unsigned)(bal >> 32) : (unsigned)bal;
                        const unsigned so = (ok && sign_lane) ? (unsigned)((row * p.W + col) * (p.Cout / 32) + nh * 2 + ni) * 4u : CV_OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(word, rs, so, 0, 0);
                    }
                    v = pos ? v : v * p.slope;
                    if (ok && !parking) {
                        const unsigned a = abits(v);
                        amax_run = amax_run > a ? amax_run : a;
                    }
                }
                if constexpr (ADJ) {
                    // (the 32 lanes of a half-wave are the 32 channels of one sign word)
                    const unsigned so = ok ? (unsigned)((row * p.W + col) * (p.Cout / 32) + nh * 2 + ni) * 4u : CV_OOB;
                    const unsigned word = __builtin_amdgcn_raw_buffer_load_b32(rs, so, 0, 0);
                    v = ((word >> l31) & 1u) ? v : v * p.adj_slope;
                    if (ok) {
                        const unsigned a = abits(v);
                        amax_run = amax_run > a ? amax_run : a;
                        bsum[ni] += v;
                    }
                }
                if constexpr (HF)
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), ry, voff, 0, 0);
                else
                    buf_store(v, ry, voff, 0);
                if constexpr (CLR) acc[mi][ni][r] = 0.f;
            }
            if constexpr (EPI) __builtin_amdgcn_sched_barrier(0);      // (one register's ballots and stores at a time)
        }
    }
    adj_commit();
}

// a workgroup's accumulators <-> a slab of the stream-K workspace (element e of thread t at e * 256 + t; buffer
// addressing: one lane-offset register and immediate element offsets)
__device__ __forceinline__ void cv_slab_store(const f32x16 (&acc)[4][2], float* base) {
    const rsrc_t rb = cv_rsrc(base, (unsigned)CV_SLAB * 4u);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store(acc[mi][ni][r], rb, threadIdx.x * 4u, (unsigned)(((mi * 2 + ni) * 16 + r) * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
}
template <bool ADD, bool PACE = true>
__device__ __forceinline__ void cv_slab_load(f32x16 (&acc)[4][2], const float* base) {
    const rsrc_t rb = cv_rsrc(base, (unsigned)CV_SLAB * 4u);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = buf_load(rb, threadIdx.x * 4u, (unsigned)(((mi * 2 + ni) * 16 + r) * 1024));
                acc[mi][ni][r] = ADD ? acc[mi][ni][r] + v : v;
            }
            if (PACE) __builtin_amdgcn_sched_barrier(0);
        }
}

// Stream-K fix-up (the launch behind a conv3_kernel that ran with a workspace): workgroup g of that launch parked the
// tail of its first tile if its range of (tile, chunk) units began inside that tile and reached the tile's end; the
// workgroups before it left the tile's head (and middle) in their partial slabs.  Tail + partials in a FIXED order
// (g - 1, g - 2, ... back to the workgroup whose range holds the tile's first unit), then the tile's epilogue.
template <bool EPI, bool HF, bool ADJ = false>
__global__ __launch_bounds__(256) void conv3_fixup_kernel(Conv3Params p, unsigned G) {
    const unsigned g = blockIdx.x, nchunks = (unsigned)p.Cin / 32u;
    const unsigned long long U = (unsigned long long)p.ntiles * nchunks;
    const unsigned long long u0 = U * g / G, u1 = U * (g + 1) / G;
    const unsigned first = (unsigned)(u0 / nchunks), c0 = (unsigned)(u0 % nchunks);
    __shared__ float bacc[ADJ ? 2 * CV_ADJ_MAXC : 1];
    if (c0 == 0 || u0 + (nchunks - c0) > u1) return;      // began on a tile boundary, or never reached its tile's end
    if constexpr (ADJ) {
        for (int i = threadIdx.x; i < 2 * p.Cout; i += 256) bacc[i] = 0.f;
        __syncthreads();
    }
    float* const slabs = reinterpret_cast<float*>(static_cast<char*>(p.ws) + CV_WS_HDR);
    // (a slab's 128 loads per thread all in flight -- this kernel has the registers --: issued 16 at a time it took 20 us,
    // on latency alone, where the convolution it completes takes 120)
    f32x16 acc[4][2], part[4][2];
    cv_slab_load<false, false>(acc, slabs + (size_t)(G + g) * CV_SLAB);
    const unsigned long long head = (unsigned long long)first * nchunks;
    for (unsigned gp = g; gp-- > 0;) {
        cv_slab_load<false, false>(part, slabs + (size_t)gp * CV_SLAB);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] += part[mi][ni];
        if (U * gp / G <= head) break;
    }
    CvTile t;
    t.ct = (int)(first % (unsigned)p.ncot);
    const unsigned pt = first / (unsigned)p.ncot;
    t.x0 = (int)(pt % (unsigned)p.tiles_x) * CV_TS;
    const unsigned rest = pt / (unsigned)p.tiles_x;
    t.y0 = (int)(rest % (unsigned)p.tiles_y) * CV_TS;
    t.n = (int)(rest / (unsigned)p.tiles_y);
    const float cx = HF ? 1.f : cv_scale_of(*p.xmax);
    const float oscale = (1.f / cx) * (1.f / *p.wscale);
    unsigned amax_run = 0;
    cv_finish<EPI, HF, false, ADJ>(p, acc, t, false, nullptr, oscale, amax_run, bacc);
    if constexpr (EPI || ADJ) {
        if (p.amax) amax_publish(amax_run, p.amax);         // (a barrier inside)
    }
    if constexpr (ADJ) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * p.Cout; i += 256) p.adj_partial[(size_t)(2 * G + 2 * g) * p.Cout + i] = bacc[i];
    }
}

// HF: half activations ("fp16 activations", BASELINE configs[4]; torch.autocast(float16) semantics: half inputs, the
// weights rounded to half once, fp32 accumulation, half output): x and y are _Float16 tensors, the patch goes to LDS
// as it is (one plane, no split, no scale), the weights are the prepared weights' HIGH plane (f16 of the scaled
// weight: the rounding autocast applies, with a power-of-two scale that is divided out again), ONE matrix product per
// term instead of three.
//
// WS ("wave-specialised", 512 threads): the same tiles, stages and LDS protocol with the two jobs of a stage in different
// waves -- waves 0-3 (one per SIMD) issue nothing but operand reads and MFMAs, waves 4-7 (their SIMD partners) request,
// split and write the patches and weight stages; one barrier per stage for all eight.  An in-order wave cannot overlap
// its staging instructions with its MFMAs; two waves of one SIMD can (the matrix pipe beside the vector-memory and
// LDS-write work).  256 registers per wave: 128 accumulators + the two taps' operands in the MFMA waves.
template <bool EPI, bool HF = false, bool SK = false, bool ADJ = false, bool WS = false>
__global__ __launch_bounds__(WS ? 512 : 256) void conv3_kernel(Conv3Params p) {
    constexpr unsigned ES = HF ? 2u : 4u;              // bytes of an activation element
    extern __shared__ float4 cv_lds[];
    u32x4* As = reinterpret_cast<u32x4*>(cv_lds);
    u32x4* Ws = As + 2 * CV_ABUF;
    static_assert(!(WS && HF), "WS: the fp32 form");
    static_assert(72 % (CV_WS_AHEAD + 1) == 0, "the slots of the rolling patch operands must come round with the chunk");
    const int tid = WS ? (int)(threadIdx.x & 255u) : (int)threadIdx.x, lane = tid & 63, wave = WS ? (wave_id() & 3) : wave_id();
    const bool producer = WS && wave_id() >= 4;       // (wave-uniform, scalar)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int mh = wave & 1, nh = wave >> 1;
    // Row i of a 32 x 32 block is pixel (i / 16, column) of the block's 2 rows x 16 columns.  ds_read_b128 serves
    // lanes {0-3, 12-15, 20-27} (and so on) together: with the patch rows 18 entries apart, the second row's
    // lanes must be rotated by 18 mod 16 = 2 columns for every such group to cover 16 different 16-byte slots of
    // the 256-byte bank row: lane 16 + j reads column (j + 14) mod 16.
    const int pcol = (l31 + 14 * (l31 >> 4)) & 15;
    const float cx = HF ? 1.f : cv_scale_of(*p.xmax);
    const float oscale = (1.f / cx) * (1.f / *p.wscale);
    const unsigned nchunks = (unsigned)p.Cin / 32u;

    // The work of this workgroup, in (tile, 32-channel chunk) units.
    //   whole tiles, round-robin (p.ws == nullptr): tiles blockIdx.x, + gridDim.x, ...: a launch whose tile count is not a
    //     multiple of the CU count idles part of the chip in its last round -- 320 tiles on 256 CUs run at 62 %, which is
    //     what the slabs of a frame sharded over 8 GPUs look like at the U-net's coarser levels;
    //   stream-K (p.ws): the units of all tiles, tile after tile, are cut into gridDim.x EQUAL contiguous ranges.  A range
    //     that starts inside a tile holds that tile's TAIL: the workgroup parks those accumulators in its slab when the
    //     tile's last chunk is through and goes on with whole tiles; a range that ends inside a tile leaves that tile's
    //     head in the workgroup's other slab.  conv3_fixup_kernel, the next launch, adds heads to tails in a fixed
    //     order and runs those tiles' epilogues: no flags, no waiting between workgroups, nothing that depends on which
    //     of them the dispatcher started first (a first version that combined inside the launch needed both).
    constexpr bool sk = SK;                              // (the launcher: SK <=> p.ws != nullptr)
    const unsigned G = gridDim.x, g = blockIdx.x;
    const unsigned long long U = (unsigned long long)p.ntiles * nchunks;
    unsigned first, stride, total, cc;
    if (sk) {
        const unsigned long long u0 = U * g / G, u1 = U * (g + 1) / G;
        first = (unsigned)(u0 / nchunks);
        cc = (unsigned)(u0 % nchunks);
        stride = 1;
        total = (unsigned)(u1 - u0);
    } else {
        first = g;
        stride = G;
        cc = 0;
        const unsigned my_tiles = first < p.ntiles ? (p.ntiles - first + stride - 1) / stride : 0;
        total = my_tiles * nchunks;
    }
    const unsigned c0 = cc;                             // chunk the first tile starts at (stream-K: may be inside it)
    using Tile = CvTile;
    auto tile_at = [&](unsigned i) -> Tile {
        // output-channel tiles of one pixel tile are neighbours in the walk (they read the same patch)
        const unsigned t = first + i * stride;
        Tile r;
        r.ct = (int)(t % (unsigned)p.ncot);
        const unsigned pt = t / (unsigned)p.ncot;
        r.x0 = (int)(pt % (unsigned)p.tiles_x) * CV_TS;
        const unsigned rest = pt / (unsigned)p.tiles_x;
        r.y0 = (int)(rest % (unsigned)p.tiles_y) * CV_TS;
        r.n = (int)(rest / (unsigned)p.tiles_y);
        return r;
    };
    // the tile of the current chunk and the one after it (the divisions above: once per tile, not per stage)
    Tile tcur = tile_at(0), tnext = tile_at(1);
    unsigned ti = 0;                                    // tiles this workgroup has completed

    // ---- staging of a patch: unit u = (pixel u / 2, channels 16 (u % 2) ..), 64 contiguous bytes ----
    u32x4 areg[CV_AROUNDS][HF ? 2 : 4];                // 16 channels of one pixel: 64 bytes of float, 32 of _Float16
    // (valid = false: a request beyond the workgroup's last chunk -- an empty descriptor, every lane reads zero.  NO
    // load of the main loop sits under a branch: at a control-flow join the compiler can no longer count which
    // loads are outstanding and waits for ALL of them, i.e. for the request it has just made.)
    auto issue_a = [&](const Tile& t, unsigned cc, bool valid) {
        const char* xb = static_cast<const char*>(p.x) + (((long)t.n * p.H + (t.y0 - 1)) * (long)p.W + (t.x0 - 1)) * (long)p.Cin * ES;
        const rsrc_t rx = cv_rsrc(valid ? (const void*)xb : p.x, valid ? 0x7FFFFFF0u : 0u);
#pragma unroll
        for (int j = 0; j < CV_AROUNDS; ++j) {
            const int u = tid + 256 * j, q = u >> 1, half = u & 1;
            const int qr = q / CV_PS, qc = q - qr * CV_PS;
            const int gy = t.y0 - 1 + qr, gx = t.x0 - 1 + qc;
            const bool in = u < 2 * CV_PP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const unsigned voff = in ? (unsigned)((qr * p.W + qc) * p.Cin + half * 16) * ES : CV_OOB;
#pragma unroll
            for (int i = 0; i < (HF ? 2 : 4); ++i)
                areg[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, cc * 32u * ES + 16u * i, 0);
        }
    };
    auto commit_a_round = [&](int abuf, int j) {
        const int u = tid + 256 * j, q = u >> 1, half = u & 1;
        if (u < 2 * CV_PP) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                u32x4* d = As + abuf * CV_ABUF + (2 * half + o) * CV_PR + q;
                if constexpr (HF) {
                    d[0] = areg[j][o];               // 8 channels of the pixel: an operand entry as it is
                } else {
                    u32x4 hh, ll;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned w0 = areg[j][2 * o + (2 * c) / 4][(2 * c) % 4];        // (copies: bit_cast of the element expression itself reads element 0)
                        const unsigned w1 = areg[j][2 * o + (2 * c + 1) / 4][(2 * c + 1) % 4];
                        unsigned hp, lp;
                        cv_split_pair(__builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1), cx, hp, lp);
                        hh[c] = hp;
                        ll[c] = lp;
                    }
                    d[0] = hh;
                    d[4 * CV_PR] = ll;
                }
            }
        }
    };
    auto commit_a = [&](int abuf) {
#pragma unroll
        for (int j = 0; j < CV_AROUNDS; ++j) commit_a_round(abuf, j);
    };
    // ---- staging of a weight stage (h, st): 1536 entries, 6 per thread; two register sets: a stage's loads are
    // in flight for two stages ----
    u32x4 wregA[6], wregB[6];
    const unsigned wbytes = (unsigned)p.ncot * ((unsigned)p.Cin / 16u) * 3u * (unsigned)CV_WSTAGE * 16u;
    auto issue_w = [&](u32x4 (&wreg)[6], const Tile& t, unsigned cc, int st, bool valid) {
        const rsrc_t rw = cv_rsrc(p.wp, valid ? wbytes : 0u);
        const unsigned k16 = cc * 2u + (unsigned)(st / 3), ky = (unsigned)(st % 3);
        const unsigned block = valid ? ((unsigned)t.ct * ((unsigned)p.Cin / 16u) + k16) * 3u + ky : 0u;
        // (HF: the high plane only -- entries kx 512 + 0..255 of the stage: half the L2 and LDS traffic of the weights)
#pragma unroll
        for (int i = 0; i < (HF ? 3 : 6); ++i)
            wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(HF ? tid + 512 * i : tid + 256 * i) * 16u,
                                                            block * (unsigned)(CV_WSTAGE * 16), 0);
    };
    auto commit_w = [&](const u32x4 (&wreg)[6], int wbuf) {
#pragma unroll
        for (int i = 0; i < (HF ? 3 : 6); ++i) Ws[wbuf * CV_WSTAGE + (HF ? tid + 512 * i : tid + 256 * i)] = wreg[i];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    unsigned amax_run = 0;
    // ADJ: the workgroup's bias sums [wave & 1][Cout], in LDS behind the weight stages
    float* const bacc = reinterpret_cast<float*>(Ws + 3 * CV_WSTAGE);
    if constexpr (ADJ) {
        for (int i = tid; i < 2 * p.Cout; i += 256) bacc[i] = 0.f;      // (visible after the prologue's barrier)
    }
    if (total == 0) return;

    // operands of one tap: A 4 m-blocks x 2 planes, B 2 n-blocks x 2 planes
    struct Ops { u32x4 ah[4], al[4], bh[2], bl[2]; };
    // (in the order the tap's MFMAs want them -- high planes, then the weights' low plane, then the patch's: the reads go out
    // in this order between the previous tap's MFMAs and return in order; with the weights last, every tap began by waiting
    // for the reads issued two MFMAs earlier)
    auto load_ops = [&](Ops& o, const u32x4* Ab, const u32x4* Wb, int kx) {
        o.bh[0] = Wb[kx * 512];
        o.ah[0] = Ab[kx];
        o.bh[1] = Wb[kx * 512 + 32];
#pragma unroll
        for (int mi = 1; mi < 4; ++mi) o.ah[mi] = Ab[mi * 2 * CV_PS + kx];
        if constexpr (!HF) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) o.bl[ni] = Wb[kx * 512 + 256 + ni * 32];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) o.al[mi] = Ab[4 * CV_PR + mi * 2 * CV_PS + kx];
        }
    };
    auto mfmas_on = [&](f32x16 (&A)[4][2], const Ops& o) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) A[mi][ni] = cv_mfma(o.ah[mi], o.bh[ni], A[mi][ni]);
        if constexpr (!HF) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) A[mi][ni] = cv_mfma(o.ah[mi], o.bl[ni], A[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) A[mi][ni] = cv_mfma(o.al[mi], o.bh[ni], A[mi][ni]);
        }
    };
    auto mfmas = [&](const Ops& o) { mfmas_on(acc, o); };
    // The next tap's 12 operand fetches go BETWEEN the current tap's 24 MFMAs (2 MFMAs, 1 fetch, ...): left to
    // itself the compiler sinks them to just before their first use and the matrix pipe waits for LDS every tap.
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < (HF ? 6 : 12); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, HF ? 1 : 2, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // DS read
        }
    };
    // LDS addresses of stage st of chunk h for this lane (tap kx adds kx / 512 kx entries)
    auto a_ptr = [&](unsigned h, int st) -> const u32x4* {
        return As + (int)(h & 1u) * CV_ABUF + (2 * (st / 3) + lhi) * CV_PR + (mh * 8 + (l31 >> 4) + st % 3) * CV_PS + pcol;
    };
    auto w_ptr = [&](int st) -> const u32x4* { return Ws + (st % 3) * CV_WSTAGE + lhi * 128 + nh * 64 + l31; };

    float* const slabs = sk ? reinterpret_cast<float*>(static_cast<char*>(p.ws) + CV_WS_HDR) : nullptr;

    // Pipeline.  Stage s = 6 h + st reads weight buffer s % 3 (= st % 3: a chunk has 6 stages).  The weights of
    // stage s + 3 are REQUESTED at the start of stage s (register set (s + 1) % 2), those of stage s + 2 are
    // WRITTEN to LDS at its end (set s % 2; buffer (s + 2) % 3, last read in stage s - 1, i.e. before the previous
    // barrier), so every stage finds its weights -- and its patch: written in stage 3 of the previous chunk --
    // in LDS one whole stage early, and the operands of its first tap are fetched during the last tap of the
    // stage before: after a barrier the matrix pipe continues at once.
    if constexpr (WS) {
        if (producer) {
            issue_a(tcur, cc, true);
            issue_w(wregA, tcur, cc, 0, true);
            issue_w(wregB, tcur, cc, 1, true);
            commit_a(0);
            commit_w(wregA, 0);
            commit_w(wregB, 1);
            __syncthreads();
            issue_w(wregA, tcur, cc, 2, true);
            for (unsigned h = 0; h < total; ++h) {
                const bool more = h + 1 < total;
                const bool last = cc + 1 == nchunks;
                const Tile tn = last ? tnext : tcur;
                const unsigned ccn = last ? 0u : cc + 1;
                const int nbuf = (int)((h + 1) & 1u);
                auto pstage = [&](u32x4 (&wfill)[6], const u32x4 (&wdone)[6], const int st) {
                    if (st < 3) issue_w(wfill, tcur, cc, st + 3, true);
                    else issue_w(wfill, tn, ccn, st - 3, more);
                    if (st == 2) issue_a(tn, ccn, more);
                    if (st == 4) commit_a(nbuf);
                    commit_w(wdone, (st + 2) % 3);
                    cv_lds_barrier();
                };
                pstage(wregB, wregA, 0);
                pstage(wregA, wregB, 1);
                pstage(wregB, wregA, 2);
                pstage(wregA, wregB, 3);
                pstage(wregB, wregA, 4);
                pstage(wregA, wregB, 5);
                if (last) {
                    tcur = tnext;
                    ++ti;
                    tnext = tile_at(ti + 1);
                    cc = 0;
                } else {
                    ++cc;
                }
            }
        } else {
            __syncthreads();
            if constexpr (CV_WS_PRIO > 0) __builtin_amdgcn_s_setprio(CV_WS_PRIO);
            // The MFMA waves' operand stream.  A GROUP = one 32-pixel block of one tap: its patch operand pair (high, low) against
            // the tap's four weight operands, 6 MFMAs on two accumulators alternately.  72 groups per chunk (6 stages x 3 taps
            // x 4 blocks).  The operands ROLL: a group's patch pair is fetched two groups (12 MFMAs) ahead into one of three
            // slots, the next tap's weights one operand per group into the other of two sets -- 14 operands (56 registers) live
            // instead of two whole taps' 24, the same 12 reads per tap.  Per accumulator the order of the products is the
            // one-wave kernel's (high x high, high x low, low x high per tap): the same bits.
            struct APair { u32x4 h, l; };
            struct BSet { u32x4 h[2], l[2]; };
            constexpr int NA = CV_WS_AHEAD + 1;
            APair Ap[NA];
            BSet Bs[2];
            // LDS addresses: ONE lane offset for the patches (+ the chunk's buffer), two for the weight stages (buffers 0-1 | 2:
            // the instruction's offset field ends at 64 KB), everything else an immediate.  Opaque to the compiler, which
            // otherwise keeps the sums' parts in five to ten registers and, short of registers, reloads them mid-stream.
            unsigned a_lane = (unsigned)((lhi * CV_PR + (mh * 8 + (l31 >> 4)) * CV_PS + pcol) * 16);
            unsigned w_lane01 = (unsigned)((2 * CV_ABUF + lhi * 128 + nh * 64 + l31) * 16);
            unsigned w_lane2 = w_lane01 + (unsigned)(2 * CV_WSTAGE * 16);
            asm volatile("" : "+v"(a_lane), "+v"(w_lane01), "+v"(w_lane2));
            auto lds_at = [&](const unsigned off) -> u32x4 {
                return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(cv_lds) + off);
            };
            auto a_base = [&](const unsigned h) -> unsigned { return a_lane + (h & 1u) * (unsigned)(CV_ABUF * 16); };
            auto load_a = [&](APair& d, const unsigned ab, const int J) {    // group J of the chunk whose patch is at ab
                const int st = J / 12, kx = (J / 4) % 3, mi = J % 4;
                const unsigned off = (unsigned)((2 * (st / 3) * CV_PR + (st % 3) * CV_PS + mi * 2 * CV_PS + kx) * 16);
                d.h = lds_at(ab + off);
                d.l = lds_at(ab + off + (unsigned)(4 * CV_PR * 16));
            };
            auto load_b = [&](BSet& d, const int T, const int c) {          // operand c of tap T (weights: the same addresses in every chunk)
                const int wb = (T / 3) % 3, kx = T % 3;
                const unsigned base = wb == 2 ? w_lane2 : w_lane01;
                const unsigned off = (unsigned)(((wb == 2 ? 0 : wb) * CV_WSTAGE + kx * 512 + (c & 1) * 32 + (c >> 1) * 256) * 16);
                const u32x4 v = lds_at(base + off);
                if (c == 0) d.h[0] = v;
                else if (c == 1) d.h[1] = v;
                else if (c == 2) d.l[0] = v;
                else d.l[1] = v;
            };
            auto prime = [&](const unsigned h) {
                const unsigned ab = a_base(h);
                load_b(Bs[0], 0, 0);
                load_a(Ap[0], ab, 0);
                load_b(Bs[0], 0, 1);
                load_b(Bs[0], 0, 2);
                load_b(Bs[0], 0, 3);
#pragma unroll
                for (int j = 1; j < CV_WS_AHEAD; ++j) load_a(Ap[j], ab, j);
            };
            prime(0);
            auto chunk = [&](const unsigned h, f32x16 (&A)[4][2]) {
                const unsigned ab = a_base(h);
                cv_unrolled<72>([&](auto jc) {
                    constexpr int J = decltype(jc)::value;
                    constexpr int T = J / 4, mi = J % 4;
                    // fetches: the next tap's weight operand `mi`, the patch pair of the group after next
                    if constexpr (T + 1 < 18) load_b(Bs[(T + 1) % 2], T + 1, mi);
                    else load_b(Bs[0], 0, mi);
                    if constexpr (J + CV_WS_AHEAD < 72) load_a(Ap[(J + CV_WS_AHEAD) % NA], ab, J + CV_WS_AHEAD);
                    else load_a(Ap[(J + CV_WS_AHEAD) % NA], a_base(h + 1), J + CV_WS_AHEAD - 72);
                    const APair& a = Ap[J % NA];
                    const BSet& b = Bs[T % 2];
                    A[mi][0] = cv_mfma(a.h, b.h[0], A[mi][0]);
                    A[mi][1] = cv_mfma(a.h, b.h[1], A[mi][1]);
                    A[mi][0] = cv_mfma(a.h, b.l[0], A[mi][0]);
                    A[mi][1] = cv_mfma(a.h, b.l[1], A[mi][1]);
                    A[mi][0] = cv_mfma(a.l, b.h[0], A[mi][0]);
                    A[mi][1] = cv_mfma(a.l, b.h[1], A[mi][1]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // DS read
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);               // MFMA
                    }
                    // (no wait for the reads in flight at a stage's end: they are the NEXT stage's operands, from buffers nobody
                    // writes in that stage; every read of this stage's buffers has fed an MFMA above)
                    if constexpr (J % 12 == 11) asm volatile("s_barrier" ::: "memory");
                });
            };
            // a tile is through: the next chunk's first operands are fetched AGAIN behind the epilogue (what the last groups
            // fetched is dead then: 32 registers the epilogue may use)
            auto next_tile = [&](const unsigned h) {
                tcur = tnext;
                ++ti;
                tnext = tile_at(ti + 1);
                cc = 0;
                __builtin_amdgcn_sched_barrier(0);        // (not above the epilogue, where the scheduler would like them)
                prime(h + 1);
            };
            unsigned h = 0;
            if constexpr (SK) {
                // stream-K: the range starts inside a tile -- that tile's tail, parked raw in this workgroup's second slab (its
                // epilogue runs in the fix-up kernel).  A loop and ACCUMULATORS of its own: parking stores beside the epilogue in
                // one loop body, or one set of accumulators through both loops, cost the 256-register wave spills.
                if (c0 > 0) {
                    f32x16 pacc[4][2];
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int r = 0; r < 16; ++r) pacc[mi][ni][r] = 0.f;
                    for (; h < total && cc < nchunks; ++h, ++cc) chunk(h, pacc);
                    if (cc == nchunks) {
                        cv_slab_store(pacc, slabs + (size_t)(G + g) * CV_SLAB);
                        next_tile(h - 1);
                    } else {
                        // (the range ends inside the tile it began in: a "middle", in the slab of heads)
                        cv_slab_store(pacc, slabs + (size_t)g * CV_SLAB);
                        cc = 0;
                    }
                }
            }
            for (; h < total; ++h) {
                const bool last = cc + 1 == nchunks;
                chunk(h, acc);
                if (last) {
                    cv_finish<EPI, HF, false, ADJ, false>(p, acc, tcur, false, nullptr, oscale, amax_run, bacc);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    next_tile(h);
                } else {
                    ++cc;
                }
            }
            if (sk && cc != 0) cv_slab_store(acc, slabs + (size_t)g * CV_SLAB);
        }
    } else {
        issue_a(tcur, cc, true);
        issue_w(wregA, tcur, cc, 0, true);
        issue_w(wregB, tcur, cc, 1, true);
        commit_a(0);
        commit_w(wregA, 0);
        commit_w(wregB, 1);
        __syncthreads();
        issue_w(wregA, tcur, cc, 2, true);
        Ops o0, o1;
        load_ops(o0, a_ptr(0, 0), w_ptr(0), 0);

        for (unsigned h = 0; h < total; ++h) {
            const bool more = h + 1 < total;
            const bool last = cc + 1 == nchunks;               // the next chunk opens the next tile
            const Tile tn = last ? tnext : tcur;
            const unsigned ccn = last ? 0u : cc + 1;
            // (the patch of the next chunk: requested in stage 2 AFTER that stage's weight request -- loads return in
            // order, and no weight stage may have to wait for the patch's HBM burst -- and written, round by round
            // between the taps' MFMAs, in stage 4: visible when stage 5 fetches the next chunk's first operands)
            auto stage = [&](Ops& cur, Ops& other, u32x4 (&wfill)[6], const u32x4 (&wdone)[6], const int st) {
                if (st < 3) issue_w(wfill, tcur, cc, st + 3, true);
                else issue_w(wfill, tn, ccn, st - 3, more);
                if (st == 2) issue_a(tn, ccn, more);
                const bool ca = st == 4;
                const int nbuf = (int)((h + 1) & 1u);
                const u32x4* Ab = a_ptr(h, st);
                const u32x4* Wb = w_ptr(st);
                load_ops(other, Ab, Wb, 1);
                mfmas(cur);
                interleave();
                if (ca) commit_a_round(nbuf, 0);
                load_ops(cur, Ab, Wb, 2);
                mfmas(other);
                interleave();
                if (ca) commit_a_round(nbuf, 1);
                if (st < 5) load_ops(other, a_ptr(h, st + 1), w_ptr(st + 1), 0);
                else load_ops(other, a_ptr(h + 1, 0), w_ptr(0), 0);
                mfmas(cur);
                interleave();
                if (ca) commit_a_round(nbuf, 2);
                commit_w(wdone, (st + 2) % 3);
                cv_lds_barrier();
            };
            stage(o0, o1, wregB, wregA, 0);
            stage(o1, o0, wregA, wregB, 1);
            stage(o0, o1, wregB, wregA, 2);
            stage(o1, o0, wregA, wregB, 3);
            stage(o0, o1, wregB, wregA, 4);
            stage(o1, o0, wregA, wregB, 5);
            if (last) {
                // (stream-K: the tail of a tile whose head other workgroups hold is parked until this one's range is through)
                const bool park = sk && ti == 0 && c0 > 0;
                cv_finish<EPI, HF, SK, ADJ>(p, acc, tcur, park, park ? slabs + (size_t)(G + g) * CV_SLAB : nullptr, oscale, amax_run, bacc);
            }
            if (last) {
                tcur = tnext;
                ++ti;
                tnext = tile_at(ti + 1);
                cc = 0;
            } else {
                ++cc;
            }
        }
        if (sk && cc != 0) {
            // the range ended inside a tile: its head (or middle) goes to this workgroup's "partial" slab; the fix-up kernel
            // (the next launch: no flags, no waiting) adds it to the tail its neighbour parked and runs that tile's epilogue
            cv_slab_store(acc, slabs + (size_t)g * CV_SLAB);
        }
    }
    if constexpr (EPI || ADJ) {
        if (p.amax) amax_publish(amax_run, p.amax);          // (a barrier inside)
    }
    if constexpr (ADJ) {
        __syncthreads();
        if (!producer)
            for (int i = tid; i < 2 * p.Cout; i += 256) p.adj_partial[(size_t)(2 * g) * p.Cout + i] = bacc[i];
    }
}

// ---- largest magnitude of a tensor (bit pattern; 0 for an empty or all-zero tensor) ----
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    unsigned m = 0;
    const long n4 = n / 4;
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const uint4 v = x4[i];
        const unsigned a = v.x & 0x7FFFFFFFu, b = v.y & 0x7FFFFFFFu, c = v.z & 0x7FFFFFFFu, d = v.w & 0x7FFFFFFFu;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        const unsigned q = ab > cd ? ab : cd;
        m = m > q ? m : q;
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - n4 * 4)) {
        const unsigned a = __builtin_bit_cast(unsigned, x[n4 * 4 + threadIdx.x]) & 0x7FFFFFFFu;
        m = m > a ? m : a;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, s, 64);
        m = m > o ? m : o;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ---- weights [cout][cin][3][3] (any strides) -> the two f16 planes in stage order ----
// entry e = ((((ct K16 + k16) 3 + ky) 3 + kx) 2 + plane) 2 + khalf) 128 + co holds input channels
// 16 k16 + 8 khalf .. + 7 of output channel 128 ct + co at tap (ky, kx) [flip: tap (2 - ky, 2 - kx)]
struct PrepParams {
    const float* w;
    long s_co, s_ci, s_ky, s_kx;
    int cout, cin, flip;
    const unsigned* wmax;
    u32x4* wp;
    float* wscale;
};
__global__ __launch_bounds__(256) void prep_weights_kernel(PrepParams p) {
    const float c = cv_scale_of(*p.wmax);
    const long K16 = p.cin / 16, units = (long)(p.cout / 128) * K16 * 9 * 2 * 128;
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u == 0) *p.wscale = c;
    if (u >= units) return;
    const int co = (int)(u % 128);
    long r = u / 128;
    const int khalf = (int)(r % 2);
    r /= 2;
    const int kx = (int)(r % 3);
    r /= 3;
    const int ky = (int)(r % 3);
    r /= 3;
    const int k16 = (int)(r % K16), ct = (int)(r / K16);
    const int sy = p.flip ? 2 - ky : ky, sx = p.flip ? 2 - kx : kx;
    const float* src = p.w + (long)(ct * 128 + co) * p.s_co + (long)(k16 * 16 + khalf * 8) * p.s_ci + sy * p.s_ky + sx * p.s_kx;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[i * p.s_ci] * c;
    u32x4 h, l;
    cv_split(v, h, l);
    const long base = ((((long)(ct * K16 + k16) * 3 + ky) * 3 + kx) * 2) * 256 + khalf * 128 + co;
    p.wp[base] = h;
    p.wp[base + 256] = l;
}

// compute units of the CURRENT device (cached per device ordinal: a process may hold several)
static int cu_count() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    if (dev >= 0 && dev < 64 && cache[dev]) return cache[dev];
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    else (void)hipGetLastError();
    if (dev >= 0 && dev < 64) cache[dev] = cus;
    return cus;
}

static bool conv3_dims_ok(int n, int h, int w, int cin, int cout) {
    if (n < 1 || h < 1 || w < 1 || cin < 32 || cout < 128 || cin % 32 || cout % 128) return false;
    // byte offsets inside one tile's rows stay below 2^31
    if ((long long)(CV_PS + 1) * w * (cin > cout ? cin : cout) * 4 >= 0x7FFFFFF0ll) return false;
    const long long tiles = (long long)n * ((h + CV_TS - 1) / CV_TS) * ((w + CV_TS - 1) / CV_TS) * (cout / 128);
    return tiles < 0x7FFFFFFFll / 64 && (long long)cout * cin * 9 * 4 < 0x7FFFFFF0ll;
}

}  // namespace sbmc

using namespace sbmc;

extern "C" int sbmc_conv3x3_supported(int n, int h, int w, int cin, int cout) {
    return conv3_dims_ok(n, h, w, cin, cout) ? 1 : 0;
}

extern "C" size_t sbmc_conv3x3_weights_bytes(int cin, int cout) {
    if (cin < 32 || cout < 128 || cin % 32 || cout % 128) return 0;
    // prepared planes + [scale, bit pattern of the largest magnitude]
    return (size_t)(cout / 128) * (cin / 16) * 3 * CV_WSTAGE * 16 + 16;
}

extern "C" int sbmc_conv3x3_absmax_f32(const float* x, long n, unsigned* out, void* stream) {
    if (n < 0 || (n && !x) || !out || (uintptr_t)x % 16) return SBMC_HIP_EINVAL;
    hipError_t e = hipMemsetAsync(out, 0, 4, (hipStream_t)stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    if (n == 0) return 0;
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return (int)hipGetLastError();
}

// the same, RAISING *out (an atomic maximum: *out keeps what it held if that is larger) -- the rows a neighbour sent
// folded into the word of the slab they pad (sbmc_amd/dist.py, torch.distributed transport)
extern "C" int sbmc_conv3x3_absmax_raise_f32(const float* x, long n, unsigned* out, void* stream) {
    if (n < 0 || (n && !x) || !out || (uintptr_t)x % 16) return SBMC_HIP_EINVAL;
    if (n == 0) return 0;
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return (int)hipGetLastError();
}

extern "C" int sbmc_conv3x3_prepare_weights_f32(const float* w, long s_co, long s_ci, long s_ky, long s_kx,
                                                 long storage_elems, int cin, int cout, int flip, void* wp,
                                                 void* stream) {
    const size_t bytes = sbmc_conv3x3_weights_bytes(cin, cout);
    if (!bytes || !w || !wp || (uintptr_t)wp % 16 || (uintptr_t)w % 16 || storage_elems < 1) return SBMC_HIP_EINVAL;
    char* tail = static_cast<char*>(wp) + bytes - 16;
    float* wscale = reinterpret_cast<float*>(tail);
    unsigned* wmax = reinterpret_cast<unsigned*>(tail + 4);
    int rc = sbmc_conv3x3_absmax_f32(w, storage_elems, wmax, stream);
    if (rc != 0) return rc;
    PrepParams p;
    p.w = w; p.s_co = s_co; p.s_ci = s_ci; p.s_ky = s_ky; p.s_kx = s_kx;
    p.cout = cout; p.cin = cin; p.flip = flip; p.wmax = wmax;
    p.wp = static_cast<u32x4*>(wp); p.wscale = wscale;
    const long units = (long)(cout / 128) * (cin / 16) * 9 * 2 * 128;
    hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

struct Conv3Adj { const unsigned* signs; float slope; float* partial; };

static int conv3_launch(const void* x, const unsigned* xmax, const void* wp, void* y, int n, int h, int w, int cin,
                        int cout, const float* bias, int act, float slope, unsigned* signs, unsigned* amax, bool epi,
                        void* ws, void* stream, bool hf = false, const Conv3Adj* adj = nullptr) {
    if (!conv3_dims_ok(n, h, w, cin, cout) || !x || (!xmax && !hf) || !wp || !y) return SBMC_HIP_EINVAL;
    if (adj && (epi || hf || cout > CV_ADJ_MAXC || !adj->signs || !adj->partial || !amax)) return SBMC_HIP_EINVAL;
    if ((uintptr_t)x % 16 || (uintptr_t)wp % 16) return SBMC_HIP_EINVAL;
    Conv3Params p;
    p.x = x; p.wp = static_cast<const u32x4*>(wp); p.y = y; p.xmax = xmax;
    p.wscale = reinterpret_cast<const float*>(static_cast<const char*>(wp) + sbmc_conv3x3_weights_bytes(cin, cout) - 16);
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout;
    p.tiles_x = (w + CV_TS - 1) / CV_TS; p.tiles_y = (h + CV_TS - 1) / CV_TS; p.ncot = cout / 128;
    p.ntiles = (unsigned)((long long)n * p.tiles_y * p.tiles_x * p.ncot);
    p.bias = bias; p.slope = act == 0 ? 1.f : (act == 1 ? 0.f : slope); p.signs = signs; p.amax = amax;
    p.adj_signs = adj ? adj->signs : nullptr; p.adj_slope = adj ? adj->slope : 1.f; p.adj_partial = adj ? adj->partial : nullptr;
    const int cus = cu_count();
    unsigned grid = p.ntiles < (unsigned)cus ? p.ntiles : (unsigned)cus;
    p.ws = nullptr;
    const unsigned rounds = (p.ntiles + (unsigned)cus - 1) / (unsigned)cus;
    // stream-K: (tile, chunk) units in equal ranges over ALL compute units, where whole tiles dealt round-robin
    //   * would leave more than 15 % of the chip idle in their last round (320 or 160 tiles on 256 CUs: the slabs of a
    //     sharded frame at the U-net's coarser levels), or
    //   * would take at least SBMC_CONV3X3_SK_SAVED (default 3) chunk times longer: a chunk (32 input channels of a
    //     tile) takes ~12 us whatever the layer, the slabs and the fix-up launch of a split launch ~30 us whatever its
    //     size -- so what decides is the ABSOLUTE time the last round wastes: 1800 tiles of 8 chunks (the whole frame
    //     at the U-net's second level) waste 7 chunk times of 64, the 480 tiles of a rank of 8 none (round 5 tried
    //     stream-K everywhere: -2.4 ms on the whole frame, +1-3 ms on a rank of 8).  0: this rule off.
    // Nearly every workgroup of a split launch holds a split tile, and the slabs they meet in are as many bytes as a
    // small convolution's own tensors.
    const unsigned long long units_all = (unsigned long long)p.ntiles * (unsigned)(cin / 32);
    const unsigned long long sk_chunks = (units_all + (unsigned)cus - 1) / (unsigned)cus;
    const unsigned long long rr_chunks = (unsigned long long)rounds * (unsigned)(cin / 32);
    const int sk_saved = env_knob("SBMC_CONV3X3_SK_SAVED", 3);
    const bool sk_pays = sk_saved > 0 && rr_chunks >= sk_chunks + (unsigned long long)sk_saved;
    if (ws != nullptr && (sk_pays || (unsigned long long)p.ntiles * 100 < (unsigned long long)rounds * (unsigned)cus * 85)) {
        if ((uintptr_t)ws % 16) return SBMC_HIP_EINVAL;
        const unsigned long long units = units_all;
        grid = units < (unsigned long long)cus ? (unsigned)units : (unsigned)cus;
        p.ws = ws;
    }
    auto kern = hf ? (epi ? conv3_kernel<true, true> : conv3_kernel<false, true>)
                   : (epi ? conv3_kernel<true, false> : conv3_kernel<false, false>);
    if (p.ws != nullptr)
        kern = hf ? (epi ? conv3_kernel<true, true, true> : conv3_kernel<false, true, true>)
                  : (epi ? conv3_kernel<true, false, true> : conv3_kernel<false, false, true>);
    if (adj) kern = p.ws != nullptr ? conv3_kernel<false, false, true, true> : conv3_kernel<false, false, false, true>;
    // wave-specialised form (fp32 activations): SBMC_CONV3X3_WS
    const bool wsp = !hf && env_knob("SBMC_CONV3X3_WS", 1) != 0;
    if (wsp) {
        if (adj) kern = p.ws != nullptr ? conv3_kernel<false, false, true, true, true> : conv3_kernel<false, false, false, true, true>;
        else if (p.ws != nullptr) kern = epi ? conv3_kernel<true, false, true, false, true> : conv3_kernel<false, false, true, false, true>;
        else kern = epi ? conv3_kernel<true, false, false, false, true> : conv3_kernel<false, false, false, false, true>;
    }
    const unsigned lds = CV_LDS_BYTES + (adj ? CV_ADJ_LDS_BYTES : 0u);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wsp ? 512 : 256), lds, (hipStream_t)stream, p);
    e = hipGetLastError();
    if (e != hipSuccess || p.ws == nullptr) return (int)e;
    auto fix = hf ? (epi ? conv3_fixup_kernel<true, true> : conv3_fixup_kernel<false, true>)
                  : (epi ? conv3_fixup_kernel<true, false> : conv3_fixup_kernel<false, false>);
    if (adj) fix = conv3_fixup_kernel<false, false, true>;
    hipLaunchKernelGGL(fix, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, grid);
    return (int)hipGetLastError();
}

// rows of the bias gradient's partial sums an ADJ launch may write (the caller hands them in zeroed): two per workgroup
// of the main kernel, and as many again for the fix-up launch of a split (stream-K) launch -- which the launcher decides,
// so the caller sizes for it
extern "C" int sbmc_conv3x3_adj_partial_rows(void) { return 4 * cu_count(); }

extern "C" size_t sbmc_conv3x3_workspace_bytes(void) {
    return (size_t)CV_WS_HDR + (size_t)2 * cu_count() * CV_SLAB * sizeof(float);
}

extern "C" int sbmc_conv3x3_nhwc_f32(const float* x, const unsigned* xmax, const void* wp, float* y, int n, int h,
                                      int w, int cin, int cout, void* ws, void* stream) {
    return conv3_launch(x, xmax, wp, y, n, h, w, cin, cout, nullptr, 0, 1.f, nullptr, nullptr, false, ws, stream);
}

extern "C" int sbmc_conv3x3_bias_act_nhwc_f32(const float* x, const unsigned* xmax, const void* wp, const float* bias,
                                               float* y, unsigned* signs, unsigned* amax, int n, int h, int w, int cin,
                                               int cout, int act, float slope, void* ws, void* stream) {
    if (!bias || act < 0 || act > 2 || (uintptr_t)signs % 4) return SBMC_HIP_EINVAL;
    // (*amax is RAISED to max |y|: the caller hands in a zeroed word -- one memset launch per convolution was a
    // cost that does not shrink with the slab of a sharded frame)
    return conv3_launch(x, xmax, wp, y, n, h, w, cin, cout, bias, act, slope, signs, amax, true, ws, stream);
}

// The data gradient of a convolution whose input is another convolution's activated output that nothing else reads,
// with that layer's activation adjoint in the epilogue (csrc: ADJ):  gz = (gy * w^T) . act'(z),  act' from the sign words
// the producing layer's forward left (sbmc_conv3x3_bias_act_nhwc_f32), partial[rows][cout]: the bias gradient's partial
// sums (sbmc_conv3x3_adj_partial_rows() rows, zeroed by the caller, who also adds them up -- e.g. through
// sbmc_conv3x3_wgrad_bias_f32), *amax raised to max |gz|.  x = gy [n, h, w, cin], wp: the adjoint's
// prepared weights, cout = the producing layer's output channels (<= 512).
extern "C" int sbmc_conv3x3_adj_nhwc_f32(const float* x, const unsigned* xmax, const void* wp, const unsigned* signs,
                                          float slope, float* y, float* partial, unsigned* amax, int n, int h, int w,
                                          int cin, int cout, void* ws, void* stream) {
    if ((uintptr_t)signs % 4) return SBMC_HIP_EINVAL;
    const Conv3Adj adj{signs, slope, partial};
    return conv3_launch(x, xmax, wp, y, n, h, w, cin, cout, nullptr, 0, 1.f, nullptr, amax, false, ws, stream, false, &adj);
}
extern "C" int sbmc_conv3x3_adj_supported(int n, int h, int w, int cin, int cout) {
    return (conv3_dims_ok(n, h, w, cin, cout) && cout <= sbmc::CV_ADJ_MAXC) ? 1 : 0;
}

// Half activations (torch.autocast(float16) semantics): x, y _Float16 channels-last; wp: the SAME prepared weights
// (their high plane is f16 of the scaled weight); bias fp32; fp32 accumulation, one rounding to half on the store.
extern "C" int sbmc_conv3x3_nhwc_f16(const void* x, const void* wp, void* y, int n, int h, int w, int cin, int cout,
                                      void* ws, void* stream) {
    return conv3_launch(x, nullptr, wp, y, n, h, w, cin, cout, nullptr, 0, 1.f, nullptr, nullptr, false, ws, stream, true);
}
extern "C" int sbmc_conv3x3_bias_act_nhwc_f16(const void* x, const void* wp, const float* bias, void* y, unsigned* signs,
                                               int n, int h, int w, int cin, int cout, int act, float slope, void* ws,
                                               void* stream) {
    if (!bias || act < 0 || act > 2 || (uintptr_t)signs % 4) return SBMC_HIP_EINVAL;
    return conv3_launch(x, nullptr, wp, y, n, h, w, cin, cout, bias, act, slope, signs, nullptr, true, ws, stream, true);
}

// =============================================================================================================
// Weight gradient:  gw[co][ci][ky][kx] = sum over pixels  gy[n][y][x][co] * x[n][y + ky - 1][x + kx - 1][ci]
//
// A GEMM whose REDUCTION runs over the pixels: the matrix instruction wants 8 consecutive pixels of one channel
// per lane, the channels-last image has them Cin floats apart.  The staging does the transposition: a thread owns
// a pixel octet and a channel quad (16-byte loads: 32 lanes = the 512 contiguous bytes of one pixel's 128
// channels), splits its 8 pixels x 4 channels and writes ONE 16-byte entry per channel and plane,
// [plane][pixel octet][channel] in LDS -- conflict free both ways (see the swizzle in the kernel).  The tap's
// column shift kx - 1 would break the octets' alignment: a workgroup owns ONE tap column and folds the shift into
// its addresses; the row shift ky - 1 is another row of a ring of four x rows in LDS.
//   * a workgroup owns 128 output x 128 input channels x the 3 taps of one kernel column (a wave: 64 x 64 x 3 =
//     12 accumulators of 32 x 32 -- 9 taps would be 288 registers, more than the 256 accumulation registers) and
//     a contiguous range of ROW STAGES (32 pixels of one image row: 2 k-steps of 16), walking down the rows: each
//     x row is fetched once and used by three gy rows; the three kx workgroups of a range are neighbours on one
//     XCD and read the same rows through its L2;
//   * per k-step a wave reads 4 gy operands + 12 x operands for 36 MFMAs; the next stages' rows are in flight
//     meanwhile (two register sets, no load under a branch);
//   * the ranges' partial sums go to a scratch buffer and a second kernel adds them in a FIXED order (no
//     atomics: the result does not depend on the run) and scales back.
namespace sbmc {

constexpr int WG_TW = 32;                       // pixels of a row stage
constexpr int WG_OCT = WG_TW / 8;
constexpr int WG_G = 2 * WG_OCT * 128;          // entries of a gy row: [plane][octet][co 128]
constexpr int WG_XR = 2 * WG_OCT * 128;         // entries of an x row: [plane][octet][ci 128]
constexpr unsigned WG_LDS_BYTES = (2 * WG_G + 4 * WG_XR) * 16;     // 98304: two gy rows + a ring of four x rows
constexpr int WG_TILE = 128 * 128 * 3;          // outputs of a workgroup

struct WgradParams {
    const void* gy;          // [N, H, W, Cout] float (HF: _Float16)
    const void* x;           // [N, H, W, Cin]
    float* partial;          // [combo][split][WG_TILE]
    const unsigned* gmax;    // bit patterns of max |gy|, max |x|
    const unsigned* xmax;
    int N, H, W, Cin, Cout;
    int nstrips, ncot, ncit, nsplit;
    unsigned long long total;   // row stages: N * nstrips * H
};

__device__ __forceinline__ unsigned cv_pack_hh(_Float16 a, _Float16 b) {
    h2 v;
    v[0] = a;
    v[1] = b;
    return __builtin_bit_cast(unsigned, v);
}

// HF: half activations -- gy and x are _Float16 tensors: the staging is a pure 8 x 4 transposition (no split, no scale,
// one plane), one matrix product per term; gw stays fp32.
// W8 (512 threads, fp32 activations): the same workgroup tile, LDS rows and pipeline on EIGHT waves of 64 x 32 channels x 3 taps
// (6 accumulators, 256 registers): waves 0-3 stage the rows AND multiply as before, waves 4-7 -- their SIMD partners -- only
// multiply: while a staging wave converts and writes, the matrix pipe of its SIMD runs its partner's MFMAs (one wave per SIMD
// issues in order and cannot overlap the two).  The partial sums keep their layout: wave (wm, wn4) writes what wave
// (wm, wn4 / 2) wrote as its block ni = wn4 % 2.
template <bool HF, bool W8 = false>
__global__ __launch_bounds__(W8 ? 512 : 256) void conv3_wgrad_kernel(WgradParams p) {
    static_assert(!(HF && W8), "W8: the fp32 form");
    constexpr int NI = W8 ? 1 : 2;                      // 32-channel blocks of x a wave multiplies
    constexpr unsigned ES = HF ? 2u : 4u;
    extern __shared__ float4 cv_lds[];
    u32x4* Gs = reinterpret_cast<u32x4*>(cv_lds);       // [2][WG_G]
    u32x4* Xs = Gs + 2 * WG_G;                          // [4][WG_XR]: ring of x rows
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id();
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;            // (W8: wn = 0 .. 3, a block of 32 input channels)
    const bool stager = !W8 || wave < 4;
    constexpr int WNC = W8 ? 32 : 64;                   // input channels per wn
    const float cg = HF ? 1.f : cv_scale_of(*p.gmax), cx = HF ? 1.f : cv_scale_of(*p.xmax);

    const unsigned logical = logical_block_id();
    const int ncombo = p.ncot * p.ncit * 3;
    const int combo = (int)(logical % (unsigned)ncombo), split = (int)(logical / (unsigned)ncombo);
    const int kx = combo % 3, cit = (combo / 3) % p.ncit, cot = combo / (3 * p.ncit);
    const unsigned long long t0 = p.total * (unsigned)split / (unsigned)p.nsplit;
    const unsigned long long t1 = p.total * (unsigned)(split + 1) / (unsigned)p.nsplit;

    // ---- loaders.  Waves 0, 1 stage x rows, waves 2, 3 gy rows: a thread owns pixel octet o (of 4) and channel
    // quad q (of 32) -- its loads are 16-byte pieces, 32 lanes cover the 512 contiguous bytes of one pixel's 128
    // channels -- and turns its 8 pixels x 4 channels into one 16-byte entry per channel and plane.  For x the 8
    // pixels are the octet shifted by kx - 1: the workgroup's tap column is folded into the addresses.  Entry of
    // channel 4 q + c inside its [128] row: 4 q + (c ^ ((q >> 1) & 3)) -- the four entries of a thread stay in its own
    // 64 bytes, but 8 neighbouring lanes of one store instruction hit 8 different 16-byte slots of the 128-byte bank
    // row, and the readers' lane groups still see 16 different slots of theirs. ----
    const bool xrole = wave < 2;
    const int uo = (tid & 127) >> 5, uq = tid & 31;
    const int usw = (uq >> 1) & 3;
    const int C = xrole ? p.Cin : p.Cout;
    const int shift = xrole ? kx - 1 : 0;               // first pixel of the stage's 32, relative to x0
    unsigned lvoff[8];                                  // byte offsets of the thread's pixels from that pixel
#pragma unroll
    for (int j = 0; j < 8; ++j) lvoff[j] = (unsigned)(uq * 4 + (8 * uo + j) * C) * ES;
    struct Rows { u32x4 v[8]; };
    // the stage the next request is for: walks down the rows of a strip, then to the next strip, the next image
    int qy, qx0, qn;
    {
        qy = (int)(t0 % (unsigned)p.H);
        const unsigned long long rest = t0 / (unsigned)p.H;
        qx0 = (int)(rest % (unsigned)p.nstrips) * WG_TW;
        qn = (int)(rest / (unsigned)p.nstrips);
    }
    // Row `row` (x role: of x, may lie outside the image; gy role: of gy) of strip x0: eight 16-byte loads under no
    // branch.  What lies outside the image reads as zero: the descriptor ends with the image row (and is empty for a
    // row outside or a request beyond the workgroup's range), the pixel before the row is masked.
    auto load_row = [&](Rows& rr, int n, int row, int x0, bool valid) {
        const int xs = x0 + shift;
        // (xs >= W: the last strip's single column shifted right -- nothing of the row is left, and the byte count
        // below must not wrap)
        const bool rowin = valid && row >= 0 && row < p.H && xs < p.W;
        const char* base = static_cast<const char*>(xrole ? p.x : p.gy) +
                           ((((long)n * p.H + (rowin ? row : 0)) * (long)p.W + xs) * (long)C + (xrole ? cit : cot) * 128) * ES;
        const rsrc_t r = cv_rsrc(rowin ? (const void*)base : p.x,
                                 rowin ? (unsigned)((p.W - xs) * C - (xrole ? cit : cot) * 128) * ES : 0u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool off = j == 0 && xs < 0 && uo == 0;
            if constexpr (HF) {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off ? CV_OOB : lvoff[j], 0, 0);
                rr.v[j][0] = v[0];
                rr.v[j][1] = v[1];
            } else {
                rr.v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, off ? CV_OOB : lvoff[j], 0, 0);
            }
        }
    };
    // One request = what stage (qn, qx0, qy) still lacks in the steady state: its gy row, and x row qy + 1 (rows
    // qy - 1 and qy are in the ring from the two stages before; a stage that opens a strip primes them itself)
    auto issue = [&](Rows& rr, bool valid) {
        const int y = qy, x0 = qx0, n = qn;
        if (++qy == p.H) {
            qy = 0;
            qx0 += WG_TW;
            if (qx0 >= p.nstrips * WG_TW) {
                qx0 = 0;
                ++qn;
            }
        }
        load_row(rr, n, xrole ? y + 1 : y, x0, valid);
    };
    // two scaled values -> their h and l halves, packed (element 0 in the low half)
    auto pair = [&](unsigned a, unsigned b, float c, unsigned& hp, unsigned& lp) {
        cv_split_pair(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), c, hp, lp);
    };
    // the thread's 8 pixels x 4 channels -> entries of row buffer `dst` ([plane][octet][128])
    // (channels ch0 .. ch1 - 1 of the thread's four: the steady state deals them out between its MFMA groups)
    auto write_row = [&](const Rows& rr, u32x4* dst, float c, const int ch0 = 0, const int ch1 = 4) {
        u32x4* d = dst + uo * 128 + 4 * uq;
#pragma unroll
        for (int ch = ch0; ch < ch1; ++ch) {
            u32x4 hh, ll;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (HF) {
                    // channel ch of pixels 2 i, 2 i + 1: halves of dword ch / 2 of their two loads
                    const unsigned a = rr.v[2 * i][ch >> 1], b = rr.v[2 * i + 1][ch >> 1];
                    hh[i] = (ch & 1) ? ((a >> 16) | (b & 0xFFFF0000u)) : ((a & 0xFFFFu) | (b << 16));
                } else {
                    const unsigned a = rr.v[2 * i][ch], b = rr.v[2 * i + 1][ch];
                    unsigned hp, lp;
                    pair(a, b, c, hp, lp);
                    hh[i] = hp;
                    ll[i] = lp;
                }
            }
            u32x4* e = d + (ch ^ usw);
            e[0] = hh;
            if constexpr (!HF) e[WG_OCT * 128] = ll;
        }
    };
    // stage k's rows: gy -> Gs[k % 2]; x row (y + 1) -> ring slot (k + 1) % 4 (y - 1, y: slots (k - 1) % 4, k % 4)
    auto commit = [&](const Rows& rr, unsigned k, const int ch0 = 0, const int ch1 = 4) {
        write_row(rr, xrole ? Xs + ((k + 1) & 3u) * WG_XR : Gs + (k & 1u) * WG_G, xrole ? cx : cg, ch0, ch1);
    };
    // where this lane's channel of a [128] row sits (the swizzle above), as a reader: channel w 64 + 32 i + l31
    const int rsw = (l31 & ~3) + ((l31 & 3) ^ ((l31 >> 3) & 3));

    f32x16 acc[2][NI][3];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][ky][r] = 0.f;

    // operands: gy (A) per k-step, x (B) per (k-step, ky) sub-step
    struct OpA { u32x4 h[2], l[2]; };
    struct OpB { u32x4 h[2], l[2]; };
    auto load_a = [&](OpA& o, unsigned k, int ks) {
        const u32x4* Gb = Gs + (k & 1u) * WG_G + (2 * ks + lhi) * 128 + wm * 64 + rsw;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            o.h[mi] = Gb[mi * 32];
            if constexpr (!HF) o.l[mi] = Gb[WG_OCT * 128 + mi * 32];
        }
    };
    auto load_b = [&](OpB& o, unsigned k, int ks, int ky) {
        const u32x4* Xb = Xs + ((k + (unsigned)ky + 3u) & 3u) * WG_XR + (2 * ks + lhi) * 128 + wn * WNC + rsw;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            o.h[ni] = Xb[ni * 32];
            if constexpr (!HF) o.l[ni] = Xb[WG_OCT * 128 + ni * 32];
        }
    };
    auto mfmas = [&](const OpA& a, const OpB& b, int ky) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni][ky] = cv_mfma(a.h[mi], b.h[ni], acc[mi][ni][ky]);
        if constexpr (!HF) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni][ky] = cv_mfma(a.h[mi], b.l[ni], acc[mi][ni][ky]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni][ky] = cv_mfma(a.l[mi], b.h[ni], acc[mi][ni][ky]);
        }
    };

    // Pipeline.  Stage k (= t - t0): its gy row and its last x row are REQUESTED in the middle of stage k - 3
    // (two register sets) and WRITTEN in the middle of stage k - 1, between two barriers: the first makes them
    // visible, the second -- at the end of stage k - 1 -- frees what stage k - 1 read.  Operands are fetched one
    // sub-step (12 MFMAs) ahead, across both barriers.  A stage that opens a strip (or the range) loads its first
    // two x rows itself, not overlapped (once or twice per workgroup).  An odd number of stages is rounded up:
    // the extra one reads zeros and adds nothing.
    const unsigned long long t1e = t1 + ((t1 - t0) & 1ull);
    Rows r0, r1;
    OpA a0, a1;
    OpB b0, b1;
    int sy, sx0, sn;                                    // the stage being computed
    {
        sy = (int)(t0 % (unsigned)p.H);
        const unsigned long long rest = t0 / (unsigned)p.H;
        sx0 = (int)(rest % (unsigned)p.nstrips) * WG_TW;
        sn = (int)(rest / (unsigned)p.nstrips);
    }
    // PURE (W8, waves 4-7): the stage without its staging -- the same barriers, operand fetches and MFMAs.
    auto stage = [&](auto pure_c, unsigned long long t, Rows& done, const unsigned k) {
        constexpr bool PURE = decltype(pure_c)::value;
        if (k == 0 || sy == 0) {
            // x rows sy - 1 and sy of a new strip
            cv_lds_barrier();
            if constexpr (!PURE) {
                Rows pr;
                load_row(pr, sn, sy - 1, sx0, xrole && t < t1);
                if (xrole) write_row(pr, Xs + ((k + 3u) & 3u) * WG_XR, cx);
                load_row(pr, sn, sy, sx0, xrole && t < t1);
                if (xrole) write_row(pr, Xs + (k & 3u) * WG_XR, cx);
            }
            __syncthreads();
            load_a(a0, k, 0);
            load_b(b0, k, 0, 0);
        }
        if (++sy == p.H) {
            sy = 0;
            sx0 += WG_TW;
            if (sx0 >= p.nstrips * WG_TW) {
                sx0 = 0;
                ++sn;
            }
        }
        // One wave per SIMD issues in order: a block of MFMAs followed by a block of conversions runs the two pipes one
        // after the other (measured: 53 % matrix-pipe busy = 2304 of 4224 cycles per stage).  The staging of the next
        // stage's rows (the split: ~6 vector instructions per MFMA, 8 LDS stores) and the operand fetches are therefore
        // DEALT OUT between this half's MFMAs; what it writes was last read before the previous stage's middle barrier.
        // (a channel's two LDS stores come BEFORE the next group's operand fetches in program order: LDS reads are
        // not moved above LDS stores, so a commit in one piece would hold back the fetches -- and the MFMAs behind
        // them -- until all of its conversions were through)
        constexpr int SUB = HF ? 4 : (W8 ? 6 : 12);    // MFMAs of a sub-step (one k-step of one tap)
        auto deal = [&](auto nv) {
            constexpr int nvalu = decltype(nv)::value;
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
                if constexpr (!PURE) {
                    __builtin_amdgcn_sched_group_barrier(0x002, W8 ? 2 * nvalu : nvalu, 0);   // VALU
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);         // DS write
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
            }
        };
        load_b(b1, k, 0, 1);
        if constexpr (!PURE) commit(done, k + 1, 0, 1);
        mfmas(a0, b0, 0);
        deal(std::integral_constant<int, 4>{});
        load_b(b0, k, 0, 2);
        if constexpr (!PURE) commit(done, k + 1, 1, 2);
        mfmas(a0, b1, 1);
        deal(std::integral_constant<int, 4>{});
        load_a(a1, k, 1);
        load_b(b1, k, 1, 0);
        if constexpr (!PURE) commit(done, k + 1, 2, 4);
        mfmas(a0, b0, 2);
        deal(std::integral_constant<int, HF ? 8 : 7>{});
        cv_lds_barrier();
        // (the next request's address arithmetic -- ~60 scalar instructions -- and its loads: behind the barrier, between
        // the second half's MFMAs)
        if constexpr (!PURE) issue(done, t + 3 < t1);
        load_b(b0, k, 1, 1);
        mfmas(a1, b1, 0);
        load_b(b1, k, 1, 2);
        mfmas(a1, b0, 1);
        load_a(a0, k + 1, 0);
        load_b(b0, k + 1, 0, 0);
        mfmas(a1, b1, 2);
        constexpr int HALF = 3 * SUB / 2;              // MFMAs of half the second half
        if constexpr (PURE) {
#pragma unroll
            for (int i = 0; i < 2 * HALF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
            }
        } else {
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
                __builtin_amdgcn_sched_group_barrier(0x004, HF ? 12 : (W8 ? 8 : 4), 0);   // SALU (the request's descriptor first)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
            }
#pragma unroll
            for (int i = 0; i < HALF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, HF ? 2 : 1, 0);    // VMEM read (its loads)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             // DS read
            }
        }
        cv_lds_barrier();
    };
    if (stager) {
        issue(r0, t0 < t1);
        commit(r0, 0);
        issue(r1, t0 + 1 < t1);
        issue(r0, t0 + 2 < t1);
        unsigned k = 0;
        for (unsigned long long t = t0; t < t1e; t += 2, k += 2) {
            stage(std::false_type{}, t, r1, k);
            stage(std::false_type{}, t + 1, r0, k + 1);
        }
    } else {
        unsigned k = 0;
        for (unsigned long long t = t0; t < t1e; t += 2, k += 2) {
            stage(std::true_type{}, t, r1, k);
            stage(std::true_type{}, t + 1, r0, k + 1);
        }
    }

    // ---- partial sums of this pixel range: [wave][mi][ni][ky][r][lane] ----
    // (W8: wave (wm, wn) holds block ni = wn % 2 of what wave (wm, wn / 2) of the four-wave form holds)
    const int wave4 = W8 ? wm + 2 * (wn >> 1) : wave, ni0 = W8 ? (wn & 1) : 0;
    float* out = p.partial + ((size_t)combo * p.nsplit + split) * WG_TILE + (size_t)wave4 * (12 * 16 * 64) + lane;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(((mi * 2 + ni0 + ni) * 3 + ky) * 16 + r) * 64] = acc[mi][ni][ky][r];
}

struct WreduceParams {
    const float* partial;
    float* gw;
    long s_co, s_ci, s_ky, s_kx;
    const unsigned* gmax;
    const unsigned* xmax;
    int ncit, nsplit;
    // optional: the bias gradient's per-workgroup partial sums [bchunks][bc] (bias_act_nhwc_bwd) -> gbias[bc], added up
    // in a fixed order by the extra row of workgroups blockIdx.y == ncombo (a reduction launch of its own otherwise)
    const float* bpartial;
    float* gbias;
    int bchunks, bc, ncombo;
};
__global__ __launch_bounds__(256) void conv3_wgrad_reduce_kernel(WreduceParams p) {
    if ((int)blockIdx.y == p.ncombo) {
        // 16 channels x 16 groups of chunks per workgroup, eight loads in flight per thread (a first version with 4
        // groups and one load at a time took 200 us for 2048 chunks -- 9 ms of a 240 ms step -- on latency alone)
        __shared__ float bred[256];
        const int c0 = (int)blockIdx.x * 16;
        if (c0 >= p.bc) return;
        const int ch = c0 + (threadIdx.x & 15), grp = threadIdx.x >> 4;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ch < p.bc) {
            int i = grp;
            for (; i + 16 * 7 < p.bchunks; i += 16 * 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += p.bpartial[(size_t)(i + 16 * u) * p.bc + ch];
            }
            for (; i < p.bchunks; i += 16) a[0] += p.bpartial[(size_t)i * p.bc + ch];
        }
        bred[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        if (grp == 0 && ch < p.bc) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) s += bred[g * 16 + (threadIdx.x & 15)];
            p.gbias[ch] = s;
        }
        return;
    }
    const unsigned e = blockIdx.x * 256u + threadIdx.x;              // element of the workgroup tile
    const int combo = blockIdx.y;
    if (e >= (unsigned)WG_TILE) return;
    const float* src = p.partial + (size_t)combo * p.nsplit * WG_TILE + e;
    float s = 0.f;
    for (int i = 0; i < p.nsplit; ++i) s += src[(size_t)i * WG_TILE];
    const float oscale = p.gmax ? (1.f / cv_scale_of(*p.gmax)) * (1.f / cv_scale_of(*p.xmax)) : 1.f;    // (half inputs: unscaled)
    const int lane = e & 63, r = (e >> 6) & 15;
    const int rest = e >> 10, ky = rest % 3, ni = (rest / 3) & 1, mi = (rest / 6) & 1, wave = rest / 12;
    const int kx = combo % 3, cit = (combo / 3) % p.ncit, cot = combo / (3 * p.ncit);
    const int co = cot * 128 + (wave & 1) * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int ci = cit * 128 + (wave >> 1) * 64 + ni * 32 + (lane & 31);
    p.gw[co * p.s_co + ci * p.s_ci + ky * p.s_ky + kx * p.s_kx] = s * oscale;
}

static bool wgrad_dims_ok(int n, int h, int w, int cin, int cout) {
    if (n < 1 || h < 1 || w < 1 || cin < 128 || cout < 128 || cin % 128 || cout % 128) return false;
    if ((long long)w * (cin > cout ? cin : cout) * 4 >= 0x7FFFFFF0ll) return false;
    return (cout / 128) * (cin / 128) * 3 <= 256 && (long long)n * h * ((w + WG_TW - 1) / WG_TW) < (1ll << 40);
}
static int wgrad_splits(int cin, int cout, long long total) {
    const int cus = cu_count();
    const int ncombo = (cout / 128) * (cin / 128) * 3;
    long long s = cus / ncombo;
    s = s < 1 ? 1 : s;
    return (int)(s > total ? total : s);
}

}  // namespace sbmc

extern "C" int sbmc_conv3x3_wgrad_supported(int n, int h, int w, int cin, int cout) {
    return wgrad_dims_ok(n, h, w, cin, cout) ? 1 : 0;
}
extern "C" size_t sbmc_conv3x3_wgrad_scratch_bytes(int n, int h, int w, int cin, int cout) {
    if (!wgrad_dims_ok(n, h, w, cin, cout)) return 0;
    const long long total = (long long)n * h * ((w + WG_TW - 1) / WG_TW);
    return (size_t)(cout / 128) * (cin / 128) * 3 * wgrad_splits(cin, cout, total) * WG_TILE * 4;
}
extern "C" int sbmc_conv3x3_wgrad_bias_f32(const float* gy, const unsigned* gmax, const float* x, const unsigned* xmax,
                                            float* gw, long s_co, long s_ci, long s_ky, long s_kx, void* scratch, int n,
                                            int h, int w, int cin, int cout, const float* bias_partial, int bias_chunks,
                                            int bias_c, float* gbias, void* stream);
extern "C" int sbmc_conv3x3_wgrad_f32(const float* gy, const unsigned* gmax, const float* x, const unsigned* xmax,
                                       float* gw, long s_co, long s_ci, long s_ky, long s_kx, void* scratch, int n,
                                       int h, int w, int cin, int cout, void* stream) {
    return sbmc_conv3x3_wgrad_bias_f32(gy, gmax, x, xmax, gw, s_co, s_ci, s_ky, s_kx, scratch, n, h, w, cin, cout, nullptr, 0,
                                       0, nullptr, stream);
}
static int wgrad_launch(const void* gy, const unsigned* gmax, const void* x, const unsigned* xmax,
                        float* gw, long s_co, long s_ci, long s_ky, long s_kx, void* scratch, int n,
                        int h, int w, int cin, int cout, const float* bias_partial, int bias_chunks,
                        int bias_c, float* gbias, void* stream, bool hf);
extern "C" int sbmc_conv3x3_wgrad_bias_f32(const float* gy, const unsigned* gmax, const float* x, const unsigned* xmax,
                                            float* gw, long s_co, long s_ci, long s_ky, long s_kx, void* scratch, int n,
                                            int h, int w, int cin, int cout, const float* bias_partial, int bias_chunks,
                                            int bias_c, float* gbias, void* stream) {
    return wgrad_launch(gy, gmax, x, xmax, gw, s_co, s_ci, s_ky, s_kx, scratch, n, h, w, cin, cout, bias_partial, bias_chunks,
                        bias_c, gbias, stream, false);
}
// half activations: gy, x _Float16; gw (and the bias gradient) fp32
extern "C" int sbmc_conv3x3_wgrad_bias_f16(const void* gy, const void* x, float* gw, long s_co, long s_ci, long s_ky,
                                            long s_kx, void* scratch, int n, int h, int w, int cin, int cout,
                                            const float* bias_partial, int bias_chunks, int bias_c, float* gbias,
                                            void* stream) {
    return wgrad_launch(gy, nullptr, x, nullptr, gw, s_co, s_ci, s_ky, s_kx, scratch, n, h, w, cin, cout, bias_partial,
                        bias_chunks, bias_c, gbias, stream, true);
}
static int wgrad_launch(const void* gy, const unsigned* gmax, const void* x, const unsigned* xmax,
                        float* gw, long s_co, long s_ci, long s_ky, long s_kx, void* scratch, int n,
                        int h, int w, int cin, int cout, const float* bias_partial, int bias_chunks,
                        int bias_c, float* gbias, void* stream, bool hf) {
    if (bias_partial && (!gbias || bias_chunks < 1 || bias_c < 1 || bias_c > 16 * (WG_TILE / 256))) return SBMC_HIP_EINVAL;
    if (!wgrad_dims_ok(n, h, w, cin, cout) || !gy || (!hf && (!gmax || !xmax)) || !x || !gw || !scratch) return SBMC_HIP_EINVAL;
    if ((uintptr_t)gy % 16 || (uintptr_t)x % 16 || (uintptr_t)scratch % 16) return SBMC_HIP_EINVAL;
    WgradParams p;
    p.gy = gy; p.x = x; p.partial = static_cast<float*>(scratch); p.gmax = gmax; p.xmax = xmax;
    p.N = n; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout;
    p.nstrips = (w + WG_TW - 1) / WG_TW; p.ncot = cout / 128; p.ncit = cin / 128;
    p.total = (unsigned long long)n * h * p.nstrips;
    p.nsplit = wgrad_splits(cin, cout, (long long)p.total);
    const int ncombo = p.ncot * p.ncit * 3;
    const bool w8 = !hf && env_knob("SBMC_CONV3X3_WGRAD_W8", 1) != 0;
    auto kern = hf ? conv3_wgrad_kernel<true> : (w8 ? conv3_wgrad_kernel<false, true> : conv3_wgrad_kernel<false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS_BYTES);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(ncombo * p.nsplit)), dim3(w8 ? 512 : 256), WG_LDS_BYTES, (hipStream_t)stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    WreduceParams q;
    q.partial = p.partial; q.gw = gw; q.s_co = s_co; q.s_ci = s_ci; q.s_ky = s_ky; q.s_kx = s_kx;
    q.gmax = gmax; q.xmax = xmax; q.ncit = p.ncit; q.nsplit = p.nsplit;
    q.bpartial = bias_partial; q.gbias = gbias; q.bchunks = bias_chunks; q.bc = bias_c; q.ncombo = ncombo;
    hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(WG_TILE / 256, (unsigned)(ncombo + (bias_partial ? 1 : 0))), dim3(256), 0,
                       (hipStream_t)stream, q);
    return (int)hipGetLastError();
}

// =============================================================================================================
// Weight bank: everything a training step derives from its weight-normalised convolution weights, for MANY
// layers per launch.  The reference wraps every convolution in torch's weight norm (sbmc/modules.py:85-94,
// 178-188: w = g v / ||v||, the norm over each output channel's cin x kh x kw block); per layer and step that
// was one weight_norm kernel forward, one backward, and -- for the 3 x 3 layers -- an absmax + a preparation
// pass (the two f16 planes in stage order) per direction: 8 launches for each of the 45 + 12 layers of the
// model.  Here: two launches forward (rows: norm, w, largest magnitude of a row; prepare: both orientations of
// every 3 x 3 layer, scaled by the largest of its rows' magnitudes), one backward (the adjoint of the weight norm
// for every layer), for up to SBMC_WBANK_MAX layers each.  Costs that do not shrink when a frame is sharded over
// several GPUs are the ones that bound its strong scaling (tools/rank_cost.py).
namespace sbmc {

struct BankFwdArgs {
    sbmc_wbank_entry e[SBMC_WBANK_MAX];
    int row0[SBMC_WBANK_MAX + 1];           // first row (output channel) of entry i in the launch's row space
    int blk0[2 * SBMC_WBANK_MAX + 1];       // first block of segment 2 i + flip of the preparation launch
    int n;
};

__device__ __forceinline__ float bank_block_sum(float v, float* red) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ unsigned bank_block_max(unsigned v, unsigned* red) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, s, 64);
        v = v > o ? v : o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    return a > b ? a : b;
}

// one workgroup per output channel: ||v||, w = v (g / ||v||), bit pattern of max |w| of the row
__global__ __launch_bounds__(256) void wbank_rows_kernel(BankFwdArgs a) {
    __shared__ float red[4];
    __shared__ unsigned redu[4];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.row0[i + 1]) ++i;
    const sbmc_wbank_entry& e = a.e[i];
    const int co = (int)blockIdx.x - a.row0[i];
    const long L = (long)e.cin * e.kh * e.kw;
    const float* v = e.v + (long)co * L;
    float* w = e.w + (long)co * L;
    float ss = 0.f;
    unsigned m = 0;
    for (long j = threadIdx.x; j < L; j += 256) {
        const float x = v[j];
        ss = fmaf(x, x, ss);
        const unsigned b = abits(x);
        m = m > b ? m : b;
    }
    const float norm = sqrtf(bank_block_sum(ss, red));
    m = bank_block_max(m, redu);
    const float s = e.g[co] / norm;
    for (long j = threadIdx.x; j < L; j += 256) w[j] = v[j] * s;
    if (threadIdx.x == 0) {
        e.norm[co] = norm;
        // |v| -> |v s| is monotone under rounding: the row's largest |w| is exactly this product
        reinterpret_cast<unsigned*>(e.norm)[e.cout + co] = abits(__builtin_bit_cast(float, m) * s);
    }
}

// the two f16 planes of w in stage order (see prep_weights_kernel), both orientations of every 3 x 3 entry
__global__ __launch_bounds__(256) void wbank_prep_kernel(BankFwdArgs a) {
    __shared__ unsigned redu[4];
    int seg = 0;
    while (seg + 1 < 2 * a.n && (int)blockIdx.x >= a.blk0[seg + 1]) ++seg;
    const sbmc_wbank_entry& e = a.e[seg >> 1];
    const int flip = seg & 1;
    const unsigned* rowmax = reinterpret_cast<const unsigned*>(e.norm) + e.cout;
    unsigned m = 0;
    for (int j = threadIdx.x; j < e.cout; j += 256) m = m > rowmax[j] ? m : rowmax[j];
    m = bank_block_max(m, redu);
    const float c = cv_scale_of(m);
    const int cout = flip ? e.cin : e.cout, cin = flip ? e.cout : e.cin;           // of the prepared orientation
    const long s_co = flip ? 9 : (long)e.cin * 9, s_ci = flip ? (long)e.cin * 9 : 9;
    u32x4* wp = static_cast<u32x4*>(flip ? e.wp_bwd : e.wp_fwd);
    const long K16 = cin / 16, units = (long)(cout / 128) * K16 * 9 * 2 * 128;
    const long u = (long)((int)blockIdx.x - a.blk0[seg]) * 256 + threadIdx.x;
    if (u == 0) {
        char* tail = reinterpret_cast<char*>(wp) + (size_t)(cout / 128) * (cin / 16) * 3 * CV_WSTAGE * 16;
        *reinterpret_cast<float*>(tail) = c;
        *reinterpret_cast<unsigned*>(tail + 4) = m;
    }
    if (u >= units) return;
    const int co = (int)(u % 128);
    long r = u / 128;
    const int khalf = (int)(r % 2);
    r /= 2;
    const int kx = (int)(r % 3);
    r /= 3;
    const int ky = (int)(r % 3);
    r /= 3;
    const int k16 = (int)(r % K16), ct = (int)(r / K16);
    const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
    const float* src = e.w + (long)(ct * 128 + co) * s_co + (long)(k16 * 16 + khalf * 8) * s_ci + sy * 3 + sx;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[i * s_ci] * c;
    u32x4 h, l;
    cv_split(v, h, l);
    const long base = ((((long)(ct * K16 + k16) * 3 + ky) * 3 + kx) * 2) * 256 + khalf * 128 + co;
    wp[base] = h;
    wp[base + 256] = l;
}

struct BankBwdArgs {
    sbmc_wbank_grad e[SBMC_WBANK_MAX];
    int row0[SBMC_WBANK_MAX + 1];
    int n;
};

// adjoint of w = g v / n, n = ||v||, one workgroup per output channel:
//   gg = <gw, v> / n,   gv = (g / n) (gw - v gg / n)
__global__ __launch_bounds__(256) void wbank_bwd_kernel(BankBwdArgs a) {
    __shared__ float red[4];
    int i = 0;
    while (i + 1 < a.n && (int)blockIdx.x >= a.row0[i + 1]) ++i;
    const sbmc_wbank_grad& e = a.e[i];
    const int co = (int)blockIdx.x - a.row0[i];
    const int taps = e.kh * e.kw;
    const long L = (long)e.cin * taps;
    const float* v = e.v + (long)co * L;
    float* gv = e.gv + (long)co * L;
    if (e.gw == nullptr) {                      // a layer the loss does not depend on
        for (long j = threadIdx.x; j < L; j += 256) gv[j] = 0.f;
        if (threadIdx.x == 0) e.gg[co] = 0.f;
        return;
    }
    const float* gw = e.gw + (long)co * e.s_co;
    auto gw_at = [&](long j) -> float {
        const long ci = j / taps;
        const int t = (int)(j - ci * taps), ky = t / e.kw, kx = t - ky * e.kw;
        return gw[ci * e.s_ci + ky * e.s_ky + kx * e.s_kx];
    };
    float dot = 0.f;
    for (long j = threadIdx.x; j < L; j += 256) dot = fmaf(gw_at(j), v[j], dot);
    dot = bank_block_sum(dot, red);
    const float rn = 1.f / e.norm[co];
    const float gg = dot * rn, s = e.g[co] * rn, t = gg * rn;
    for (long j = threadIdx.x; j < L; j += 256) gv[j] = s * (gw_at(j) - v[j] * t);
    if (threadIdx.x == 0) e.gg[co] = gg;
}

}  // namespace sbmc

extern "C" int sbmc_wbank_forward_f32(const sbmc_wbank_entry* entries, int n, void* stream) {
    if (n < 0 || n > SBMC_WBANK_MAX || (n && !entries)) return SBMC_HIP_EINVAL;
    if (n == 0) return 0;
    BankFwdArgs a;
    a.n = n;
    int rows = 0, blocks = 0;
    for (int i = 0; i < n; ++i) {
        const sbmc_wbank_entry& e = entries[i];
        if (!e.v || !e.g || !e.w || !e.norm || e.cout < 1 || e.cin < 1 || e.kh < 1 || e.kw < 1) return SBMC_HIP_EINVAL;
        if ((e.wp_fwd == nullptr) != (e.wp_bwd == nullptr)) return SBMC_HIP_EINVAL;
        if (e.wp_fwd) {
            if (e.kh != 3 || e.kw != 3 || !sbmc_conv3x3_weights_bytes(e.cin, e.cout) ||
                !sbmc_conv3x3_weights_bytes(e.cout, e.cin) || (uintptr_t)e.wp_fwd % 16 || (uintptr_t)e.wp_bwd % 16)
                return SBMC_HIP_EINVAL;
        }
        a.e[i] = e;
        a.row0[i] = rows;
        rows += e.cout;
        for (int flip = 0; flip < 2; ++flip) {
            a.blk0[2 * i + flip] = blocks;
            if (e.wp_fwd) blocks += (int)((long)e.cout * e.cin * 9 / 8 / 256);      // units of either orientation / 256
        }
    }
    a.row0[n] = rows;
    a.blk0[2 * n] = blocks;
    hipLaunchKernelGGL(wbank_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess || blocks == 0) return (int)err;
    hipLaunchKernelGGL(wbank_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int sbmc_wbank_backward_f32(const sbmc_wbank_grad* entries, int n, void* stream) {
    if (n < 0 || n > SBMC_WBANK_MAX || (n && !entries)) return SBMC_HIP_EINVAL;
    if (n == 0) return 0;
    BankBwdArgs a;
    a.n = n;
    int rows = 0;
    for (int i = 0; i < n; ++i) {
        const sbmc_wbank_grad& e = entries[i];
        if (!e.v || !e.g || !e.norm || !e.gv || !e.gg || e.cout < 1 || e.cin < 1 || e.kh < 1 || e.kw < 1) return SBMC_HIP_EINVAL;
        a.e[i] = e;
        a.row0[i] = rows;
        rows += e.cout;
    }
    a.row0[n] = rows;
    hipLaunchKernelGGL(wbank_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

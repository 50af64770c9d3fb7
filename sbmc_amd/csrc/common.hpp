// Shared device/host helpers for the gfx950 SBMC splat kernels.
//
// Tiling model used by every kernel in this library (wave64, CDNA4):
//   * a workgroup owns a tile of TX = 64 pixels (one per lane) by TY rows, one
//     wavefront per row, so every global access of the big [k*k, H, W] tensors
//     is a 256-byte contiguous segment per wave instruction;
//   * the small [C, H, W] operand that is read at a per-tap *offset* (sample
//     radiance in the forward, upstream gradients in the backward) is staged
//     once per workgroup in LDS with a (k-1)/2 halo, struct-of-arrays, so the
//     per-tap LDS reads are lane-consecutive (conflict-free ds_read_b32);
//   * workgroups are numbered so that each XCD (blockIdx % 8 on MI355X) walks a
//     contiguous band of image rows: x-adjacent tiles, which share the cache
//     lines that a misaligned 256-byte segment straddles, land in the same
//     XCD-private L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

namespace sbmc {

constexpr int TX = 64;             // pixels per tile row == wavefront width
constexpr int NUM_XCD = 8;         // MI355X: 8 XCDs, block b is placed on XCD b % 8
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float OUTSIDE_MAX = 1.0e30f;  // "max" of an out-of-image destination: exp(s - it) == 0

struct TileCoord {
    int n;   // batch index
    int y0;  // first row of the tile
    int x0;  // first column of the tile
};

// Host side: number of tiles for one [H, W] image plane.
static inline int tiles_x(int w) { return (w + TX - 1) / TX; }
static inline int tiles_y(int h, int ty) { return (h + ty - 1) / ty; }

// Device side: XCD-aware decode of the linear workgroup id.
// Physical block b runs on XCD b % 8 (observed placement -- used for speed only,
// correctness does not depend on it).  XCD x receives the logical tile range
// [x*q + min(x, r), ...) so that it walks consecutive tiles (x fastest, then y).
__device__ __forceinline__ unsigned logical_block_id() {
    const unsigned nb = gridDim.x;
    const unsigned b = blockIdx.x;
    const unsigned q = nb / NUM_XCD, r = nb % NUM_XCD;
    const unsigned xcd = b % NUM_XCD, i = b / NUM_XCD;
    return xcd * q + (xcd < r ? xcd : r) + i;
}

__device__ __forceinline__ TileCoord decode_tile(int ntx, int nty, int ty_rows) {
    const unsigned logical = logical_block_id();
    TileCoord t;
    const unsigned per_img = (unsigned)ntx * (unsigned)nty;
    t.n = (int)(logical / per_img);
    const unsigned rem = logical % per_img;
    t.y0 = (int)(rem / (unsigned)ntx) * ty_rows;
    t.x0 = (int)(rem % (unsigned)ntx) * TX;
    return t;
}

// Cooperative load of a haloed tile of one [H, W] plane into LDS.
//   dst[r * tw + cc] = inside ? src[(ys0 + r) * W + xs0 + cc] : fill
template <typename T>
__device__ __forceinline__ void stage_plane(float* __restrict__ dst,
                                            const T* __restrict__ src,
                                            int h, int w, int ys0, int xs0,
                                            int th, int tw, float fill) {
    const int nthreads = blockDim.x;
    for (int r = threadIdx.x / TX; r < th; r += nthreads / TX) {
        const int ys = ys0 + r;
        const bool yin = (ys >= 0) && (ys < h);
        const T* row = src + (size_t)(yin ? ys : 0) * w;
        for (int cc = threadIdx.x % TX; cc < tw; cc += TX) {
            const int xs = xs0 + cc;
            const bool in = yin && (xs >= 0) && (xs < w);
            dst[r * tw + cc] = in ? (float)row[xs] : fill;
        }
    }
}

// Raw buffer access (SRSRC descriptor in SGPRs): address = base + soffset(SGPR) + voffset(VGPR).
// All per-tap address arithmetic is scalar; the one per-lane offset register is shared by
// every load/store of a strip.  The hardware bounds check runs on voffset only, so a lane
// is switched off by giving it BUF_OOB: its loads return 0 and its stores are dropped.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned BUF_RANGE = 0x80000000u;  // num_records: valid voffsets are below this
constexpr unsigned BUF_OOB = 0x80000000u;    // voffset of a switched-off lane

__device__ __forceinline__ rsrc_t make_rsrc(const void* uniform_base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(uniform_base), /*stride*/ 0,
                                             BUF_RANGE, 0x00020000);
}
// AUX: cache-policy bits of the buffer instruction (0 = default, 2 = nt: streaming data that
// is touched once, so that it does not evict the small re-used operands from L2).
template <int AUX = 0>
__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, AUX);
}

// Logits / logit gradients come in two storage types: float (the reference's dtype) and
// _Float16 ("fp16 activations", BASELINE configs[4]); all arithmetic is fp32 either way.
template <typename T, int AUX = 0>
__device__ __forceinline__ float logit_load(rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
    } else {
        return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, AUX));
    }
}
// The same load without the conversion: half logits stay half until the arithmetic needs them -- each in the
// low half of a 32-bit register of its own (RawLogit<_Float16> = unsigned: as `_Float16` values the compiler
// packs two per VGPR and spends a v_perm + v_lshr per pair on it).
template <typename T> struct RawLogit { using type = float; };
template <> struct RawLogit<_Float16> { using type = unsigned; };
template <typename T, int AUX = 0>
__device__ __forceinline__ typename RawLogit<T>::type logit_load_raw(rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, AUX));
    } else {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, AUX);
    }
}
__device__ __forceinline__ _Float16 raw_half(unsigned u) { return __builtin_bit_cast(_Float16, (unsigned short)u); }
template <typename T, int AUX = 0>
__device__ __forceinline__ void logit_store(float v, rsrc_t r, unsigned voff, unsigned soff) {
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), r, voff, soff, AUX);
    }
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Orders this wave's LDS traffic: LDS operations of one wavefront execute in issue
// order, so lanes of a wave may exchange data through LDS without s_barrier -- but the
// compiler must be told not to move LDS accesses across the hand-off point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_id() {
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// Largest magnitude seen by a workgroup -> *amax (bit pattern; magnitudes order like unsigned integers).  The 3 x 3
// convolution that consumes the tensor takes its power-of-two scale from it (csrc/conv3x3.hip): the producer's
// pass finds it on the way instead of a pass of its own.
__device__ __forceinline__ unsigned abits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7FFFFFFFu; }
__device__ __forceinline__ unsigned amax4(unsigned m, const float4& v) {
    const unsigned a = abits(v.x), b = abits(v.y), c = abits(v.z), d = abits(v.w);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
    return m > q ? m : q;
}
__device__ __forceinline__ void amax_publish(unsigned m, unsigned* amax) {
    __shared__ unsigned wave_max[16];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, s, 64);
        m = m > o ? m : o;
    }
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup at most -- and none once the word already holds something at least as large (a
    // stale read only costs a superfluous atomic): a hundred thousand waves on one address would take a millisecond
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x + 63) / 64; ++w) m = m > wave_max[w] ? m : wave_max[w];
        if (m > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, m);
    }
}

// ---- split precision on the f16 matrix pipe (csrc/conv3x3.hip, csrc/pointwise.hip): a tensor is scaled by a power of two
// c that brings its largest magnitude into [2^14, 2^15), then c x = h + l + e with h = f16(c x), l = f16(c x - h),
// |e| <= max(2^-23 |c x|, 2^-25), and x w = (hx hw + hx lw + lx hw) / (cx cw) + a term <= 2^-22 |x w|: three f16 MFMAs with
// exact partial products, fp32 accumulation.
// power of two that brings a tensor whose largest magnitude has the bit pattern `maxbits` into [2^14, 2^15)
__device__ __forceinline__ float pow2_scale_of(unsigned maxbits) {
    int e = 268 - (int)(maxbits >> 23);      // 127 + 14 - (exponent - 127)
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    return __builtin_bit_cast(float, (unsigned)e << 23);
}
// Two fp32 values and their (wave-uniform, power-of-two) scale -> the packed f16 pairs h = f16(c v), l = f16(c v - h): FOUR
// instructions (v_fma_mix*_f16: an fp32 fused multiply-add whose addend is read as a half and whose result is rounded to
// half once) where multiply, convert, convert back, subtract, convert take eight.  Same values: c v and c v - h are exact.
__device__ __forceinline__ void f16_split_pair(float a, float b, float c, unsigned& h, unsigned& l) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(a), "v"(b), "s"(c));
}

// An integer development knob from the environment (never part of the ABI), read by the same rule as the host
// side's sbmc_amd/utils.py `knob`: unset -> fallback; "on" / "yes" / "true" -> 1; else atoi ("off", "no" -> 0).
static inline int env_knob(const char* name, int fallback) {
    const char* v = getenv(name);
    if (!v) return fallback;
    while (*v == ' ' || *v == '\t') ++v;
    // whole tokens, as sbmc_amd/utils.py knob compares them ("only" is not "on"): trailing blanks do not count
    size_t n = strlen(v);
    while (n > 0 && (v[n - 1] == ' ' || v[n - 1] == '\t' || v[n - 1] == '\n')) --n;
    if ((n == 2 && !strncasecmp(v, "on", 2)) || (n == 3 && !strncasecmp(v, "yes", 3)) || (n == 4 && !strncasecmp(v, "true", 4))) return 1;
    return atoi(v);
}

}  // namespace sbmc

#define SBMC_DISPATCH_C(CVAL, ...)                                   \
    switch (CVAL) {                                                  \
        case 1: { constexpr int C = 1; __VA_ARGS__; } break;         \
        case 2: { constexpr int C = 2; __VA_ARGS__; } break;         \
        case 3: { constexpr int C = 3; __VA_ARGS__; } break;         \
        case 4: { constexpr int C = 4; __VA_ARGS__; } break;         \
        case 5: { constexpr int C = 5; __VA_ARGS__; } break;         \
        case 6: { constexpr int C = 6; __VA_ARGS__; } break;         \
        case 7: { constexpr int C = 7; __VA_ARGS__; } break;         \
        case 8: { constexpr int C = 8; __VA_ARGS__; } break;         \
        default: return SBMC_HIP_EINVAL;                             \
    }

#define SBMC_DISPATCH_C4(CVAL, ...)                                  \
    switch (CVAL) {                                                  \
        case 1: { constexpr int C = 1; __VA_ARGS__; } break;         \
        case 2: { constexpr int C = 2; __VA_ARGS__; } break;         \
        case 3: { constexpr int C = 3; __VA_ARGS__; } break;         \
        case 4: { constexpr int C = 4; __VA_ARGS__; } break;         \
        default: return SBMC_HIP_EINVAL;                             \
    }

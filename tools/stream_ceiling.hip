// Micro-benchmark: what HBM rate does the access PATTERN of the backward strip kernel reach when
// nothing but the loads and stores is left?  (build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/stream_ceiling.hip -o /tmp/sc && /tmp/sc)
//  V0  linear float4 copy of the whole [441,H,W] tensor (the chip's copy ceiling)
//  V1  strip pattern: wave = 64-px row strip; for ky: 21 dword loads (planes ky*21..+20), 21 dword stores
//  V2  read-only strip pattern (forward-like, aligned)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int K = 21, H = 720, W = 1280;
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ inline rsrc_t mk(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000); }

__global__ __launch_bounds__(256) void v0_copy(const float4* a, float4* b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <bool STORE, int AUX>
__global__ __launch_bounds__(256, 8) void v1_strip(const float* S, float* D, float* sink) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned nb = gridDim.x, b = blockIdx.x, q = nb / 8, r = nb % 8, xcd = b % 8, i = b / 8;
    const long item = (long)(xcd * q + (xcd < r ? xcd : r) + i) * 4 + wv;
    const int nseg = W / 64;
    if (item >= (long)H * nseg) return;
    const int y = __builtin_amdgcn_readfirstlane((int)(item / nseg)), x0 = __builtin_amdgcn_readfirstlane((int)(item % nseg) * 64);
    const size_t hw = (size_t)H * W;
    const unsigned voff = lane * 4u, ps = (unsigned)hw * 4u;
    float acc = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        const rsrc_t rs = mk(S + (size_t)(ky * K) * hw + (size_t)y * W + x0);
        const rsrc_t ws = mk(D + (size_t)(ky * K) * hw + (size_t)y * W + x0);
        float v[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) v[kx] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, kx * ps, AUX));
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            if (STORE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[kx] * 1.0001f), ws, voff, kx * ps, AUX);
            else acc += v[kx];
        }
    }
    if (!STORE && acc == 12345.678f) sink[0] = acc;
}
int main() {
    const size_t n = (size_t)K * K * H * W;
    float *a, *b, *sink;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (H * (W / 64) + 3) / 4;
    auto run = [&](const char* name, int which, double bytes) {
        float best = 1e9, tot = 0; const int reps = 20;
        for (int i = 0; i < reps + 3; ++i) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(v0_copy, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4);
            if (which == 1) hipLaunchKernelGGL((v1_strip<true, 0>), dim3(grid), dim3(256), 0, 0, a, b, sink);
            if (which == 2) hipLaunchKernelGGL((v1_strip<false, 0>), dim3(grid), dim3(256), 0, 0, a, b, sink);
            if (which == 3) hipLaunchKernelGGL((v1_strip<true, 2>), dim3(grid), dim3(256), 0, 0, a, b, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i >= 3) { tot += ms; if (ms < best) best = ms; }
        }
        printf("%-34s avg %.4f ms  best %.4f ms  -> %.0f GB/s avg\n", name, tot / reps, best, bytes / (tot / reps) / 1e6);
        return 0;
    };
    run("V0 linear float4 copy (r+w)", 0, 2.0 * n * 4);
    run("V1 strip pattern copy (r+w)", 1, 2.0 * n * 4);
    run("V3 strip pattern copy, nt", 3, 2.0 * n * 4);
    run("V2 strip pattern read only", 2, 1.0 * n * 4);
    return 0;
}

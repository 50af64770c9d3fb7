// Dev microbenchmark: sustained issue rate of v_mfma_f32_32x32x2_f32 (the fp32 matrix instruction of the 1x1
// kernels) as a function of waves per SIMD and independent accumulator chains per wave.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_rate tools/mfma_rate.hip && build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CHAINS>
__global__ void rate_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[c][j] = (float)(c + j);
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[c][j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int CHAINS>
static void run(int waves_per_simd) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 64 * 4 * waves_per_simd;      // one workgroup per CU
    const int iters = 20000;
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<CHAINS>, dim3(cus), dim3(threads), 0, 0, out, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<CHAINS>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)cus * 4 * waves_per_simd * iters * 8.0 * CHAINS;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d  chains/wave %d : %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", waves_per_simd, CHAINS, ms, tf,
           100.0 * tf / 157.3);
    hipFree(out);
}

// The inner loop of pw_fwd_kernel without global memory: per k-step one ds_read2_b32 feeds the B operands of two
// MFMAs (two accumulator chains), the A operands sit in 64 registers.  PREFETCH = k-steps the LDS reads run ahead.
template <int PREFETCH, bool BARRIER>
__global__ __launch_bounds__(512) void lds_fed_kernel(float* out, int tiles, const float* w) {
    extern __shared__ float xs[];                      // [2][128][128]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ph = wave >> 2, l31 = lane & 31, lhi = lane >> 5;
    for (int i = threadIdx.x; i < 2 * 128 * 128; i += blockDim.x) xs[i] = (float)(i & 7);
    float a[64];
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) a[kk] = w[(kk * 64 + lane) & 1023];
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc0[j] = acc1[j] = 0.f;
    int buf = 0;
    for (int t = 0; t < tiles; ++t, buf ^= 1) {
        const float* xb = xs + buf * (128 * 128) + lhi * 128 + ph * 64 + l31;
        float b0[PREFETCH + 1], b1[PREFETCH + 1];
#pragma unroll
        for (int q = 0; q < PREFETCH; ++q) { b0[q] = xb[(2 * q) * 128]; b1[q] = xb[(2 * q) * 128 + 32]; }
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) {
            if (kk + PREFETCH < 64) {
                b0[(kk + PREFETCH) % (PREFETCH + 1)] = xb[(2 * (kk + PREFETCH)) * 128];
                b1[(kk + PREFETCH) % (PREFETCH + 1)] = xb[(2 * (kk + PREFETCH)) * 128 + 32];
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b0[kk % (PREFETCH + 1)], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b1[kk % (PREFETCH + 1)], acc1, 0, 0, 0);
            if (PREFETCH > 0 && (kk & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        if (BARRIER) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int PREFETCH, bool BARRIER>
static void run_lds() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int tiles = 2000;
    float *out, *w;
    (void)hipMalloc(&out, 4096);
    (void)hipMalloc(&w, 4096);
    (void)hipMemset(w, 0, 4096);
    auto kern = lds_fed_kernel<PREFETCH, BARRIER>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 131072, 0, out, 10, w);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 131072, 0, out, tiles, w);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)cus * 8 * tiles * 128.0;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    printf("LDS-fed, reads %d k-step(s) ahead, %s : %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", PREFETCH,
           BARRIER ? "barrier per tile" : "no barrier      ", ms, tf, 100.0 * tf / 157.3);
    (void)hipFree(out); (void)hipFree(w);
}

int main() {
    run_lds<0, false>(); run_lds<0, true>(); run_lds<1, false>(); run_lds<1, true>(); run_lds<2, true>(); run_lds<4, true>();

    for (int w : {1, 2, 4}) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}

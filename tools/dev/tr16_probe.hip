// Dev probe: lane map of ds_read_b64_tr_b16 (gfx950).  LDS holds a [rows][72] image of 16-bit values v = row * 100 + col;
// within a 16-lane group lane i points at row (i / 4), columns 4 (i % 4) .. + 3 of the group's 4 x 16 block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) lds[i] = (short)((i / 72) * 100 + (i % 72));
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
    // group g: rows 8 g .. 8 g + 3, columns 16 .. 31
    const short* a = lds + (8 * g + i / 4) * 72 + 16 + 4 * (i % 4);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
    for (int j = 0; j < 4; ++j) out[4 * lane + j] = r[j];
}
int main() {
    short* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}

// A bump allocator over ONE device allocation, as a torch pluggable allocator (tools/placement_experiment4.py): every tensor of the
// process is a 2 MB-aligned carve of the same region; nothing is ever given back.   hipcc -shared -fPIC -o arena_alloc.so arena_alloc.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
static char* base = nullptr;
static size_t cap = 0, used = 0;
static std::mutex mu;
extern "C" void* arena_malloc(ssize_t size, int device, hipStream_t stream) {
    std::lock_guard<std::mutex> g(mu);
    if (!base) {
        const char* gb = getenv("ARENA_GB");
        cap = (size_t)(gb ? atoi(gb) : 100) << 30;
        const char* contig = getenv("ARENA_CONTIGUOUS");
        hipError_t e = hipErrorUnknown;
        if (contig && atoi(contig)) e = hipExtMallocWithFlags((void**)&base, cap, hipDeviceMallocContiguous);
        if (e != hipSuccess) {
            if (contig && atoi(contig)) fprintf(stderr, "arena: contiguous allocation of %zu GB failed (%d), plain hipMalloc\n", cap >> 30, (int)e);
            (void)hipGetLastError();
            if (hipMalloc((void**)&base, cap) != hipSuccess) { fprintf(stderr, "arena: hipMalloc failed\n"); return nullptr; }
        }
        fprintf(stderr, "arena: %zu GB at %p\n", cap >> 30, (void*)base);
    }
    const size_t a = ((size_t)size + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (used + a > cap) { fprintf(stderr, "arena: out of space\n"); return nullptr; }
    void* p = base + used;
    used += a;
    return p;
}
extern "C" void arena_free(void* ptr, ssize_t size, int device, hipStream_t stream) {}

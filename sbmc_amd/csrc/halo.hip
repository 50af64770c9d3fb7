// Neighbour exchange of row halos between the ranks of one node, without a communication library on the
// data path: every rank owns a MAILBOX in uncached device memory, exported once through HIP IPC and mapped by
// its two neighbours; a `put` kernel packs rows of a tensor straight into the neighbour's mailbox over xGMI
// (peer stores) and raises a flag there; a `get` kernel waits for that flag, unpacks into the consumer's tensor
// (optionally adding, and copying the slab's own rows in the same launch) and returns the slot to its sender.
// One exchange = two launches per rank and no host round trip (torch.distributed P2P: ~50 us of launches,
// stream hand-overs and host time per exchange however few bytes it moves; the sharded 720p step has ~150).
//
// New functionality: the reference is single-process; its closest analogue is the overlapped tiling of
// scripts/denoise.py:54-93 (a 256-px halo recomputed per tile).  The state merge at the end of this file is
// the cross-rank form of the reference's running-softmax merge, sbmc/modules.py:450-471.
//
// Mailbox layout (sbmc_halo_bytes(slot_bytes, nslots)); direction d: 0 = up (rank - 1), 1 = down (rank + 1):
//   header, one 128-byte line per word:
//     ready[d]  number of messages the neighbour in direction d has delivered into ring d   (written by IT)
//     ack[d]    number of MY messages towards direction d that it has consumed              (written by IT)
//     err       non-zero once a wait has timed out                                           (local)
//     cnt[..]   block-completion counters of the running launch                             (local)
//   ring[d][nslots][slot_bytes]  messages from the neighbour in direction d, dense (chunk after chunk)
// Message i of a direction uses slot i % nslots; its sender first waits until message i - nslots has been
// consumed (ack), so neither side ever overruns the other.  Sequence numbers are kept by the caller (both ends
// of a link issue the same sequence of exchanges -- the condition any matched send / receive pair has).
//
// Visibility: payload and flags live in memory both GPUs map UNCACHED (hipDeviceMallocUncached: no L2 line of
// it exists anywhere); a block's payload stores are followed by a system-scope release fence before the block
// counts itself done, the last block raises the flag with a system-scope store; the consumer polls the flag
// relaxed, then one system-scope acquire.  Waits are bounded (timeout_ticks of the 100 MHz wall clock): a rank
// whose neighbour died sets `err` and runs on instead of hanging the GPU; the host reads `err` at its next
// synchronisation point (sbmc_halo_status).
#include "common.hpp"
#include <cstring>
#include "../../include/sbmc_hip.h"

namespace sbmc {
namespace halo {

constexpr int HDR = 4096;
constexpr int OFF_READY = 0, OFF_ACK = 256, OFF_ERR = 512, OFF_CNT = 640;
// one word per (ring, slot): the bit pattern of the largest magnitude of the tensor a message's rows were cut from
// (what csrc/conv3x3.hip scales its input by: the receiver's padded map must be scaled by the larger of its own and
// its neighbours').  Written by the sender with the payload, read by the receiver before it returns the slot.
constexpr int OFF_AMAX = 1024, MAX_SLOTS = 128;
constexpr int THREADS = 256;

__device__ __forceinline__ unsigned* word(char* box, int off) { return reinterpret_cast<unsigned*>(box + off); }

// Polls *flag (system scope, relaxed) until it has reached `need` (wrap-safe).  One lane of a block calls it.
// Once ANY wait of this mailbox has timed out (`err` set) no later wait spins again: the step is lost anyway (the
// host raises at its next status read), and ~150 exchanges of a step must not each sit out their own time-out.
__device__ __forceinline__ void wait_reached(unsigned* flag, unsigned need, unsigned* err, unsigned code,
                                             long long timeout_ticks) {
    const long long t0 = (long long)wall_clock64();
    for (unsigned polls = 0;; ++polls) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int)(v - need) >= 0) break;
        if ((polls & 63u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        __builtin_amdgcn_s_sleep(8);
        if ((long long)wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// The last block of a launch part to arrive (its payload traffic fenced) signals `flag` := value.
__device__ __forceinline__ void signal_when_all_done(unsigned* cnt, unsigned nblocks, unsigned* flag, unsigned value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // this thread's payload stores / loads are complete
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nblocks - 1) {
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct PutArgs {
    char* box;                 // my mailbox
    char* peer[2];             // the neighbours' mailboxes as mapped here (NULL: no neighbour)
    const char* src[2];        // first byte of what goes up / down
    long long pitch;           // bytes between the chunks of a source
    unsigned chunk_units;      // units (sizeof(U)) per chunk
    unsigned total_units;      // chunks * chunk_units
    unsigned seq[2];           // index of this message in its direction
    unsigned nslots;
    long long slot_bytes;
    long long timeout_ticks;
    const unsigned* amax;      // optional: bit pattern of the largest magnitude of the tensor the rows belong to
};

template <typename U>
__global__ void __launch_bounds__(THREADS) put_kernel(PutArgs a) {
    const int d = blockIdx.y;
    char* peer = a.peer[d];
    if (peer == nullptr) return;
    const unsigned seq = a.seq[d];
    if (threadIdx.x == 0 && seq >= a.nslots)          // the slot's previous message must have been consumed
        wait_reached(word(a.box, OFF_ACK + 128 * d), seq + 1 - a.nslots, word(a.box, OFF_ERR), 1u + d, a.timeout_ticks);
    __syncthreads();
    U* dst = reinterpret_cast<U*>(peer + HDR + ((long long)(1 - d) * a.nslots + seq % a.nslots) * a.slot_bytes);
    const char* src = a.src[d];
    for (unsigned i = blockIdx.x * THREADS + threadIdx.x; i < a.total_units; i += gridDim.x * THREADS) {
        const unsigned chunk = i / a.chunk_units, off = i - chunk * a.chunk_units;
        dst[i] = reinterpret_cast<const U*>(src + chunk * a.pitch)[off];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.amax != nullptr)
        __hip_atomic_store(word(peer, OFF_AMAX + 4 * (int)((1 - d) * a.nslots + seq % a.nslots)), *a.amax,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    signal_when_all_done(word(a.box, OFF_CNT + 4 * d), gridDim.x, word(peer, OFF_READY + 128 * (1 - d)), seq + 1);
}

struct GetArgs {
    char* box;
    char* peer[2];             // whom to return the slot to
    char* dst[2];              // where the rows from above / below land (NULL: no such neighbour)
    const char* add[2];        // optional: dst = add + received
    long long dst_pitch, add_pitch;
    unsigned chunk_units, total_units;
    unsigned seq[2];
    unsigned nslots;
    long long slot_bytes;
    long long timeout_ticks;
    char* body_dst;            // optional plain 2-d copy in the same launch (the slab's own rows);
    const char* body_src;      // body_src == NULL: the run is filled with zeros instead
    long long body_dst_pitch, body_src_pitch;
    unsigned body_chunk_units, body_total_units;
    unsigned halo_blocks;      // blocks (of gridDim.x) that work on a halo part; the body uses all of them
    unsigned* amax;            // optional: raised (atomic max) to the words the senders attached to their messages
};

template <typename U, int ADD>   // ADD: 0 copy, 1 float add, 2 half add (on the lanes of U)
__device__ __forceinline__ U combine(U got, U base) {
    if constexpr (ADD == 0) {
        return got;
    } else if constexpr (ADD == 1) {
        constexpr int N = sizeof(U) / 4;
        typedef float V __attribute__((ext_vector_type(N)));
        return __builtin_bit_cast(U, __builtin_bit_cast(V, got) + __builtin_bit_cast(V, base));
    } else {
        constexpr int N = sizeof(U) / 2;
        typedef _Float16 V __attribute__((ext_vector_type(N)));
        return __builtin_bit_cast(U, __builtin_bit_cast(V, got) + __builtin_bit_cast(V, base));
    }
}

template <typename U, int ADD>
__global__ void __launch_bounds__(THREADS) get_kernel(GetArgs a) {
    const int part = blockIdx.y;
    if (part == 2) {
        if (a.body_dst == nullptr) return;
        for (unsigned i = blockIdx.x * THREADS + threadIdx.x; i < a.body_total_units; i += gridDim.x * THREADS) {
            const unsigned chunk = i / a.body_chunk_units, off = i - chunk * a.body_chunk_units;
            U v{};
            if (a.body_src != nullptr) v = reinterpret_cast<const U*>(a.body_src + chunk * a.body_src_pitch)[off];
            reinterpret_cast<U*>(a.body_dst + chunk * a.body_dst_pitch)[off] = v;
        }
        return;
    }
    const int d = part;
    if (a.dst[d] == nullptr || blockIdx.x >= a.halo_blocks) return;
    const unsigned seq = a.seq[d];
    if (threadIdx.x == 0)
        wait_reached(word(a.box, OFF_READY + 128 * d), seq + 1, word(a.box, OFF_ERR), 3u + d, a.timeout_ticks);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const U* src = reinterpret_cast<const U*>(a.box + HDR + ((long long)d * a.nslots + seq % a.nslots) * a.slot_bytes);
    for (unsigned i = blockIdx.x * THREADS + threadIdx.x; i < a.total_units; i += a.halo_blocks * THREADS) {
        const unsigned chunk = i / a.chunk_units, off = i - chunk * a.chunk_units;
        U v = src[i];
        if constexpr (ADD != 0) v = combine<U, ADD>(v, reinterpret_cast<const U*>(a.add[d] + chunk * a.add_pitch)[off]);
        reinterpret_cast<U*>(a.dst[d] + chunk * a.dst_pitch)[off] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.amax != nullptr) {
        const unsigned theirs = __hip_atomic_load(word(a.box, OFF_AMAX + 4 * (int)(d * a.nslots + seq % a.nslots)),
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        atomicMax(a.amax, theirs);
    }
    signal_when_all_done(word(a.box, OFF_CNT + 8 + 4 * d), a.halo_blocks, word(a.peer[d], OFF_ACK + 128 * (1 - d)), seq + 1);
}

// ---- merge of the splat's running state across a slab boundary ------------------------------------------
// (reference sbmc/modules.py:450-471: M = max(m1, m2); sum = sum1 exp(m1 - M) + sum2 exp(m2 - M).)
// A rank's partial state lives on its slab extended by p rows towards each neighbour, [bs, C + 2, hd, w] with
// hd = top + rows + bot and channels (C of sum_r, sum_w, max_w).  Its overhang rows were `put` into the
// neighbours' mailboxes; this kernel merges what the neighbours sent into the slab's own first / last p rows:
// first the rows from above, then the rows from below (the order the torch composition used; a row both
// neighbours reach -- slabs thinner than 2p -- sees both), keeps what it received for the backward.

template <int C>
struct State { float v[C + 2]; };

template <int C>
__device__ __forceinline__ State<C> merge2(const State<C>& a, const State<C>& b) {
    State<C> o;
    const float m = fmaxf(a.v[C + 1], b.v[C + 1]);
    const float sa = __expf(a.v[C + 1] - m), sb = __expf(b.v[C + 1] - m);
#pragma unroll
    for (int j = 0; j <= C; ++j) o.v[j] = a.v[j] * sa + b.v[j] * sb;
    o.v[C + 1] = m;
    return o;
}

// adjoint of merge2: given g (gradient of the merged state), returns ga, gb
template <int C>
__device__ __forceinline__ void merge2_bwd(const State<C>& a, const State<C>& b, const State<C>& g,
                                           State<C>& ga, State<C>& gb) {
    const float ma = a.v[C + 1], mb = b.v[C + 1];
    const float m = fmaxf(ma, mb);
    const float sa = __expf(ma - m), sb = __expf(mb - m);
    float dsa = 0.f, dsb = 0.f;
#pragma unroll
    for (int j = 0; j <= C; ++j) {
        ga.v[j] = g.v[j] * sa;
        gb.v[j] = g.v[j] * sb;
        dsa += g.v[j] * a.v[j];
        dsb += g.v[j] * b.v[j];
    }
    dsa *= sa;
    dsb *= sb;
    const float dm = g.v[C + 1] - dsa - dsb;
    // torch.max(a, b): the gradient goes to the larger argument, half to each on a tie
    const float wa = ma > mb ? 1.f : (ma == mb ? 0.5f : 0.f);
    ga.v[C + 1] = dsa + dm * wa;
    gb.v[C + 1] = dsb + dm * (1.f - wa);
}

struct MergeArgs {
    char* box;
    char* peer[2];
    const float* ext;          // [bs, C + 2, hd, w]
    float* out;                // [bs, C + 2, rows, w]
    float* recv[2];            // [bs, C + 2, p, w] what arrived from above / below (kept for the backward)
    int bs, rows, w, p, top, bot;
    unsigned seq[2];
    unsigned nslots;
    long long slot_bytes;
    long long timeout_ticks;
};

template <int C>
__global__ void __launch_bounds__(THREADS) merge_fwd_kernel(MergeArgs a) {
    const int row = blockIdx.y, n = blockIdx.z;
    const int x = blockIdx.x * THREADS + threadIdx.x;
    const bool up = a.top > 0 && row < a.p, down = a.bot > 0 && row >= a.rows - a.p;
    if (threadIdx.x == 0) {
        if (up) wait_reached(word(a.box, OFF_READY), a.seq[0] + 1, word(a.box, OFF_ERR), 3u, a.timeout_ticks);
        if (down) wait_reached(word(a.box, OFF_READY + 128), a.seq[1] + 1, word(a.box, OFF_ERR), 4u, a.timeout_ticks);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const int hd = a.top + a.rows + a.bot;
    if (x < a.w) {
        State<C> s;
#pragma unroll
        for (int j = 0; j < C + 2; ++j) s.v[j] = a.ext[(((size_t)n * (C + 2) + j) * hd + a.top + row) * a.w + x];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (d == 0 ? !up : !down) continue;
            const int r = d == 0 ? row : row - (a.rows - a.p);
            const float* ring = reinterpret_cast<const float*>(
                a.box + HDR + ((long long)d * a.nslots + a.seq[d] % a.nslots) * a.slot_bytes);
            State<C> o;
#pragma unroll
            for (int j = 0; j < C + 2; ++j) {
                const size_t at = (((size_t)n * (C + 2) + j) * a.p + r) * a.w + x;
                o.v[j] = ring[at];
                a.recv[d][at] = o.v[j];
            }
            s = merge2<C>(s, o);
        }
#pragma unroll
        for (int j = 0; j < C + 2; ++j) a.out[(((size_t)n * (C + 2) + j) * a.rows + row) * a.w + x] = s.v[j];
    }
    // hand the slots back once every block that read them is through
    const unsigned per_dir = gridDim.x * (unsigned)a.p * gridDim.z;
    if (up) signal_when_all_done(word(a.box, OFF_CNT + 8), per_dir, word(a.peer[0], OFF_ACK + 128), a.seq[0] + 1);
    if (down) signal_when_all_done(word(a.box, OFF_CNT + 12), per_dir, word(a.peer[1], OFF_ACK), a.seq[1] + 1);
}

struct MergeBwdArgs {
    const float* ext;          // [bs, C + 2, hd, w] the forward's input
    const float* recv[2];      // what the forward received
    const float* gout;         // [bs, C + 2, rows, w]
    float* gext;               // [bs, C + 2, hd, w]: rows [top, top + rows) are written here
    float* grecv[2];           // [bs, C + 2, p, w]: gradient of what was received (goes back to its sender)
    int bs, rows, w, p, top, bot;
};

template <int C>
__global__ void __launch_bounds__(THREADS) merge_bwd_kernel(MergeBwdArgs a) {
    const int row = blockIdx.y, n = blockIdx.z;
    const int x = blockIdx.x * THREADS + threadIdx.x;
    if (x >= a.w) return;
    const bool up = a.top > 0 && row < a.p, down = a.bot > 0 && row >= a.rows - a.p;
    const int hd = a.top + a.rows + a.bot;
    State<C> s, g, o[2];
#pragma unroll
    for (int j = 0; j < C + 2; ++j) {
        s.v[j] = a.ext[(((size_t)n * (C + 2) + j) * hd + a.top + row) * a.w + x];
        g.v[j] = a.gout[(((size_t)n * (C + 2) + j) * a.rows + row) * a.w + x];
    }
    size_t at[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (d == 0 ? !up : !down) continue;
        const int r = d == 0 ? row : row - (a.rows - a.p);
        at[d] = ((size_t)n * (C + 2) * a.p + r) * a.w + x;
#pragma unroll
        for (int j = 0; j < C + 2; ++j) o[d].v[j] = a.recv[d][at[d] + (size_t)j * a.p * a.w];
    }
    // forward: s1 = up ? merge2(s, o0) : s;  out = down ? merge2(s1, o1) : s1
    State<C> s1 = up ? merge2<C>(s, o[0]) : s;
    State<C> g1 = g, go;
    if (down) {
        merge2_bwd<C>(s1, o[1], g, g1, go);
#pragma unroll
        for (int j = 0; j < C + 2; ++j) a.grecv[1][at[1] + (size_t)j * a.p * a.w] = go.v[j];
    }
    State<C> gs = g1;
    if (up) {
        merge2_bwd<C>(s, o[0], g1, gs, go);
#pragma unroll
        for (int j = 0; j < C + 2; ++j) a.grecv[0][at[0] + (size_t)j * a.p * a.w] = go.v[j];
    }
#pragma unroll
    for (int j = 0; j < C + 2; ++j) a.gext[(((size_t)n * (C + 2) + j) * hd + a.top + row) * a.w + x] = gs.v[j];
}

// *dst = 1 if a wait of this mailbox has timed out, else 0 (on the stream: the flag can ride in a collective)
__global__ void status_to_kernel(char* box, float* dst) {
    *dst = __hip_atomic_load(word(box, OFF_ERR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 ? 1.f : 0.f;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline unsigned blocks_for(unsigned units) {
    const unsigned b = (units + THREADS * 4 - 1) / (THREADS * 4);       // ~4 units per thread
    return b < 1 ? 1 : (b > 64 ? 64 : b);
}

}  // namespace halo
}  // namespace sbmc

using namespace sbmc::halo;

extern "C" {

size_t sbmc_halo_bytes(long long slot_bytes, int nslots) {
    if (slot_bytes <= 0 || nslots <= 0 || nslots > MAX_SLOTS || (slot_bytes & 15)) return 0;
    return (size_t)HDR + 2 * (size_t)nslots * (size_t)slot_bytes;
}

int sbmc_halo_alloc(size_t bytes, void** base, unsigned char* handle) {
    if (bytes < (size_t)HDR || base == nullptr || handle == nullptr) return SBMC_HIP_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == SBMC_HALO_HANDLE_BYTES, "IPC handle size");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return (int)e; }
    memcpy(handle, &h, sizeof(h));
    *base = p;
    return 0;
}

int sbmc_halo_free(void* base) {
    if (base == nullptr) return 0;
    const hipError_t e = hipFree(base);
    if (e != hipSuccess) (void)hipGetLastError();
    return (int)e;
}

int sbmc_halo_open(const unsigned char* handle, void** base) {
    if (handle == nullptr || base == nullptr) return SBMC_HIP_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    *base = p;
    return 0;
}

int sbmc_halo_close(void* peer_base) {
    if (peer_base == nullptr) return 0;
    const hipError_t e = hipIpcCloseMemHandle(peer_base);
    if (e != hipSuccess) (void)hipGetLastError();
    return (int)e;
}

int sbmc_halo_status(void* box, unsigned* err) {
    if (box == nullptr || err == nullptr) return SBMC_HIP_EINVAL;
    const hipError_t e = hipMemcpy(err, static_cast<char*>(box) + OFF_ERR, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) (void)hipGetLastError();
    return (int)e;
}

int sbmc_halo_status_to(void* box, float* dst, void* stream) {
    if (box == nullptr || dst == nullptr) return SBMC_HIP_EINVAL;
    hipLaunchKernelGGL(status_to_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<char*>(box), dst);
    return (int)hipGetLastError();
}

int sbmc_halo_put(void* box, void* up_box, void* down_box, const void* src_up, const void* src_down,
                  long long chunks, long long chunk_bytes, long long pitch, unsigned seq_up, unsigned seq_down,
                  int nslots, long long slot_bytes, long long timeout_ticks, const unsigned* amax, void* stream) {
    if (box == nullptr || chunks <= 0 || chunk_bytes <= 0 || nslots <= 0 || nslots > MAX_SLOTS) return SBMC_HIP_EINVAL;
    if (chunks * chunk_bytes > slot_bytes || chunks * chunk_bytes >= (1ll << 32)) return SBMC_HIP_EINVAL;
    if ((up_box && !src_up) || (down_box && !src_down)) return SBMC_HIP_EINVAL;
    if (!up_box && !down_box) return 0;
    PutArgs a;
    a.box = static_cast<char*>(box);
    a.peer[0] = static_cast<char*>(up_box);
    a.peer[1] = static_cast<char*>(down_box);
    a.src[0] = static_cast<const char*>(src_up);
    a.src[1] = static_cast<const char*>(src_down);
    a.pitch = pitch;
    a.seq[0] = seq_up;
    a.seq[1] = seq_down;
    a.nslots = (unsigned)nslots;
    a.slot_bytes = slot_bytes;
    a.timeout_ticks = timeout_ticks;
    a.amax = amax;
    const bool v16 = (chunk_bytes % 16 == 0) && (pitch % 16 == 0) && (!up_box || aligned16(src_up)) &&
                     (!down_box || aligned16(src_down));
    const int unit = v16 ? 16 : (chunk_bytes % 4 == 0 && pitch % 4 == 0 ? 4 : (chunk_bytes % 2 == 0 && pitch % 2 == 0 ? 2 : 1));
    a.chunk_units = (unsigned)(chunk_bytes / unit);
    a.total_units = (unsigned)(chunks * chunk_bytes / unit);
    const dim3 grid(blocks_for(a.total_units), 2);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (unit) {
        case 16: hipLaunchKernelGGL(put_kernel<uint4>, grid, dim3(THREADS), 0, s, a); break;
        case 4: hipLaunchKernelGGL(put_kernel<unsigned>, grid, dim3(THREADS), 0, s, a); break;
        case 2: hipLaunchKernelGGL(put_kernel<unsigned short>, grid, dim3(THREADS), 0, s, a); break;
        default: hipLaunchKernelGGL(put_kernel<unsigned char>, grid, dim3(THREADS), 0, s, a); break;
    }
    return (int)hipGetLastError();
}

int sbmc_halo_get(void* box, void* up_box, void* down_box, void* dst_up, void* dst_down,
                  const void* add_up, const void* add_down, int add_elem,
                  long long chunks, long long chunk_bytes, long long dst_pitch, long long add_pitch,
                  void* body_dst, const void* body_src, long long body_chunks, long long body_chunk_bytes,
                  long long body_dst_pitch, long long body_src_pitch,
                  unsigned seq_up, unsigned seq_down, int nslots, long long slot_bytes, long long timeout_ticks,
                  unsigned* amax, void* stream) {
    if (box == nullptr || nslots <= 0 || nslots > MAX_SLOTS) return SBMC_HIP_EINVAL;
    const bool any = dst_up != nullptr || dst_down != nullptr;
    if (any && (chunks <= 0 || chunk_bytes <= 0 || chunks * chunk_bytes > slot_bytes ||
                chunks * chunk_bytes >= (1ll << 32))) return SBMC_HIP_EINVAL;
    if ((dst_up && !up_box) || (dst_down && !down_box)) return SBMC_HIP_EINVAL;
    if (add_elem != 0 && add_elem != 2 && add_elem != 4) return SBMC_HIP_EINVAL;
    if (add_elem != 0 && ((dst_up && !add_up) || (dst_down && !add_down))) return SBMC_HIP_EINVAL;
    if (body_dst != nullptr && (body_chunks <= 0 || body_chunk_bytes <= 0 ||
                                body_chunks * body_chunk_bytes >= (1ll << 32))) return SBMC_HIP_EINVAL;
    if (!any && body_dst == nullptr) return 0;
    GetArgs a;
    a.box = static_cast<char*>(box);
    a.peer[0] = static_cast<char*>(up_box);
    a.peer[1] = static_cast<char*>(down_box);
    a.dst[0] = static_cast<char*>(dst_up);
    a.dst[1] = static_cast<char*>(dst_down);
    a.add[0] = static_cast<const char*>(add_up);
    a.add[1] = static_cast<const char*>(add_down);
    a.dst_pitch = dst_pitch;
    a.add_pitch = add_pitch;
    a.seq[0] = seq_up;
    a.seq[1] = seq_down;
    a.nslots = (unsigned)nslots;
    a.slot_bytes = slot_bytes;
    a.timeout_ticks = timeout_ticks;
    a.body_dst = static_cast<char*>(body_dst);
    a.body_src = static_cast<const char*>(body_src);
    a.body_dst_pitch = body_dst_pitch;
    a.body_src_pitch = body_src_pitch;
    a.amax = amax;
    bool v16 = true;
    int min_unit = add_elem ? add_elem : 1;
    if (any) {
        v16 = v16 && chunk_bytes % 16 == 0 && dst_pitch % 16 == 0 && (!dst_up || aligned16(dst_up)) &&
              (!dst_down || aligned16(dst_down));
        if (add_elem) v16 = v16 && add_pitch % 16 == 0 && (!dst_up || aligned16(add_up)) && (!dst_down || aligned16(add_down));
    }
    if (body_src == nullptr) body_src_pitch = 0;
    if (body_dst) v16 = v16 && body_chunk_bytes % 16 == 0 && body_dst_pitch % 16 == 0 && body_src_pitch % 16 == 0 &&
                        aligned16(body_dst) && aligned16(body_src);
    int unit = 16;
    if (!v16) {
        // the widest unit every run of bytes is a multiple of (and no narrower than the element an add works on)
        unit = 4;
        for (;;) {
            bool ok = true;
            if (any) ok = ok && chunk_bytes % unit == 0 && dst_pitch % unit == 0 && (!add_elem || add_pitch % unit == 0);
            if (body_dst) ok = ok && body_chunk_bytes % unit == 0 && body_dst_pitch % unit == 0 && body_src_pitch % unit == 0;
            if (ok || unit == 1) break;
            unit /= 2;
        }
        if (unit < min_unit) return SBMC_HIP_EINVAL;
    }
    a.chunk_units = any ? (unsigned)(chunk_bytes / unit) : 1;
    a.total_units = any ? (unsigned)(chunks * chunk_bytes / unit) : 0;
    a.body_chunk_units = body_dst ? (unsigned)(body_chunk_bytes / unit) : 1;
    a.body_total_units = body_dst ? (unsigned)(body_chunks * body_chunk_bytes / unit) : 0;
    // one grid: parts 0 / 1 = the two directions (halo_blocks blocks each: a few rows), 2 = the body (a plain
    // copy of the whole slab: enough blocks to stream at the HBM rate)
    // (at most 16 workgroups per direction wait for a flag: a waiting workgroup holds its CU slot, and where several
    // ranks share ONE device -- the 8-ranks-on-one-GPU dry runs -- waiting workgroups on every CU keep the other ranks'
    // whole-CU convolution workgroups from ever starting: a resource deadlock no real node can have, but a cheap one to
    // make unlikely)
    a.halo_blocks = blocks_for(a.total_units);
    if (a.halo_blocks > 16) a.halo_blocks = 16;
    unsigned bx = (a.body_total_units + THREADS * 8 - 1) / (THREADS * 8);
    bx = bx > 2048 ? 2048 : bx;
    if (bx < a.halo_blocks) bx = a.halo_blocks;
    const dim3 grid(bx, 3);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define SBMC_GET(U, ADD) hipLaunchKernelGGL((get_kernel<U, ADD>), grid, dim3(THREADS), 0, s, a)
    if (add_elem == 0) {
        switch (unit) {
            case 16: SBMC_GET(uint4, 0); break;
            case 4: SBMC_GET(unsigned, 0); break;
            case 2: SBMC_GET(unsigned short, 0); break;
            default: SBMC_GET(unsigned char, 0); break;
        }
    } else if (add_elem == 4) {
        if (unit == 16) SBMC_GET(uint4, 1); else SBMC_GET(unsigned, 1);
    } else {
        if (unit == 16) SBMC_GET(uint4, 2); else if (unit == 4) SBMC_GET(unsigned, 2); else SBMC_GET(unsigned short, 2);
    }
#undef SBMC_GET
    return (int)hipGetLastError();
}

int sbmc_halo_merge_state_fwd_f32(void* box, void* up_box, void* down_box, const float* ext, float* out,
                                  float* recv_up, float* recv_down, int bs, int c, int rows, int w, int p,
                                  int top, int bot, unsigned seq_up, unsigned seq_down, int nslots,
                                  long long slot_bytes, long long timeout_ticks, void* stream) {
    if (box == nullptr || ext == nullptr || out == nullptr || bs <= 0 || rows <= 0 || w <= 0 || p <= 0) return SBMC_HIP_EINVAL;
    if ((top != 0 && top != p) || (bot != 0 && bot != p) || rows < p) return SBMC_HIP_EINVAL;
    if ((top && (!up_box || !recv_up)) || (bot && (!down_box || !recv_down))) return SBMC_HIP_EINVAL;
    if ((long long)bs * (c + 2) * p * w * 4 > slot_bytes) return SBMC_HIP_EINVAL;
    MergeArgs a;
    a.box = static_cast<char*>(box);
    a.peer[0] = static_cast<char*>(up_box);
    a.peer[1] = static_cast<char*>(down_box);
    a.ext = ext;
    a.out = out;
    a.recv[0] = recv_up;
    a.recv[1] = recv_down;
    a.bs = bs; a.rows = rows; a.w = w; a.p = p; a.top = top; a.bot = bot;
    a.seq[0] = seq_up;
    a.seq[1] = seq_down;
    a.nslots = (unsigned)nslots;
    a.slot_bytes = slot_bytes;
    a.timeout_ticks = timeout_ticks;
    const dim3 grid((w + THREADS - 1) / THREADS, rows, bs);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SBMC_DISPATCH_C(c, hipLaunchKernelGGL(merge_fwd_kernel<C>, grid, dim3(THREADS), 0, s, a));
    return (int)hipGetLastError();
}

int sbmc_halo_merge_state_bwd_f32(const float* ext, const float* recv_up, const float* recv_down,
                                  const float* gout, float* gext, float* grecv_up, float* grecv_down,
                                  int bs, int c, int rows, int w, int p, int top, int bot, void* stream) {
    if (ext == nullptr || gout == nullptr || gext == nullptr || bs <= 0 || rows <= 0 || w <= 0 || p <= 0) return SBMC_HIP_EINVAL;
    if ((top != 0 && top != p) || (bot != 0 && bot != p) || rows < p) return SBMC_HIP_EINVAL;
    if ((top && (!recv_up || !grecv_up)) || (bot && (!recv_down || !grecv_down))) return SBMC_HIP_EINVAL;
    MergeBwdArgs a;
    a.ext = ext;
    a.recv[0] = recv_up;
    a.recv[1] = recv_down;
    a.gout = gout;
    a.gext = gext;
    a.grecv[0] = grecv_up;
    a.grecv[1] = grecv_down;
    a.bs = bs; a.rows = rows; a.w = w; a.p = p; a.top = top; a.bot = bot;
    const dim3 grid((w + THREADS - 1) / THREADS, rows, bs);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SBMC_DISPATCH_C(c, hipLaunchKernelGGL(merge_bwd_kernel<C>, grid, dim3(THREADS), 0, s, a));
    return (int)hipGetLastError();
}

}  // extern "C"

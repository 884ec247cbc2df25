// tests/emu/hip/hip_runtime.h - HOST EMULATION of the small HIP surface the generated header uses.
// TEST INFRASTRUCTURE ONLY (lives under tests/): lets `g++ -std=c++20 -Itests/emu` compile the *unchanged* generated
// grid.cuh and run its kernels on the CPU so that the generated algorithm/indexing/topology logic can be checked against
// the oracle without a GPU (`pytest -m "not gpu"`).  One OS thread per GPU thread of a block, blocks run one after
// another, the wave-level sync is a barrier over the block's threads.  It is never used by the product.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidConfiguration = 9, hipErrorNotSupported = 801, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };

inline const char *hipGetErrorString(hipError_t) { return "emulated hip error"; }
inline hipError_t hipDeviceReset() { return hipSuccess; }
inline thread_local int g_emu_device = 0;  // (the current device is per host thread, as in HIP)
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= 8) return hipErrorInvalidValue; g_emu_device = d; return hipSuccess; }  // (8 pretend devices sharing the host)
inline hipError_t hipGetDevice(int *d) { *d = g_emu_device; return hipSuccess; }
// the emulated device has 8 GiB: lets the out-of-memory return path of the C ABI be exercised without a GPU
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = *total_b = (size_t)8 << 30; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = (n > ((size_t)8 << 30)) ? nullptr : calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
// (every emulated allocation counts as page-locked host memory: the chunked host path of the C ABI runs in the emulation)
enum { hipMemoryTypeHost = 1 };
struct hipPointerAttribute_t { int type; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeHost; return hipSuccess; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
typedef void *hipEvent_t;  // (every emulated launch is synchronous: events are no-ops)
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }

// dynamic LDS of the (single) running block
__attribute__((aligned(16))) unsigned char grid_smem_raw[160 * 1024];  // (one definition: the emulation harness is a single translation unit)

namespace hipemu {
inline std::barrier<> *g_barrier = nullptr;
inline void wave_barrier() { g_barrier->arrive_and_wait(); }

// mode 0: all threads of a block run concurrently, the wave-level sync is a barrier over the whole block (stricter ordering than the
//         hardware gives: hides nothing the generated code relies on, but also cannot show a cross-wave race);
// mode 1: the waves (64 consecutive threads) of a block run ONE AFTER ANOTHER, LAST WAVE FIRST, each with its own barrier - legal for
//         kernels that never synchronise across waves (the lane-group kernels: no __syncthreads), and it makes any read of another
//         wave's LDS data deterministic: the producer has not run yet, the reader sees the NaN poison;
// mode 2: like mode 1 but the FIRST wave of every block does not run at all: whatever lands in the records that wave owns was written by
//         another wave (from LDS the owner never filled).
inline int g_mode = 0;
template <typename F>
void launch(dim3 grid, dim3 block, size_t smem_bytes, F &&body) {
    if (smem_bytes > sizeof(grid_smem_raw)) { fprintf(stderr, "hipemu: dynamic LDS request too large\n"); abort(); }
    const unsigned nthreads = block.x * block.y * block.z;
    static std::mutex one_launch_at_a_time;  // the emulated "device" is one block's worth of global state: host threads take turns
    std::lock_guard<std::mutex> lock(one_launch_at_a_time);
    for (unsigned by = 0; by < grid.y; by++)
        for (unsigned bx = 0; bx < grid.x; bx++) {
            // LDS contents are undefined at block start on the hardware: poison them (NaN pattern) so that a read-before-write, or a
            // hand-off that only works because a previous block left the same values behind, shows up as NaN instead of passing by luck
            for (size_t w = 0; w + 4 <= smem_bytes; w += 4) { const unsigned poison = 0x7fc0dead; memcpy(grid_smem_raw + w, &poison, 4); }  // (the block's own allocation)
            const unsigned chunk = g_mode >= 1 ? 64u : nthreads;
            const unsigned nchunks = (nthreads + chunk - 1) / chunk;
            for (unsigned ci = 0; ci < nchunks; ci++) {
                const unsigned c = g_mode >= 1 ? nchunks - 1 - ci : ci;
                if (g_mode == 2 && c == 0) continue;
                const unsigned t0 = c * chunk, t1 = (t0 + chunk < nthreads) ? t0 + chunk : nthreads;
                std::barrier<> bar(t1 - t0);
                g_barrier = &bar;
                std::vector<std::thread> ts;
                for (unsigned t = t0; t < t1; t++)
                    ts.emplace_back([&, t]() {
                        threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                        blockIdx = dim3(bx, by, 0);
                        blockDim = block;
                        gridDim = grid;
                        body();
                        bar.arrive_and_drop();  // a retiring thread no longer takes part in wave-level syncs
                    });
                for (auto &t : ts) t.join();
            }
        }
}
}  // namespace hipemu

extern "C" inline __attribute__((visibility("default"), used)) void hipemu_set_mode(int mode) { hipemu::g_mode = mode; }
extern "C" inline __attribute__((visibility("default"), used)) int hipemu_get_device() { return g_emu_device; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
// cross-lane shuffle inside groups of `width` consecutive threads (width-aligned): exchange through a scratch array
namespace hipemu {
template <typename V> inline V g_shfl[1024];  // (one exchange buffer per value type: the double instantiations shuffle doubles)
template <typename V>
inline V shfl(V v, int src, int width) {
    const unsigned t = threadIdx.x + threadIdx.y * blockDim.x;
    g_shfl<V>[t] = v;
    wave_barrier();
    const V r = g_shfl<V>[(t & ~(unsigned)(width - 1)) + (unsigned)src];
    wave_barrier();
    return r;
}
}  // namespace hipemu
#define __shfl(v, src, width) hipemu::shfl((v), (src), (width))
inline int __lane_id() { return static_cast<int>((threadIdx.x + threadIdx.y * blockDim.x) & 63u); }  // lane inside the wavefront (waves = 64 consecutive threads of a block)
// DPP row shifts (row_shr:K = 0x110+K reads lane-K, row_shl:K = 0x100+K reads lane+K, inside rows of 16 consecutive lanes); a source
// outside the row yields 0 with bound_ctrl, else `old`.  Only what the generated code uses (full row/bank masks).
namespace hipemu {
inline int g_dpp[1024];
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (row_mask != 0xf || bank_mask != 0xf) { fprintf(stderr, "hipemu: DPP row/bank masks are not emulated\n"); abort(); }
    const unsigned t = threadIdx.x + threadIdx.y * blockDim.x;
    g_dpp[t] = src;
    wave_barrier();
    int from;
    if (ctrl >= 0x111 && ctrl <= 0x11f) from = (int)(t & 15u) - (ctrl - 0x110);
    else if (ctrl >= 0x101 && ctrl <= 0x10f) from = (int)(t & 15u) + (ctrl - 0x100);
    else { fprintf(stderr, "hipemu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
    const int r = (from >= 0 && from < 16) ? g_dpp[(t & ~15u) + (unsigned)from] : (bound_ctrl ? 0 : old);
    wave_barrier();
    return r;
}
}  // namespace hipemu
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu::update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_barrier()
inline void __syncthreads() { hipemu::wave_barrier(); }  // (mode 0: the barrier spans the block's threads; kernels that use it must run in that mode)
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) hipemu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })

// grid_capi.hip - C-ABI shim over the generated, robot-specialised HIP header (see include/grid_capi.h).
// Built once per robot:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -I<dir of generated grid.cuh> -I<repo>/include grid_capi.hip
//
// Error model: the generated host API follows the reference (gpuAssert prints and exit()s, reference GRiDCodeGenerator.py:279-286).
// This library must never take the host process down, so it re-binds the header's error hook to a C++ exception that every
// entry point catches and turns into the hipError_t return value.
#include <hip/hip_runtime.h>

struct grid_capi_error {
    hipError_t code;
    const char *file;
    int line;
};
#define GRID_ON_GPU_ERROR(code, file, line) throw grid_capi_error{(code), (file), (line)}

#include "grid.cuh"
#include "grid_capi.h"

#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#ifndef GRID_ROBOT_NAME
#define GRID_ROBOT_NAME "robot"
#endif

template <typename T>
struct grid_typed {
    grid::robotModel<T> *d_robotModel = nullptr;
    grid::gridData<T> *hd_data = nullptr;
};

struct grid_handle {
    int device;
    int max_timesteps;
    int blocks;   // 0 = derive from the batch
    int threads;  // 0 = the kernel's suggested block size
    int host_chunks;  // 0 = automatic: chunks of the pipelined host entry point (page-locked buffers)
    hipStream_t *streams;
    grid_typed<float> f32;
    grid_typed<double> f64;  // allocated by the first *_f64 call
    std::mutex alloc_lock;   // serialises that lazy allocation (two threads making their first *_f64 call on one handle)
    // GRID_SO_DIRECT (records beyond the LDS of a CU): fdsva_so keeps the idsva_so tensors in the handle's ONE d_idsva_so workspace.  Calls on different streams
    // (or threads) are made safe by ordering them: every launch waits for the previous one's event before it may touch the workspace
    hipEvent_t so_done = nullptr;
    bool so_pending = false;
};
template <typename T> static inline grid_typed<T> &typed(grid_handle *h);
template <> inline grid_typed<float> &typed<float>(grid_handle *h) { return h->f32; }
template <> inline grid_typed<double> &typed<double>(grid_handle *h) { return h->f64; }

static thread_local char g_err[512] = "";

static int fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
static int fail_msg(hipError_t e, const char *msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return (int)e;
}
#define GRID_TRY(expr)                                   \
    do {                                                 \
        hipError_t e__ = (expr);                         \
        if (e__ != hipSuccess) return fail(e__, #expr);  \
    } while (0)
// every entry point body runs inside this: errors raised by the generated host API come back as return codes
#define GRID_GUARDED(body)                                                                                                       \
    try {                                                                                                                        \
        body                                                                                                                     \
    } catch (const grid_capi_error &e) {                                                                                         \
        snprintf(g_err, sizeof(g_err), "%s (%s:%d)", hipGetErrorString(e.code), e.file, e.line);                                 \
        return (int)e.code;                                                                                                      \
    } catch (const std::exception &e) {                                                                                          \
        snprintf(g_err, sizeof(g_err), "%s", e.what());                                                                          \
        return (int)hipErrorUnknown;                                                                                             \
    }

// Makes the handle's device current for the duration of a call and restores the caller's device afterwards: one process may hold
// one handle per GPU and call them from any thread in any order (reference: one implicit device, default-stream launches).
struct device_guard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit device_guard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = (err == hipSuccess);
        }
    }
    ~device_guard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define GRID_ON_DEVICE(h)                  \
    device_guard guard__((h)->device);     \
    if (guard__.err != hipSuccess) return fail(guard__.err, "hipSetDevice(handle device)")

static const size_t GRID_CU_LDS_BYTES = 160 * 1024;  // LDS of one gfx950 CU: no block may ask for more

// launch geometry of one kernel: lane groups per block (capped by the kernel's own limit and by the CU's LDS), threads, blocks, LDS bytes
struct launch_cfg {
    dim3 grid, block;
    size_t lds;
};
// the second-order kernels: robots with 8-lane groups carry a second instance of the library for 16-lane groups (GRID_SO_WIDE, namespace grid::wide)
#ifdef GRID_SO_WIDE
namespace grid_so = grid::wide;
#else
namespace grid_so = grid;
#endif

template <typename T>
static int make_launch(const grid_handle *h, int num_timesteps, int default_threads, int max_groups, int lds_per_solve, int out_per_solve, launch_cfg *cfg,
                       int lanes = grid::GRID_LANES_PER_SOLVE) {
    int threads = h->threads > 0 ? h->threads : default_threads;
    if (threads < lanes) threads = lanes;  // (the second-order kernels of 8-lane robots run 16-lane groups: namespace wide)
    if (threads < grid::GRID_MIN_THREADS || threads > grid::GRID_MAX_THREADS)
        return fail_msg(hipErrorInvalidConfiguration, "threads per block out of range");
    if (lanes == grid::GRID_LANES_PER_SOLVE) threads -= threads % grid::GRID_MIN_THREADS;  // (GRID_LANE_INTERLEAVE: blocks are whole 16-lane rows; the kernels retire the rest)
    int gpb = threads / lanes;
    if (gpb > max_groups) gpb = max_groups;  // (the kernels retire the lane groups beyond their cap)
    const size_t per_group = (size_t)(lds_per_solve + out_per_solve) * sizeof(T);
    if ((size_t)gpb * per_group > GRID_CU_LDS_BYTES) {
        // e.g. the 30-DoF robot in double precision: fewer solves per block than the block size suggests
        gpb = (int)(GRID_CU_LDS_BYTES / per_group);
        if (gpb < 1) return fail_msg(hipErrorInvalidConfiguration, "one solve of this robot does not fit the LDS of a CU in this precision");
        threads = gpb * lanes;
    }
    int blocks = h->blocks > 0 ? h->blocks : (num_timesteps + gpb - 1) / gpb;
    if (blocks < 1) blocks = 1;
    cfg->grid = dim3(blocks, 1, 1);
    cfg->block = dim3(threads, 1, 1);
    cfg->lds = (size_t)gpb * per_group;
    return 0;
}
template <typename T>
static int general_launch(const grid_handle *h, int num_timesteps, launch_cfg *cfg) {
    return make_launch<T>(h, num_timesteps, grid::SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::GRID_LDS_PER_SOLVE, grid::GRID_OUT_PER_SOLVE, cfg);
}

// device entry points: required pointers must not be NULL and the stride must cover what the kernel loads per solve (a smaller or negative stride
// would make the last solves read before / past the caller's buffer: a GPU memory fault instead of an error code)
static int check_io(const void *in, int stride, int min_stride, const void *out, int num_timesteps) {
    if (num_timesteps <= 0) return 0;
    if (!in || !out) return fail_msg(hipErrorInvalidValue, "null input or output pointer");
    if (stride < min_stride) {
        snprintf(g_err, sizeof(g_err), "stride %d is smaller than the %d values the kernel reads per solve", stride, min_stride);
        return (int)hipErrorInvalidValue;
    }
    return 0;
}

static int check_args(const grid_handle *h, int num_timesteps) {
    if (!h) return fail_msg(hipErrorInvalidValue, "null handle");
    if (num_timesteps < 0) return fail_msg(hipErrorInvalidValue, "negative num_timesteps");
    if (h->threads != 0 && (h->threads < grid::GRID_MIN_THREADS || h->threads > grid::GRID_MAX_THREADS)) {
        snprintf(g_err, sizeof(g_err), "threads per block must be in [%d, %d]", grid::GRID_MIN_THREADS, grid::GRID_MAX_THREADS);
        return (int)hipErrorInvalidConfiguration;
    }
    return 0;
}

// device + pinned host bytes init_gridData<T>(N) will ask for (same list as the generated function)
template <typename T>
static size_t grid_data_bytes(int N) {
    const size_t n = grid::NUM_JOINTS;
    size_t per = 3 * n + 2 * n + n + n + n * n + n + 2 * n * n + 2 * n * n;
    size_t bytes = per * (size_t)N * sizeof(T);
#if GRID_HAS_IDSVA_SO
    const int so = N < grid::grid_so_max_timesteps<T>() ? N : grid::grid_so_max_timesteps<T>();
    bytes += 8 * n * n * n * (size_t)so * sizeof(T);
#endif
    return bytes;
}

// solves per call the second-order entry points of this handle accept (their records are 4 n^3 values: init_gridData caps those buffers)
template <typename T>
static int so_capacity(const grid_handle *h) {
#if GRID_HAS_IDSVA_SO
    return h->max_timesteps < grid::grid_so_max_timesteps<T>() ? h->max_timesteps : grid::grid_so_max_timesteps<T>();
#else
    (void)h;
    return 0;
#endif
}

template <typename T>
static int ensure_typed(grid_handle *h) {
    grid_typed<T> &t = typed<T>(h);
    std::lock_guard<std::mutex> lock(h->alloc_lock);
    if (t.hd_data) return 0;
    size_t free_b = 0, total_b = 0;
    GRID_TRY(hipMemGetInfo(&free_b, &total_b));
    if (grid_data_bytes<T>(h->max_timesteps) > free_b) {
        snprintf(g_err, sizeof(g_err), "grid_init: max_timesteps = %d needs %zu bytes of device memory, %zu are free", h->max_timesteps,
                 grid_data_bytes<T>(h->max_timesteps), free_b);
        return (int)hipErrorOutOfMemory;
    }
    if (!t.d_robotModel) t.d_robotModel = grid::init_robotModel<T>();
    t.hd_data = grid::init_gridData<T>(h->max_timesteps);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- device entry points
template <typename T>
static int fd_grad_device(grid_handle *h, const T *d_q_qd_u, int stride, int N, T gravity, T *d_df_du, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd_u, stride, 3*(int)grid::NUM_JOINTS, d_df_du, N))) return rc;
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    // this kernel has its own (smaller) LDS slice and suggested block size: FD_DU_LDS_PER_SOLVE, FD_DU_SUGGESTED_THREADS
    if ((rc = make_launch<T>(h, N, grid::FD_DU_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::FD_DU_LDS_PER_SOLVE, grid::FD_DU_OUT_PER_SOLVE, &c))) return rc;
    hipLaunchKernelGGL((grid::forward_dynamics_gradient_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_df_du, d_q_qd_u, stride,
                       typed<T>(h).d_robotModel, gravity, N);
    GRID_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int fd_grad_qdd_minv_device(grid_handle *h, const T *d_q_qd, int stride, const T *d_qdd, const T *d_Minv, int N, T gravity, T *d_df_du, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd, stride, 2*(int)grid::NUM_JOINTS, d_df_du, N))) return rc;
    if (N > 0 && (!d_qdd || !d_Minv)) return fail_msg(hipErrorInvalidValue, "null qdd or Minv pointer");
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = general_launch<T>(h, N, &c))) return rc;
    hipLaunchKernelGGL((grid::forward_dynamics_gradient_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_df_du, d_q_qd, stride, d_qdd, d_Minv,
                       typed<T>(h).d_robotModel, gravity, N);
    GRID_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int id_device(grid_handle *h, const T *d_q_qd, int stride, const T *d_qdd, int N, T gravity, T *d_c, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd, stride, 2*(int)grid::NUM_JOINTS, d_c, N))) return rc;
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = make_launch<T>(h, N, grid::ID_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::ID_LDS_PER_SOLVE, grid::ID_OUT_PER_SOLVE, &c))) return rc;
    if (d_qdd) {
        hipLaunchKernelGGL((grid::inverse_dynamics_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_c, d_q_qd, stride, d_qdd, typed<T>(h).d_robotModel, gravity, N);
    } else {
        hipLaunchKernelGGL((grid::inverse_dynamics_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_c, d_q_qd, stride, typed<T>(h).d_robotModel, gravity, N);
    }
    GRID_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int id_grad_device(grid_handle *h, const T *d_q_qd, int stride, const T *d_qdd, int N, T gravity, T *d_dc_du, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd, stride, 2*(int)grid::NUM_JOINTS, d_dc_du, N))) return rc;
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = make_launch<T>(h, N, grid::ID_DU_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::ID_DU_LDS_PER_SOLVE, grid::ID_DU_OUT_PER_SOLVE, &c))) return rc;
    if (d_qdd) {
        hipLaunchKernelGGL((grid::inverse_dynamics_gradient_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_dc_du, d_q_qd, stride, d_qdd,
                           typed<T>(h).d_robotModel, gravity, N);
    } else {
        hipLaunchKernelGGL((grid::inverse_dynamics_gradient_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_dc_du, d_q_qd, stride,
                           typed<T>(h).d_robotModel, gravity, N);
    }
    GRID_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int minv_device(grid_handle *h, const T *d_q, int stride, int N, T *d_Minv, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q, stride, 1*(int)grid::NUM_JOINTS, d_Minv, N))) return rc;
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = make_launch<T>(h, N, grid::MINV_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::MINV_LDS_PER_SOLVE, grid::MINV_OUT_PER_SOLVE, &c))) return rc;
    hipLaunchKernelGGL((grid::direct_minv_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_Minv, d_q, stride, typed<T>(h).d_robotModel, N);
    GRID_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int fd_device(grid_handle *h, const T *d_q_qd_u, int stride, int N, T gravity, T *d_qdd, void *stream, bool aba) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd_u, stride, 3*(int)grid::NUM_JOINTS, d_qdd, N))) return rc;
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = aba ? make_launch<T>(h, N, grid::ABA_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::ABA_LDS_PER_SOLVE, grid::ABA_OUT_PER_SOLVE, &c) : make_launch<T>(h, N, grid::FD_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::FD_LDS_PER_SOLVE, grid::FD_OUT_PER_SOLVE, &c))) return rc;
    if (aba) {
        hipLaunchKernelGGL((grid::aba_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_qdd, d_q_qd_u, stride, typed<T>(h).d_robotModel, gravity, N);
    } else {
        hipLaunchKernelGGL((grid::forward_dynamics_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_qdd, d_q_qd_u, stride, typed<T>(h).d_robotModel, gravity, N);
    }
    GRID_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int idsva_so_device(grid_handle *h, const T *d_q_qd_u, int stride, const T *d_qdd, int N, T gravity, T *d_idsva_so, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd_u, stride, 2*(int)grid::NUM_JOINTS, d_idsva_so, N))) return rc;
#if GRID_HAS_IDSVA_SO
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = make_launch<T>(h, N, grid_so::IDSVA_SO_SUGGESTED_THREADS, grid_so::IDSVA_SO_MAX_SOLVES_PER_BLOCK, grid_so::IDSVA_SO_LDS_PER_SOLVE, grid_so::IDSVA_SO_STAGE_PER_SOLVE, &c, grid_so::GRID_LANES_PER_SOLVE))) return rc;
    if (d_qdd) {
        hipLaunchKernelGGL((grid_so::idsva_so_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_idsva_so, d_q_qd_u, stride, d_qdd, reinterpret_cast<const grid_so::robotModel<T> *>(typed<T>(h).d_robotModel), gravity, N);
    } else {
        hipLaunchKernelGGL((grid_so::idsva_so_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_idsva_so, d_q_qd_u, stride, reinterpret_cast<const grid_so::robotModel<T> *>(typed<T>(h).d_robotModel), gravity, N);
    }
    GRID_TRY(hipGetLastError());
    return 0;
#else
    (void)d_q_qd_u; (void)stride; (void)d_qdd; (void)gravity; (void)d_idsva_so; (void)stream;
    return fail_msg(hipErrorNotSupported, "idsva_so is not emitted for this library's robot (see GRID_HAS_IDSVA_SO in the generated header)");
#endif
}

template <typename T>
static int fdsva_so_device(grid_handle *h, const T *d_q_qd_u, int stride, int N, T gravity, T *d_df2, void *stream) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if ((rc = check_io(d_q_qd_u, stride, 3*(int)grid::NUM_JOINTS, d_df2, N))) return rc;
#if GRID_HAS_IDSVA_SO
    if (N == 0) return 0;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    launch_cfg c;
    if ((rc = make_launch<T>(h, N, grid_so::FDSVA_SO_SUGGESTED_THREADS, grid_so::FDSVA_SO_MAX_SOLVES_PER_BLOCK, grid_so::FDSVA_SO_LDS_PER_SOLVE, grid_so::FDSVA_SO_STAGE_PER_SOLVE, &c, grid_so::GRID_LANES_PER_SOLVE))) return rc;
#if GRID_SO_DIRECT
    // the idsva_so tensors of a solve do not fit LDS: the kernel keeps them in the handle's d_idsva_so buffer
    if (N > so_capacity<T>(h)) return fail_msg(hipErrorInvalidValue, "num_timesteps exceeds the handle's second-order workspace (grid_second_order_capacity)");
    {
        std::lock_guard<std::mutex> lock(h->alloc_lock);
        if (!h->so_done) GRID_TRY(hipEventCreateWithFlags(&h->so_done, hipEventDisableTiming));
        if (h->so_pending) GRID_TRY(hipStreamWaitEvent((hipStream_t)stream, h->so_done, 0));  // (the previous launch may be on another stream)
#if GRID_SO_SPLIT
        // two kernels: gradient, M^-1 and the tensors by lane groups into the handle's buffers, then the contraction with one block per solve
        hipLaunchKernelGGL((grid_so::fdsva_so_prepare_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, typed<T>(h).hd_data->d_idsva_so, typed<T>(h).hd_data->d_df_du,
                           typed<T>(h).hd_data->d_Minv, d_q_qd_u, stride, reinterpret_cast<const grid_so::robotModel<T> *>(typed<T>(h).d_robotModel), gravity, N);
        GRID_TRY(hipGetLastError());
        hipLaunchKernelGGL((grid_so::fdsva_so_contract_kernel<T>), dim3(N < 4096 ? N : 4096), dim3(grid_so::FDSVA_SO_CONTRACT_THREADS), (size_t)grid_so::FDSVA_SO_CONTRACT_LDS * sizeof(T),
                           (hipStream_t)stream, d_df2, typed<T>(h).hd_data->d_idsva_so, typed<T>(h).hd_data->d_df_du, typed<T>(h).hd_data->d_Minv, N);
#else
        hipLaunchKernelGGL((grid_so::fdsva_so_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_df2, typed<T>(h).hd_data->d_idsva_so, d_q_qd_u, stride, reinterpret_cast<const grid_so::robotModel<T> *>(typed<T>(h).d_robotModel), gravity, N);
#endif
        GRID_TRY(hipGetLastError());
        GRID_TRY(hipEventRecord(h->so_done, (hipStream_t)stream));
        h->so_pending = true;
    }
#else
    hipLaunchKernelGGL((grid_so::fdsva_so_kernel<T>), c.grid, c.block, c.lds, (hipStream_t)stream, d_df2, d_q_qd_u, stride, reinterpret_cast<const grid_so::robotModel<T> *>(typed<T>(h).d_robotModel), gravity, N);
#endif
    GRID_TRY(hipGetLastError());
    return 0;
#else
    (void)d_q_qd_u; (void)stride; (void)gravity; (void)d_df2; (void)stream;
    return fail_msg(hipErrorNotSupported, "fdsva_so is not emitted for this library's robot (see GRID_HAS_IDSVA_SO in the generated header)");
#endif
}

// ---------------------------------------------------------------------------------------------------------------- host entry points
// Host buffers in, host buffers out, synchronous: H2D on the handle's stream, launch, D2H, stream sync - the semantics of the
// reference's host wrappers (e.g. reference algorithms/_inverse_dynamics.py:440-512), with the handle's device buffers as staging.
template <typename T>
static int host_prologue(grid_handle *h, int N) {
    int rc = check_args(h, N);
    if (rc) return rc;
    if (N > h->max_timesteps) return fail_msg(hipErrorInvalidValue, "num_timesteps exceeds grid_init's max_timesteps");
    return 0;
}
#define GRID_H2D(dst, src, count) GRID_TRY(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(T), hipMemcpyHostToDevice, s))
#define GRID_D2H(dst, src, count) GRID_TRY(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(T), hipMemcpyDeviceToHost, s))

// true where `ptr` is page-locked host memory the GPU's copy engines can reach directly (hipHostMalloc / grid_host_alloc / hipHostRegister): only then is
// hipMemcpyAsync asynchronous.  (Pinning a caller's pageable buffer for one call was tried: hipHostRegister of a freshly allocated 6.4 MB result buffer costs more
// than the overlap gains - 64 instead of 77 M solves/s end to end.)
static bool is_pinned_host(const void *ptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// The hot path's host entry point.  Semantics of the reference's wrapper (H2D, launch, D2H, synchronous on return; reference
// algorithms/_forward_dynamics_gradient.py:221-245).  Where BOTH of the caller's buffers are page-locked (grid_host_alloc) the batch is cut into chunks that
// travel on the handle's three streams: the copy engines run H2D of chunk c+1 and D2H of chunk c-1 beside the kernel of chunk c.  Pageable buffers take the
// strictly sequential form (hipMemcpyAsync blocks on them anyway).
template <typename T>
static int fd_grad_host(grid_handle *h, const T *h_q_qd_u, int N, T gravity, T *h_df_du) {
    int rc = host_prologue<T>(h, N);
    if (rc || N == 0) return rc;
    if (!h_q_qd_u || !h_df_du) return fail_msg(hipErrorInvalidValue, "null input or output pointer");
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    const size_t n = grid::NUM_JOINTS;
    grid::gridData<T> *d = typed<T>(h).hd_data;
    const int min_chunk = 2048;  // (below that a chunk's kernel is all launch latency)
    int chunks = N / min_chunk;
    if (chunks > 3) chunks = 3;  // (one chunk per stream; measured at 16 384 solves, page-locked buffers: 1 chunk 178 us, 2: 176, 3: 172.5, 4: 206, 8: 244 - tools/bench_host_pipeline.py)
    if (h->host_chunks > 0) chunks = h->host_chunks < N ? h->host_chunks : N;
    if (chunks < 2 || !is_pinned_host(h_q_qd_u) || !is_pinned_host(h_df_du)) {
        hipStream_t s = h->streams[0];
        GRID_H2D(d->d_q_qd_u, h_q_qd_u, 3 * n * N);
        if ((rc = fd_grad_device<T>(h, d->d_q_qd_u, 3 * (int)n, N, gravity, d->d_df_du, (void *)s))) return rc;
        GRID_D2H(h_df_du, d->d_df_du, 2 * n * n * N);
        GRID_TRY(hipStreamSynchronize(s));
        return 0;
    }
    const int per = (N + chunks - 1) / chunks;
    for (int c = 0; c < chunks; c++) {
        const int k0 = c * per, cnt = (k0 + per <= N) ? per : N - k0;
        if (cnt <= 0) break;
        hipStream_t s = h->streams[c % 3];
        hipError_t e = hipMemcpyAsync(d->d_q_qd_u + (size_t)k0 * 3 * n, h_q_qd_u + (size_t)k0 * 3 * n, 3 * n * (size_t)cnt * sizeof(T), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { rc = fail(e, "hipMemcpyAsync(H2D)"); break; }  // (no early return: the streams are drained below)
        if ((rc = fd_grad_device<T>(h, d->d_q_qd_u + (size_t)k0 * 3 * n, 3 * (int)n, cnt, gravity, d->d_df_du + (size_t)k0 * 2 * n * n, (void *)s))) break;
        e = hipMemcpyAsync(h_df_du + (size_t)k0 * 2 * n * n, d->d_df_du + (size_t)k0 * 2 * n * n, 2 * n * n * (size_t)cnt * sizeof(T), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) { rc = fail(e, "hipMemcpyAsync(D2H)"); break; }
    }
    for (int c = 0; c < 3; c++) {  // (always drained: the call is synchronous on return)
        hipError_t e = hipStreamSynchronize(h->streams[c]);
        if (e != hipSuccess && !rc) rc = fail(e, "hipStreamSynchronize");
    }
    return rc;
}

template <typename T>
static int fd_grad_qdd_minv_host(grid_handle *h, const T *h_q_qd, int stride, const T *h_qdd, const T *h_Minv, int N, T gravity, T *h_df_du) {
    int rc = host_prologue<T>(h, N);
    if (rc || N == 0) return rc;
    const size_t n = grid::NUM_JOINTS;
    if (stride < 2 * (int)n || stride > 3 * (int)n) return fail_msg(hipErrorInvalidValue, "stride must be in [2n, 3n] for host buffers");
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    grid::gridData<T> *d = typed<T>(h).hd_data;
    hipStream_t s = h->streams[0];
    GRID_H2D(d->d_q_qd_u, h_q_qd, (size_t)stride * N);
    GRID_H2D(d->d_qdd, h_qdd, n * N);
    GRID_H2D(d->d_Minv, h_Minv, n * n * N);
    if ((rc = fd_grad_qdd_minv_device<T>(h, d->d_q_qd_u, stride, d->d_qdd, d->d_Minv, N, gravity, d->d_df_du, (void *)s))) return rc;
    GRID_D2H(h_df_du, d->d_df_du, 2 * n * n * N);
    GRID_TRY(hipStreamSynchronize(s));
    return 0;
}

// which: 0 = inverse dynamics (c), 1 = its gradient (dc_du)
template <typename T>
static int id_host(grid_handle *h, const T *h_q_qd, int stride, const T *h_qdd, int N, T gravity, T *h_out, int which) {
    int rc = host_prologue<T>(h, N);
    if (rc || N == 0) return rc;
    const size_t n = grid::NUM_JOINTS;
    if (stride < 2 * (int)n || stride > 3 * (int)n) return fail_msg(hipErrorInvalidValue, "stride must be in [2n, 3n] for host buffers (USE_COMPRESSED_MEM: 2n)");
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    grid::gridData<T> *d = typed<T>(h).hd_data;
    hipStream_t s = h->streams[0];
    GRID_H2D(d->d_q_qd_u, h_q_qd, (size_t)stride * N);
    if (h_qdd) GRID_H2D(d->d_qdd, h_qdd, n * N);
    if (which == 0) {
        if ((rc = id_device<T>(h, d->d_q_qd_u, stride, h_qdd ? d->d_qdd : nullptr, N, gravity, d->d_c, (void *)s))) return rc;
        GRID_D2H(h_out, d->d_c, n * N);
    } else {
        if ((rc = id_grad_device<T>(h, d->d_q_qd_u, stride, h_qdd ? d->d_qdd : nullptr, N, gravity, d->d_dc_du, (void *)s))) return rc;
        GRID_D2H(h_out, d->d_dc_du, 2 * n * n * N);
    }
    GRID_TRY(hipStreamSynchronize(s));
    return 0;
}

template <typename T>
static int minv_host(grid_handle *h, const T *h_q, int stride, int N, T *h_Minv) {
    int rc = host_prologue<T>(h, N);
    if (rc || N == 0) return rc;
    const size_t n = grid::NUM_JOINTS;
    if (stride < (int)n || stride > 3 * (int)n) return fail_msg(hipErrorInvalidValue, "stride must be in [n, 3n] for host buffers (USE_COMPRESSED_MEM: n)");
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    grid::gridData<T> *d = typed<T>(h).hd_data;
    hipStream_t s = h->streams[0];
    GRID_H2D(d->d_q_qd_u, h_q, (size_t)stride * N);
    if ((rc = minv_device<T>(h, d->d_q_qd_u, stride, N, d->d_Minv, (void *)s))) return rc;
    GRID_D2H(h_Minv, d->d_Minv, n * n * N);
    GRID_TRY(hipStreamSynchronize(s));
    return 0;
}

template <typename T>
static int fd_host(grid_handle *h, const T *h_q_qd_u, int N, T gravity, T *h_qdd, bool aba) {
    int rc = host_prologue<T>(h, N);
    if (rc || N == 0) return rc;
    const size_t n = grid::NUM_JOINTS;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    grid::gridData<T> *d = typed<T>(h).hd_data;
    hipStream_t s = h->streams[0];
    GRID_H2D(d->d_q_qd_u, h_q_qd_u, 3 * n * N);
    if ((rc = fd_device<T>(h, d->d_q_qd_u, 3 * (int)n, N, gravity, d->d_qdd, (void *)s, aba))) return rc;
    GRID_D2H(h_qdd, d->d_qdd, n * N);
    GRID_TRY(hipStreamSynchronize(s));
    return 0;
}

// which: 0 = idsva_so (h_qdd may be NULL), 1 = fdsva_so
template <typename T>
static int so_host(grid_handle *h, const T *h_q_qd_u, const T *h_qdd, int N, T gravity, T *h_out, int which) {
    int rc = host_prologue<T>(h, N);
    if (rc) return rc;
#if GRID_HAS_IDSVA_SO
    if (N == 0) return 0;
    if (N > so_capacity<T>(h)) return fail_msg(hipErrorInvalidValue, "num_timesteps exceeds the handle's second-order buffers (grid_second_order_capacity)");
    const size_t n = grid::NUM_JOINTS;
    GRID_ON_DEVICE(h);
    if ((rc = ensure_typed<T>(h))) return rc;
    grid::gridData<T> *d = typed<T>(h).hd_data;
    hipStream_t s = h->streams[0];
    GRID_H2D(d->d_q_qd_u, h_q_qd_u, 3 * n * N);
    if (which == 0) {
        if (h_qdd) GRID_H2D(d->d_qdd, h_qdd, n * N);
        if ((rc = idsva_so_device<T>(h, d->d_q_qd_u, 3 * (int)n, h_qdd ? d->d_qdd : nullptr, N, gravity, d->d_idsva_so, (void *)s))) return rc;
        GRID_D2H(h_out, d->d_idsva_so, 4 * n * n * n * N);
    } else {
        if ((rc = fdsva_so_device<T>(h, d->d_q_qd_u, 3 * (int)n, N, gravity, d->d_df2, (void *)s))) return rc;
        GRID_D2H(h_out, d->d_df2, 4 * n * n * n * N);
    }
    GRID_TRY(hipStreamSynchronize(s));
    return 0;
#else
    (void)h_q_qd_u; (void)h_qdd; (void)gravity; (void)h_out; (void)which;
    return fail_msg(hipErrorNotSupported, "the second-order kernels are not emitted for this library's robot (see GRID_HAS_IDSVA_SO in the generated header)");
#endif
}

// ---------------------------------------------------------------------------------------------------------------- multi-GPU driver
// One process, G handles (one per GPU): the batch [0, N) is cut into G contiguous ranges of ceil(N/G) solves (SURVEY.md section 8(e),
// 16 384 total -> 16 384/G per GPU, no collective).  Every device has its own robotModel copy and stream.
static inline void multi_range(int N, int G, int g, int *k0, int *cnt) {
    const int per = (N + G - 1) / G;
    int a = g * per, b = a + per;
    if (a > N) a = N;
    if (b > N) b = N;
    *k0 = a;
    *cnt = b - a;
}

template <typename T>
static int fd_grad_multi_host(grid_handle **hs, int G, const T *h_q_qd_u, int N, T gravity, T *h_df_du) {
    if (!hs || G < 1 || N < 0) return fail_msg(hipErrorInvalidValue, "grid_forward_dynamics_gradient_multi_host: bad arguments");
    const size_t n = grid::NUM_JOINTS;
    std::vector<int> rcs(G, 0);
    std::vector<std::string> msgs(G);
    std::vector<std::thread> ts;
    // one host thread per device: pageable host memory makes hipMemcpyAsync block, threads keep the G copy engines busy at once
    for (int g = 0; g < G; g++) {
        ts.emplace_back([&, g]() {
            int k0, cnt;
            multi_range(N, G, g, &k0, &cnt);
            rcs[g] = fd_grad_host<T>(hs[g], h_q_qd_u + (size_t)k0 * 3 * n, cnt, gravity, h_df_du + (size_t)k0 * 2 * n * n);
            if (rcs[g]) msgs[g] = g_err;  // (g_err is thread-local)
        });
    }
    for (auto &t : ts) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[g]) {
            snprintf(g_err, sizeof(g_err), "device slot %d: %s", g, msgs[g].c_str());
            return rcs[g];
        }
    return 0;
}

extern "C" {

int grid_num_joints(void) { return grid::NUM_JOINTS; }
const char *grid_robot_name(void) { return GRID_ROBOT_NAME; }
int grid_lanes_per_solve(void) { return grid::GRID_LANES_PER_SOLVE; }
int grid_suggested_threads(void) { return grid::SUGGESTED_THREADS; }
int grid_lds_bytes_per_block(void) {  // what a default forward_dynamics_gradient launch asks for (same arithmetic as make_launch)
    grid_handle h{};
    launch_cfg c;
    if (make_launch<float>(&h, 1, grid::FD_DU_SUGGESTED_THREADS, grid::GRID_MAX_SOLVES_PER_BLOCK, grid::FD_DU_LDS_PER_SOLVE, grid::FD_DU_OUT_PER_SOLVE, &c)) return -1;
    return (int)c.lds;
}
int grid_has_second_order(void) { return GRID_HAS_IDSVA_SO; }
int grid_second_order_capacity(const grid_handle *h, int f64) { return h ? (f64 ? so_capacity<double>(h) : so_capacity<float>(h)) : 0; }
const char *grid_last_error(void) { return g_err; }

int grid_init(int device, int max_timesteps, grid_handle **out) {
    if (!out || max_timesteps < 1) return fail_msg(hipErrorInvalidValue, "grid_init: bad arguments");
    *out = nullptr;
    device_guard guard(device);
    if (guard.err != hipSuccess) return fail(guard.err, "hipSetDevice(device)");
    grid_handle *h = new (std::nothrow) grid_handle();
    if (!h) return fail_msg(hipErrorOutOfMemory, "grid_init: out of host memory");
    h->device = device;
    h->max_timesteps = max_timesteps;
    h->blocks = h->threads = h->host_chunks = 0;
    h->streams = nullptr;
    int rc = 0;
    try {
        rc = ensure_typed<float>(h);
        if (!rc) h->streams = grid::init_grid<float>();
    } catch (const grid_capi_error &e) {
        snprintf(g_err, sizeof(g_err), "grid_init: %s (%s:%d)", hipGetErrorString(e.code), e.file, e.line);
        rc = (int)e.code;
    }
    if (rc) {  // (buffers a failed init_* allocated before the failing call are not tracked: they stay allocated until the process ends)
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

int grid_close(grid_handle *h) {
    if (!h) return 0;
    GRID_ON_DEVICE(h);
    GRID_GUARDED(
        if (h->f64.hd_data) {  // close_grid frees the streams too: give the f64 state its own (empty) set
            hipStream_t *none = grid::init_grid<double>();
            grid::close_grid<double>(none, h->f64.d_robotModel, h->f64.hd_data);
        }
        grid::close_grid<float>(h->streams, h->f32.d_robotModel, h->f32.hd_data);
        if (h->so_done) (void)hipEventDestroy(h->so_done);
    )
    delete h;
    return 0;
}

int grid_device(const grid_handle *h) { return h ? h->device : -1; }

int grid_set_host_chunks(grid_handle *h, int chunks) {
    if (!h || chunks < 0 || chunks > 64) return fail_msg(hipErrorInvalidValue, "grid_set_host_chunks: chunks must be 0 (automatic) .. 64");
    h->host_chunks = chunks;
    return 0;
}
int grid_host_alloc(size_t bytes, void **out) {
    if (!out) return fail_msg(hipErrorInvalidValue, "grid_host_alloc: null result pointer");
    *out = nullptr;
    GRID_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}
int grid_host_free(void *p) {
    if (p) GRID_TRY(hipHostFree(p));
    return 0;
}

int grid_set_launch_dims(grid_handle *h, int blocks, int threads) {
    if (!h) return (int)hipErrorInvalidValue;
    if (blocks < 0 || (threads != 0 && (threads < grid::GRID_MIN_THREADS || threads > grid::GRID_MAX_THREADS))) {
        snprintf(g_err, sizeof(g_err), "threads per block must be 0 or in [%d, %d] (whole lane groups; robots with 8-lane groups: whole 16-lane rows), blocks >= 0", grid::GRID_MIN_THREADS, grid::GRID_MAX_THREADS);
        return (int)hipErrorInvalidConfiguration;
    }
    h->blocks = blocks;
    h->threads = threads;
    return 0;
}

// ---- float
int grid_forward_dynamics_gradient_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_df_du, void *stream) {
    GRID_GUARDED(return fd_grad_device<float>(h, d_q_qd_u, stride_q_qd_u, num_timesteps, gravity, d_df_du, stream);)
}
int grid_forward_dynamics_gradient_qdd_minv_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, const float *d_Minv,
                                                   int num_timesteps, float gravity, float *d_df_du, void *stream) {
    GRID_GUARDED(return fd_grad_qdd_minv_device<float>(h, d_q_qd, stride_q_qd, d_qdd, d_Minv, num_timesteps, gravity, d_df_du, stream);)
}
int grid_inverse_dynamics_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, int num_timesteps, float gravity, float *d_c, void *stream) {
    GRID_GUARDED(return id_device<float>(h, d_q_qd, stride_q_qd, d_qdd, num_timesteps, gravity, d_c, stream);)
}
int grid_inverse_dynamics_gradient_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, int num_timesteps, float gravity,
                                          float *d_dc_du, void *stream) {
    GRID_GUARDED(return id_grad_device<float>(h, d_q_qd, stride_q_qd, d_qdd, num_timesteps, gravity, d_dc_du, stream);)
}
int grid_direct_minv_device(grid_handle *h, const float *d_q, int stride_q, int num_timesteps, float *d_Minv, void *stream) {
    GRID_GUARDED(return minv_device<float>(h, d_q, stride_q, num_timesteps, d_Minv, stream);)
}
int grid_forward_dynamics_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_qdd, void *stream) {
    GRID_GUARDED(return fd_device<float>(h, d_q_qd_u, stride_q_qd_u, num_timesteps, gravity, d_qdd, stream, false);)
}
int grid_aba_device(grid_handle *h, const float *d_q_qd_tau, int stride_q_qd, int num_timesteps, float gravity, float *d_qdd, void *stream) {
    GRID_GUARDED(return fd_device<float>(h, d_q_qd_tau, stride_q_qd, num_timesteps, gravity, d_qdd, stream, true);)
}
int grid_idsva_so_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, const float *d_qdd, int num_timesteps, float gravity, float *d_idsva_so, void *stream) {
    GRID_GUARDED(return idsva_so_device<float>(h, d_q_qd_u, stride_q_qd_u, d_qdd, num_timesteps, gravity, d_idsva_so, stream);)
}
int grid_fdsva_so_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_df2, void *stream) {
    GRID_GUARDED(return fdsva_so_device<float>(h, d_q_qd_u, stride_q_qd_u, num_timesteps, gravity, d_df2, stream);)
}

int grid_forward_dynamics_gradient_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df_du) {
    GRID_GUARDED(return fd_grad_host<float>(h, h_q_qd_u, num_timesteps, gravity, h_df_du);)
}
int grid_forward_dynamics_gradient_qdd_minv_host(grid_handle *h, const float *h_q_qd, int stride_q_qd, const float *h_qdd, const float *h_Minv, int num_timesteps,
                                                 float gravity, float *h_df_du) {
    GRID_GUARDED(return fd_grad_qdd_minv_host<float>(h, h_q_qd, stride_q_qd, h_qdd, h_Minv, num_timesteps, gravity, h_df_du);)
}
int grid_inverse_dynamics_host(grid_handle *h, const float *h_q_qd, int stride_q_qd, const float *h_qdd, int num_timesteps, float gravity, float *h_c) {
    GRID_GUARDED(return id_host<float>(h, h_q_qd, stride_q_qd, h_qdd, num_timesteps, gravity, h_c, 0);)
}
int grid_inverse_dynamics_gradient_host(grid_handle *h, const float *h_q_qd, int stride_q_qd, const float *h_qdd, int num_timesteps, float gravity, float *h_dc_du) {
    GRID_GUARDED(return id_host<float>(h, h_q_qd, stride_q_qd, h_qdd, num_timesteps, gravity, h_dc_du, 1);)
}
int grid_direct_minv_host(grid_handle *h, const float *h_q, int stride_q, int num_timesteps, float *h_Minv) {
    GRID_GUARDED(return minv_host<float>(h, h_q, stride_q, num_timesteps, h_Minv);)
}
int grid_forward_dynamics_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_qdd) {
    GRID_GUARDED(return fd_host<float>(h, h_q_qd_u, num_timesteps, gravity, h_qdd, false);)
}
int grid_aba_host(grid_handle *h, const float *h_q_qd_tau, int num_timesteps, float gravity, float *h_qdd) {
    GRID_GUARDED(return fd_host<float>(h, h_q_qd_tau, num_timesteps, gravity, h_qdd, true);)
}
int grid_idsva_so_host(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, int num_timesteps, float gravity, float *h_idsva_so) {
    GRID_GUARDED(return so_host<float>(h, h_q_qd_u, h_qdd, num_timesteps, gravity, h_idsva_so, 0);)
}
int grid_fdsva_so_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df2) {
    GRID_GUARDED(return so_host<float>(h, h_q_qd_u, nullptr, num_timesteps, gravity, h_df2, 1);)
}
int grid_forward_dynamics_gradient_multi_host(grid_handle **handles, int num_handles, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df_du) {
    GRID_GUARDED(return fd_grad_multi_host<float>(handles, num_handles, h_q_qd_u, num_timesteps, gravity, h_df_du);)
}

// ---- double: the T = double instantiations of the same generated kernels (buffers allocated by the first *_f64 call)
int grid_forward_dynamics_gradient_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, int num_timesteps, double gravity, double *d_df_du, void *stream) {
    GRID_GUARDED(return fd_grad_device<double>(h, d_q_qd_u, stride_q_qd_u, num_timesteps, gravity, d_df_du, stream);)
}
int grid_forward_dynamics_gradient_qdd_minv_device_f64(grid_handle *h, const double *d_q_qd, int stride_q_qd, const double *d_qdd, const double *d_Minv,
                                                       int num_timesteps, double gravity, double *d_df_du, void *stream) {
    GRID_GUARDED(return fd_grad_qdd_minv_device<double>(h, d_q_qd, stride_q_qd, d_qdd, d_Minv, num_timesteps, gravity, d_df_du, stream);)
}
int grid_inverse_dynamics_device_f64(grid_handle *h, const double *d_q_qd, int stride_q_qd, const double *d_qdd, int num_timesteps, double gravity, double *d_c, void *stream) {
    GRID_GUARDED(return id_device<double>(h, d_q_qd, stride_q_qd, d_qdd, num_timesteps, gravity, d_c, stream);)
}
int grid_inverse_dynamics_gradient_device_f64(grid_handle *h, const double *d_q_qd, int stride_q_qd, const double *d_qdd, int num_timesteps, double gravity,
                                              double *d_dc_du, void *stream) {
    GRID_GUARDED(return id_grad_device<double>(h, d_q_qd, stride_q_qd, d_qdd, num_timesteps, gravity, d_dc_du, stream);)
}
int grid_direct_minv_device_f64(grid_handle *h, const double *d_q, int stride_q, int num_timesteps, double *d_Minv, void *stream) {
    GRID_GUARDED(return minv_device<double>(h, d_q, stride_q, num_timesteps, d_Minv, stream);)
}
int grid_forward_dynamics_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, int num_timesteps, double gravity, double *d_qdd, void *stream) {
    GRID_GUARDED(return fd_device<double>(h, d_q_qd_u, stride_q_qd_u, num_timesteps, gravity, d_qdd, stream, false);)
}
int grid_aba_device_f64(grid_handle *h, const double *d_q_qd_tau, int stride_q_qd, int num_timesteps, double gravity, double *d_qdd, void *stream) {
    GRID_GUARDED(return fd_device<double>(h, d_q_qd_tau, stride_q_qd, num_timesteps, gravity, d_qdd, stream, true);)
}
int grid_idsva_so_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, const double *d_qdd, int num_timesteps, double gravity, double *d_idsva_so, void *stream) {
    GRID_GUARDED(return idsva_so_device<double>(h, d_q_qd_u, stride_q_qd_u, d_qdd, num_timesteps, gravity, d_idsva_so, stream);)
}
int grid_fdsva_so_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, int num_timesteps, double gravity, double *d_df2, void *stream) {
    GRID_GUARDED(return fdsva_so_device<double>(h, d_q_qd_u, stride_q_qd_u, num_timesteps, gravity, d_df2, stream);)
}
int grid_forward_dynamics_gradient_host_f64(grid_handle *h, const double *h_q_qd_u, int num_timesteps, double gravity, double *h_df_du) {
    GRID_GUARDED(return fd_grad_host<double>(h, h_q_qd_u, num_timesteps, gravity, h_df_du);)
}
int grid_inverse_dynamics_host_f64(grid_handle *h, const double *h_q_qd, int stride_q_qd, const double *h_qdd, int num_timesteps, double gravity, double *h_c) {
    GRID_GUARDED(return id_host<double>(h, h_q_qd, stride_q_qd, h_qdd, num_timesteps, gravity, h_c, 0);)
}
int grid_inverse_dynamics_gradient_host_f64(grid_handle *h, const double *h_q_qd, int stride_q_qd, const double *h_qdd, int num_timesteps, double gravity, double *h_dc_du) {
    GRID_GUARDED(return id_host<double>(h, h_q_qd, stride_q_qd, h_qdd, num_timesteps, gravity, h_dc_du, 1);)
}
int grid_direct_minv_host_f64(grid_handle *h, const double *h_q, int stride_q, int num_timesteps, double *h_Minv) {
    GRID_GUARDED(return minv_host<double>(h, h_q, stride_q, num_timesteps, h_Minv);)
}
int grid_forward_dynamics_host_f64(grid_handle *h, const double *h_q_qd_u, int num_timesteps, double gravity, double *h_qdd) {
    GRID_GUARDED(return fd_host<double>(h, h_q_qd_u, num_timesteps, gravity, h_qdd, false);)
}
int grid_aba_host_f64(grid_handle *h, const double *h_q_qd_tau, int num_timesteps, double gravity, double *h_qdd) {
    GRID_GUARDED(return fd_host<double>(h, h_q_qd_tau, num_timesteps, gravity, h_qdd, true);)
}
int grid_idsva_so_host_f64(grid_handle *h, const double *h_q_qd_u, const double *h_qdd, int num_timesteps, double gravity, double *h_idsva_so) {
    GRID_GUARDED(return so_host<double>(h, h_q_qd_u, h_qdd, num_timesteps, gravity, h_idsva_so, 0);)
}
int grid_fdsva_so_host_f64(grid_handle *h, const double *h_q_qd_u, int num_timesteps, double gravity, double *h_df2) {
    GRID_GUARDED(return so_host<double>(h, h_q_qd_u, nullptr, num_timesteps, gravity, h_df2, 1);)
}

int grid_forward_dynamics_gradient_single_timing(grid_handle *h, const float *h_q_qd_u, int reps, float gravity, float *h_df_du, double *us_per_call) {
    int rc = check_args(h, reps);
    if (rc) return rc;
    const int n = grid::NUM_JOINTS;
    GRID_ON_DEVICE(h);
    GRID_TRY(hipMemcpy(h->f32.hd_data->d_q_qd_u, h_q_qd_u, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice));
    GRID_TRY(hipDeviceSynchronize());
    struct timespec start, end;
    clock_gettime(CLOCK_MONOTONIC, &start);
    hipLaunchKernelGGL((grid::forward_dynamics_gradient_kernel_single_timing<float>), dim3(1), dim3(grid::GRID_LANES_PER_SOLVE < 64 ? 64 : grid::GRID_LANES_PER_SOLVE),
                       (size_t)(64 / grid::GRID_LANES_PER_SOLVE > 0 ? 64 / grid::GRID_LANES_PER_SOLVE : 1) * (grid::GRID_LDS_PER_SOLVE + grid::GRID_OUT_PER_SOLVE) * sizeof(float),
                       0, h->f32.hd_data->d_df_du, h->f32.hd_data->d_q_qd_u, 3 * n, h->f32.d_robotModel, gravity, reps);
    GRID_TRY(hipGetLastError());
    GRID_TRY(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &end);
    GRID_TRY(hipMemcpy(h_df_du, h->f32.hd_data->d_df_du, (size_t)2 * n * n * sizeof(float), hipMemcpyDeviceToHost));
    if (us_per_call) *us_per_call = time_delta_us_timespec(start, end) / (double)(reps > 0 ? reps : 1);
    return 0;
}

}  // extern "C"

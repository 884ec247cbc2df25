// grid_capi.hip - C-ABI shim over the generated, robot-specialised HIP header (see include/grid_capi.h).
// Built once per robot:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC -I<dir of generated grid.cuh> -I<repo>/include grid_capi.hip
#include "grid.cuh"
#include "grid_capi.h"

#include <string.h>

#ifndef GRID_ROBOT_NAME
#define GRID_ROBOT_NAME "robot"
#endif

struct grid_handle {
    int device;
    int max_timesteps;
    int blocks;   // 0 = derive from the batch
    int threads;  // 0 = SUGGESTED_THREADS
    grid::robotModel<float> *d_robotModel;
    grid::gridData<float> *hd_data;
    hipStream_t *streams;
};

static thread_local char g_err[512] = "";

static int fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}
#define GRID_TRY(expr)                                   \
    do {                                                 \
        hipError_t e__ = (expr);                         \
        if (e__ != hipSuccess) return fail(e__, #expr);  \
    } while (0)

// dynamic LDS actually needed by a block of `threads` threads: one slice + one staging record per lane group.
// (the *_DYNAMIC_SHARED_MEM_COUNT constants cover SUGGESTED_THREADS; smaller blocks must not reserve that much or
//  they lose occupancy: measured 64.8 us vs 22.6 us per 16384-solve launch for 64-thread blocks)
static inline size_t lds_bytes(const grid_handle *h) {
    int threads = h->threads > 0 ? h->threads : grid::SUGGESTED_THREADS;
    int gpb = threads / grid::GRID_LANES_PER_SOLVE;
    if (gpb > grid::GRID_MAX_SOLVES_PER_BLOCK) gpb = grid::GRID_MAX_SOLVES_PER_BLOCK;
    if (gpb < 1) gpb = 1;
    return (size_t)gpb * (grid::GRID_LDS_PER_SOLVE + grid::GRID_OUT_PER_SOLVE) * sizeof(float);
}

static inline void launch_dims(const grid_handle *h, int num_timesteps, dim3 *grid, dim3 *block) {
    int threads = h->threads > 0 ? h->threads : grid::SUGGESTED_THREADS;
    int gpb = threads / grid::GRID_LANES_PER_SOLVE;
    if (gpb > grid::GRID_MAX_SOLVES_PER_BLOCK) gpb = grid::GRID_MAX_SOLVES_PER_BLOCK;
    if (gpb < 1) gpb = 1;
    int blocks = h->blocks > 0 ? h->blocks : (num_timesteps + gpb - 1) / gpb;
    if (blocks < 1) blocks = 1;
    *grid = dim3(blocks, 1, 1);
    *block = dim3(threads, 1, 1);
}

static int check_args(const grid_handle *h, int num_timesteps) {
    if (!h) { snprintf(g_err, sizeof(g_err), "null handle"); return (int)hipErrorInvalidValue; }
    if (num_timesteps < 0) { snprintf(g_err, sizeof(g_err), "negative num_timesteps"); return (int)hipErrorInvalidValue; }
    int threads = h->threads > 0 ? h->threads : grid::SUGGESTED_THREADS;
    if (threads < grid::GRID_LANES_PER_SOLVE || threads > grid::GRID_MAX_THREADS) {
        snprintf(g_err, sizeof(g_err), "threads per block must be in [%d, %d]", grid::GRID_LANES_PER_SOLVE, grid::GRID_MAX_THREADS);
        return (int)hipErrorInvalidConfiguration;
    }
    return 0;
}

extern "C" {

int grid_num_joints(void) { return grid::NUM_JOINTS; }
const char *grid_robot_name(void) { return GRID_ROBOT_NAME; }
int grid_lanes_per_solve(void) { return grid::GRID_LANES_PER_SOLVE; }
int grid_suggested_threads(void) { return grid::SUGGESTED_THREADS; }
int grid_lds_bytes_per_block(void) { return grid::FD_DU_DYNAMIC_SHARED_MEM_COUNT * (int)sizeof(float); }
const char *grid_last_error(void) { return g_err; }

int grid_init(int device, int max_timesteps, grid_handle **out) {
    if (!out || max_timesteps < 1) { snprintf(g_err, sizeof(g_err), "grid_init: bad arguments"); return (int)hipErrorInvalidValue; }
    GRID_TRY(hipSetDevice(device));
    grid_handle *h = (grid_handle *)calloc(1, sizeof(grid_handle));
    h->device = device;
    h->max_timesteps = max_timesteps;
    h->d_robotModel = grid::init_robotModel<float>();
    h->streams = grid::init_grid<float>();
    h->hd_data = grid::init_gridData<float>(max_timesteps);
    *out = h;
    return 0;
}

int grid_close(grid_handle *h) {
    if (!h) return 0;
    GRID_TRY(hipSetDevice(h->device));
    grid::close_grid<float>(h->streams, h->d_robotModel, h->hd_data);
    free(h);
    return 0;
}

int grid_set_launch_dims(grid_handle *h, int blocks, int threads) {
    if (!h) return (int)hipErrorInvalidValue;
    h->blocks = blocks;
    h->threads = threads;
    return check_args(h, 0);
}

int grid_forward_dynamics_gradient_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity,
                                          float *d_df_du, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    // this kernel has its own (smaller) LDS slice and suggested block size: FD_DU_LDS_PER_SOLVE, FD_DU_SUGGESTED_THREADS
    int threads = h->threads > 0 ? h->threads : grid::FD_DU_SUGGESTED_THREADS;
    int gpb = threads / grid::GRID_LANES_PER_SOLVE;
    if (gpb > grid::GRID_MAX_SOLVES_PER_BLOCK) gpb = grid::GRID_MAX_SOLVES_PER_BLOCK;
    if (gpb < 1) gpb = 1;
    int blocks = h->blocks > 0 ? h->blocks : (num_timesteps + gpb - 1) / gpb;
    const dim3 grid(blocks < 1 ? 1 : blocks, 1, 1), block(threads, 1, 1);
    hipLaunchKernelGGL((grid::forward_dynamics_gradient_kernel<float>), grid, block, (size_t)gpb * (grid::FD_DU_LDS_PER_SOLVE + grid::GRID_OUT_PER_SOLVE) * sizeof(float),
                       (hipStream_t)stream, d_df_du, d_q_qd_u, stride_q_qd_u, h->d_robotModel, gravity, num_timesteps);
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_forward_dynamics_gradient_qdd_minv_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, const float *d_Minv,
                                                   int num_timesteps, float gravity, float *d_df_du, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    dim3 grid, block;
    launch_dims(h, num_timesteps, &grid, &block);
    hipLaunchKernelGGL((grid::forward_dynamics_gradient_kernel<float>), grid, block, lds_bytes(h),
                       (hipStream_t)stream, d_df_du, d_q_qd, stride_q_qd, d_qdd, d_Minv, h->d_robotModel, gravity, num_timesteps);
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_forward_dynamics_gradient_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df_du) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps > h->max_timesteps) { snprintf(g_err, sizeof(g_err), "num_timesteps exceeds grid_init's max_timesteps"); return (int)hipErrorInvalidValue; }
    if (num_timesteps == 0) return 0;
    const int n = grid::NUM_JOINTS;
    GRID_TRY(hipSetDevice(h->device));
    hipStream_t s = h->streams[0];
    GRID_TRY(hipMemcpyAsync(h->hd_data->d_q_qd_u, h_q_qd_u, (size_t)3 * n * num_timesteps * sizeof(float), hipMemcpyHostToDevice, s));
    rc = grid_forward_dynamics_gradient_device(h, h->hd_data->d_q_qd_u, 3 * n, num_timesteps, gravity, h->hd_data->d_df_du, (void *)s);
    if (rc) return rc;
    GRID_TRY(hipMemcpyAsync(h_df_du, h->hd_data->d_df_du, (size_t)2 * n * n * num_timesteps * sizeof(float), hipMemcpyDeviceToHost, s));
    GRID_TRY(hipStreamSynchronize(s));
    return 0;
}

int grid_inverse_dynamics_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, int num_timesteps, float gravity,
                                 float *d_c, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    dim3 grid, block;
    launch_dims(h, num_timesteps, &grid, &block);
    if (d_qdd) {
        hipLaunchKernelGGL((grid::inverse_dynamics_kernel<float>), grid, block, lds_bytes(h), (hipStream_t)stream,
                           d_c, d_q_qd, stride_q_qd, d_qdd, h->d_robotModel, gravity, num_timesteps);
    } else {
        hipLaunchKernelGGL((grid::inverse_dynamics_kernel<float>), grid, block, lds_bytes(h), (hipStream_t)stream,
                           d_c, d_q_qd, stride_q_qd, h->d_robotModel, gravity, num_timesteps);
    }
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_direct_minv_device(grid_handle *h, const float *d_q, int stride_q, int num_timesteps, float *d_Minv, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    dim3 grid, block;
    launch_dims(h, num_timesteps, &grid, &block);
    hipLaunchKernelGGL((grid::direct_minv_kernel<float>), grid, block, lds_bytes(h), (hipStream_t)stream,
                       d_Minv, d_q, stride_q, h->d_robotModel, num_timesteps);
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_forward_dynamics_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_qdd, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    dim3 grid, block;
    launch_dims(h, num_timesteps, &grid, &block);
    hipLaunchKernelGGL((grid::forward_dynamics_kernel<float>), grid, block, lds_bytes(h), (hipStream_t)stream,
                       d_qdd, d_q_qd_u, stride_q_qd_u, h->d_robotModel, gravity, num_timesteps);
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_aba_device(grid_handle *h, const float *d_q_qd_tau, int stride_q_qd, int num_timesteps, float gravity, float *d_qdd, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    dim3 grid, block;
    launch_dims(h, num_timesteps, &grid, &block);
    hipLaunchKernelGGL((grid::aba_kernel<float>), grid, block, lds_bytes(h), (hipStream_t)stream,
                       d_qdd, d_q_qd_tau, stride_q_qd, h->d_robotModel, gravity, num_timesteps);
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_idsva_so_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, const float *d_qdd, int num_timesteps, float gravity,
                         float *d_idsva_so, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
#if GRID_HAS_IDSVA_SO
    if (num_timesteps == 0) return 0;
    int threads = h->threads > 0 ? h->threads : grid::IDSVA_SO_SUGGESTED_THREADS;
    if (threads > grid::IDSVA_SO_SUGGESTED_THREADS) threads = grid::IDSVA_SO_SUGGESTED_THREADS;
    int gpb = threads / grid::GRID_LANES_PER_SOLVE;
    if (gpb < 1) { gpb = 1; threads = grid::GRID_LANES_PER_SOLVE; }
    const int blocks = h->blocks > 0 ? h->blocks : (num_timesteps + gpb - 1) / gpb;
    const size_t lds = (size_t)gpb * (grid::IDSVA_SO_LDS_PER_SOLVE + grid::IDSVA_SO_STAGE_PER_SOLVE) * sizeof(float);
    const dim3 grid(blocks, 1, 1), block(threads, 1, 1);
    if (d_qdd) {
        hipLaunchKernelGGL((grid::idsva_so_kernel<float>), grid, block, lds, (hipStream_t)stream,
                           d_idsva_so, d_q_qd_u, stride_q_qd_u, d_qdd, h->d_robotModel, gravity, num_timesteps);
    } else {
        hipLaunchKernelGGL((grid::idsva_so_kernel<float>), grid, block, lds, (hipStream_t)stream,
                           d_idsva_so, d_q_qd_u, stride_q_qd_u, h->d_robotModel, gravity, num_timesteps);
    }
    GRID_TRY(hipGetLastError());
    return 0;
#else
    (void)d_q_qd_u; (void)stride_q_qd_u; (void)d_qdd; (void)gravity; (void)d_idsva_so; (void)stream;
    snprintf(g_err, sizeof(g_err), "idsva_so is emitted for serial revolute chains only; this library's robot is not one");
    return (int)hipErrorNotSupported;
#endif
}

int grid_fdsva_so_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_df2, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
#if GRID_HAS_IDSVA_SO
    if (num_timesteps == 0) return 0;
    int threads = h->threads > 0 ? h->threads : grid::FDSVA_SO_SUGGESTED_THREADS;
    if (threads > grid::FDSVA_SO_SUGGESTED_THREADS) threads = grid::FDSVA_SO_SUGGESTED_THREADS;
    int gpb = threads / grid::GRID_LANES_PER_SOLVE;
    if (gpb < 1) { gpb = 1; threads = grid::GRID_LANES_PER_SOLVE; }
    int blocks = h->blocks > 0 ? h->blocks : (num_timesteps + gpb - 1) / gpb;
    const size_t lds = (size_t)gpb * (grid::GRID_LDS_PER_SOLVE + grid::FDSVA_SO_STAGE_PER_SOLVE) * sizeof(float);
    hipLaunchKernelGGL((grid::fdsva_so_kernel<float>), dim3(blocks, 1, 1), dim3(threads, 1, 1), lds, (hipStream_t)stream,
                       d_df2, d_q_qd_u, stride_q_qd_u, h->d_robotModel, gravity, num_timesteps);
    GRID_TRY(hipGetLastError());
    return 0;
#else
    (void)d_q_qd_u; (void)stride_q_qd_u; (void)gravity; (void)d_df2; (void)stream;
    snprintf(g_err, sizeof(g_err), "fdsva_so is emitted for serial revolute chains only; this library's robot is not one");
    return (int)hipErrorNotSupported;
#endif
}

int grid_inverse_dynamics_gradient_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, int num_timesteps, float gravity,
                                          float *d_dc_du, void *stream) {
    int rc = check_args(h, num_timesteps);
    if (rc) return rc;
    if (num_timesteps == 0) return 0;
    dim3 grid, block;
    launch_dims(h, num_timesteps, &grid, &block);
    if (d_qdd) {
        hipLaunchKernelGGL((grid::inverse_dynamics_gradient_kernel<float>), grid, block, lds_bytes(h),
                           (hipStream_t)stream, d_dc_du, d_q_qd, stride_q_qd, d_qdd, h->d_robotModel, gravity, num_timesteps);
    } else {
        hipLaunchKernelGGL((grid::inverse_dynamics_gradient_kernel<float>), grid, block, lds_bytes(h),
                           (hipStream_t)stream, d_dc_du, d_q_qd, stride_q_qd, h->d_robotModel, gravity, num_timesteps);
    }
    GRID_TRY(hipGetLastError());
    return 0;
}

int grid_forward_dynamics_gradient_single_timing(grid_handle *h, const float *h_q_qd_u, int reps, float gravity, float *h_df_du, double *us_per_call) {
    int rc = check_args(h, reps);
    if (rc) return rc;
    const int n = grid::NUM_JOINTS;
    GRID_TRY(hipSetDevice(h->device));
    GRID_TRY(hipMemcpy(h->hd_data->d_q_qd_u, h_q_qd_u, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice));
    GRID_TRY(hipDeviceSynchronize());
    struct timespec start, end;
    clock_gettime(CLOCK_MONOTONIC, &start);
    hipLaunchKernelGGL((grid::forward_dynamics_gradient_kernel_single_timing<float>), dim3(1), dim3(grid::GRID_LANES_PER_SOLVE < 64 ? 64 : grid::GRID_LANES_PER_SOLVE),
                       (size_t)(64 / grid::GRID_LANES_PER_SOLVE > 0 ? 64 / grid::GRID_LANES_PER_SOLVE : 1) * (grid::GRID_LDS_PER_SOLVE + grid::GRID_OUT_PER_SOLVE) * sizeof(float),
                       0, h->hd_data->d_df_du, h->hd_data->d_q_qd_u, 3 * n, h->d_robotModel, gravity, reps);
    GRID_TRY(hipGetLastError());
    GRID_TRY(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &end);
    GRID_TRY(hipMemcpy(h_df_du, h->hd_data->d_df_du, (size_t)2 * n * n * sizeof(float), hipMemcpyDeviceToHost));
    if (us_per_call) *us_per_call = time_delta_us_timespec(start, end) / (double)(reps > 0 ? reps : 1);
    return 0;
}

}  // extern "C"

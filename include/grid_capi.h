/*
 * grid_capi.h - C ABI of a robot-specialised GRiD library for AMD MI355X (gfx950).
 *
 * One shared library is built per robot (libgrid_<robot>.so) from the header emitted by
 * gridcodegenerator_amd.GRiDCodeGenerator(robot).gen_all_code() plus gridcodegenerator_amd/csrc/grid_capi.hip.
 * The entry points below are what a foreign-function binding (ctypes, cgo, JNI ...) of the reference's emitted
 * C++ host API would bind; each cites the reference interface it replaces.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success or a non-zero hipError_t value; grid_last_error() gives the text (per calling thread).
 *     No entry point ever exit()s or aborts the host process: the reference's host API prints "GPUassert: ..." and exit()s
 *     (reference GRiDCodeGenerator.py:279-286); the shim re-binds the generated header's error hook (GRID_ON_GPU_ERROR) instead;
 *   - T is float ("Suggested Type T is float", reference GRiDCodeGenerator.py:378); every entry point also exists as *_f64
 *     (the T = double instantiation of the same generated kernels; its buffers are allocated by the first *_f64 call);
 *   - layouts (reference algorithms/_forward_dynamics_gradient.py:50,61,168 and SURVEY.md section 8(a) a1):
 *       q_qd_u [k*stride + {0..n | n..2n | 2n..3n}]            inputs, array-of-structs over the batch index k
 *       df_du  [k*2n^2 + col*n + row], col in [0,2n)           = [d qdd/d q | d qdd/d qd], column-major n x 2n
 *       dc_du  same shape as df_du;  Minv [k*n^2 + col*n + row] (upper triangle, column-major);  c, qdd [k*n + i]
 *   - gravity is passed POSITIVE (9.81), as in the reference's emitted code (reference _inverse_dynamics.py:123);
 *   - *_device entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) without synchronising; *_host entry points take HOST pointers, copy in, run, copy out and synchronise.
 *   - one handle per GPU.  Every entry point makes the handle's device current for the duration of the call and restores the
 *     caller's device afterwards, so one process may hold a handle per GPU and call them in any order from any thread;
 *     a single handle is not thread-safe.  grid_forward_dynamics_gradient_multi_host drives G handles at once.
 */
#ifndef GRID_CAPI_H
#define GRID_CAPI_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct grid_handle grid_handle;

/* robot the library was generated for (constants NUM_JOINTS etc., reference GRiDCodeGenerator.py:94-111) */
int grid_num_joints(void);
const char *grid_robot_name(void);
int grid_lanes_per_solve(void);
int grid_suggested_threads(void);
int grid_lds_bytes_per_block(void);
int grid_has_second_order(void); /* 1 if idsva_so / fdsva_so are emitted for this robot (GRID_HAS_IDSVA_SO of the generated header) */
const char *grid_last_error(void);

/* replaces init_robotModel<T>() + init_grid<T>() + init_gridData<T>(max_timesteps)
 * (reference helpers/_topology_helpers.py:715-730, GRiDCodeGenerator.py:160-271) */
int grid_init(int device, int max_timesteps, grid_handle **out);
/* replaces close_grid<T>() (reference GRiDCodeGenerator.py:252-271) */
int grid_close(grid_handle *h);
/* device the handle was created on (-1 for NULL) */
int grid_device(const grid_handle *h);
/* Page-locked host buffers (hipHostMalloc / hipHostFree).  Replaces the pinned h_* members of gridData that the reference's callers fill and read
 * (reference GRiDCodeGenerator.py:160-213 allocates them with malloc; ours with hipHostMalloc).  grid_forward_dynamics_gradient_host overlaps its
 * copies with the kernel when BOTH of its buffers come from here (or are otherwise page-locked); pageable buffers take the sequential form. */
int grid_set_host_chunks(grid_handle *h, int chunks); /* chunks of the pipelined host entry point; 0 = automatic (tuning aid, like grid_set_launch_dims) */
int grid_host_alloc(size_t bytes, void **out);
int grid_host_free(void *p);
/* solves per call the second-order entry points accept on this handle: min(max_timesteps, 1 GiB / record) - the generated init_gridData<T>()
 * caps the d_idsva_so / d_df2 buffers (4 n^3 values per solve; grid_so_max_timesteps<T>() of the generated header).  f64 != 0: the _f64 forms */
int grid_second_order_capacity(const grid_handle *h, int f64);

/* replaces forward_dynamics_gradient<T,false>(hd_data, d_robotModel, gravity, num_timesteps, block, thread, streams)
 * (reference algorithms/_forward_dynamics_gradient.py:186-249): host buffers in, host buffers out, synchronous */
int grid_forward_dynamics_gradient_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df_du);
/* replaces forward_dynamics_gradient_compute_only<T,false> / a direct forward_dynamics_gradient_kernel<T> launch
 * (reference algorithms/_forward_dynamics_gradient.py:113-184,203-234): device buffers, asynchronous on `stream` */
int grid_forward_dynamics_gradient_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity,
                                          float *d_df_du, void *stream);
/* the (q,qd,qdd,Minv)-input overload, USE_QDD_MINV_FLAG=true (reference :126-130,160,233) */
int grid_forward_dynamics_gradient_qdd_minv_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, const float *d_Minv,
                                                   int num_timesteps, float gravity, float *d_df_du, void *stream);
/* launch geometry override for the *_device entry points (0 = library default).  The reference API takes
 * block_dimms/thread_dimms from the caller on every call (reference :198-199). */
int grid_set_launch_dims(grid_handle *h, int blocks, int threads);

/* SURVEY.md section 8(f) "next" rows: the stand-alone algorithms the hot path is composed of */
/* replaces inverse_dynamics_kernel<T> (reference algorithms/_inverse_dynamics.py:371-438); d_qdd may be NULL (qdd = 0) */
int grid_inverse_dynamics_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, int num_timesteps, float gravity,
                                 float *d_c, void *stream);
/* replaces direct_minv_kernel<T> (reference algorithms/_direct_minv.py:478-525) */
int grid_direct_minv_device(grid_handle *h, const float *d_q, int stride_q, int num_timesteps, float *d_Minv, void *stream);
/* replaces forward_dynamics_kernel<T> (reference algorithms/_forward_dynamics.py:149-199) */
int grid_forward_dynamics_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_qdd, void *stream);
/* replaces aba_kernel<T> (reference algorithms/_aba.py:482-537): O(n) articulated-body forward dynamics, same result as grid_forward_dynamics_device */
int grid_aba_device(grid_handle *h, const float *d_q_qd_tau, int stride_q_qd, int num_timesteps, float gravity, float *d_qdd, void *stream);
/* replaces idsva_so_kernel<T> (reference algorithms/_idsva_so.py:958-1028): second-order derivatives of inverse dynamics, 4 n^3 values per solve
 * [d2tau_dq2 | d2tau_dqd2 | d2tau_dvdq | dM_dq]; d_qdd may be NULL (qdd = 0).  Robots without the second-order kernels (grid_has_second_order() == 0) return hipErrorNotSupported;
 * launches with at most IDSVA_SO_SUGGESTED_THREADS threads per block whatever grid_set_launch_dims() says */
int grid_idsva_so_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, const float *d_qdd, int num_timesteps, float gravity,
                         float *d_idsva_so, void *stream);
/* replaces fdsva_so_kernel<T> (reference algorithms/_fdsva_so.py:159-230): second-order derivatives of forward dynamics, 4 n^3 values per solve
 * [d2a_dqdq | d2a_dvdv | d2a_dvdq | d2a_dtdq].  hipErrorNotSupported when grid_has_second_order() == 0; launches with at most
 * FDSVA_SO_SUGGESTED_THREADS threads per block whatever grid_set_launch_dims() says.  Robots whose 4 n^3 record is larger than the LDS of a
 * CU (30 joints: 432 KB; GRID_SO_DIRECT of the generated header) keep the idsva_so tensors in the handle's own workspace: num_timesteps must
 * not exceed grid_second_order_capacity() and calls on one handle must not overlap */
int grid_fdsva_so_device(grid_handle *h, const float *d_q_qd_u, int stride_q_qd_u, int num_timesteps, float gravity, float *d_df2, void *stream);
/* replaces inverse_dynamics_gradient_kernel<T> (reference algorithms/_inverse_dynamics_gradient.py:817-888); d_qdd may be NULL */
int grid_inverse_dynamics_gradient_device(grid_handle *h, const float *d_q_qd, int stride_q_qd, const float *d_qdd, int num_timesteps, float gravity,
                                          float *d_dc_du, void *stream);

/* Host-buffer forms of the stand-alone algorithms: H2D, launch, D2H, synchronous - the semantics of the reference's host wrappers
 * inverse_dynamics<T,USE_QDD_FLAG,USE_COMPRESSED_MEM> (reference algorithms/_inverse_dynamics.py:440-512: stride 2n = compressed, 3n = q_qd_u; NULL h_qdd =
 * USE_QDD_FLAG false), inverse_dynamics_gradient<T,...> (_inverse_dynamics_gradient.py:890-962), direct_minv<T,USE_COMPRESSED_MEM> (_direct_minv.py:527-588:
 * stride n or 3n), forward_dynamics<T> (_forward_dynamics.py:199-265), aba<T> (_aba.py:539-600), idsva_so_host<T,USE_QDD_FLAG> (_idsva_so.py:1030-1090),
 * fdsva_so<T> (_fdsva_so.py:246-316), forward_dynamics_gradient<T,true> (_forward_dynamics_gradient.py:186-249).  num_timesteps <= grid_init's max_timesteps. */
int grid_inverse_dynamics_host(grid_handle *h, const float *h_q_qd, int stride_q_qd, const float *h_qdd, int num_timesteps, float gravity, float *h_c);
int grid_inverse_dynamics_gradient_host(grid_handle *h, const float *h_q_qd, int stride_q_qd, const float *h_qdd, int num_timesteps, float gravity, float *h_dc_du);
int grid_direct_minv_host(grid_handle *h, const float *h_q, int stride_q, int num_timesteps, float *h_Minv);
int grid_forward_dynamics_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_qdd);
int grid_aba_host(grid_handle *h, const float *h_q_qd_tau, int num_timesteps, float gravity, float *h_qdd);
int grid_idsva_so_host(grid_handle *h, const float *h_q_qd_u, const float *h_qdd, int num_timesteps, float gravity, float *h_idsva_so);
int grid_fdsva_so_host(grid_handle *h, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df2);
int grid_forward_dynamics_gradient_qdd_minv_host(grid_handle *h, const float *h_q_qd, int stride_q_qd, const float *h_qdd, const float *h_Minv, int num_timesteps,
                                                 float gravity, float *h_df_du);

/* Multi-GPU form of the hot path: ONE host batch of num_timesteps solves is cut into num_handles
 * contiguous ranges of ceil(num_timesteps/num_handles) solves, range g runs on handles[g] (one handle per GPU, one host thread per handle:
 * H2D, kernel, D2H on that handle's stream), results land in the matching ranges of h_df_du.  No collective: every solve is independent
 * (reference helpers/_code_generation_helpers.py:46-47, the kernels' only cross-k structure is the grid-stride loop).  Synchronous. */
int grid_forward_dynamics_gradient_multi_host(grid_handle **handles, int num_handles, const float *h_q_qd_u, int num_timesteps, float gravity, float *h_df_du);

/* T = double instantiations (reference: template <typename T> on every emitted function, GRiDCodeGenerator.py:312-380): the same entry points with
 * double buffers and a double gravity; their device and pinned host buffers are allocated by the first *_f64 call on a handle.  A block never asks
 * for more LDS than a CU has: where the float block size does not fit in double precision the library launches fewer solves per block. */
int grid_forward_dynamics_gradient_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, int num_timesteps, double gravity,
                                              double *d_df_du, void *stream);
int grid_forward_dynamics_gradient_qdd_minv_device_f64(grid_handle *h, const double *d_q_qd, int stride_q_qd, const double *d_qdd, const double *d_Minv,
                                                       int num_timesteps, double gravity, double *d_df_du, void *stream);
int grid_inverse_dynamics_device_f64(grid_handle *h, const double *d_q_qd, int stride_q_qd, const double *d_qdd, int num_timesteps, double gravity,
                                     double *d_c, void *stream);
int grid_inverse_dynamics_gradient_device_f64(grid_handle *h, const double *d_q_qd, int stride_q_qd, const double *d_qdd, int num_timesteps, double gravity,
                                              double *d_dc_du, void *stream);
int grid_direct_minv_device_f64(grid_handle *h, const double *d_q, int stride_q, int num_timesteps, double *d_Minv, void *stream);
int grid_forward_dynamics_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, int num_timesteps, double gravity, double *d_qdd, void *stream);
int grid_aba_device_f64(grid_handle *h, const double *d_q_qd_tau, int stride_q_qd, int num_timesteps, double gravity, double *d_qdd, void *stream);
int grid_idsva_so_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, const double *d_qdd, int num_timesteps, double gravity,
                             double *d_idsva_so, void *stream);
int grid_fdsva_so_device_f64(grid_handle *h, const double *d_q_qd_u, int stride_q_qd_u, int num_timesteps, double gravity, double *d_df2, void *stream);
int grid_forward_dynamics_gradient_host_f64(grid_handle *h, const double *h_q_qd_u, int num_timesteps, double gravity, double *h_df_du);
int grid_inverse_dynamics_host_f64(grid_handle *h, const double *h_q_qd, int stride_q_qd, const double *h_qdd, int num_timesteps, double gravity, double *h_c);
int grid_inverse_dynamics_gradient_host_f64(grid_handle *h, const double *h_q_qd, int stride_q_qd, const double *h_qdd, int num_timesteps, double gravity, double *h_dc_du);
int grid_direct_minv_host_f64(grid_handle *h, const double *h_q, int stride_q, int num_timesteps, double *h_Minv);
int grid_forward_dynamics_host_f64(grid_handle *h, const double *h_q_qd_u, int num_timesteps, double gravity, double *h_qdd);
int grid_aba_host_f64(grid_handle *h, const double *h_q_qd_tau, int num_timesteps, double gravity, double *h_qdd);
int grid_idsva_so_host_f64(grid_handle *h, const double *h_q_qd_u, const double *h_qdd, int num_timesteps, double gravity, double *h_idsva_so);
int grid_fdsva_so_host_f64(grid_handle *h, const double *h_q_qd_u, int num_timesteps, double gravity, double *h_df2);

/* in-kernel timing probe: replaces forward_dynamics_gradient_single_timing<T> (reference :236-248); returns microseconds per solve */
int grid_forward_dynamics_gradient_single_timing(grid_handle *h, const float *h_q_qd_u, int reps, float gravity, float *h_df_du, double *us_per_call);

#ifdef __cplusplus
}
#endif
#endif /* GRID_CAPI_H */

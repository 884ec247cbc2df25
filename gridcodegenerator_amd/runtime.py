"""Build + ctypes binding of the robot-specialised C-ABI library (include/grid_capi.h).

``build_library(robot)`` runs the generator, then hipcc for gfx950, and leaves ``libgrid_<robot>.so`` in-tree
(gridcodegenerator_amd/_build/<robot>/) so that it travels with the repository snapshot.  ``GridLibrary`` loads it.
There is no CPU fallback: if the library is missing or does not load, GridLibrary raises.
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np

from .GRiDCodeGenerator import GRiDCodeGenerator
from .robot import RobotModel

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
BUILD_DIR = os.path.join(PKG_DIR, "_build")
CAPI_SRC = os.path.join(PKG_DIR, "csrc", "grid_capi.hip")
INCLUDE_DIR = os.path.join(REPO_DIR, "include")

# -fno-slp-vectorize: the SLP vectorizer pairs the unrolled 6-vector arithmetic into v_pk_*_f32, which on gfx950 has the
# same FLOP rate as scalar v_fma_f32 but needs even-aligned register pairs, extra v_mov shuffles and constants held in
# VGPR pairs (measured: +60 VGPRs and scratch spills in the RNEA kernel).
# -fno-signed-zeros -ffinite-math-only: lets the compiler fold the structural zeros of root-link vectors (0*x, x+0); no
# reassociation is enabled, results for finite inputs are unchanged except for the sign of exact zeros (-8.6 % VALU instructions).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fno-signed-zeros", "-ffinite-math-only",
               "-shared", "-fPIC", "-Wno-unused-value",
               "-mllvm", "-amdgpu-kernarg-preload-count=8"]  # kernel arguments arrive in SGPRs with the dispatch (no s_load at the head of every wave): -0.15 us per launch


def library_path(robot_name, build_dir=None):
    return os.path.join(build_dir or BUILD_DIR, robot_name, "libgrid_%s.so" % robot_name)


def generate_header(robot, out_dir, namespace="grid", cols_per_lane=None, tuning=None, debug_mode=False):
    """Runs GRiDCodeGenerator(robot).gen_all_code() with out_dir as the working directory (the generator writes
    <namespace>.cuh into the cwd, like the reference does)."""
    os.makedirs(out_dir, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(out_dir)
    try:
        GRiDCodeGenerator(robot, DEBUG_MODE=debug_mode, FILE_NAMESPACE=namespace, COLS_PER_LANE=cols_per_lane, tuning=tuning).gen_all_code()
    finally:
        os.chdir(cwd)
    return os.path.join(out_dir, namespace + ".cuh")


def build_library(robot, build_dir=None, force=False, extra_flags=(), verbose=False, cols_per_lane=None, tuning=None):
    """robot: a fixture name, a RobotModel, or any URDFParser-style robot object.  Returns the .so path.
    tuning: generation-time knobs (GRiDCodeGenerator.TUNING_DEFAULTS); nothing is read from the environment."""
    if isinstance(robot, str):
        robot = RobotModel.from_fixture(robot)
    name = robot.name
    out_dir = os.path.join(build_dir or BUILD_DIR, name)
    so = library_path(name, build_dir)
    header = generate_header(robot, out_dir, cols_per_lane=cols_per_lane, tuning=tuning)
    stamp = so + ".stamp"
    srcs_mtime = max(os.path.getmtime(p) for p in (header, CAPI_SRC, os.path.join(INCLUDE_DIR, "grid_capi.h")))
    sig = open(header).read() + open(CAPI_SRC).read() + " ".join(HIPCC_FLAGS + list(extra_flags))
    import hashlib
    digest = hashlib.sha256(sig.encode()).hexdigest()
    if not force and os.path.exists(so) and os.path.exists(stamp) and open(stamp).read() == digest:
        return so
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-I" + out_dir, "-I" + INCLUDE_DIR, '-DGRID_ROBOT_NAME="%s"' % name, CAPI_SRC, "-o", so]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(digest)
    return so


_c_float_p = ctypes.POINTER(ctypes.c_float)


def _ptr(x):
    """Accepts a raw device/host address (int), a numpy float32 array, or anything with data_ptr() (torch tensors)."""
    if x is None:
        return ctypes.c_void_p(None)
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        if x.dtype != np.float32 or not x.flags["C_CONTIGUOUS"]:
            raise TypeError("expected a C-contiguous float32 array")
        return ctypes.c_void_p(x.ctypes.data)
    raise TypeError("cannot interpret %r as a pointer" % type(x))


class GridError(RuntimeError):
    pass


class GridLibrary:
    """Thin ctypes mirror of include/grid_capi.h for one robot."""

    def __init__(self, path, device=0, max_timesteps=16384):
        if not os.path.exists(path):
            raise GridError("robot library %s is missing - run gridcodegenerator_amd.runtime.build_library() / __graft_entry__.build(); "
                            "there is no CPU fallback" % path)
        self.path = path
        self.lib = ctypes.CDLL(path)
        L = self.lib
        if not hasattr(L, "grid_second_order_capacity"):
            raise GridError("%s was built from an older grid_capi.hip - rebuild it (gridcodegenerator_amd.runtime.build_library)" % path)
        L.grid_robot_name.restype = ctypes.c_char_p
        L.grid_last_error.restype = ctypes.c_char_p
        self.n = L.grid_num_joints()
        self.robot_name = L.grid_robot_name().decode()
        self.lanes_per_solve = L.grid_lanes_per_solve()
        self.suggested_threads = L.grid_suggested_threads()
        self.lds_bytes_per_block = L.grid_lds_bytes_per_block()
        self.has_second_order = bool(L.grid_has_second_order())
        self.handle = ctypes.c_void_p()
        self._check(L.grid_init(ctypes.c_int(device), ctypes.c_int(max_timesteps), ctypes.byref(self.handle)))
        self.max_timesteps = max_timesteps

    def pinned_empty(self, shape, dtype=np.float32):
        """A NumPy array in page-locked host memory (grid_host_alloc): with such buffers forward_dynamics_gradient_host(..., out=...) overlaps its copies
        with the kernel.  The memory is released when the array (and every view of it) is gone."""
        import weakref

        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        p = ctypes.c_void_p()
        self._check(self.lib.grid_host_alloc(ctypes.c_size_t(max(1, count * dtype.itemsize)), ctypes.byref(p)))
        buf = (ctypes.c_char * max(1, count * dtype.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)
        weakref.finalize(buf, self.lib.grid_host_free, ctypes.c_void_p(p.value))
        return arr

    def second_order_capacity(self, f64=False):
        """Solves per call the second-order entry points accept on this handle (their buffers are capped at 1 GiB each)."""
        return self.lib.grid_second_order_capacity(self.handle, ctypes.c_int(1 if f64 else 0))

    def _check(self, rc):
        if rc != 0:
            raise GridError("%s (code %d)" % (self.lib.grid_last_error().decode(), rc))

    def close(self):
        if self.handle:
            self._check(self.lib.grid_close(self.handle))
            self.handle = ctypes.c_void_p()

    def set_launch_dims(self, blocks=0, threads=0):
        self._check(self.lib.grid_set_launch_dims(self.handle, ctypes.c_int(blocks), ctypes.c_int(threads)))

    # ---- host-buffer entry points (H2D, launch, D2H, synchronous): NumPy arrays in, NumPy arrays out
    def _host_in(self, a, cols, what, dtype=np.float32):
        x = np.ascontiguousarray(a, dtype=dtype)
        if x.ndim != 2 or x.shape[1] not in (cols if isinstance(cols, tuple) else (cols,)):
            raise ValueError("%s must have shape (N, %s)" % (what, cols))
        return x

    def forward_dynamics_gradient_host(self, q_qd_u, gravity=9.81, out=None):
        """out: optional (N, 2n^2) float32 result array (e.g. from pinned_empty(): with page-locked input AND output the C entry point pipelines its copies)."""
        x = q_qd_u if (isinstance(q_qd_u, np.ndarray) and q_qd_u.dtype == np.float32 and q_qd_u.flags["C_CONTIGUOUS"] and q_qd_u.ndim == 2 and q_qd_u.shape[1] == 3 * self.n) \
            else self._host_in(q_qd_u, 3 * self.n, "q_qd_u")  # (no copy of an array that already has the right layout: a copy would leave page-locked memory)
        N = x.shape[0]
        if out is None:
            out = np.empty((N, 2 * self.n * self.n), dtype=np.float32)
        elif out.dtype != np.float32 or out.shape != (N, 2 * self.n * self.n) or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous float32 array of shape (N, 2n^2)")
        self._check(self.lib.grid_forward_dynamics_gradient_host(self.handle, _ptr(x), ctypes.c_int(N), ctypes.c_float(gravity), _ptr(out)))
        return out

    def forward_dynamics_gradient_host_f64(self, q_qd_u, gravity=9.81):
        x = self._host_in(q_qd_u, 3 * self.n, "q_qd_u", np.float64)
        N = x.shape[0]
        out = np.empty((N, 2 * self.n * self.n), dtype=np.float64)
        self._check(self.lib.grid_forward_dynamics_gradient_host_f64(self.handle, ctypes.c_void_p(x.ctypes.data), ctypes.c_int(N), ctypes.c_double(gravity),
                                                                     ctypes.c_void_p(out.ctypes.data)))
        return out

    def host_f64(self, algorithm, *arrays, gravity=9.81):
        """T = double host-buffer entry points: algorithm in {inverse_dynamics, inverse_dynamics_gradient, direct_minv, forward_dynamics, aba,
        idsva_so, fdsva_so}; arrays as for the float methods (q_qd[, qdd] / q / q_qd_u[, qdd]) in float64."""
        n = self.n
        P = lambda a: ctypes.c_void_p(None) if a is None else ctypes.c_void_p(a.ctypes.data)
        # same column rules as the float methods; entry points without a stride argument read exactly 3n values per solve
        first = {"inverse_dynamics": (2 * n, 3 * n), "inverse_dynamics_gradient": (2 * n, 3 * n), "direct_minv": (n, 2 * n, 3 * n)}.get(algorithm, 3 * n)
        a0 = self._host_in(arrays[0], first, "first input of " + algorithm, np.float64)
        N = a0.shape[0]

        def D(a):
            if a is None:
                return None
            x = self._host_in(a, n, "qdd", np.float64)
            if x.shape[0] != N:
                raise ValueError("qdd must have as many rows as the first input")
            return x
        cols = {"inverse_dynamics": n, "inverse_dynamics_gradient": 2 * n * n, "direct_minv": n * n, "forward_dynamics": n, "aba": n,
                "idsva_so": 4 * n ** 3, "fdsva_so": 4 * n ** 3}[algorithm]
        out = np.empty((N, cols), dtype=np.float64)
        fn = getattr(self.lib, "grid_%s_host_f64" % algorithm)
        g = ctypes.c_double(gravity)
        if algorithm in ("inverse_dynamics", "inverse_dynamics_gradient"):
            qdd = D(arrays[1]) if len(arrays) > 1 else None
            rc = fn(self.handle, P(a0), ctypes.c_int(a0.shape[1]), P(qdd), ctypes.c_int(N), g, P(out))
        elif algorithm == "direct_minv":
            rc = fn(self.handle, P(a0), ctypes.c_int(a0.shape[1]), ctypes.c_int(N), P(out))
        elif algorithm == "idsva_so":
            qdd = D(arrays[1]) if len(arrays) > 1 else None
            rc = fn(self.handle, P(a0), P(qdd), ctypes.c_int(N), g, P(out))
        else:
            rc = fn(self.handle, P(a0), ctypes.c_int(N), g, P(out))
        self._check(rc)
        return out

    def forward_dynamics_gradient_qdd_minv_host(self, q_qd, qdd, Minv, gravity=9.81):
        n = self.n
        x = self._host_in(q_qd, (2 * n, 3 * n), "q_qd")
        N = x.shape[0]
        a, M = self._host_in(qdd, n, "qdd"), self._host_in(Minv, n * n, "Minv")
        out = np.empty((N, 2 * n * n), dtype=np.float32)
        self._check(self.lib.grid_forward_dynamics_gradient_qdd_minv_host(self.handle, _ptr(x), ctypes.c_int(x.shape[1]), _ptr(a), _ptr(M), ctypes.c_int(N),
                                                                          ctypes.c_float(gravity), _ptr(out)))
        return out

    def inverse_dynamics_host(self, q_qd, qdd=None, gravity=9.81):
        n = self.n
        x = self._host_in(q_qd, (2 * n, 3 * n), "q_qd")
        a = None if qdd is None else self._host_in(qdd, n, "qdd")
        out = np.empty((x.shape[0], n), dtype=np.float32)
        self._check(self.lib.grid_inverse_dynamics_host(self.handle, _ptr(x), ctypes.c_int(x.shape[1]), _ptr(a), ctypes.c_int(x.shape[0]), ctypes.c_float(gravity), _ptr(out)))
        return out

    def inverse_dynamics_gradient_host(self, q_qd, qdd=None, gravity=9.81):
        n = self.n
        x = self._host_in(q_qd, (2 * n, 3 * n), "q_qd")
        a = None if qdd is None else self._host_in(qdd, n, "qdd")
        out = np.empty((x.shape[0], 2 * n * n), dtype=np.float32)
        self._check(self.lib.grid_inverse_dynamics_gradient_host(self.handle, _ptr(x), ctypes.c_int(x.shape[1]), _ptr(a), ctypes.c_int(x.shape[0]), ctypes.c_float(gravity), _ptr(out)))
        return out

    def direct_minv_host(self, q):
        n = self.n
        x = self._host_in(q, (n, 2 * n, 3 * n), "q")
        out = np.empty((x.shape[0], n * n), dtype=np.float32)
        self._check(self.lib.grid_direct_minv_host(self.handle, _ptr(x), ctypes.c_int(x.shape[1]), ctypes.c_int(x.shape[0]), _ptr(out)))
        return out

    def forward_dynamics_host(self, q_qd_u, gravity=9.81, aba=False):
        x = self._host_in(q_qd_u, 3 * self.n, "q_qd_u")
        out = np.empty((x.shape[0], self.n), dtype=np.float32)
        fn = self.lib.grid_aba_host if aba else self.lib.grid_forward_dynamics_host
        self._check(fn(self.handle, _ptr(x), ctypes.c_int(x.shape[0]), ctypes.c_float(gravity), _ptr(out)))
        return out

    def idsva_so_host(self, q_qd_u, qdd=None, gravity=9.81):
        n = self.n
        x = self._host_in(q_qd_u, 3 * n, "q_qd_u")
        a = None if qdd is None else self._host_in(qdd, n, "qdd")
        out = np.empty((x.shape[0], 4 * n ** 3), dtype=np.float32)
        self._check(self.lib.grid_idsva_so_host(self.handle, _ptr(x), _ptr(a), ctypes.c_int(x.shape[0]), ctypes.c_float(gravity), _ptr(out)))
        return out

    def fdsva_so_host(self, q_qd_u, gravity=9.81):
        n = self.n
        x = self._host_in(q_qd_u, 3 * n, "q_qd_u")
        out = np.empty((x.shape[0], 4 * n ** 3), dtype=np.float32)
        self._check(self.lib.grid_fdsva_so_host(self.handle, _ptr(x), ctypes.c_int(x.shape[0]), ctypes.c_float(gravity), _ptr(out)))
        return out

    def forward_dynamics_gradient_single_timing(self, q_qd_u_one, reps, gravity=9.81):
        x = np.ascontiguousarray(q_qd_u_one, dtype=np.float32).reshape(3 * self.n)
        out = np.empty(2 * self.n * self.n, dtype=np.float32)
        us = ctypes.c_double()
        self._check(self.lib.grid_forward_dynamics_gradient_single_timing(self.handle, _ptr(x), ctypes.c_int(reps), ctypes.c_float(gravity), _ptr(out), ctypes.byref(us)))
        return out, us.value

    # ---- device-pointer entry points (asynchronous on `stream`)
    def forward_dynamics_gradient_device(self, d_q_qd_u, N, d_df_du, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_forward_dynamics_gradient_device(self.handle, _ptr(d_q_qd_u), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N),
                                                                   ctypes.c_float(gravity), _ptr(d_df_du), ctypes.c_void_p(stream)))

    def prepare_forward_dynamics_gradient_device(self, d_q_qd_u, N, d_df_du, stride=None, gravity=9.81, stream=0):
        """Returns a zero-argument callable that enqueues the same launch every time it is called: the ctypes argument objects are built
        once, so a call costs one foreign-function call (an MPC loop re-launching on the same buffers; bench.py's step)."""
        fn = self.lib.grid_forward_dynamics_gradient_device
        args = (self.handle, _ptr(d_q_qd_u), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N), ctypes.c_float(gravity), _ptr(d_df_du), ctypes.c_void_p(stream))
        check = self._check

        def launch():
            rc = fn(*args)
            if rc:
                check(rc)
        return launch

    def forward_dynamics_gradient_device_f64(self, d_q_qd_u, N, d_df_du, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_forward_dynamics_gradient_device_f64(self.handle, _ptr(d_q_qd_u), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N),
                                                                       ctypes.c_double(gravity), _ptr(d_df_du), ctypes.c_void_p(stream)))

    def forward_dynamics_gradient_qdd_minv_device(self, d_q_qd, d_qdd, d_Minv, N, d_df_du, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_forward_dynamics_gradient_qdd_minv_device(self.handle, _ptr(d_q_qd), ctypes.c_int(stride or 3 * self.n), _ptr(d_qdd), _ptr(d_Minv),
                                                                            ctypes.c_int(N), ctypes.c_float(gravity), _ptr(d_df_du), ctypes.c_void_p(stream)))

    def inverse_dynamics_device(self, d_q_qd, d_qdd, N, d_c, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_inverse_dynamics_device(self.handle, _ptr(d_q_qd), ctypes.c_int(stride or 3 * self.n), _ptr(d_qdd), ctypes.c_int(N),
                                                          ctypes.c_float(gravity), _ptr(d_c), ctypes.c_void_p(stream)))

    def direct_minv_device(self, d_q, N, d_Minv, stride=None, stream=0):
        self._check(self.lib.grid_direct_minv_device(self.handle, _ptr(d_q), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N), _ptr(d_Minv), ctypes.c_void_p(stream)))

    def forward_dynamics_device(self, d_q_qd_u, N, d_qdd, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_forward_dynamics_device(self.handle, _ptr(d_q_qd_u), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N),
                                                          ctypes.c_float(gravity), _ptr(d_qdd), ctypes.c_void_p(stream)))

    def aba_device(self, d_q_qd_tau, N, d_qdd, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_aba_device(self.handle, _ptr(d_q_qd_tau), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N),
                                             ctypes.c_float(gravity), _ptr(d_qdd), ctypes.c_void_p(stream)))

    def idsva_so_device(self, d_q_qd_u, d_qdd, N, d_idsva_so, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_idsva_so_device(self.handle, _ptr(d_q_qd_u), ctypes.c_int(stride or 3 * self.n), _ptr(d_qdd), ctypes.c_int(N),
                                                  ctypes.c_float(gravity), _ptr(d_idsva_so), ctypes.c_void_p(stream)))

    def fdsva_so_device(self, d_q_qd_u, N, d_df2, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_fdsva_so_device(self.handle, _ptr(d_q_qd_u), ctypes.c_int(stride or 3 * self.n), ctypes.c_int(N),
                                                  ctypes.c_float(gravity), _ptr(d_df2), ctypes.c_void_p(stream)))

    def inverse_dynamics_gradient_device(self, d_q_qd, d_qdd, N, d_dc_du, stride=None, gravity=9.81, stream=0):
        self._check(self.lib.grid_inverse_dynamics_gradient_device(self.handle, _ptr(d_q_qd), ctypes.c_int(stride or 3 * self.n), _ptr(d_qdd), ctypes.c_int(N),
                                                                   ctypes.c_float(gravity), _ptr(d_dc_du), ctypes.c_void_p(stream)))


class MultiGpuGrid:
    """One process driving G GPUs: G handles of one robot library, the batch cut into G contiguous ranges.
    `devices` may repeat a device (several handles on one GPU: how the split is rehearsed on a one-GPU box)."""

    def __init__(self, path, devices, max_timesteps=16384):
        self.parts = [GridLibrary(path, device=d, max_timesteps=max_timesteps) for d in devices]
        self.n = self.parts[0].n
        self.lib = self.parts[0].lib

    @staticmethod
    def ranges(N, G):
        """[k0, k1) of every device slot - the same arithmetic as csrc/grid_capi.hip: multi_range."""
        per = (N + G - 1) // G
        return [(min(g * per, N), min((g + 1) * per, N)) for g in range(G)]

    def forward_dynamics_gradient_host(self, q_qd_u, gravity=9.81):
        x = self.parts[0]._host_in(q_qd_u, 3 * self.n, "q_qd_u")
        N = x.shape[0]
        out = np.empty((N, 2 * self.n * self.n), dtype=np.float32)
        hs = (ctypes.c_void_p * len(self.parts))(*[p.handle for p in self.parts])
        rc = self.lib.grid_forward_dynamics_gradient_multi_host(hs, ctypes.c_int(len(self.parts)), _ptr(x), ctypes.c_int(N), ctypes.c_float(gravity), _ptr(out))
        self.parts[0]._check(rc)
        return out

    def close(self):
        for p in self.parts:
            p.close()


def load(robot_name, device=0, max_timesteps=16384, build_dir=None):
    return GridLibrary(library_path(robot_name, build_dir), device=device, max_timesteps=max_timesteps)

"""ctypes binding of the CPU oracle (oracle/rbd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (gridcodegenerator_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(march="x86-64-v3", out=None, force=False):
    out = out or os.path.join(HERE, "librbd_oracle.so")
    src = os.path.join(HERE, "rbd_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "MARCH=" + march, "OUT=" + out])
    return out


def _lib(path=None):
    path = path or build()
    if path not in _LIBS:
        _LIBS[path] = ctypes.CDLL(path)
    return _LIBS[path]


class _Model(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("parent", ctypes.c_void_p), ("S_index", ctypes.c_void_p),
                ("X_tree", ctypes.c_void_p), ("I", ctypes.c_void_p), ("damping", ctypes.c_void_p)]


class Oracle:
    """Per-robot handle.  dtype float64 (default) or float32 selects the *_f64 / *_f32 build."""

    def __init__(self, robot, dtype=np.float64, lib_path=None):
        self.lib = _lib(lib_path)
        self.dtype = np.dtype(dtype)
        self.sfx = "_f64" if self.dtype == np.float64 else "_f32"
        self.creal = ctypes.c_double if self.dtype == np.float64 else ctypes.c_float
        arr = robot.to_arrays(self.dtype)
        self.n = int(arr["n"])
        self._keep = arr
        self.model = _Model(self.n, arr["parent"].ctypes.data, arr["S_index"].ctypes.data,
                            arr["X_tree"].ctypes.data, arr["I"].ctypes.data, arr["damping"].ctypes.data)

    def _fn(self, name):
        return getattr(self.lib, name + self.sfx)

    def _a(self, x):
        return np.ascontiguousarray(x, dtype=self.dtype)

    def rnea(self, q, qd, qdd=None, gravity=9.81):
        n = self.n
        q, qd = self._a(q), self._a(qd)
        qdd_a = None if qdd is None else self._a(qdd)
        c = np.zeros(n, self.dtype)
        v, a, f = (np.zeros((n, 6), self.dtype) for _ in range(3))
        self._fn("rbd_rnea")(ctypes.byref(self.model), q.ctypes, qd.ctypes, None if qdd_a is None else qdd_a.ctypes,
                             self.creal(gravity), c.ctypes, v.ctypes, a.ctypes, f.ctypes)
        return c, v.T.copy(), a.T.copy(), f.T.copy()  # (6,n) like the reference

    def minv(self, q, dense=True):
        n = self.n
        q = self._a(q)
        M = np.zeros((n, n), self.dtype)
        self._fn("rbd_minv")(ctypes.byref(self.model), q.ctypes, M.ctypes, ctypes.c_int(1 if dense else 0))
        return M

    def rnea_grad(self, q, qd, qdd, gravity=9.81):
        n = self.n
        q, qd, qdd = self._a(q), self._a(qd), self._a(qdd)
        out = np.zeros((n, 2 * n), self.dtype)
        self._fn("rbd_rnea_grad")(ctypes.byref(self.model), q.ctypes, qd.ctypes, qdd.ctypes, self.creal(gravity), out.ctypes)
        return out

    def fd_grad(self, q, qd, u, gravity=9.81, full=False):
        n = self.n
        q, qd, u = self._a(q), self._a(qd), self._a(u)
        out = np.zeros((n, 2 * n), self.dtype)
        qdd = np.zeros(n, self.dtype)
        Minv = np.zeros((n, n), self.dtype)
        dc = np.zeros((n, 2 * n), self.dtype)
        self._fn("rbd_fd_grad")(ctypes.byref(self.model), q.ctypes, qd.ctypes, u.ctypes, self.creal(gravity), out.ctypes,
                                qdd.ctypes, Minv.ctypes, dc.ctypes)
        return (out, qdd, Minv, dc) if full else out

    def fd_grad_batch(self, q_qd_u, gravity=9.81, nthreads=0):
        """q_qd_u: (N, 3n) AoS -> (N, 2n*n) in the device layout df_du[k][col*n+row]; returns (out, threads_used)."""
        x = self._a(q_qd_u)
        N = x.shape[0]
        out = np.zeros((N, 2 * self.n * self.n), self.dtype)
        used = self._fn("rbd_fd_grad_batch")(ctypes.byref(self.model), ctypes.c_int(N), x.ctypes, ctypes.c_int(x.shape[1]),
                                             self.creal(gravity), out.ctypes, ctypes.c_int(nthreads))
        return out, int(used)

    def rnea_batch(self, q_qd, qdd=None, gravity=9.81):
        x = self._a(q_qd)
        N = x.shape[0]
        qdd_a = None if qdd is None else self._a(qdd)
        out = np.zeros((N, self.n), self.dtype)
        self._fn("rbd_rnea_batch")(ctypes.byref(self.model), ctypes.c_int(N), x.ctypes, ctypes.c_int(x.shape[1]),
                                   None if qdd_a is None else qdd_a.ctypes, self.creal(gravity), out.ctypes)
        return out

    def rnea_grad_batch(self, q_qd, qdd=None, gravity=9.81):
        """(N, >=2n) AoS (+ optional (N, n) qdd) -> (N, 2n*n) in the device layout dc_du[k][col*n+row]."""
        x = self._a(q_qd)
        N = x.shape[0]
        qdd_a = None if qdd is None else self._a(qdd)
        out = np.zeros((N, 2 * self.n * self.n), self.dtype)
        self._fn("rbd_rnea_grad_batch")(ctypes.byref(self.model), ctypes.c_int(N), x.ctypes, ctypes.c_int(x.shape[1]),
                                        None if qdd_a is None else qdd_a.ctypes, self.creal(gravity), out.ctypes)
        return out

    def minv_batch(self, q):
        x = self._a(q)
        N = x.shape[0]
        out = np.zeros((N, self.n * self.n), self.dtype)
        self._fn("rbd_minv_batch")(ctypes.byref(self.model), ctypes.c_int(N), x.ctypes, ctypes.c_int(x.shape[1]), out.ctypes)
        return out

"""Register-resident 6-vector device functions for the HIP/CDNA4 backend.

Replaces the reference's helpers/_spatial_algebra_helpers.py (dot_prod / mx0..5 / mxX / fx_times_v emitters that
operate in place on shared memory, reference lines 122-344) with functions on per-lane register 6-vectors
(``T v[6]`` fully unrolled -> VGPRs).  Every joint transform is applied in its block form
``X = [[E, 0], [B, E]]`` (18 numbers, 27 FMAs) instead of a dense 6x6 (reference `_topology_helpers.py:224,324-328`
already notes TL == BR and TR == 0).  Per-link spatial inertias are emitted as literal-constant matvecs so the
constants ride in the instruction stream (structural zeros are pruned at generation time).
"""

_SPATIAL_LIBRARY = r"""
// ---------------------------------------------------------------------------------------------
// wave-level hand-off: LDS operations of one wavefront execute in issue order, so a lane group
// (which never straddles a wave) only needs the compiler fenced, not an s_barrier.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void grid_wave_sync() {
    // (an LDS-only fence - third argument "local" - was tried: it lets the compiler hoist the 6n global loads of the inertia columns
    //  above every sync, which costs 900 B of scratch per lane on the 30-DoF robot and gains nothing on the 7-DoF one)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Returns x unchanged but opaque to the optimizer: used once per batch-loop iteration on the lane index so that
// lane-dependent (loop-invariant) values are recomputed per iteration instead of being hoisted and kept live - on gfx950 the
// hoisted values of a 5k-instruction loop body ended up in scratch.
__device__ __forceinline__ int grid_loop_variant(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Pins a value (or 6-vector) in registers at this point of the program.  Without it LLVM sinks whole dependency chains
// (thousands of FMAs) down into the final `if (lane < n)` store block because that is their only use: the LDS loads that feed
// them have to stay above the wave-level syncs, so everything loaded in between was spilled to scratch.
// The asm is deliberately NOT volatile: a plain asm with an in/out operand is enough to keep LLVM from sinking the chain (it never
// duplicates or sinks inline asm), while volatile asms are additionally ordered among themselves, which serialised independent
// work for the scheduler (measured: 14.7 us -> 13.4 us per launch without `volatile`).
template <typename T>
__device__ __forceinline__ void grid_pin(T &x) { asm("" : "+v"(x)); }
template <typename T>
__device__ __forceinline__ void grid_pin6(T (&v)[6]) {
    #pragma unroll
    for (int r = 0; r < 6; r++) { asm("" : "+v"(v[r])); }
}

// value of x on the lane that owns joint `src` of THIS solve (cross-lane read inside the lane group, no LDS storage)
template <typename V>
__device__ __forceinline__ V grid_group_shfl(const V x, const int src) { @@GROUP_SHFL@@ }

// 16-byte global store of a finished output record chunk (the address is only 4-byte aligned in general; gfx950 handles that)
template <typename T>
__device__ __forceinline__ void grid_store4(T *dst, const T (&v)[4]) { @@STORE4@@ }

// reciprocal: one v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence
__device__ __forceinline__ float grid_rcp(const float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double grid_rcp(const double x) { return 1.0/x; }

// X is stored compactly per joint: X[0..8] = E (row-major 3x3, top-left == bottom-right block),
// X[9..17] = B (bottom-left block); the top-right block is identically zero.
// scheduling fence: nothing is moved across it by the instruction scheduler (keeps software-pipelined loops from being re-clustered)
#if defined(__HIP_DEVICE_COMPILE__)
#define GRID_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GRID_SCHED_FENCE() do {} while (0)
#endif
#define GRID_X_STRIDE 20  // floats per joint in LDS (18 used; keeps every joint 16-byte aligned)

template <typename T>
__device__ __forceinline__ void grid_load_X(T (&X)[18], const T *s_Xj) {
    #pragma unroll
    for (int i = 0; i < 18; i++) { X[i] = s_Xj[i]; }
}

// y = X * v   (motion vector, parent -> child coordinates)
template <typename T>
__device__ __forceinline__ void grid_xmul(T (&y)[6], const T (&X)[18], const T (&v)[6]) {
    #pragma unroll
    for (int r = 0; r < 3; r++) {
        y[r]   = X[3*r]*v[0] + X[3*r+1]*v[1] + X[3*r+2]*v[2];
        y[r+3] = X[9+3*r]*v[0] + X[9+3*r+1]*v[1] + X[9+3*r+2]*v[2] + X[3*r]*v[3] + X[3*r+1]*v[4] + X[3*r+2]*v[5];
    }
}

// y += X^T * f   (force vector, child -> parent coordinates)
template <typename T>
__device__ __forceinline__ void grid_xtmul_peq(T (&y)[6], const T (&X)[18], const T (&f)[6]) {
    #pragma unroll
    for (int r = 0; r < 3; r++) {
        y[r]   += X[r]*f[0] + X[3+r]*f[1] + X[6+r]*f[2] + X[9+r]*f[3] + X[12+r]*f[4] + X[15+r]*f[5];
        y[r+3] += X[r]*f[3] + X[3+r]*f[4] + X[6+r]*f[5];
    }
}

// y = X^T * f
template <typename T>
__device__ __forceinline__ void grid_xtmul(T (&y)[6], const T (&X)[18], const T (&f)[6]) {
    #pragma unroll
    for (int r = 0; r < 3; r++) {
        y[r]   = X[r]*f[0] + X[3+r]*f[1] + X[6+r]*f[2] + X[9+r]*f[3] + X[12+r]*f[4] + X[15+r]*f[5];
        y[r+3] = X[r]*f[3] + X[3+r]*f[4] + X[6+r]*f[5];
    }
}

// out += alpha * crm(vec) * S   for the unit motion subspace S = e_SIND  (the reference's mx0..mx5 family)
template <typename T, int SIND>
__device__ __forceinline__ void grid_mxS_peq(T (&out)[6], const T (&vec)[6], const T alpha) {
    if (SIND == 0) { out[1] += vec[2]*alpha; out[2] -= vec[1]*alpha; out[4] += vec[5]*alpha; out[5] -= vec[4]*alpha; }
    if (SIND == 1) { out[0] -= vec[2]*alpha; out[2] += vec[0]*alpha; out[3] -= vec[5]*alpha; out[5] += vec[3]*alpha; }
    if (SIND == 2) { out[0] += vec[1]*alpha; out[1] -= vec[0]*alpha; out[3] += vec[4]*alpha; out[4] -= vec[3]*alpha; }
    if (SIND == 3) { out[4] += vec[2]*alpha; out[5] -= vec[1]*alpha; }
    if (SIND == 4) { out[3] -= vec[2]*alpha; out[5] += vec[0]*alpha; }
    if (SIND == 5) { out[3] += vec[1]*alpha; out[4] -= vec[0]*alpha; }
}

// out += crf(a) * b   (force cross product; the reference's fx_times_v_peq)
template <typename T>
__device__ __forceinline__ void grid_fxv_peq(T (&out)[6], const T (&a)[6], const T (&b)[6]) {
    out[0] += -a[2]*b[1] + a[1]*b[2] - a[5]*b[4] + a[4]*b[5];
    out[1] +=  a[2]*b[0] - a[0]*b[2] + a[5]*b[3] - a[3]*b[5];
    out[2] += -a[1]*b[0] + a[0]*b[1] - a[4]*b[3] + a[3]*b[4];
    out[3] += -a[2]*b[4] + a[1]*b[5];
    out[4] +=  a[2]*b[3] - a[0]*b[5];
    out[5] += -a[1]*b[3] + a[0]*b[4];
}

template <typename T>
__device__ __forceinline__ void grid_zero6(T (&v)[6]) {
    #pragma unroll
    for (int r = 0; r < 6; r++) { v[r] = static_cast<T>(0); }
}

template <typename T>
__device__ __forceinline__ T grid_dot6(const T (&a)[6], const T (&b)[6]) {
    return a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3] + a[4]*b[4] + a[5]*b[5];
}
"""


def _lit(x):
    """C++ literal for a model constant (double literal, cast to T at compile time by the caller)."""
    return repr(float(x))


def _rigid_body_params(I, tol=1e-12):
    """If I is a rigid-body spatial inertia [[Ibar, h~],[h~^T, m 1]] return (Ibar(3x3), h(3), m), else None."""
    import numpy as np
    m = I[3, 3]
    if not (np.allclose(I[3:, 3:], m * np.eye(3), atol=tol) and np.allclose(I, I.T, atol=tol)):
        return None
    H = I[:3, 3:]
    h = np.array([H[2, 1], H[0, 2], H[1, 0]])
    Hs = np.array([[0, -h[2], h[1]], [h[2], 0, -h[0]], [-h[1], h[0], 0]])
    if not np.allclose(H, Hs, atol=tol):
        return None
    return I[:3, :3], h, m


def gen_spatial_algebra_helpers(self):
    """Emit the register 6-vector library plus one literal-constant inertia matvec per link.

    Link inertias are rigid-body inertias (10 parameters: Ibar symmetric 3x3, h = m*c, m), so
    I*[w;v] = [Ibar w + h x v ; m v - h x w] costs 24 FMAs with 10 constants instead of a dense 36/36;
    a general symmetric 6x6 (e.g. a caller-supplied composite inertia) falls back to the dense form."""
    lib = _SPATIAL_LIBRARY
    if self.tuning["no_pins"]:  # experiment: no pins at all
        lib = lib.replace('__device__ __forceinline__ void grid_pin(T &x) { asm("" : "+v"(x)); }', '__device__ __forceinline__ void grid_pin(T &x) { (void)x; }')
        lib = lib.replace('for (int r = 0; r < 6; r++) { asm("" : "+v"(v[r])); }', 'for (int r = 0; r < 6; r++) { (void)v[r]; }')
    if self.tuning["no_wave_barrier"]:  # experiment: fences only
        lib = lib.replace("    __builtin_amdgcn_wave_barrier();\n", "")
    store4 = "__builtin_memcpy(dst, v, 4*sizeof(T));"
    if self.tuning["nt_store"]:  # streaming (non-temporal) output stores: the record is never re-read by the kernel (+2 % on the 7-DoF arm)
        store4 = ("\n#if defined(__HIP_DEVICE_COMPILE__)\n    typedef T vec4_t __attribute__((ext_vector_type(4), aligned(4))); vec4_t x = {v[0], v[1], v[2], v[3]}; "
                  "__builtin_nontemporal_store(x, reinterpret_cast<vec4_t *>(dst));\n#else\n    __builtin_memcpy(dst, v, 4*sizeof(T));\n#endif\n")
    if getattr(self, "lane_interleave", False):  # thread t of a 16-lane row: solve t & 1, joint t >> 1
        lib = lib.replace("@@GROUP_SHFL@@", "const int wl = __lane_id(); return __shfl(x, (wl & 0x31) | (src << 1), 64);")
    else:
        lib = lib.replace("@@GROUP_SHFL@@", "return __shfl(x, src, %d);" % self.lanes_per_solve)
    for line in lib.replace("@@STORE4@@", store4).strip("\n").split("\n"):
        self.gen_add_code_line(line)
    self.gen_add_code_line("")
    m = self.model
    C = lambda x: "static_cast<T>(" + _lit(x) + ")"
    ZERO = "static_cast<T>(0)"
    for i in range(m.n):
        PE, PB = m.X_pattern[i]
        nnz = 2 * int(PE.sum()) + int(PB.sum())
        self.gen_add_code_line("// joint %d transform products, specialised to the %d of 27 block entries of X_%d(q) that can be non-zero" % (i, nnz, i))
        # y = X v
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__device__ __forceinline__ void grid_xmul_%d(T (&y)[6], const T (&X)[18], const T (&v)[6]) {" % i, True)
        for r in range(3):
            top = ["X[%d]*v[%d]" % (3 * r + c, c) for c in range(3) if PE[r, c]]
            bot = ["X[%d]*v[%d]" % (9 + 3 * r + c, c) for c in range(3) if PB[r, c]] + ["X[%d]*v[%d]" % (3 * r + c, 3 + c) for c in range(3) if PE[r, c]]
            self.gen_add_code_line("y[%d] = %s; y[%d] = %s;" % (r, " + ".join(top) if top else ZERO, r + 3, " + ".join(bot) if bot else ZERO))
        self.gen_add_end_function()
        # y (+)= X^T f
        for peq in (False, True):
            self.gen_add_code_line("template <typename T>")
            self.gen_add_code_line("__device__ __forceinline__ void grid_xtmul%s_%d(T (&y)[6], const T (&X)[18], const T (&f)[6]) {" % ("_peq" if peq else "", i), True)
            for r in range(3):
                top = ["X[%d]*f[%d]" % (3 * k + r, k) for k in range(3) if PE[k, r]] + ["X[%d]*f[%d]" % (9 + 3 * k + r, 3 + k) for k in range(3) if PB[k, r]]
                bot = ["X[%d]*f[%d]" % (3 * k + r, 3 + k) for k in range(3) if PE[k, r]]
                if peq:
                    line = ""
                    if top:
                        line += "y[%d] += %s; " % (r, " + ".join(top))
                    if bot:
                        line += "y[%d] += %s;" % (r + 3, " + ".join(bot))
                    if line:
                        self.gen_add_code_line(line)
                else:
                    self.gen_add_code_line("y[%d] = %s; y[%d] = %s;" % (r, " + ".join(top) if top else ZERO, r + 3, " + ".join(bot) if bot else ZERO))
            self.gen_add_end_function()
    for i in range(m.n):
        I = m.I[i]
        rb = _rigid_body_params(I)
        self.gen_add_code_line("// y = I[" + str(i) + "] * x  (spatial inertia of link " + str(i) + " as instruction-stream constants)")
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__device__ __forceinline__ void grid_imul_" + str(i) + "(T (&y)[6], const T (&x)[6]) {", True)
        if rb is None:
            for r in range(6):
                terms = [(C(I[r, c]) + "*x[" + str(c) + "]") for c in range(6) if I[r, c] != 0.0]
                self.gen_add_code_line("y[" + str(r) + "] = " + (" + ".join(terms) if terms else "static_cast<T>(0)") + ";")
        else:
            Ib, h, mass = rb
            self.gen_add_code_line("// rigid body: y = [Ibar w + h x v ; m v - h x w]")
            cross = [(1, 2), (2, 0), (0, 1)]  # (h x v)[r] = h[a] v[b] - h[b] v[a]
            for r in range(3):
                a, b = cross[r]
                terms = [(C(Ib[r, c]) + "*x[" + str(c) + "]") for c in range(3) if Ib[r, c] != 0.0]
                if h[a] != 0.0:
                    terms.append(C(h[a]) + "*x[" + str(3 + b) + "]")
                if h[b] != 0.0:
                    terms.append(C(-h[b]) + "*x[" + str(3 + a) + "]")
                self.gen_add_code_line("y[" + str(r) + "] = " + (" + ".join(terms) if terms else "static_cast<T>(0)") + ";")
            for r in range(3):
                a, b = cross[r]
                terms = [C(mass) + "*x[" + str(3 + r) + "]"]
                if h[a] != 0.0:
                    terms.append(C(-h[a]) + "*x[" + str(b) + "]")
                if h[b] != 0.0:
                    terms.append(C(h[b]) + "*x[" + str(a) + "]")
                self.gen_add_code_line("y[" + str(3 + r) + "] = " + " + ".join(terms) + ";")
        self.gen_add_end_function()


def gen_mx_func_call_for_cpp(self, S_index, out_name, vec_name, alpha="static_cast<T>(1)"):
    """Generation-time dispatch on the joint's motion subspace (reference helpers/_spatial_algebra_helpers.py:1-33
    picks mx<S> at generation time when all S are equal, else a runtime switch; joints are unrolled here so the
    subspace is always a compile-time constant)."""
    return "grid_mxS_peq<T," + str(S_index) + ">(" + out_name + ", " + vec_name + ", " + alpha + ");"

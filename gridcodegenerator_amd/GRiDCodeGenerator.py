"""GRiDCodeGenerator for AMD MI355X (gfx950 / CDNA4).

Keeps the reference's Python entry point (reference GRiDCodeGenerator.py:54 ctor, :309 gen_all_code) and the
emitted ``ALGORITHM_inner / _device / _kernel / host`` surface (reference :312-380), but emits HIP designed for
wave64 hardware instead of CUDA:

    from gridcodegenerator_amd import GRiDCodeGenerator
    GRiDCodeGenerator(robot).gen_all_code()      # writes <FILE_NAMESPACE>.cuh in the cwd (HIP source)

``robot`` is any object offering the URDFParser-style getters; gridcodegenerator_amd.robot
provides one (RobotModel) for JSON/URDF descriptions.
"""
import numpy as np

from .robot import DuckRobot

# Generation-time tuning knobs.  They are reachable ONLY through the constructor (``GRiDCodeGenerator(robot, tuning={...})``): nothing in
# the process environment changes the generated code.  The "ablation" group emits kernels whose RESULTS ARE WRONG on purpose (timing
# experiments of tools/); it must be requested with the extra key ``"allow_wrong_results": True``.
TUNING_DEFAULTS = {
    "cols_per_lane": 2,         # 2: lane j owns d/dq_j and d/dqd_j; 1: one derivative column per lane (lane group = next pow2 >= 2n)
    "gradient_walk": "auto",    # auto | registers | lds | tipframe | branch: which formulation forward_dynamics_gradient is emitted in
    "fuse_fd": True,            # fused M^-1 || RNEA sweep of the column-walk forward dynamics (robots with <= 9 joints)
    "reuse_rnea": False,        # column walk re-uses v, I v, fx(v) I v of the RNEA(qdd=0) pass (measured slower: 16.6 vs 15.0 us)
    "min_waves": 0,             # second __launch_bounds__ argument (minimum waves per SIMD); 0 = compiler's choice
    "so_unroll": None,          # inner-loop unrolling of idsva_so (None = full; subtree mapping only)
    "so_mapping": "balanced",   # balanced | subtree: work distribution of the idsva_so main loops (algorithms/_idsva_so.py: gen_idsva_so_items)
    "so_loops": "dots",         # dots | mxm: loop bodies of the balanced idsva_so main loops - every cross product folded into per-item vectors, a step is dot products
                                # only (algorithms/_idsva_so.py: _SO_FOLD), or the round-2 bodies with motion cross products per step
    "so_origin": "joint",       # joint | base: tree form of the second-order kernels - link inertias, Coriolis matrices, forces and their subtree composites about the origin of every
                                # joint's own frame (shifted from child to parent on the way up; the vectors of joints m, l are shifted to joint c's origin inside the items), or
                                # everything about the base origin (round 2: entries of light distal links are then small differences of m d^2 terms - fp32 errors of 4e-7 of
                                # max|dM_dq| became 8e-4 of max|d2a_dtdq| behind the two products with M^-1 in fdsva_so on the 12-DoF tree)
    "so_split": True,           # fdsva_so of robots whose records do not fit LDS and that have several base-rooted components: host wrappers / C ABI run two kernels
                                # (algorithms/_fdsva_so.py: gen_fdsva_so_split) - the contraction with one block per solve instead of a lane group
    "so_hold": 0,               # fdsva_so contraction from the compact record: the results of so_hold adjacent k of a lane stay in registers and are stored together - whole runs of the result written at once (0: every k stores inside the k loop).
                                # Measured on the 7-DoF arm @65 536 (profiles/ab/r3_fdsva_hold.jsonl): 4 cuts the HBM write traffic from 2.03x to 1.35x the result but its rolled L loop (the unrolled one spills) costs 17 % more instructions: 245 -> 258 us
    "so_blocked": True,         # fdsva_so of robots with several base-rooted components: the contraction runs per component (M^-1, df/du and the idsva_so tensors are block
                                # diagonal over them: 7.5x fewer multiply-adds on the 30-DoF humanoid, 64x on the quadruped; algorithms/_fdsva_so.py: _BLOCKED)
    "so_fused": True,           # fdsva_so of serial chains: the forward-dynamics-gradient inner goes straight on to the idsva_so main loops with the per-joint quantities it holds
                                # in registers (one pass over frames, velocities and composites instead of two; algorithms/_tip_frame_gradient.py: with_so)
    "so_stage": "auto",         # auto | compact | dense: LDS staging of the idsva_so record - compact = every value once (symmetric entries, no structural zeros), expanded
                                # through the table grid_so_expand on the way out (auto: serial chains whose record is staged in LDS)
    "min_lanes": 8,             # smallest lane group (8 | 16 | 32 | 64): wider groups than the joint count needs leave lanes idle in the first-order kernels but
                                # give the item loops of the second-order kernels more lanes per solve (and fewer solves' staging per wave)
    "so_lanes": "auto",         # auto | off | 16 | 32: lane-group width of the second-order KERNELS.  Robots with 8- or 16-lane groups get a second instance of the
                                # library for groups twice as wide in the nested namespace `wide` and idsva_so_host / fdsva_so (and the C ABI) launch its kernels:
                                # half the staging per wave, twice the lanes in the item loops (7-DoF arm, 65 536 solves: idsva_so 289 -> 195 us)
    "so_direct": "auto",        # auto | True: second-order kernels write their 4 n^3 record straight to global memory instead of staging it in LDS
                                # (auto = only where the record does not fit LDS, algorithms/_idsva_so.py: gen_idsva_so_direct)
    "lane_interleave": True,    # 8-lane groups: the two solves of a 16-lane DPP row interleave (solve = lane parity, joint = lane / 2), so that the row's end IS
                                # the solve's end for the lane-group scans: no masked multiply with the neighbouring solve's values (0 * NaN), no masks at all
    "scan_form": "add",         # add | fmac: instruction of an unmasked scan step (interleaved 8-lane groups): v_add_f32_dpp x, x, x or v_fmac_f32_dpp x, x, 1.0
    "dpp_asm": True,            # lane-group scans as single v_fmac_f32_dpp instructions (inline asm) instead of builtin DPP move + FMA
    "tip_chain": "select",      # select | lds: how the tip-frame chain hands (R, p) to the owning lane
    "nt_store": True,           # non-temporal output stores
    "stagger": 0,               # forward_dynamics_gradient kernel: waves in odd wave slots of their SIMD start k*64 cycles late (0 = off), so that the two
                                # waves of a SIMD do not load, compute and store in lock-step
    "lds_pad": 0,               # occupancy experiment: extra elements per solve in the forward-dynamics-gradient slice of branch-frame robots
    "stream_out": "auto",       # auto | True | False: forward_dynamics_gradient kernel of branch-frame robots stages ONE half of the record in LDS at a time
                                # (dc/dqd columns parked compactly while dc/dq is assembled, both solved together, stored half after half);
                                # auto = where that raises the resident waves per CU (the 30-DoF humanoid: 7 -> 8)
    "branch_walk": "auto",      # path | owner | auto: entries of the branch-frame path - every lane walks its root path with running sums, or dot products in the frame of the ancestor's branch with the owner's vectors fetched across lanes (algorithms/_branch_frame_gradient.branch_owner_walk)
    "factor_preload": True,     # branch-frame path: the column solves read the factors of a component into registers with 16-byte LDS loads (False: one 4-byte read per use)
    "branch_chain": "auto",     # auto | walk | scan: frames of the branch-frame path - every lane walks its root path (D steps), or a log-step scan of rigid transforms
                                # over the lanes of a branch + re-expression of the ancestors' joint axes level by level (algorithms/_branch_frame_gradient.py)
    "factor_split": "auto",     # auto | branch | component: who eliminates which pivots of the tree-sparse factorisation on the branch-frame path
                                # (algorithms/_branch_frame_gradient.py: branch_factor_by_branch)
    "fast_sincos": True,        # fp32 joint angles: branch-free Cody-Waite + minimax polynomials (29 instructions) instead of the math library's sincosf (120)
    "composite_scan": "f32",    # f32 | f64: precision of the suffix sums of the link inertias (tip/branch-frame paths); f64 = exact sums, rounded once
    "base_origin": "auto",      # auto | off | <joint position>: tip-frame path - the joint-space inertia entries of the base half of a chain are
                                # evaluated about the origin of this joint instead of the tip (fp32 accuracy); auto = L // 2 for L >= 5
}
TUNING_ABLATION = {
    "debug_stop": 0,            # truncate the kernel after a phase / cycle stamps (tools/prof_ablation.sh, tools/phase_stamps.py); 21 = tip-frame inner without its frame chain
    "no_pins": False,           # drop the register pins
    "no_wave_barrier": False,   # drop the wave barrier of grid_wave_sync (fences only)
    "out_half": False,          # timing experiment: the output image of a solve overlaps its neighbour's (half the staging LDS, wrong results): what would more resident waves buy?
    "no_store": False,          # skip the final global store of every kernel (how much of a launch is the output leaving the chip?)
    "round_probe": (),          # accuracy diagnosis (tools/precision_probe.py): stages of the tip-frame inner whose results are rounded to fp32
                                # inside the T = double instantiation - shows which stage's fp32 rounding the final error comes from
}


def resolve_tuning(tuning=None, COLS_PER_LANE=None):
    t = dict(TUNING_DEFAULTS)
    t.update(TUNING_ABLATION)
    tuning = dict(tuning or {})
    allow = bool(tuning.pop("allow_wrong_results", False))
    for k, v in tuning.items():
        if k in TUNING_ABLATION:
            if v != TUNING_ABLATION[k] and not allow:
                raise ValueError("tuning key %r produces wrong results by design; pass allow_wrong_results=True with it (timing experiments only)" % k)
        elif k not in TUNING_DEFAULTS:
            raise ValueError("unknown tuning key %r (known: %s)" % (k, ", ".join(sorted(TUNING_DEFAULTS))))
        t[k] = v
    if COLS_PER_LANE is not None:
        t["cols_per_lane"] = int(COLS_PER_LANE)
    if t["gradient_walk"] not in ("auto", "registers", "lds", "tipframe", "branch"):
        raise ValueError("tuning['gradient_walk'] must be auto, registers, lds, tipframe or branch")
    if t["cols_per_lane"] not in (1, 2):
        raise ValueError("tuning['cols_per_lane'] must be 1 or 2")
    if t["tip_chain"] not in ("select", "lds"):
        raise ValueError("tuning['tip_chain'] must be select or lds")
    if t["so_origin"] not in ("joint", "base"):
        raise ValueError("tuning['so_origin'] must be joint or base")
    if t["so_loops"] not in ("dots", "mxm"):
        raise ValueError("tuning['so_loops'] must be dots or mxm")
    return t


class GRiDCodeGenerator:
    # emission primitives, device math, model constants (free functions taking self, like the reference's layout)
    from .helpers import gen_add_code_line, gen_add_code_lines, gen_add_end_control_flow, gen_add_end_function, \
        gen_add_func_doc, gen_add_serial_ops, gen_add_parallel_loop, gen_add_sync, gen_var_in_list, gen_var_not_in_list, gen_add_multi_threaded_select, \
        gen_lane_mask_test, gen_kernel_prologue, gen_kernel_load_inputs, gen_kernel_save_result, gen_kernel_save_result_expanded, gen_kernel_load_inputs_single_timing, gen_kernel_save_result_single_timing, \
        gen_static_array_ind_2d, gen_static_array_ind_3d, gen_add_debug_print_code_line, gen_add_debug_print_code_lines, \
        gen_spatial_algebra_helpers, gen_mx_func_call_for_cpp, \
        gen_lds_layout, gen_model_constant_table, gen_get_XI_size, gen_topology_helpers_size, gen_init_XImats, gen_init_topology_helpers, gen_init_robotModel, \
        gen_load_update_XImats_helpers_function_call, gen_load_update_XImats_helpers, gen_topology_sparsity_helpers_python, gen_topology_helpers_pointers_for_cpp

    # algorithms on the forward-dynamics-gradient path
    from .algorithms import gen_tree_traversal, \
        gen_inverse_dynamics_inner_temp_mem_size, gen_inverse_dynamics_inner_function_call, gen_inverse_dynamics_inner, \
        gen_inverse_dynamics_device, gen_inverse_dynamics_kernel, gen_inverse_dynamics_host, gen_inverse_dynamics, \
        gen_direct_minv_inner_temp_mem_size, gen_direct_minv_inner_function_call, gen_direct_minv_inner, gen_direct_minv_inner_header, gen_direct_minv_inner_body, \
        gen_direct_minv_device, gen_direct_minv_kernel, gen_direct_minv_host, gen_direct_minv, \
        gen_forward_dynamics_inner_temp_mem_size, gen_forward_dynamics_finish_function_call, gen_forward_dynamics_finish, \
        gen_forward_dynamics_inner_function_call, gen_forward_dynamics_inner, gen_forward_dynamics_device, gen_forward_dynamics_kernel, \
        gen_forward_dynamics_host, gen_forward_dynamics, \
        gen_aba_inner_temp_mem_size, gen_aba_inner_function_call, gen_aba_inner, gen_aba_device, gen_aba_kernel, gen_aba_host, gen_aba, \
        gen_idsva_so_available, gen_idsva_so_mode, gen_idsva_so_direct, gen_idsva_so_compact, gen_idsva_so_blocks, gen_idsva_so_packed, gen_idsva_so_blocks_layout, gen_idsva_so_chain_compact_layout, gen_idsva_so_compact_layout, gen_idsva_so_rec, gen_idsva_so_tree_tables, gen_idsva_so_items, gen_idsva_so_items_table, gen_idsva_so_lds_layout, gen_idsva_so_inner_temp_mem_size, gen_idsva_so_inner_function_call, gen_idsva_so_inner, gen_idsva_so_device, gen_idsva_so_kernel, gen_idsva_so_host, gen_idsva_so, \
        gen_fdsva_so_inner_temp_mem_size, gen_fdsva_so_stage_size, gen_fdsva_so_fused_layout, gen_fdsva_so_components, gen_fdsva_so_split, gen_fdsva_so_split_kernels, gen_fdsva_so_lds_per_solve, gen_fdsva_so_fused_device, gen_fdsva_so_inner, gen_fdsva_so_device, gen_fdsva_so_kernel, gen_fdsva_so_host, gen_fdsva_so, \
        gen_inverse_dynamics_gradient_inner_temp_mem_size, gen_inverse_dynamics_gradient_kernel_max_temp_mem_size, \
        gen_inverse_dynamics_gradient_inner_function_call, gen_inverse_dynamics_gradient_inner, gen_dc_du_to_lds, gen_gradient_slots, gen_gradient_outputs_decl, \
        gen_inverse_dynamics_gradient_device, gen_inverse_dynamics_gradient_kernel, gen_inverse_dynamics_gradient_host, gen_inverse_dynamics_gradient, \
        gen_forward_dynamics_gradient_inner_temp_mem_size, gen_forward_dynamics_gradient_kernel_max_temp_mem_size, \
        gen_forward_dynamics_gradient_inner_python, gen_forward_dynamics_gradient_device, gen_forward_dynamics_gradient_stream_device, gen_forward_dynamics_gradient_kernel, \
        gen_forward_dynamics_gradient_host, gen_forward_dynamics_gradient, gen_forward_dynamics_gradient_device_function_call, \
        gen_tip_frame_link_constants, gen_tip_frame_joint_offset, gen_tip_frame_library, gen_forward_dynamics_gradient_inner_tip, \
        gen_forward_dynamics_gradient_inner_tip_function_call, gen_tip_frame_gradient, gen_tip_frame_fused_so, \
        gen_inverse_dynamics_inner_tip, gen_inverse_dynamics_gradient_inner_tip, gen_forward_dynamics_inner_tip, gen_direct_minv_inner_tip, gen_tip_frame_components, \
        gen_branch_frame_plan, gen_branch_frame_constants, gen_branch_frame_library, gen_branch_frame_components, gen_forward_dynamics_gradient_inner_branch, gen_forward_dynamics_gradient_inner_branch_stream, \
        gen_forward_dynamics_gradient_inner_branch_function_call

    # NumPy debug helpers with the reference's names and signatures (reference GRiDCodeGenerator.py:50-51, README "Additional Features")
    from ._test import test_rnea, test_minv, test_rnea_grad, test_fd_grad

    def __init__(self, robotObj, DEBUG_MODE=False, NEED_PRINT_MAT=False, USE_DYNAMIC_SHARED_MEM=True, FILE_NAMESPACE="grid", COLS_PER_LANE=None, tuning=None):
        if not USE_DYNAMIC_SHARED_MEM:
            # reference GRiDCodeGenerator.py:54,61 / helpers/_topology_helpers.py:151-153: static `__shared__ T s_XImats[...]` per function.
            # The lane-group kernels size their LDS by the caller's block size (solves per block), which a static array cannot express.
            raise NotImplementedError("USE_DYNAMIC_SHARED_MEM=False (static __shared__ arrays) is not supported by the lane-group kernels: "
                                      "their LDS slice count depends on the launch's block size; keep the default (dynamic LDS)")
        self.tuning = resolve_tuning(tuning, COLS_PER_LANE)
        self._ctor = dict(DEBUG_MODE=DEBUG_MODE, NEED_PRINT_MAT=NEED_PRINT_MAT, COLS_PER_LANE=COLS_PER_LANE, tuning=dict(tuning or {}))
        self.nested = False  # True on the instance that emits the nested `wide` library of another generator
        self.robot = robotObj
        self.model = DuckRobot(robotObj)  # numeric tables; raises for robots outside the supported joint models
        self.code_str = ""
        self.indent_level = 0
        self._cur_joint = None
        self.DEBUG_MODE = DEBUG_MODE
        self.gen_print_mat = DEBUG_MODE or NEED_PRINT_MAT
        self.use_dynamic_shared_mem_flag = True  # the lane-group kernels always carve their LDS slice from dynamic LDS
        self.file_namespace = FILE_NAMESPACE
        n = self.model.n
        if n > 64:
            raise NotImplementedError("robots with more than 64 joints need more than one wavefront per solve")
        # COLS_PER_LANE (tuning knob, not in the reference API): 2 -> lane j owns d/dq_j and d/dqd_j (lane group = next pow2 >= n);
        # 1 -> first half of the group owns the d/dq columns, second half the d/dqd columns (lane group = next pow2 >= 2n)
        COLS_PER_LANE = self.tuning["cols_per_lane"]
        if COLS_PER_LANE == 1 and 2 * n > 64:
            COLS_PER_LANE = 2
        self.cols_per_lane = COLS_PER_LANE
        need = n if COLS_PER_LANE == 2 else 2 * n
        lanes = max(8 if COLS_PER_LANE == 2 else 16, int(self.tuning["min_lanes"]))
        if lanes not in (8, 16, 32, 64):
            raise ValueError("tuning['min_lanes'] must be 8, 16, 32 or 64")
        while lanes < need:
            lanes *= 2
        self.lanes_per_solve = lanes          # lane j of a group <-> joint j; 6 lanes also carry the articulated inertia columns
        # 8-lane groups: two solves share one 16-lane DPP row.  Interleaved (thread t of a row: solve t & 1, joint t >> 1) a DPP shift by 2k moves k joints inside ONE
        # solve and leaves the row exactly where the solve ends; blocks are then made of whole rows (a multiple of 16 threads, GRID_MIN_THREADS)
        self.lane_interleave = bool(lanes == 8 and self.tuning["lane_interleave"])
        # derivative-walk form (see algorithms/_inverse_dynamics_gradient.py): VGPR-resident backward sweep for shallow trees, LDS-assisted
        # forward accumulation for deep ones; and whether the gradient walk of forward_dynamics_gradient re-uses v, I v, fx(v) I v of the RNEA(qdd=0) pass
        depth_max = max(self.model.depth) + 1
        nslots = len(self.gen_gradient_slots())
        mode = self.tuning["gradient_walk"]
        self.register_walk = (mode == "registers") or (mode in ("auto", "tipframe") and depth_max * 6 * (1 + nslots) <= 200)
        # forward_dynamics_gradient of serial revolute chains is assembled in the tip link's frame (algorithms/_tip_frame_gradient.py);
        # every other robot, and every other kernel, uses the column walk selected above
        m_ = self.model
        # ... and so do forests of equal chains hanging off the fixed base (a quadruped's legs): one tip frame per chain
        segs = [(r_, len(m_.subtree[r_])) for r_ in m_.roots]
        chains = all(len(m_.children[j]) <= 1 for j in range(n))
        equal = chains and len(set(L_ for _, L_ in segs)) == 1 and all([m_.S_index[st + i] for i in range(L_)] == [m_.S_index[i] for i in range(L_)] for st, L_ in segs)
        tip_ok = bool(chains and equal and all(s_ < 3 for s_ in m_.S_index) and COLS_PER_LANE == 2 and lanes <= 16)
        self.tip_L, self.tip_nseg = (segs[0][1], len(segs)) if tip_ok else (n, 1)
        bo = self.tuning["base_origin"]
        if bo == "auto":
            self.tip_jB = self.tip_L // 2 if (tip_ok and self.tip_L >= 5) else None
        elif bo in ("off", None, False):
            self.tip_jB = None
        else:
            self.tip_jB = int(bo)
            if not (tip_ok and 0 <= self.tip_jB < self.tip_L - 1):
                raise ValueError("tuning['base_origin'] must be auto, off or a joint position 0 .. L-2 of a tip-frame robot")
        self.tip_rec = 20 if self.tip_jB is not None else 16  # values per joint in the hand-off records of the tip-frame inner
        if mode == "tipframe" and not tip_ok:
            raise NotImplementedError("gradient_walk=tipframe needs a serial chain of revolute joints (or a forest of equal such chains) with at most 16 joints")
        self.tip_frame = tip_ok and mode in ("auto", "tipframe") and not DEBUG_MODE  # (DEBUG_MODE prints M^-1, which this path never forms)
        # branched robots with revolute joints: forward_dynamics_gradient with every branch in the frame of its own tip link
        # (algorithms/_branch_frame_gradient.py); the other kernels of such robots stay on the column walk
        # Long chains too: the tip-frame inner replicates the dense factorisation in registers (12-DoF chain: 552 B of scratch, 75 us per 16 384
        # solves) where the branch-frame inner keeps its factors in LDS (46 us); measured the other way round for 8 joints (18 vs 13 us).
        want_branch = mode == "branch" or (mode == "auto" and (not self.tip_frame or self.tip_L >= 10))
        self.branch_plan = self.gen_branch_frame_plan() if (COLS_PER_LANE == 2 and not DEBUG_MODE and want_branch) else None
        if mode == "auto" and self.branch_plan is not None and self.branch_plan["factor_work"] > 1000:
            # every lane factors its whole component: beyond ~1000 multiply-adds (a dense 18-joint chain) that costs more than it saves
            # (20-joint chain, 4 096 solves: 99 us against 90 us on the column walk; 27-joint random tree with 651: 141 against 154 us)
            self.branch_plan = None
        if mode == "branch" and self.branch_plan is None:
            raise NotImplementedError("gradient_walk=branch needs revolute joints and branches that fit the 16-lane rows of the lane group")
        self.branch_frame = self.branch_plan is not None
        if mode == "branch":
            self.tip_frame = False
        if self.branch_frame and not self.tip_frame:
            self.tip_L = self.branch_plan["maxLb"]  # (length of the DPP scans)
        self.branch_tab_offset = 54 * n + (len(self.gen_tip_frame_link_constants()) if self.tip_frame else 0)
        self.branch_components = self.branch_frame and not self.tip_frame  # the stand-alone kernels of branched robots run the same path
        self.fd_stream_out = False  # (decided in gen_lds_layout: it needs the slice size)
        self.reuse_rnea = self.register_walk and n <= 9 and self.tuning["fuse_fd"] and self.tuning["reuse_rnea"]  # measured: 16.6 us vs 15.0 us per launch with re-use (extra LDS traffic on the critical path), so off by default
        # tuning knob: minimum waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument); 0 = compiler's choice
        self.min_waves_per_eu = int(self.tuning["min_waves"])
        self.minv_ld = (n + 3) // 4 * 4  # leading dimension of the dense M^-1 in LDS
        self.suggested_threads = 256
        self.max_threads = 512                # __launch_bounds__: keeps 256 VGPRs available per lane

    # ------------------------------------------------------------------ file prologue
    def gen_add_includes(self, use_thread_group=False):
        self.gen_add_code_line("")
        self.gen_add_code_lines(["#include <assert.h>", "#include <stdio.h>", "#include <stdlib.h>", "#include <math.h>", "#include <time.h>",
                                 "#include <hip/hip_runtime.h>"])
        self.gen_add_code_lines(["// single kernel timing helper code",
                                 "#define time_delta_us_timespec(start,end) (1e6*static_cast<double>(end.tv_sec - start.tv_sec)+1e-3*static_cast<double>(end.tv_nsec - start.tv_nsec))"])
        self.gen_add_code_line("")
        self.gen_add_code_line("#define XIMAT_SIZE 36")

    def gen_add_gpu_err(self):
        self.gen_add_func_doc("Check for runtime errors using the HIP API",
                              ["default: print and exit like the reference's gpuAssert; a host program (or the C-ABI shim, which must not take its",
                               "caller's process down) may define GRID_ON_GPU_ERROR(code, file, line) before including this header to do something else"], [], None)
        self.gen_add_code_line("#ifndef GRID_ON_GPU_ERROR")
        self.gen_add_code_line("#define GRID_ON_GPU_ERROR(code, file, line) { fprintf(stderr,\"GPUassert: %s %s %d\\n\", hipGetErrorString(code), file, line); hipDeviceReset(); exit(code); }")
        self.gen_add_code_line("#endif")
        self.gen_add_code_line("__host__")
        self.gen_add_code_line("inline void gpuAssert(hipError_t code, const char *file, const int line, bool abort=true){", True)
        self.gen_add_code_line("if (code != hipSuccess){", True)
        self.gen_add_code_line("if (abort) { GRID_ON_GPU_ERROR(code, file, line); }")
        self.gen_add_code_line("else { fprintf(stderr,\"GPUassert: %s %s %d\\n\", hipGetErrorString(code), file, line); }")
        self.gen_add_end_control_flow()
        self.gen_add_end_control_flow()
        self.gen_add_code_line("#define gpuErrchk(err) {gpuAssert(err, __FILE__, __LINE__);}")
        self.gen_add_code_line("")
        if self.gen_print_mat:
            for const in ("", "const "):
                self.gen_add_code_line("template <typename T, int M, int N>")
                self.gen_add_code_line("__host__ __device__")
                self.gen_add_code_line("void printMat(" + const + "T *A, int lda){", True)
                self.gen_add_code_line("for(int i=0; i<M; i++){", True)
                self.gen_add_code_line("for(int j=0; j<N; j++){printf(\"%.4f \",A[i + lda*j]);}")
                self.gen_add_code_line("printf(\"\\n\");")
                self.gen_add_end_control_flow()
                self.gen_add_end_function()

    def gen_add_constants_helpers(self, include_base_inertia=False, include_homogenous_transforms=False):
        n = self.model.n
        lds = self.lds = self.gen_lds_layout()
        G = self.lanes_per_solve
        max_groups = self.suggested_threads // G
        count = max_groups * (lds["TOTAL"] + lds["OUT_PER_SOLVE"])
        _sp = self.gen_topology_sparsity_helpers_python(); dva_cols, df_cols = _sp[0], _sp[3]
        self.gen_add_code_lines(["const int NUM_JOINTS = " + str(n) + ";",
                                 "const int NUM_VEL = " + str(n) + ";",
                                 "const int NUM_EES = " + str(sum(1 for c in self.model.children if not c)) + ";",
                                 "// lane-group decomposition: GRID_LANES_PER_SOLVE consecutive lanes of one wavefront own one solve",
                                 "const int GRID_LANES_PER_SOLVE = " + str(G) + ";",
                                 "const int GRID_SOLVES_PER_WAVE = " + str(64 // G) + ";",
                                 "const int GRID_LANE_INTERLEAVE = %d; // 1: the two solves of a 16-lane row interleave (thread t of the row: solve t & 1, joint t >> 1); blocks are whole rows" % (1 if self.lane_interleave else 0),
                                 "const int GRID_MIN_THREADS = %d; // smallest block (and block-size granule): one %s" % ((16, "16-lane row = two solves") if self.lane_interleave else (G, "lane group")),
                                 "const int GRID_MAX_THREADS = " + str(self.max_threads) + "; // __launch_bounds__ of every kernel",
                                 "#define GRID_LAUNCH_BOUNDS __launch_bounds__(" + str(self.max_threads) + (", " + str(self.min_waves_per_eu) if self.min_waves_per_eu else "") + ")",
                                 "const int SUGGESTED_THREADS = " + str(self.suggested_threads) + ";",
                                 "const int GRID_MAX_SOLVES_PER_BLOCK = SUGGESTED_THREADS/GRID_LANES_PER_SOLVE; // what the *_DYNAMIC_SHARED_MEM_COUNT constants cover",
                                 "// per-solve LDS slice (elements of T) and the offsets of its parts",
                                 "const int GRID_LDS_PER_SOLVE = " + str(lds["TOTAL"]) + ";",
                                 "const int GRID_MINV_LD = " + str(self.minv_ld) + "; // leading dimension of the dense symmetric M^-1 kept in LDS (s_Minv[row*GRID_MINV_LD + col])",
                                 "const int GRID_OUT_PER_SOLVE = " + str(lds["OUT_PER_SOLVE"]) + "; // output staging per lane group, placed behind the block's slices",
                                 "// a block of t threads needs (t/GRID_LANES_PER_SOLVE)*(GRID_LDS_PER_SOLVE+GRID_OUT_PER_SOLVE)*sizeof(T) bytes of dynamic LDS;",
                                 "// the *_DYNAMIC_SHARED_MEM_COUNT constants below are that amount for SUGGESTED_THREADS"])
        for k in ("IN", "X", "U", "T", "MINV", "QDD", "F", "J", "SP"):
            self.gen_add_code_line("const int GRID_OFF_" + k + " = " + str(lds[k]) + ";")
        fd_threads = self.suggested_threads
        if lds["FD_TOTAL"] < lds["TOTAL"]:  # LDS-capacity-bound kernels: one wave per block packs the CU's 160 KB best
            fd_threads = 64
        self.gen_add_code_lines(["// the forward_dynamics_gradient kernel with (q, qd, u) input carves slices of FD_DU_LDS_PER_SOLVE elements (a prefix-compatible subset of the",
                                 "// general slice: same GRID_OFF_IN / GRID_OFF_X) and is best launched with FD_DU_SUGGESTED_THREADS threads per block",
                                 "const int FD_DU_LDS_PER_SOLVE = " + str(lds["FD_TOTAL"]) + ";",
                                 "const int FD_DU_OFF_SP = " + str(lds["FD_SP"]) + "; const int FD_DU_OFF_QDD = " + str(lds["FD_QDD"]) + "; const int FD_DU_OFF_YPARK = " + str(lds["FD_YPARK"]) + ";",
                                 "const int FD_DU_SUGGESTED_THREADS = " + str(fd_threads) + ";",
                                 "const int FD_DU_OUT_PER_SOLVE = " + str(lds["FD_OUT_PER_SOLVE"]) + "; // staging of that kernel" + (": ONE half of the record at a time (d/dqd leaves after the factorisation, then d/dq: more resident waves per CU)" if self.fd_stream_out else "") + "; the _single_timing twin stages GRID_OUT_PER_SOLVE"])
        self.gen_add_code_line("// per-kernel slices of the stand-alone kernels (ID = inverse_dynamics, ID_DU = its gradient, MINV = direct_minv, FD = forward_dynamics, ABA = aba): elements per")
        self.gen_add_code_line("// solve, staging elements per solve, offsets of the path-axis scratch and of M^-1 inside the slice, best block size.  Robots on the branch-frame path")
        self.gen_add_code_line("// carve compact slices (IN | X | path axes [| M^-1]); everyone else uses the general slice")
        for k in ("ID", "ID_DU", "MINV", "FD", "ABA"):
            K = lds["KERNELS"][k]
            self.gen_add_code_line("const int %s_LDS_PER_SOLVE = %d; const int %s_OUT_PER_SOLVE = %d; const int %s_OFF_SP = %d; const int %s_OFF_MINV = %d; const int %s_SUGGESTED_THREADS = %s;"
                                   % (k, K["LDS"], k, K["OUT"], k, K["SP"], k, K["MINV"], k, "64" if K["compact"] else "SUGGESTED_THREADS"))
        for k in ("ID", "MINV", "FD", "ABA", "ID_DU", "FD_DU"):
            self.gen_add_code_line("const int " + k + "_DYNAMIC_SHARED_MEM_COUNT = " + str(count) + ";")
        self.gen_add_code_lines(["const int ID_DU_MAX_SHARED_MEM_COUNT = " + str(count) + ";",
                                 "const int FD_DU_MAX_SHARED_MEM_COUNT = " + str(count) + ";",
                                 "// (reference bookkeeping) derivative columns that are structurally non-zero: dv/da " + str(dva_cols) + ", df " + str(df_cols)])
        self.gen_add_code_line("#define GRID_HAS_IDSVA_SO %d // second-order derivatives (idsva_so, fdsva_so): emitted for fixed-base robots with revolute joints" % (1 if self.gen_idsva_so_available() else 0))
        if self.gen_idsva_so_available():
            sl_, scr_, stg_, thr_ = self.gen_idsva_so_lds_layout()
            if scr_ > lds["MINV"] - lds["X"]:
                raise ValueError("idsva_so scratch (%d) does not fit between X(q) and M^-1 of the general slice (fdsva_so_device runs it there)" % scr_)
            self.gen_add_code_line("#define GRID_SO_COMPACT %d // 1: the kernels stage the idsva_so record of a solve in compact form (every value once) and expand it through grid_so_expand" % (1 if self.gen_idsva_so_packed() else 0))
            self.gen_add_code_line("#define GRID_SO_SPLIT %d // 1: the host wrappers (and the C ABI) run fdsva_so as two kernels: fdsva_so_prepare_kernel + fdsva_so_contract_kernel" % (1 if self.gen_fdsva_so_split() is not None else 0))
            self.gen_add_code_lines(["#define GRID_SO_DIRECT %d // 1: the 4 n^3 record of one solve does not fit LDS - idsva_so writes it entry by entry to global memory, fdsva_so_kernel takes a d_idsva_so workspace" % (1 if self.gen_idsva_so_direct() else 0),
                                     "// init_gridData sizes the second-order buffers (d_idsva_so, d_df2 and their pinned host twins: 4 n^3 values per solve) for at most this many solves",
                                     "// (1 GiB per buffer); the second-order host wrappers reject longer batches with hipErrorInvalidValue",
                                     "template <typename T> constexpr int grid_so_max_timesteps() { return static_cast<int>((static_cast<size_t>(1) << 30)/(sizeof(T)*4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS)); }"])
            self.gen_add_code_lines(["const int IDSVA_SO_SUGGESTED_THREADS = %d; // idsva_so stages the 4 n^3 record of every solve in LDS: fewer solves per block" % thr_,
                                     "const int IDSVA_SO_LDS_PER_SOLVE = %d; // compact slice of the idsva_so kernels: q | qd | qdd | scratch" % sl_,
                                     "const int IDSVA_SO_SCRATCH_PER_SOLVE = %d; // X(q) / per-joint records + a zero qdd vector" % scr_,
                                     "const int IDSVA_SO_STAGE_PER_SOLVE = %d;" % stg_,
                                     "const int IDSVA_SO_MAX_SOLVES_PER_BLOCK = IDSVA_SO_SUGGESTED_THREADS/GRID_LANES_PER_SOLVE; // lane groups of larger blocks retire",
                                     "const int IDSVA_SO_DYNAMIC_SHARED_MEM_COUNT = IDSVA_SO_MAX_SOLVES_PER_BLOCK*(IDSVA_SO_LDS_PER_SOLVE + IDSVA_SO_STAGE_PER_SOLVE);"])
            fsl_, st_ = self.gen_fdsva_so_lds_per_solve()
            # block size of fdsva_so: the number of lane groups per block that lets the most solves be resident on a CU (its LDS holds the workspace,
            # df/du and the 4 n^3 idsva_so tensors of every solve); ties go to the smaller block = more waves (measured on the 7-DoF arm, 65 536
            # solves: 64 threads 707 us, 48: 652, 32: 606, 24: 596)
            per = (fsl_ + st_) * 4
            best_g, best_res = 1, 0
            for g_ in range(max(1, -(-24 // G)), max(1, 64 // G) + 1):  # (at least 24 lanes of a wave in use)
                res = min(155 * 1024 // (g_ * per), 8) * g_  # (blocks of at most one wave; the kernel holds > 168 VGPRs: at most 8 waves per CU)
                if res > best_res:  # (ties go to the smaller block: more waves for the same number of resident solves)
                    best_g, best_res = g_, res
            fd_so_threads = best_g * G
            self.gen_add_code_lines(["const int FDSVA_SO_SUGGESTED_THREADS = %d; // fdsva_so keeps the 4 n^3 idsva_so tensors of every solve in LDS: fewer solves per block" % fd_so_threads,
                                     "const int FDSVA_SO_LDS_PER_SOLVE = " + str(fsl_) + "; // slice of the fdsva_so kernels" + (": per-joint records | df/du | M^-1 (the gradient works inside the staging record)" if fsl_ != lds["TOTAL"] else " (the general slice)"),
                                     "const int FDSVA_SO_STAGE_PER_SOLVE = " + str(st_) + "; // " + ("the idsva_so tensors" if fsl_ != lds["TOTAL"] else "df/du (2 n^2, padded) + the idsva_so tensors") + ", behind the block's slices",
                                     "const int FDSVA_SO_MAX_SOLVES_PER_BLOCK = FDSVA_SO_SUGGESTED_THREADS/GRID_LANES_PER_SOLVE;",
                                     "const int FDSVA_SO_DYNAMIC_SHARED_MEM_COUNT = FDSVA_SO_MAX_SOLVES_PER_BLOCK*(FDSVA_SO_LDS_PER_SOLVE + FDSVA_SO_STAGE_PER_SOLVE);"])
        self.gen_add_code_lines(["// dynamic LDS (bytes) a launch with `threads` threads per block needs: one slice + one staging record per lane group of the block.",
                                 "// The host wrappers size their launches with it (the *_DYNAMIC_SHARED_MEM_COUNT constants are this amount for SUGGESTED_THREADS in",
                                 "// elements of T; a block may not exceed the CU's 160 KB: large robots in double precision need fewer threads per block)",
                                 "template <typename T>",
                                 "__host__ inline size_t grid_lds_bytes(const dim3 threads, const int lds_per_solve = GRID_LDS_PER_SOLVE, const int out_per_solve = GRID_OUT_PER_SOLVE,",
                                 "                                      const int max_groups = GRID_MAX_SOLVES_PER_BLOCK) {",
                                 "    int gpb = static_cast<int>(threads.x*threads.y)/GRID_LANES_PER_SOLVE;",
                                 "    if (gpb > max_groups) {gpb = max_groups;} // (the kernels retire lane groups beyond their cap)",
                                 "    if (gpb < 1) {gpb = 1;}",
                                 "    return static_cast<size_t>(gpb)*(lds_per_solve + out_per_solve)*sizeof(T);",
                                 "}"])
        self.gen_add_code_line("// Define custom structs" + (" (same layout as the enclosing namespace's - own types, so that argument-dependent lookup stays inside this namespace; init_* and close_grid are the enclosing namespace's)" if self.nested else ""))
        self.gen_add_code_lines(["template <typename T>", "struct robotModel {", "    T *d_XImats;", "    int *d_topology_helpers;", "};"])
        self.gen_add_code_lines(["template <typename T>", "struct gridData {",
                                 "    // GPU INPUTS", "    T *d_q_qd_u;", "    T *d_q_qd;", "    T *d_q;",
                                 "    // CPU INPUTS", "    T *h_q_qd_u;", "    T *h_q_qd;", "    T *h_q;",
                                 "    // GPU OUTPUTS", "    T *d_c;", "    T *d_Minv;", "    T *d_qdd;", "    T *d_M;", "    T *d_dc_du;", "    T *d_df_du;",
                                 "    T *d_eePos;", "    T *d_deePos;", "    T *d_d2eePos;", "    T *d_idsva_so;", "    T *d_df2;",
                                 "    // CPU OUTPUTS", "    T *h_c;", "    T *h_Minv;", "    T *h_qdd;", "    T *h_M;", "    T *h_dc_du;", "    T *h_df_du;",
                                 "    T *h_eePos;", "    T *h_deePos;", "    T *h_d2eePos;", "    T *h_idsva_so;", "    T *h_df2;",
                                 "};"])

    def gen_init_gridData(self):
        dev = [("d_q_qd_u", "3*NUM_JOINTS"), ("d_q_qd", "2*NUM_JOINTS"), ("d_q", "NUM_JOINTS"), ("d_c", "NUM_JOINTS"),
               ("d_Minv", "NUM_JOINTS*NUM_JOINTS"), ("d_qdd", "NUM_JOINTS"), ("d_dc_du", "NUM_JOINTS*2*NUM_JOINTS"), ("d_df_du", "NUM_JOINTS*2*NUM_JOINTS")]
        host = [("h_q_qd_u", "3*NUM_JOINTS"), ("h_q_qd", "2*NUM_JOINTS"), ("h_q", "NUM_JOINTS"), ("h_c", "NUM_JOINTS"),
                ("h_Minv", "NUM_JOINTS*NUM_JOINTS"), ("h_qdd", "NUM_JOINTS"), ("h_dc_du", "NUM_JOINTS*2*NUM_JOINTS"), ("h_df_du", "NUM_JOINTS*2*NUM_JOINTS")]
        unused = ["d_M", "d_eePos", "d_deePos", "d_d2eePos", "d_idsva_so", "d_df2", "h_M", "h_eePos", "h_deePos", "h_d2eePos", "h_idsva_so", "h_df2"]
        if self.gen_idsva_so_available():
            dev += [("d_idsva_so", "4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS"), ("d_df2", "4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS")]
            host += [("h_idsva_so", "4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS"), ("h_df2", "4*NUM_JOINTS*NUM_JOINTS*NUM_JOINTS")]
            unused = [u for u in unused if u not in ("d_idsva_so", "h_idsva_so", "d_df2", "h_df2")]
        code = ["gridData<T> *hd_data = (gridData<T> *)malloc(sizeof(gridData<T>));",
                "// device buffers of the dynamics algorithms"]
        cnt = lambda nm: "SO_TIMESTEPS" if nm[2:] in ("idsva_so", "df2") else "NUM_TIMESTEPS"
        if self.gen_idsva_so_available():
            code += ["const int SO_TIMESTEPS = NUM_TIMESTEPS < grid_so_max_timesteps<T>() ? NUM_TIMESTEPS : grid_so_max_timesteps<T>(); // (second-order records: capped, see grid_so_max_timesteps)"]
        code += ["gpuErrchk(hipMalloc((void**)&hd_data->" + nm + ", static_cast<size_t>(" + cnt(nm) + ")*" + sz + "*sizeof(T)));" for nm, sz in dev]
        code += ["// pinned host buffers (so the host wrappers' hipMemcpyAsync really is asynchronous)"]
        code += ["gpuErrchk(hipHostMalloc((void**)&hd_data->" + nm + ", static_cast<size_t>(" + cnt(nm) + ")*" + sz + "*sizeof(T), hipHostMallocDefault));" for nm, sz in host]
        code += ["// buffers of algorithms that this generator does not emit (kinematics, CRBA, second order) stay null"]
        code += ["hd_data->" + nm + " = nullptr;" for nm in unused]
        code += ["return hd_data;"]
        self.gen_add_func_doc("Allocated device and host memory for all computations", [], [], "A pointer to the gridData struct of pointers")
        self.gen_add_code_line("template <typename T, int NUM_TIMESTEPS>")
        self.gen_add_code_line("__host__")
        self.gen_add_code_line("gridData<T> *init_gridData(){", True)
        self.gen_add_code_lines(code)
        self.gen_add_end_function()
        self.gen_add_func_doc("Allocated device and host memory for all computations", [], ["Max number of timesteps in the trajectory"],
                              "A pointer to the gridData struct of pointers")
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__")
        self.gen_add_code_line("gridData<T> *init_gridData(int NUM_TIMESTEPS){", True)
        self.gen_add_code_lines(code)
        self.gen_add_end_function()

    def gen_init_close_grid(self):
        MAX_STREAMS = 3
        self.gen_add_func_doc("Initializes streams for host functions",
                              ["the per-solve LDS slices fit the default dynamic LDS limit, so no kernel attribute needs raising"], [],
                              "A pointer to the array of streams")
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__")
        self.gen_add_code_line("hipStream_t *init_grid(){", True)
        self.gen_add_code_lines(["hipStream_t *streams = (hipStream_t *)malloc(" + str(MAX_STREAMS) + "*sizeof(hipStream_t));",
                                 "int priority, minPriority, maxPriority;",
                                 "gpuErrchk(hipDeviceGetStreamPriorityRange(&minPriority, &maxPriority));",
                                 "for(int i=0; i<" + str(MAX_STREAMS) + "; i++){",
                                 "    int adjusted_max = maxPriority - i; priority = adjusted_max > minPriority ? adjusted_max : minPriority;",
                                 "    gpuErrchk(hipStreamCreateWithPriority(&(streams[i]),hipStreamNonBlocking,priority));",
                                 "}", "return streams;"])
        self.gen_add_end_function()
        self.gen_add_func_doc("Frees the memory used by grid", [],
                              ["streams allocated by init_grid", "robotModel allocated by init_robotModel", "data allocated by init_gridData"], None)
        self.gen_add_code_line("template <typename T>")
        self.gen_add_code_line("__host__")
        self.gen_add_code_line("void close_grid(hipStream_t *streams, robotModel<T> *d_robotModel, gridData<T> *hd_data){", True)
        self.gen_add_code_lines(["robotModel<T> h_robotModel; gpuErrchk(hipMemcpy(&h_robotModel,d_robotModel,sizeof(robotModel<T>),hipMemcpyDeviceToHost));",
                                 "gpuErrchk(hipFree(h_robotModel.d_XImats)); gpuErrchk(hipFree(h_robotModel.d_topology_helpers)); gpuErrchk(hipFree(d_robotModel));",
                                 "gpuErrchk(hipFree(hd_data->d_q_qd_u)); gpuErrchk(hipFree(hd_data->d_q_qd)); gpuErrchk(hipFree(hd_data->d_q));",
                                 "gpuErrchk(hipFree(hd_data->d_c)); gpuErrchk(hipFree(hd_data->d_Minv)); gpuErrchk(hipFree(hd_data->d_qdd));",
                                 "gpuErrchk(hipFree(hd_data->d_dc_du)); gpuErrchk(hipFree(hd_data->d_df_du));",
                                 "if (hd_data->d_idsva_so) {gpuErrchk(hipFree(hd_data->d_idsva_so));} if (hd_data->h_idsva_so) {gpuErrchk(hipHostFree(hd_data->h_idsva_so));}",
                                 "if (hd_data->d_df2) {gpuErrchk(hipFree(hd_data->d_df2));} if (hd_data->h_df2) {gpuErrchk(hipHostFree(hd_data->h_df2));}",
                                 "gpuErrchk(hipHostFree(hd_data->h_q_qd_u)); gpuErrchk(hipHostFree(hd_data->h_q_qd)); gpuErrchk(hipHostFree(hd_data->h_q));",
                                 "gpuErrchk(hipHostFree(hd_data->h_c)); gpuErrchk(hipHostFree(hd_data->h_Minv)); gpuErrchk(hipHostFree(hd_data->h_qdd));",
                                 "gpuErrchk(hipHostFree(hd_data->h_dc_du)); gpuErrchk(hipHostFree(hd_data->h_df_du));",
                                 "free(hd_data);",
                                 "for(int i=0; i<" + str(MAX_STREAMS) + "; i++){gpuErrchk(hipStreamDestroy(streams[i]));} free(streams);"])
        self.gen_add_end_function()

    def so_wide_lanes(self):
        """Lane-group width of the nested `wide` instance that carries the second-order kernels (tuning so_lanes), or None."""
        want = self.tuning["so_lanes"]
        if self.nested or want == "off" or self.lanes_per_solve > 16 or not self.gen_idsva_so_available() or int(self.tuning["debug_stop"]) not in (0, 30, 31, 32):
            return None
        if hasattr(self, "_so_wide_cache"):
            return self._so_wide_cache
        if want == "auto":
            # the wider instance must run the same second-order form (a 12-joint chain would fall from the tip-frame chain form to the base-frame tree
            # form at 32 lanes: 10x the fp32 error for no gain)
            probe = GRiDCodeGenerator(self.robot, COLS_PER_LANE=self._ctor["COLS_PER_LANE"], tuning=dict(self._ctor["tuning"], min_lanes=2 * self.lanes_per_solve, so_lanes="off"))
            self._so_wide_cache = 2 * self.lanes_per_solve if probe.gen_idsva_so_mode() == self.gen_idsva_so_mode() else None
            return self._so_wide_cache
        # measured (profiles/ab/r3e_*, r3g_*): 7-DoF arm 8 -> 16 lanes idsva_so 289 -> 195 us per 65 536 solves (32 lanes: 298); quadruped 16 -> 32 lanes
        # 414 -> 275 us per 16 384, 12-DoF tree 578 -> 431 (their 28 KB records leave ONE wave per CU at 4 solves per wave)
        if want not in (16, 32) or want <= self.lanes_per_solve:
            raise ValueError("tuning['so_lanes'] must be auto, off, or 16 / 32 and wider than the robot's lane groups")
        return want

    def _gen_library_body(self, use_thread_group=False, include_base_inertia=False, include_homogenous_transforms=False):
        """Everything inside the namespace.  The nested `wide` instance (so_wide_lanes) emits the same body minus structs, init_* and close_grid."""
        self.gen_add_constants_helpers(include_base_inertia, include_homogenous_transforms)
        self.gen_spatial_algebra_helpers()
        if self.tip_frame or self.branch_frame or self.gen_idsva_so_mode() is not None:  # (the second-order kernels use its cross products, 10-parameter inertias, Coriolis matrices)
            self.gen_tip_frame_library()
        if self.branch_frame:
            self.gen_branch_frame_library()
        self.gen_model_constant_table()
        if not self.nested:
            self.gen_init_topology_helpers()
            self.gen_init_XImats(include_base_inertia, include_homogenous_transforms)
            self.gen_init_robotModel()
            self.gen_init_gridData()
        self.gen_load_update_XImats_helpers(use_thread_group)
        if self.tip_frame:
            self.gen_tip_frame_components(use_thread_group)
        elif self.branch_frame:
            self.gen_branch_frame_components(use_thread_group)
        # the dynamics algorithms on (and next to) the forward-dynamics-gradient path
        self.gen_inverse_dynamics(use_thread_group)
        self.gen_direct_minv(use_thread_group)
        self.gen_forward_dynamics(use_thread_group)
        self.gen_aba(use_thread_group)
        self.gen_inverse_dynamics_gradient(use_thread_group)
        if self.tip_frame:
            self.gen_tip_frame_gradient(use_thread_group)
        if self.branch_frame:
            self.gen_forward_dynamics_gradient_inner_branch(use_thread_group)
            if self.fd_stream_out:
                self.gen_forward_dynamics_gradient_inner_branch_stream(use_thread_group)
        self.gen_forward_dynamics_gradient(use_thread_group)
        wide = self.so_wide_lanes()
        if wide:
            # second instance of the library for wider lane groups, in a nested namespace; the second-order host wrappers below launch ITS kernels
            sub = GRiDCodeGenerator(self.robot, FILE_NAMESPACE="wide", DEBUG_MODE=self._ctor["DEBUG_MODE"], NEED_PRINT_MAT=self._ctor["NEED_PRINT_MAT"], COLS_PER_LANE=self._ctor["COLS_PER_LANE"],
                                    tuning=dict(self._ctor["tuning"], min_lanes=wide, so_lanes="off"))
            sub.nested, sub.parent_namespace, sub.indent_level = True, self.file_namespace, self.indent_level
            sub.gen_add_func_doc("The same library for lane groups of %d lanes: what the second-order host wrappers (idsva_so_host, fdsva_so) and the C ABI launch" % wide,
                                 ["half as many solves stage their 4 n^3 records per wave and the item loops of idsva_so spread over twice the lanes",
                                  "init_* and close_grid are the enclosing namespace's (robotModel / gridData have the same layout here); launch its kernels with wide::GRID_LANES_PER_SOLVE lanes per solve"])
            sub.gen_add_code_line("namespace wide {", True)
            sub._gen_library_body(use_thread_group, include_base_inertia, include_homogenous_transforms)
            sub.gen_add_end_control_flow()
            self.code_str += sub.code_str
            self.gen_add_code_line("#define GRID_SO_WIDE 1 // the second-order host wrappers launch the kernels of namespace wide (%d lanes per solve)" % wide)
        self.gen_idsva_so(use_thread_group)
        self.gen_fdsva_so(use_thread_group)
        if not self.nested:
            self.gen_init_close_grid()

    # ------------------------------------------------------------------ everything
    def gen_all_code(self, use_thread_group=False, include_base_inertia=False, include_homogenous_transforms=False, fixed_target_name=""):
        if use_thread_group:
            raise NotImplementedError("cooperative-groups mode is unfinished in the reference (cgrps::thread_group tgrp = TBD) and has no HIP counterpart here")
        self.code_str = ""
        self.indent_level = 0
        n = self.model.n
        file_notes = ["Interface is:",
                      "    __host__   robotModel<T> *d_robotModel = init_robotModel<T>()",
                      "    __host__   hipStream_t *streams = init_grid<T>()",
                      "    __host__   gridData<T> *hd_data = init_gridData<T,NUM_TIMESTEPS>();  (or init_gridData<T>(NUM_TIMESTEPS))",
                      "    __host__   close_grid<T>(hipStream_t *streams, robotModel<T> *d_robotModel, gridData<T> *hd_data)",
                      "",
                      "    __device__ inverse_dynamics_inner<T>(T (&c)[NUM_JOINTS], const T *s_qd, [const T *s_qdd,] const T *s_X, const T gravity)",
                      "    __global__ inverse_dynamics_kernel<T>(T *d_c, const T *d_q_qd, const int stride_q_qd, [const T *d_qdd,] const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   inverse_dynamics<T,USE_QDD_FLAG=false,USE_COMPRESSED_MEM=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      "",
                      "    __device__ direct_minv_inner<T>(T *s_Minv, const T *s_X, T *s_U, T *s_T, const robotModel<T> *d_robotModel, const int lane)",
                      "    __global__ direct_minv_kernel<T>(T *d_Minv, const T *d_q, const int stride_q, const robotModel<T> *d_robotModel, const int NUM_TIMESTEPS)",
                      "    __host__   direct_minv<T,USE_COMPRESSED_MEM=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      "",
                      "    __device__ forward_dynamics_inner<T>(T *s_qdd, const T *s_qd, const T *s_u, const T *s_X, T *s_U, T *s_T, T *s_Minv, const robotModel<T> *d_robotModel, const T gravity, const int lane)",
                      "    __global__ forward_dynamics_kernel<T>(T *d_qdd, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   forward_dynamics<T>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      "",
                      "    __device__ inverse_dynamics_gradient_inner<T>(T (&dc_dq)[NUM_JOINTS], T (&dc_dqd)[NUM_JOINTS], const T *s_qd, const T *s_qdd, const T *s_X, const T gravity, const int lane)",
                      "    __global__ inverse_dynamics_gradient_kernel<T>(T *d_dc_du, const T *d_q_qd, const int stride_q_qd, [const T *d_qdd,] const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   inverse_dynamics_gradient<T,USE_QDD_FLAG=false,USE_COMPRESSED_MEM=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      "",
                      "    __device__ forward_dynamics_gradient_device<T>(T *s_df_du, const T *s_q, const T *s_qd, const T *s_u, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane)",
                      "    __device__ forward_dynamics_gradient_device<T>(T *s_df_du, const T *s_q, const T *s_qd, const T *s_qdd, const T *s_Minv, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane)",
                      "    __global__ forward_dynamics_gradient_kernel<T>(T *d_df_du, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __global__ forward_dynamics_gradient_kernel<T>(T *d_df_du, const T *d_q_qd, const int stride_q_qd, const T *d_qdd, const T *d_Minv, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   forward_dynamics_gradient<T,USE_QDD_MINV_FLAG=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      "",
                      "    __device__ aba_device<T>(T *s_qdd, const T *s_q, const T *s_qd, const T *s_tau, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane)",
                      "    __global__ aba_kernel<T>(T *d_qdd, const T *d_q_qd_tau, const int stride_q_qd, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   aba<T>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      ""] + ([
                      "    second order (launch with IDSVA_SO_/FDSVA_SO_SUGGESTED_THREADS threads; the host wrappers size the LDS from thread_dimms):",
                      "    __device__ idsva_so_device<T>(T *so, const T *s_q, const T *s_qd, [const T *s_qdd,] T *s_scratch, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active)",
                      "    __global__ idsva_so_kernel<T>(T *d_idsva_so, const T *d_q_qd_u, const int stride_q_qd_u, [const T *d_qdd,] const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   idsva_so_host<T,USE_QDD_FLAG=false>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      "    __device__ fdsva_so_device<T>(T *df2, T *s_df_du, T *s_idsva_so, const T *s_q, const T *s_qd, const T *s_u, T *s_work, const robotModel<T> *d_robotModel, const T gravity, const int lane, const bool active)",
                      "    __global__ fdsva_so_kernel<T>(T *d_df2, const T *d_q_qd_u, const int stride_q_qd_u, const robotModel<T> *d_robotModel, const T gravity, const int NUM_TIMESTEPS)",
                      "    __host__   fdsva_so<T>(gridData<T> *hd_data, const robotModel<T> *d_robotModel, const T gravity, const int num_timesteps, const dim3 block_dimms, const dim3 thread_dimms, hipStream_t *streams)",
                      ""] if self.gen_idsva_so_available() else []) + [
                      "Every host function also exists as NAME_single_timing and NAME_compute_only (no streams argument).",
                      "",
                      "Execution model (differs from the CUDA original by design): a lane group of GRID_LANES_PER_SOLVE = " + str(self.lanes_per_solve) + " consecutive lanes of one",
                      "wavefront owns a solve (lane j <-> joint j), a block of T threads therefore advances T/GRID_LANES_PER_SOLVE solves at a time and",
                      "grid-strides over the batch.  Launch with thread_dimms <= SUGGESTED_THREADS (" + str(self.suggested_threads) + ") threads and <FUNC_CODE>_DYNAMIC_SHARED_MEM_COUNT*sizeof(T)",
                      "bytes of dynamic LDS; a natural grid is ceil(num_timesteps*GRID_LANES_PER_SOLVE/threads) blocks.",
                      "", "Suggested Type T is float"]
        self.gen_add_func_doc("This instance of grid.cuh (HIP, gfx950) is optimized for the urdf: " + str(self.model.name), file_notes)
        self.gen_add_code_line("#pragma once")
        self.gen_add_includes(use_thread_group)
        self.gen_add_gpu_err()
        self.gen_add_code_line("// dynamic LDS of the block; every kernel carves one GRID_LDS_PER_SOLVE slice per lane group out of it")
        self.gen_add_code_line("extern __shared__ __attribute__((aligned(16))) unsigned char grid_smem_raw[];")
        self.gen_add_code_line("")
        self.gen_add_func_doc("All functions are kept in this namespace")
        self.gen_add_code_line("namespace " + self.file_namespace + " {", True)
        self._gen_library_body(use_thread_group, include_base_inertia, include_homogenous_transforms)
        self.gen_add_end_control_flow()
        with open(self.file_namespace + ".cuh", "w") as f:
            f.write(self.code_str)
        return self.code_str

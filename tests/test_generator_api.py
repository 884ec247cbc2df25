"""The Python face of the drop-in boundary: constructor / gen_all_code signature, the public
gen_* helpers, the emitted C++ surface, and the NumPy debug helpers test_* bound to the generator."""
import inspect
import os

import numpy as np
import pytest

from gridcodegenerator_amd import GRiDCodeGenerator, RobotModel


@pytest.fixture(scope="module")
def code(tmp_path_factory):
    d = tmp_path_factory.mktemp("gen")
    cwd = os.getcwd()
    os.chdir(d)
    try:
        g = GRiDCodeGenerator(RobotModel.from_fixture("hyq"), FILE_NAMESPACE="hyqgrid")
        text = g.gen_all_code()
        assert os.path.exists(d / "hyqgrid.cuh")  # written into the cwd like the reference (GRiDCodeGenerator.py:435-437)
        assert open(d / "hyqgrid.cuh").read() == text == g.code_str
    finally:
        os.chdir(cwd)
    return text


def test_signatures_match_the_reference():
    sig = inspect.signature(GRiDCodeGenerator.__init__)
    assert list(sig.parameters)[:6] == ["self", "robotObj", "DEBUG_MODE", "NEED_PRINT_MAT", "USE_DYNAMIC_SHARED_MEM", "FILE_NAMESPACE"]
    assert sig.parameters["FILE_NAMESPACE"].default == "grid"
    sig = inspect.signature(GRiDCodeGenerator.gen_all_code)
    assert list(sig.parameters) == ["self", "use_thread_group", "include_base_inertia", "include_homogenous_transforms", "fixed_target_name"]


def test_emitted_surface(code):
    assert "namespace hyqgrid {" in code
    for name in ("init_robotModel", "init_grid", "init_gridData", "close_grid", "struct robotModel", "struct gridData",
                 "load_update_XImats_helpers", "inverse_dynamics_inner", "inverse_dynamics_device", "inverse_dynamics_kernel",
                 "inverse_dynamics_kernel_single_timing", "direct_minv_inner", "direct_minv_device", "direct_minv_kernel",
                 "forward_dynamics_finish", "forward_dynamics_inner", "forward_dynamics_device", "forward_dynamics_kernel",
                 "aba_inner", "aba_device", "aba_kernel", "aba_kernel_single_timing", "void aba(", "void aba_single_timing(", "void aba_compute_only(", "ABA_DYNAMIC_SHARED_MEM_COUNT",
                 "inverse_dynamics_gradient_inner", "inverse_dynamics_gradient_device", "inverse_dynamics_gradient_kernel",
                 "forward_dynamics_gradient_device", "forward_dynamics_gradient_kernel", "forward_dynamics_gradient_kernel_single_timing",
                 "void forward_dynamics_gradient(", "void forward_dynamics_gradient_single_timing(", "void forward_dynamics_gradient_compute_only(",
                 "NUM_JOINTS = 12", "FD_DU_DYNAMIC_SHARED_MEM_COUNT", "FD_DU_MAX_SHARED_MEM_COUNT", "SUGGESTED_THREADS", "#define XIMAT_SIZE 36",
                 "gpuErrchk", "USE_QDD_MINV_FLAG", "USE_QDD_FLAG", "USE_COMPRESSED_MEM"):
        assert name in code, name
    assert "GRID_HAS_IDSVA_SO 1" in code and "idsva_so_kernel" in code and "fdsva_so_kernel" in code  # trees get the tree form of the second-order surface
    assert "cuda_runtime" not in code and "cudaMalloc" not in code and "cudaStream_t" not in code  # HIP only, no CUDA shims
    assert "__syncthreads" not in code  # wave-level hand-offs only


def test_public_gen_helpers_exist_and_compose():
    g = GRiDCodeGenerator(RobotModel.from_fixture("iiwa14"))
    for name in ("gen_add_code_line", "gen_add_code_lines", "gen_add_end_control_flow", "gen_add_end_function", "gen_add_func_doc",
                 "gen_add_serial_ops", "gen_add_parallel_loop", "gen_add_sync", "gen_var_in_list", "gen_var_not_in_list",
                 "gen_add_multi_threaded_select", "gen_kernel_load_inputs", "gen_kernel_save_result", "gen_mx_func_call_for_cpp",
                 "gen_topology_helpers_pointers_for_cpp", "gen_topology_sparsity_helpers_python",
                 "gen_forward_dynamics_gradient_inner_temp_mem_size", "gen_forward_dynamics_gradient_device", "gen_forward_dynamics_gradient_kernel",
                 "gen_forward_dynamics_gradient_host", "gen_forward_dynamics_gradient", "gen_inverse_dynamics", "gen_direct_minv", "gen_forward_dynamics",
                 "gen_inverse_dynamics_gradient"):
        assert callable(getattr(g, name)), name
    g.gen_add_func_doc("doc", ["note"], ["p"], "r")
    g.gen_add_code_line("void f(){", True)
    g.gen_add_multi_threaded_select("ind", "<", ["6", "12"], [("int", "a", ["0", "1", "2"]), ("int", "b", ["3", "4", "5"])])
    g.gen_add_multi_threaded_select("ind", "<", ["6"], [("int", "c", ["0", "1"])])
    g.gen_add_end_function()
    assert g.code_str.count("{") == g.code_str.count("}")
    assert g.gen_var_in_list("x", ["1", "2"]) == "((x == 1) || (x == 2))"
    vals = g.gen_topology_sparsity_helpers_python()
    assert vals[0] == 28 and vals[3] == 49
    assert g.gen_topology_helpers_pointers_for_cpp() == ("(jid-1)", "2")


@pytest.mark.parametrize("name", ["iiwa14", "hyq", "mixed5"])
def test_numpy_debug_helpers_match_reference_goldens(name, golden):
    g = GRiDCodeGenerator(RobotModel.from_fixture(name))
    gd = golden(name)
    for k in range(3):
        q, qd, u = gd["q"][k], gd["qd"][k], gd["u"][k]
        c, v, a, f = g.test_rnea(q, qd)
        assert np.allclose(c, gd["c"][k], rtol=1e-10, atol=1e-10) and np.allclose(f, gd["f"][k], rtol=1e-10, atol=1e-9)
        assert np.allclose(g.test_minv(q, True), gd["Minv"][k], rtol=1e-9, atol=1e-9)
        assert np.allclose(g.test_rnea_grad(q, qd, gd["qdd"][k]), gd["dc_du"][k], rtol=1e-9, atol=1e-8)
        assert np.allclose(g.test_fd_grad(q, qd, u), gd["df_du"][k], rtol=1e-8, atol=1e-7)


def test_formulation_chosen_per_robot():
    """Which forward-dynamics-gradient formulation every fixture gets (a robot silently falling back to the column walk would still pass
    the parity tests, only slower): tip frame for short revolute chains and forests of equal chains, branch frames for branched revolute
    robots (all kernels) and long chains (gradient only), column walk for robots with prismatic joints."""
    from gridcodegenerator_amd import GRiDCodeGenerator, RobotModel
    expect = {  # name: (tip_frame, branch_frame, branch_components)
        "iiwa14": (True, False, False), "arm6": (True, False, False), "chain8": (True, False, False), "hyq": (True, False, False),
        "chain12": (True, True, False), "atlas": (False, True, True), "tree12": (False, True, True), "mixed5": (False, False, False),
    }
    for name, want in expect.items():
        g = GRiDCodeGenerator(RobotModel.from_fixture(name))
        assert (g.tip_frame, g.branch_frame, g.branch_components) == want, name
    plan = GRiDCodeGenerator(RobotModel.from_fixture("tree12")).branch_plan
    assert plan["maxlevel"] == 2 and len(plan["shapes"]) == 2 and plan["nb"] == 6 and plan["place"]["G"][0] == "sp" and plan["place"]["U"][0] == "x"
    plan = GRiDCodeGenerator(RobotModel.from_fixture("atlas")).branch_plan
    # (150 entries of the tree-sparse M; every base-rooted component starts at a multiple of 4 values: 0, 108, 132 -> 156 with the padding)
    assert plan["maxlevel"] == 1 and plan["D"] == 10 and plan["nnz"] == 156 and sorted(j for j in plan["joint_of_lane"] if j >= 0) == list(range(30))
    assert sorted(plan["ubase"].values()) == [0, 108, 132]
    # which walk the branch-frame robots get (algorithms/_branch_frame_gradient.branch_owner_walk): long root paths -> owner walk
    assert plan["owner"] and plan["sp_size"] == 0
    assert GRiDCodeGenerator(RobotModel.from_fixture("chain12")).branch_plan["owner"] and not GRiDCodeGenerator(RobotModel.from_fixture("tree12")).branch_plan["owner"]
    # every branch sits inside one 16-lane row
    for b, J in enumerate(plan["branches"]):
        lanes = [plan["joint_of_lane"].index(j) for j in J]
        assert lanes == list(range(lanes[0], lanes[0] + len(J))) and lanes[0] // 16 == lanes[-1] // 16

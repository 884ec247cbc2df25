"""The C-ABI library builds for gfx950 and exports every symbol include/grid_capi.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from gridcodegenerator_amd.runtime import INCLUDE_DIR, build_library


def declared_symbols():
    text = open(os.path.join(INCLUDE_DIR, "grid_capi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grid_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_hot_path_entry_points():
    syms = declared_symbols()
    for s in ("grid_init", "grid_close", "grid_forward_dynamics_gradient_host", "grid_forward_dynamics_gradient_device",
              "grid_forward_dynamics_gradient_qdd_minv_device", "grid_last_error"):
        assert s in syms


@pytest.mark.parametrize("robot", ["iiwa14"])
def test_library_builds_and_exports_all_symbols(robot):
    so = build_library(robot)  # hipcc cross-compiles gfx950 without a GPU
    lib = ctypes.CDLL(so)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.grid_robot_name.restype = ctypes.c_char_p
    assert lib.grid_robot_name().decode() == robot
    assert lib.grid_num_joints() == 7
    assert lib.grid_lanes_per_solve() == 8
    assert lib.grid_lds_bytes_per_block() <= 80 * 1024  # two SUGGESTED_THREADS blocks fit the 160 KB LDS of a CU


def test_missing_library_fails_loudly(tmp_path):
    from gridcodegenerator_amd.runtime import GridError, GridLibrary

    with pytest.raises(GridError):
        GridLibrary(str(tmp_path / "libgrid_nope.so"))

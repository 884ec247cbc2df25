from ._inverse_dynamics import *
from ._direct_minv import *
from ._forward_dynamics import *
from ._aba import *
from ._idsva_so import gen_idsva_so_available, gen_idsva_so_mode, gen_idsva_so_direct, gen_idsva_so_compact, gen_idsva_so_compact_layout, gen_idsva_so_rec, gen_idsva_so_tree_tables, gen_idsva_so_items, gen_idsva_so_items_table, gen_idsva_so_lds_layout, gen_idsva_so_inner_temp_mem_size, gen_idsva_so_inner_function_call, gen_idsva_so_inner, \
    gen_idsva_so_device, gen_idsva_so_kernel, gen_idsva_so_host, gen_idsva_so
from ._fdsva_so import gen_fdsva_so_inner_temp_mem_size, gen_fdsva_so_stage_size, gen_fdsva_so_fused_layout, gen_fdsva_so_components, gen_fdsva_so_split, gen_fdsva_so_split_kernels, gen_fdsva_so_lds_per_solve, gen_fdsva_so_fused_device, gen_fdsva_so_inner, gen_fdsva_so_device, gen_fdsva_so_kernel, gen_fdsva_so_host, gen_fdsva_so
from ._inverse_dynamics_gradient import *
from ._forward_dynamics_gradient import *
from ._tip_frame_gradient import gen_tip_frame_link_constants, gen_tip_frame_joint_offset, gen_tip_frame_library, \
    gen_forward_dynamics_gradient_inner_tip, gen_forward_dynamics_gradient_inner_tip_function_call, gen_tip_frame_gradient, gen_tip_frame_fused_so, \
    gen_inverse_dynamics_inner_tip, gen_inverse_dynamics_gradient_inner_tip, gen_forward_dynamics_inner_tip, gen_direct_minv_inner_tip, gen_tip_frame_components
from ._branch_frame_gradient import gen_branch_frame_plan, gen_branch_frame_constants, gen_branch_frame_library, gen_branch_frame_components, \
    gen_forward_dynamics_gradient_inner_branch, gen_forward_dynamics_gradient_inner_branch_stream, gen_forward_dynamics_gradient_inner_branch_function_call

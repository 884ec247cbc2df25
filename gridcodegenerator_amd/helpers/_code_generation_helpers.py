"""Emission primitives for the HIP/CDNA4 backend.

Replaces the reference's helpers/_code_generation_helpers.py (parallel-loop / sync / select / load-save emitters,
reference lines 1-199) with wave64-native equivalents.  The execution model these primitives emit is NOT the
reference's "one block per solve, block-wide strided loops + __syncthreads": here a *lane group* of
GRID_LANES_PER_SOLVE (8/16/32/64) consecutive lanes of one wavefront owns a solve, lane j owns joint j
(its X update, its Minv column, its two gradient columns), groups never straddle a wave, and every
hand-off goes through LDS followed by a wave-level sync (no s_barrier).
"""


def gen_add_code_line(self, new_code_line, add_indent_after=False):
    cur = getattr(self, "_cur_joint", None)
    if cur is not None and "grid_x" in new_code_line:
        # inside a joint's code block the joint transform calls bind to that joint's sparsity-specialised functions
        for op in ("grid_xmul(", "grid_xtmul_peq(", "grid_xtmul("):
            new_code_line = new_code_line.replace(op, op[:-1] + "_" + str(cur) + "(")
    self.code_str += self.indent_level * "    " + new_code_line + "\n"
    if add_indent_after:
        self.indent_level += 1


def gen_add_code_lines(self, new_code_lines, add_indent_after=False):
    for line in new_code_lines:
        self.gen_add_code_line(line)
    if add_indent_after:
        self.indent_level += 1


def gen_add_end_control_flow(self):
    self.indent_level -= 1
    self.gen_add_code_line("}")


def gen_add_end_function(self):
    self.indent_level -= 1
    self.gen_add_code_line("}\n")


def gen_add_func_doc(self, func_desc, notes=[], params=[], return_val=None):
    self.gen_add_code_line("/**")
    self.gen_add_code_line(" * " + func_desc)
    self.gen_add_code_line(" *")
    if len(notes) > 0:
        self.gen_add_code_line(" * Notes:")
        for note in notes:
            self.gen_add_code_line(" *   " + note)
        self.gen_add_code_line(" *")
    for param in params:
        self.gen_add_code_line(" * @param " + param)
    if return_val is not None:
        self.gen_add_code_line(" * @return " + return_val)
    self.gen_add_code_line(" */")


def gen_add_serial_ops(self, use_thread_group=False):
    """One lane of the solve's lane group (reference: thread 0 of the block)."""
    self.gen_add_code_line("if(lane == 0){", True)


def gen_add_parallel_loop(self, var_name, max_val, use_thread_group=False, block_level=False):
    """block_level: grid-stride loop over *batches* of solves (one solve per lane group, gpb groups per block).
    otherwise: loop strided over the lanes of this solve's lane group."""
    if block_level:
        self.gen_add_code_line("for(int " + var_name + "0 = (blockIdx.x + blockIdx.y*gridDim.x)*gpb; " + var_name + "0 < " + max_val +
                               "; " + var_name + "0 += gridDim.x*gridDim.y*gpb){", True)
        self.gen_add_code_line("const int " + var_name + " = " + var_name + "0 + grp; const bool valid = " + var_name + " < " + max_val +
                               "; const int " + var_name + "c = valid ? " + var_name + " : " + max_val + " - 1;")
        self.gen_add_code_line("const int NUM_TIMESTEPS_OUT = " + max_val + ";")
        self.gen_add_code_line("const int lane = grid_loop_variant(lane_id); // keeps lane-dependent values from being hoisted out of the batch loop (and spilled)")
    else:
        self.gen_add_code_line("for(int " + var_name + " = lane; " + var_name + " < " + max_val + "; " + var_name +
                               " += GRID_LANES_PER_SOLVE){", True)


def gen_add_sync(self, use_thread_group=False):
    """Wave-level hand-off point: LDS is in-order per wave, so only the compiler needs fencing."""
    self.gen_add_code_line("grid_wave_sync();")


def gen_var_in_list(self, var_name, option_list):
    if len(option_list) == 1:
        return "(" + var_name + " == " + option_list[0] + ")"
    return "(" + " || ".join(["(" + var_name + " == " + o + ")" for o in option_list]) + ")"


def gen_var_not_in_list(self, var_name, option_list):
    if len(option_list) == 1:
        return "(" + var_name + " != " + option_list[0] + ")"
    return "(" + " && ".join(["(" + var_name + " != " + o + ")" for o in option_list]) + ")"


def gen_add_multi_threaded_select(self, loop_counter, comparator, counts, select_tuples, USE_NON_BRANCH_ALWAYS=False):
    """Selects values by the range `loop_counter` falls into (reference helpers/_code_generation_helpers.py:96-145, same arguments).
    select_tuples = [(type, name, [value per range])]; ranges are `loop_counter <comparator> counts[i]` tried in order, the last
    value is the default.  One variable (or USE_NON_BRANCH_ALWAYS) gives a branch-free sum of predicated terms, otherwise an
    if / else-if chain - on wave64 hardware both compile to v_cndmask selects."""
    decls = []
    for (dtype, dvar, _) in select_tuples:
        if dtype is None:
            decls.append(dvar)
        elif "|" in dtype:
            pre, post = dtype.split("|")
            decls.append(pre + dvar + ")" + post)
        else:
            decls.append(dtype + " " + dvar)
    k = len(counts)
    conds = []
    for i in range(k):
        c = "(" + loop_counter + " " + comparator + " " + counts[i] + ")"
        if comparator != "==" and i > 0:
            c = "(" + c + " && !(" + loop_counter + " " + comparator + " " + counts[i - 1] + "))"
        conds.append(c)
    if len(select_tuples) > 1 and not USE_NON_BRANCH_ALWAYS:
        self.gen_add_code_line("// branch to get pointer locations")
        self.gen_add_code_line("; ".join(decls) + ";")
        for i in range(k):
            head = ("if " if i == 0 else "else if ") + conds[i] if i < k - 1 or comparator == "==" else "else"
            if i == k - 1 and comparator != "==":
                head = "else"
            body = " ".join(dvar + " = " + vals[i] + ";" for (_, dvar, vals) in select_tuples)
            self.gen_add_code_line(head + " { " + body + " }")
    else:
        self.gen_add_code_line("// non-branching pointer selector")
        for t, (_, dvar, vals) in enumerate(select_tuples):
            terms = []
            for i in range(k):
                cond = conds[i]
                if i == k - 1 and comparator != "==":
                    cond = "!(" + loop_counter + " " + comparator + " " + counts[k - 2] + ")" if k > 1 else "1"
                terms.append(cond + " * " + vals[i])
            self.gen_add_code_line(decls[t] + " = " + " + ".join(terms) + ";")


def gen_lane_mask_test(self, ids, var_name="lane"):
    """Compile-time lane-set membership as a 64-bit mask test (used for per-joint-type dispatch)."""
    mask = 0
    for i in ids:
        mask |= 1 << i
    return "((0x%xull >> %s) & 1ull)" % (mask, var_name)


def gen_kernel_prologue(self, lds_per_solve_const, max_groups_const="GRID_MAX_SOLVES_PER_BLOCK"):
    """Lane-group decomposition shared by every kernel.  Threads beyond the last whole lane group (or beyond
    max_groups_const groups, which is what the kernel's *_DYNAMIC_SHARED_MEM_COUNT constant is sized for) retire, so a launch
    with more threads than the kernel's suggested block size stays inside the LDS the host wrappers allocate."""
    if getattr(self, "lane_interleave", False):
        head = ["const int tid = threadIdx.x + threadIdx.y*blockDim.x;",
                "// (GRID_LANE_INTERLEAVE) the two solves of a 16-lane row interleave: a DPP row shift by 2k moves k joints inside one solve and never reaches the other one",
                "const int lane_id = (tid & 15) >> 1; // lane j of a solve's lane group owns joint j",
                "const int grp = ((tid >> 4) << 1) | (tid & 1);",
                "int gpb = ((blockDim.x*blockDim.y) >> 4) << 1; if (gpb > " + max_groups_const + ") {gpb = " + max_groups_const + ";} // (whole rows only)"]
    else:
        head = ["const int tid = threadIdx.x + threadIdx.y*blockDim.x;",
                "const int lane_id = tid & (GRID_LANES_PER_SOLVE-1); // lane j of a solve's lane group owns joint j",
                "const int grp = tid / GRID_LANES_PER_SOLVE;",
                "int gpb = (blockDim.x*blockDim.y) / GRID_LANES_PER_SOLVE; if (gpb > " + max_groups_const + ") {gpb = " + max_groups_const + ";}"]
    self.gen_add_code_lines(head + [
        "if (grp >= gpb) {return;}",
        "T *s_mem = reinterpret_cast<T *>(grid_smem_raw) + grp*" + lds_per_solve_const + ";",
        "// output staging lives behind this block's slices so that the records of a wave's lane groups are contiguous",
        "T *s_out_all = reinterpret_cast<T *>(grid_smem_raw) + gpb*" + lds_per_solve_const + ";",
    ])


def gen_kernel_load_inputs(self, name, stride, amount, use_thread_group=False, name2=None, stride2=1, amount2=1,
                           name3=None, stride3=1, amount3=1, symmetrize3=None):
    """global -> this solve's LDS slice.  Consecutive lanes read consecutive floats and consecutive lane groups read
    consecutive solves, so a wave's loads cover one contiguous span of the AoS input (SURVEY.md 8(a) a1 layout)."""
    self.gen_add_code_line("// load this solve's inputs to LDS (coalesced across the lane groups of the wave)")
    G = self.lanes_per_solve
    for (nm, st, am) in ((name, stride, amount), (name2, stride2, amount2)):
        if nm is None:
            continue
        trips = (int(am) + G - 1) // G
        self.gen_add_code_line("const T *d_" + nm + "_k = &d_" + nm + "[static_cast<size_t>(kc)*" + str(st) + "];")
        self.gen_add_code_line("{ // all global loads are issued before the first LDS write (one memory latency instead of %d)" % trips)
        self.gen_add_code_line("  T r_in[%d];" % trips)
        self.gen_add_code_line("  #pragma unroll")
        self.gen_add_code_line("  for (int it = 0; it < %d; it++) { const int ind = lane + it*GRID_LANES_PER_SOLVE; r_in[it] = (ind < %s) ? d_%s_k[ind] : static_cast<T>(0); }" % (trips, str(am), nm))
        self.gen_add_code_line("  #pragma unroll")
        self.gen_add_code_line("  for (int it = 0; it < %d; it++) { const int ind = lane + it*GRID_LANES_PER_SOLVE; if (ind < %s) { s_%s[ind] = r_in[it]; } }" % (trips, str(am), nm))
        self.gen_add_code_line("}")
    if name3 is not None:
        n = symmetrize3
        self.gen_add_code_line("const T *d_" + name3 + "_k = &d_" + name3 + "[static_cast<size_t>(kc)*" + str(stride3) + "];")
        self.gen_add_parallel_loop("ind", str(amount3), use_thread_group)
        if n is None:
            self.gen_add_code_line("s_" + name3 + "[ind] = d_" + name3 + "_k[ind];")
        else:
            self.gen_add_code_line("// only the upper triangle of the caller's matrix is dereferenced (reference _forward_dynamics_gradient.py:54)")
            self.gen_add_code_line("const int row = ind % " + str(n) + "; const int col = ind / " + str(n) + ";")
            self.gen_add_code_line("s_" + name3 + "[row*" + str(self.minv_ld) + " + col] = d_" + name3 + "_k[(row <= col) ? (col*" + str(n) + " + row) : (row*" + str(n) + " + col)];")
        self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)


def gen_static_array_ind_2d(self, col, row, col_stride=6):
    """Flat index of (row, col) in a column-major block (reference helpers/_code_generation_helpers.py:57)."""
    return col_stride * col + row


def gen_static_array_ind_3d(self, ind, col, row, ind_stride=36, col_stride=6):
    """Flat index of (row, col) of the ind-th column-major block (reference helpers/_code_generation_helpers.py:60)."""
    return ind_stride * ind + col_stride * col + row


def gen_add_debug_print_code_line(self, print_code_string, use_thread_group=False):
    self.gen_add_debug_print_code_lines([print_code_string], use_thread_group)


def gen_add_debug_print_code_lines(self, print_code_string_arr, use_thread_group=False):
    """printf/printMat lines executed by lane 0 of the lane group that owns solve 0 (reference: thread 0 of block 0)."""
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("if (lane == 0 && k == 0) {", True)
    for line in print_code_string_arr:
        self.gen_add_code_line(line)
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)


def gen_kernel_load_inputs_single_timing(self, name, amount, use_thread_group=False, name2=None, amount2=1, name3=None, amount3=1):
    """Timing kernels read ONE solve (record 0): same staging as gen_kernel_load_inputs with kc == 0."""
    self.gen_kernel_load_inputs(name, 0, amount, use_thread_group, name2, 0, amount2, name3, 0, amount3)


def gen_kernel_save_result_single_timing(self, store_to_name, amount, use_thread_group=False, load_from_name=None):
    """Timing kernels run ONE solve on ONE lane group: plain lane-strided store of its record."""
    if load_from_name is None:
        load_from_name = "s_" + store_to_name
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("// save down to global")
    self.gen_add_parallel_loop("ind", str(amount), use_thread_group)
    self.gen_add_code_line("d_" + store_to_name + "[ind] = " + load_from_name + "[ind];")
    self.gen_add_end_control_flow()


def gen_kernel_save_result(self, store_to_name, stride, amount, use_thread_group=False, load_from_name=None):
    """LDS staging -> global.  The lane groups of one wavefront own CONSECUTIVE solves and their staging records are
    contiguous in LDS (s_out_wave), so the wave's output is one contiguous span of global memory: every lane moves 16 bytes
    per trip (ds_read_b128 + global_store_dwordx4), with a dword tail.  `stride` must equal `amount` (contiguous records)."""
    assert str(stride) == str(amount)
    if load_from_name is None:
        load_from_name = "s_" + store_to_name
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("// save down to global: wave-cooperative, coalesced")
    self.gen_add_code_line("{", True)
    self.gen_add_code_lines(["const int gw0 = grp & ~(GRID_SOLVES_PER_WAVE-1); // first lane group of this wave",
                             "int nv = NUM_TIMESTEPS_OUT - (k - grp + gw0); { const int ng = gpb - gw0; nv = nv < ng ? nv : ng; nv = nv < GRID_SOLVES_PER_WAVE ? nv : GRID_SOLVES_PER_WAVE; nv = nv > 0 ? nv : 0; } // (a wave whose lane groups are all past the end of the batch writes nothing)",
                             "const int total = nv*" + str(amount) + "; // elements this wave writes",
                             "const T *src = " + load_from_name + " - (grp - gw0)*" + str(amount) + ";",
                             "T *dst = &d_" + store_to_name + "[static_cast<size_t>(k - grp + gw0)*" + str(amount) + "];",
                             "const int wl = tid & 63;",
                             "int wn = gpb*GRID_LANES_PER_SOLVE - (tid & ~63); wn = wn < 64 ? wn : 64; // lanes of this wave that take part (the last wave of a block may be partial)",
                             ("if (NUM_TIMESTEPS_OUT < 0) " if self.tuning["no_store"] else "") + "for (int e = 4*wl; e + 3 < total; e += 4*wn) { T tmp[4]; __builtin_memcpy(tmp, __builtin_assume_aligned(src + e, 4*sizeof(T) < 16 ? 4*sizeof(T) : 16), 4*sizeof(T)); grid_store4(dst + e, tmp); }",
                             "{ const int e = (total & ~3) + wl; if (wl < 3 && e < total) { dst[e] = src[e]; } }"])
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)


def gen_kernel_save_result_expanded(self, store_to_name, amount, stage, table_name, use_thread_group=False, load_from_name=None):
    """Compact LDS staging -> dense global record.  As gen_kernel_save_result (the lane groups of a wave own consecutive solves, their staging records
    are contiguous in LDS, the wave's output is one contiguous span of global memory written 16 bytes per lane and trip), but a staging record holds
    every value of a solve ONCE (`stage` values) and the dense record (`amount` values, a multiple of 4) is gathered from it through `table_name`
    (unsigned short slot per dense element, the same for every solve): a lane reads the four slots of its 16-byte piece once (8 bytes) and moves that
    piece of every solve of the wave."""
    assert int(amount) % 4 == 0
    if load_from_name is None:
        load_from_name = "s_" + store_to_name
    self.gen_add_sync(use_thread_group)
    self.gen_add_code_line("// save down to global: wave-cooperative, coalesced; the dense record is gathered from the compact staging through " + table_name)
    self.gen_add_code_line("{", True)
    self.gen_add_code_lines(["const int gw0 = grp & ~(GRID_SOLVES_PER_WAVE-1); // first lane group of this wave",
                             "int nv = NUM_TIMESTEPS_OUT - (k - grp + gw0); { const int ng = gpb - gw0; nv = nv < ng ? nv : ng; nv = nv < GRID_SOLVES_PER_WAVE ? nv : GRID_SOLVES_PER_WAVE; nv = nv > 0 ? nv : 0; } // (a wave whose lane groups are all past the end of the batch writes nothing)",
                             "const T *src = " + load_from_name + " - (grp - gw0)*" + str(stage) + ";",
                             "T *dst = &d_" + store_to_name + "[static_cast<size_t>(k - grp + gw0)*" + str(amount) + "];",
                             "const int wl = tid & 63;",
                             "int wn = gpb*GRID_LANES_PER_SOLVE - (tid & ~63); wn = wn < 64 ? wn : 64; // lanes of this wave that take part (the last wave of a block may be partial)",
                             ("if (NUM_TIMESTEPS_OUT < 0) " if self.tuning["no_store"] else "") + "for (int g = wl; g < " + str(int(amount) // 4) + "; g += wn) { // 16-byte piece g of every record", ])
    self.indent_level += 1
    self.gen_add_code_lines(["unsigned short ix[4]; __builtin_memcpy(ix, __builtin_assume_aligned(&" + table_name + "[4*g], 8), 8);",
                             "#pragma unroll",
                             "for (int s = 0; s < GRID_SOLVES_PER_WAVE; s++) {",
                             "    if (s < nv) { const T *rec = src + s*" + str(stage) + "; T tmp[4] = {rec[ix[0]], rec[ix[1]], rec[ix[2]], rec[ix[3]]}; grid_store4(dst + static_cast<size_t>(s)*" + str(amount) + " + 4*g, tmp); }",
                             "}"])
    self.gen_add_end_control_flow()
    self.gen_add_end_control_flow()
    self.gen_add_sync(use_thread_group)

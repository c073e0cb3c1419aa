"""The build's DPP hazard pass (hcr_genesis_lr_cl_amd/dpp_hazard_pass.py) on hand-written instruction streams: it may only ever remove a
wait state that the hardware rule "VALU writes VGPR -> DPP source read: 2 wait states" does not ask for."""
import glob
import json
import os

import pytest

from hcr_genesis_lr_cl_amd import dpp_hazard_pass as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QP = "quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf"


def kernel(body):
    return "\t.text\nk:\n" + "\n".join("\t" + l if not l.endswith(":") else l for l in body) + "\n.Lfunc_end0:\n"


def nops(text):
    return [l.strip().split(";")[0].strip() for l in text.split("\n") if l.strip().startswith("s_nop")]


def test_marked_nop_goes_when_the_source_is_old():
    src = kernel(["v_mul_f32_e32 v5, v1, v2", "v_add_f32_e32 v6, v1, v2", "v_add_f32_e32 v7, v1, v2",
                  "s_nop 1 ; lg-dpp-hazard", f"v_mul_f32_dpp v9, v5, v3 {QP}", "s_endpgm"])
    out, st = P.fix(src)
    assert nops(out) == [] and st["marked"] == 1 and st["kept_or_inserted"] == 0
    assert P.check(out) == []


def test_marked_nop_stays_when_the_source_was_just_written():
    src = kernel(["v_mul_f32_e32 v5, v1, v2", "s_nop 1 ; lg-dpp-hazard", f"v_mul_f32_dpp v9, v5, v3 {QP}", "s_endpgm"])
    out, st = P.fix(src)
    assert nops(out) == ["s_nop 1"] and P.check(out) == []


def test_one_instruction_in_between_leaves_one_wait_state_to_pad():
    src = kernel(["v_mul_f32_e32 v5, v1, v2", "v_add_f32_e32 v6, v1, v2", "s_nop 1 ; lg-dpp-hazard", f"v_fmac_f32_dpp v9, v5, v3 {QP}", "s_endpgm"])
    out, _ = P.fix(src)
    assert nops(out) == ["s_nop 0"] and P.check(out) == []


def test_only_the_dpp_routed_operand_counts():
    # v3 (plain operand) and v9 (accumulator) are fresh, the DPP source v5 is old: nothing to pad
    src = kernel(["v_mul_f32_e32 v5, v1, v2", "v_add_f32_e32 v6, v1, v2", "v_mul_f32_e32 v3, v1, v2", "v_mul_f32_e32 v9, v1, v2",
                  "s_nop 1 ; lg-dpp-hazard", f"v_fmac_f32_dpp v9, v5, v3 {QP}", "s_endpgm"])
    out, _ = P.fix(src)
    assert nops(out) == []


def test_a_write_inside_a_register_pair_is_seen():
    src = kernel(["v_pk_fma_f32 v[4:5], v[0:1], v[2:3], v[6:7]", "s_nop 1 ; lg-dpp-hazard", f"v_mov_b32_dpp v9, v5 {QP}", "s_endpgm"])
    out, _ = P.fix(src)
    assert nops(out) == ["s_nop 1"]


def test_a_branch_into_the_block_is_a_predecessor():
    # fall-through path: v5 written three instructions before; the branch path writes it right before jumping
    src = kernel(["v_mul_f32_e32 v5, v1, v2", "s_cbranch_scc1 .LBB0_2", "v_add_f32_e32 v6, v1, v2", "v_add_f32_e32 v7, v1, v2", "v_add_f32_e32 v8, v1, v2",
                  ".LBB0_2:", "s_nop 1 ; lg-dpp-hazard", f"v_mul_f32_dpp v9, v5, v3 {QP}", "s_endpgm"])
    out, _ = P.fix(src)
    assert nops(out) == ["s_nop 0"]          # write, branch (one wait state), DPP read: one more is missing on that path
    assert P.check(out) == []


def test_compiler_visible_dpp_behind_an_asm_write_is_padded():
    src = kernel([f"v_fmac_f32_dpp v9, v5, v3 {QP}", f"v_mov_b32_dpp v10, v9 {QP}", "s_endpgm"])
    assert len(P.check(src)) == 1
    out, st = P.fix(src)
    assert nops(out) == ["s_nop 1"] and st["kept_or_inserted"] == 1 and P.check(out) == []


def test_exec_written_by_a_valu_compare_needs_five():
    src = kernel(["v_cmpx_lt_f32_e32 v1, v2", "v_add_f32_e32 v6, v1, v2", f"v_mov_b32_dpp v10, v9 {QP}", "s_endpgm"])
    out, _ = P.fix(src)
    assert nops(out) == ["s_nop 3"] and P.check(out) == []    # five wait states, one instruction already in between


def test_unmarked_nops_are_left_alone_and_counted():
    src = kernel(["v_mul_f32_e32 v5, v1, v2", "s_nop 0", f"v_mul_f32_dpp v9, v5, v3 {QP}", "s_endpgm"])
    out, _ = P.fix(src)
    assert nops(out) == ["s_nop 0", "s_nop 0"]   # the compiler's one wait state stays, the missing one is added


FILL = ["v_add_f32_e32 v20, v1, v2", "v_add_f32_e32 v21, v1, v2", "v_add_f32_e32 v22, v1, v2", "v_add_f32_e32 v23, v1, v2", "v_add_f32_e32 v24, v1, v2",
        "v_add_f32_e32 v25, v1, v2"]


def test_compiler_padding_for_a_plain_operand_is_dropped():
    # the compiler's s_nop 1 is there because v60 (PLAIN operand of the DPP add) was just written; the DPP source v59 is old
    src = kernel(FILL + ["v_mul_f32_e32 v59, v1, v2", "v_add_f32_e32 v30, v1, v2", "v_add_f32_e32 v60, v1, v2", "s_nop 1",
                         "v_add_f32_dpp v61, v59, v60 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_endpgm"])
    out, st = P.fix(src)
    assert nops(out) == [] and st["compiler_nops_relaxed"] == 1 and st["compiler_wait_states_relaxed"] == 2
    assert nops(P.fix(src, relax=False)[0]) == ["s_nop 1"]


def test_compiler_padding_comes_back_when_the_dpp_source_needs_it():
    src = kernel(FILL + ["v_mul_f32_e32 v59, v1, v2", "s_nop 1", "v_add_f32_dpp v61, v59, v60 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_endpgm"])
    out, st = P.fix(src)
    assert nops(out) == ["s_nop 1"] and st["compiler_nops_relaxed"] == 1 and st["kept_or_inserted"] == 1


@pytest.mark.parametrize("blocker", ["v_cmp_lt_f32_e32 vcc, v1, v2", "v_rcp_f32_e32 v40, v1", "v_readlane_b32 s6, v18, 0", "global_store_dword v1, v2, s[0:1]",
                                     "v_pk_fma_f32 v[40:41], v[0:1], v[2:3], v[6:7]", "s_waitcnt vmcnt(0)", "v_add_co_u32_e32 v40, vcc, v1, v2"])
def test_compiler_padding_is_left_alone_next_to_anything_with_rules_of_its_own(blocker):
    src = kernel(FILL + [blocker, "v_add_f32_e32 v30, v1, v2", "v_add_f32_e32 v60, v1, v2", "s_nop 1",
                         "v_add_f32_dpp v61, v59, v60 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_endpgm"])
    out, st = P.fix(src)
    assert nops(out) == ["s_nop 1"] and st["compiler_nops_relaxed"] == 0


def test_compiler_padding_is_left_alone_behind_a_label_and_in_front_of_other_instructions():
    src = kernel(FILL[:3] + [".LBB0_4:"] + FILL[:3] + ["s_nop 1", "v_add_f32_dpp v61, v59, v60 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf bound_ctrl:1", "s_endpgm"])
    assert nops(P.fix(src)[0]) == ["s_nop 1"]
    src = kernel(FILL + ["s_nop 0", "v_cndmask_b32_dpp v61, v59, v60, vcc quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf", "s_endpgm"])
    assert nops(P.fix(src)[0]) == ["s_nop 0"]
    src = kernel(FILL + ["s_nop 0", "v_mul_f32_e32 v61, v59, v60", "s_endpgm"])
    assert nops(P.fix(src)[0]) == ["s_nop 0"]


def test_the_built_library_went_through_the_pass():
    side = os.path.join(ROOT, "hcr_genesis_lr_cl_amd", "csrc", "liblgsim.build.json")
    if not os.path.exists(side):
        pytest.skip("library not built")
    with open(side) as f:
        meta = json.load(f)
    st = meta.get("dpp_hazard_pass")
    assert st and st["marked"] > 0 and st["dpp"] > 10000, st
    # whatever the pass wrote next to the objects is clean under its own check (these files stay in the build tree)
    for path in glob.glob(os.path.join(ROOT, "hcr_genesis_lr_cl_amd", "csrc", "obj", "lg_inst_*.fix.s"))[:3]:
        with open(path) as f:
            assert P.check(f.read()) == [], path

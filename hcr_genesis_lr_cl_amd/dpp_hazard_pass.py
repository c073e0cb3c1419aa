"""DPP read-after-write hazards of the kernels, settled on the compiler's assembly (hcr_genesis_lr_cl_amd/build.py runs this between
`hipcc -S` and the assembler for every kernel translation unit).

gfx9 does not interlock "VALU writes a VGPR -> a DPP instruction reads that VGPR as its DPP source": two wait states (issue slots of the
wave) have to lie between the two.  The compiler pads its own DPP instructions, but it does not look inside inline asm, and the
products of csrc/lg_quad.h (v_mul_f32_dpp / v_fmac_f32_dpp sequences) are inline asm: each block used to open with its own `s_nop 1`,
needed or not.  One wave per SIMD hides nothing: an `s_nop 1` is two issue slots, ~8.6 cycles (tools/ubench/pk_issue.hip), and the
physics loop had 50 of them per sub-step -- 1.0 us of the 25.6 us go2 step.

Only the DPP-routed operand (src0) is subject to the rule: the accumulator of a v_fmac_f32_dpp chain is written and re-read in
consecutive slots throughout lg_quad.h's blocks, and the physics matches its f64 CPU restatement.  The compiler pads for every VGPR operand of a
DPP instruction; where that padding stands in front of a plain f32 / mov DPP instruction and nothing nearby has wait-state rules of its
own (_blocks_relaxing: SGPR / VCC / EXEC writers, transcendentals, partial-register writes, memory, ...), it is dropped too
(`_relax_compiler_nops`: 44 of the loop's remaining wait states, another 0.4 us).

This pass removes those marked nops (`s_nop 1 ; lg-dpp-hazard`) and then walks every function: for EVERY `*_dpp` instruction -- the
compiler's and the asm blocks' alike -- it measures the wait states since the last VALU write of the DPP source register over all
paths that reach the instruction (branch targets resolved; an unknown predecessor counts as a write), and inserts exactly the `s_nop`
that is missing.  It also covers what the compiler cannot see in the other direction: a compiler-emitted DPP read right behind an asm
block that wrote its source ("VALU writes EXEC -> DPP", five wait states, is checked the same way).  The result never has fewer wait
states in front of a DPP read than the hardware asks for; `check()` re-verifies the rewritten text and the build fails on a finding.
"""
import re
import sys

MARK = "lg-dpp-hazard"
VGPR_DPP_WAIT = 2       # VALU writes VGPR -> DPP reads it
EXEC_DPP_WAIT = 5       # VALU writes EXEC -> DPP op

_LABEL = re.compile(r"^([.\w$]+):")
_REG = re.compile(r"^v(\d+)$")
_RANGE = re.compile(r"^v\[(\d+):(\d+)\]$")


def _operands(rest):
    """Split the operand text at top-level commas; drop trailing modifiers (quad_perm:[..] row_mask:.. etc. stay attached to the last)."""
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _vgprs(op):
    """VGPR numbers named by one operand (`v12`, `-v12`, `|v12|`, `v[2:3]`), else an empty set."""
    op = op.split()[0] if op else ""
    op = op.strip("-|")
    if op.startswith("abs(") or op.startswith("neg("):
        op = op[4:].rstrip(")")
    m = _REG.match(op)
    if m:
        return {int(m.group(1))}
    m = _RANGE.match(op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


class Inst:
    __slots__ = ("mn", "ops", "line", "marked", "text")

    def __init__(self, mn, ops, line, marked, text):
        self.mn, self.ops, self.line, self.marked, self.text = mn, ops, line, marked, text

    @property
    def is_dpp(self):
        return self.mn.endswith("_dpp")

    @property
    def wait_states(self):
        if self.mn == "s_nop":
            return int(self.ops[0], 0) + 1 if self.ops else 1
        return 1

    def vgpr_writes(self):
        mn = self.mn
        if not mn.startswith("v_") or mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            return set()
        w = _vgprs(self.ops[0]) if self.ops else set()
        if mn.startswith("v_swap") and len(self.ops) > 1:
            w |= _vgprs(self.ops[1])
        return w

    def writes_exec(self):
        return self.mn.startswith("v_cmpx") or (self.mn.startswith("v_") and bool(self.ops) and self.ops[0].split()[0] in ("exec", "exec_lo", "exec_hi"))

    def dpp_source(self):
        return _vgprs(self.ops[1]) if len(self.ops) > 1 else set()

    def branch_target(self):
        if self.mn.startswith(("s_cbranch", "s_branch")) and self.ops:
            return self.ops[0]
        return None


def _parse(lines):
    """-> list of functions; a function is a list of items: ('label', name, line) | Inst.  Only text between a global symbol label and
    its .Lfunc_end is a function."""
    funcs, cur = [], None
    for i, raw in enumerate(lines):
        s = raw.strip()
        if not s or s.startswith((";", "//")):
            continue
        m = _LABEL.match(s)
        if m:
            name = m.group(1)
            if name.startswith(".Lfunc_end"):
                if cur is not None:
                    funcs.append(cur)
                cur = None
            elif not name.startswith("."):
                cur = [("label", name, i)]
            elif cur is not None:
                cur.append(("label", name, i))
            continue
        if s.startswith(".") or cur is None:
            continue
        code = s.split(";")[0].split("//")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        mn = parts[0]
        ops = _operands(parts[1]) if len(parts) > 1 else []
        cur.append(Inst(mn, ops, i, MARK in s and mn == "s_nop", s))
    if cur is not None:
        funcs.append(cur)
    return funcs


def _needed(items, idx, labels, branches):
    """Wait states missing in front of the DPP instruction items[idx] (0 = none)."""
    inst = items[idx]
    src = inst.dpp_source()
    worst = 0

    def walk(j, ws, depth):
        # scan backwards from item j (exclusive) with `ws` wait states already between; returns nothing, updates `worst`
        nonlocal worst
        while j > 0:
            j -= 1
            it = items[j]
            if isinstance(it, tuple):          # a label: whoever branches here is a predecessor too
                for b in branches.get(it[1], ()):  # the branch instruction itself is one wait state
                    if depth < 8:
                        walk(b + 1, ws, depth + 1)
                    else:                      # never seen; if it happens the path counts as an immediate writer
                        worst = max(worst, VGPR_DPP_WAIT - min(ws, VGPR_DPP_WAIT))
                continue                       # ... and so is the fall-through path above the label (if it falls through)
            if it.mn == "s_branch" or it.mn == "s_endpgm" or it.mn.startswith("s_setpc"):
                return                         # nothing falls through an unconditional branch
            if ws < VGPR_DPP_WAIT and (it.vgpr_writes() & src):
                worst = max(worst, VGPR_DPP_WAIT - ws)
            if ws < EXEC_DPP_WAIT and it.writes_exec():
                worst = max(worst, EXEC_DPP_WAIT - ws)
            ws += it.wait_states
            if ws >= EXEC_DPP_WAIT:
                return
        # ran off the top of the function with the window still open: the kernel entry, no writer before it
    walk(idx, 0, 0)
    return worst


def _analyse(items):
    labels = {it[1]: k for k, it in enumerate(items) if isinstance(it, tuple)}
    branches = {}
    for k, it in enumerate(items):
        if isinstance(it, Inst):
            t = it.branch_target()
            if t is not None:
                branches.setdefault(t, []).append(k)
    return labels, branches


_PLAIN_DPP = ("v_mov_b32_dpp", "v_add_f32_dpp", "v_sub_f32_dpp", "v_subrev_f32_dpp", "v_mul_f32_dpp", "v_fmac_f32_dpp", "v_max_f32_dpp", "v_min_f32_dpp")
_TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")
RELAX_WINDOW = 6


def _blocks_relaxing(it):
    """True for anything in front of a compiler nop that could be the reason for it other than the DPP rule, or whose own consumers count
    wait states across it: VALU writes of SGPR / VCC / EXEC (lane selects, v_div_fmas, VMEM / VALU reads of that SGPR: up to 5 wait states),
    transcendental and partial-register (op_sel, SDWA, packed) results (forwarding hazards, 1 wait state), stores (their data registers
    may not be overwritten in the next slot), hardware-register accesses, and any instruction this pass does not know."""
    mn = it.mn
    if mn.startswith("v_"):
        if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_writelane", "v_div_", "v_mad_u64", "v_mad_i64", "v_mfma", "v_accvgpr", "v_permlane") + _TRANS):
            return True
        if "_co_" in mn or "_pk_" in mn or "sdwa" in mn or "op_sel" in it.text or "16" in mn:
            return True
        if it.ops and it.ops[0].split()[0].startswith(("s", "vcc", "exec", "a", "m0")):
            return True
        return False
    if mn in ("s_nop", "s_mov_b32", "s_mov_b64", "s_and_b64", "s_or_b64", "s_andn2_b64", "s_xor_b64", "s_add_i32", "s_sub_i32", "s_cmp_eq_u32", "s_cmp_lg_u32",
              "s_cselect_b32", "s_cselect_b64", "s_lshl_b32", "s_and_b32", "s_or_b32", "s_cmp_lt_i32", "s_cmp_ge_u32", "s_cmp_lg_u64", "s_mul_i32"):
        return False
    return True            # memory, branches, waitcnt, setreg, message, ... : leave the nop alone


def _relax_compiler_nops(lines, stats):
    """The compiler pads a DPP instruction when ANY of its VGPR operands was written in the two slots before; the hardware rule concerns the
    DPP-routed source only (the accumulators of lg_quad.h's blocks are written and re-read back to back).  Drop a compiler s_nop that stands
    directly in front of a plain f32 / mov DPP instruction when nothing in the RELAX_WINDOW instructions before it is of a kind that could
    need wait states for another reason (_blocks_relaxing) and no label lies in between; the main loop of fix() then re-inserts whatever
    the DPP source itself needs."""
    drop = set()
    for items in _parse(lines):
        for k, it in enumerate(items):
            if not (isinstance(it, Inst) and it.mn == "s_nop" and not it.marked and MARK not in it.text):
                continue
            nxt = items[k + 1] if k + 1 < len(items) else None
            if not (isinstance(nxt, Inst) and nxt.mn in _PLAIN_DPP):
                continue
            ok, j, seen = True, k, 0
            while seen < RELAX_WINDOW:
                j -= 1
                if j < 0 or isinstance(items[j], tuple) or _blocks_relaxing(items[j]):
                    ok = False
                    break
                seen += 1
            if ok:
                drop.add(it.line)
                stats["compiler_nops_relaxed"] += 1
                stats["compiler_wait_states_relaxed"] += it.wait_states
    return [raw for i, raw in enumerate(lines) if i not in drop]


def fix(text, relax=True):
    """-> (new text, stats).  Removes the marked nops (and, `relax`, the compiler's over-wide padding in front of plain DPP instructions),
    then inserts the minimal s_nop in front of every DPP read that needs one."""
    lines = text.split("\n")
    stats = {"marked": 0, "kept_or_inserted": 0, "wait_states_inserted": 0, "dpp": 0, "functions": 0, "compiler_nops_relaxed": 0, "compiler_wait_states_relaxed": 0}
    # 1. drop the marked nops
    keep = []
    for raw in lines:
        s = raw.strip()
        if MARK in s and s.startswith("s_nop"):
            stats["marked"] += 1
            continue
        keep.append(raw)
    lines = keep
    if relax:
        lines = _relax_compiler_nops(lines, stats)
    # 2. insert what is missing, function by function, top to bottom (an insertion only lengthens later distances)
    funcs = _parse(lines)
    stats["functions"] = len(funcs)
    inserts = {}                                # line index -> s_nop operand
    for items in funcs:
        labels, branches = _analyse(items)
        k = 0
        while k < len(items):
            it = items[k]
            if isinstance(it, Inst) and it.is_dpp:
                stats["dpp"] += 1
                need = _needed(items, k, labels, branches)
                if need > 0:
                    nop = Inst("s_nop", [str(need - 1)], it.line, False, "")
                    items.insert(k, nop)
                    # branch indices behind k shift by one
                    for t in branches:
                        branches[t] = [b + 1 if b >= k else b for b in branches[t]]
                    inserts[it.line] = max(inserts.get(it.line, 0), need)
                    stats["kept_or_inserted"] += 1
                    stats["wait_states_inserted"] += need
                    k += 1
            k += 1
    out = []
    for i, raw in enumerate(lines):
        if i in inserts:
            out.append(f"\ts_nop {inserts[i] - 1} ; {MARK} (pass)")
        out.append(raw)
    return "\n".join(out), stats


def check(text):
    """Findings (function, line number, instruction, missing wait states) of a text as it stands; [] = clean."""
    lines = text.split("\n")
    bad = []
    for items in _parse(lines):
        labels, branches = _analyse(items)
        for k, it in enumerate(items):
            if isinstance(it, Inst) and it.is_dpp:
                need = _needed(items, k, labels, branches)
                if need > 0:
                    bad.append((items[0][1], it.line + 1, it.text, need))
    return bad


if __name__ == "__main__":
    src = open(sys.argv[1]).read()
    if len(sys.argv) > 2 and sys.argv[2] == "--check":
        f = check(src)
        for x in f[:50]:
            print(x)
        print(len(f), "finding(s)")
        sys.exit(1 if f else 0)
    new, st = fix(src)
    print(st, "findings after:", len(check(new)))
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(new)

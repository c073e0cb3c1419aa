"""DPP read-after-write hazards of the kernels, settled on the compiler's assembly (hcr_genesis_lr_cl_amd/build.py runs this between
`hipcc -S` and the assembler for every kernel translation unit).

gfx9 does not interlock "VALU writes a VGPR -> a DPP instruction reads that VGPR as its DPP source": two wait states (issue slots of the
wave) have to lie between the two.  The compiler pads its own DPP instructions, but it does not look inside inline asm, and the
products of csrc/lg_quad.h (v_mul_f32_dpp / v_fmac_f32_dpp sequences) are inline asm: each block used to open with its own `s_nop 1`,
needed or not.  One wave per SIMD hides nothing: an `s_nop 1` is two issue slots, ~8.6 cycles (tools/ubench/pk_issue.hip), and the
physics loop had 50 of them per sub-step -- 1.0 us of the 25.6 us go2 step.

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


def fix(text):
    """-> (new text, stats).  Removes the marked nops, then inserts the minimal s_nop in front of every DPP read that needs one."""
    lines = text.split("\n")
    stats = {"marked": 0, "kept_or_inserted": 0, "wait_states_inserted": 0, "dpp": 0, "functions": 0}
    # 1. drop the marked nops
    keep = []
    for raw in lines:
        s = raw.strip()
        if MARK in s and s.startswith("s_nop"):
            stats["marked"] += 1
            continue
        keep.append(raw)
    lines = keep
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

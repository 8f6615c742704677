"""Static checks on the gfx950 code of the product (test infrastructure; CPU only: hipcc cross-compiles without a GPU).

Two guards for the hand-scheduled field arithmetic of csrc/gl.h and csrc/poseidon_fast.h:

1. `salu_in_asm_strings(path)`: no scalar-ALU mnemonic other than `s_nop` inside an inline-asm string.  The compiler does not
   see inside an asm string: an `s_or_b64` there clobbered SCC behind its back in round 2 and ended in a GPU memory-access
   fault (gpurun_out/bench_h1.log).

2. `sgpr_hazards(asm_text)`: on gfx950 (the gfx940 family) a VALU instruction that reads an SGPR -- a carry-in, a
   v_cndmask mask, a constant operand -- needs TWO wait states after the last VALU instruction that wrote that SGPR
   (LLVM's GCNHazardRecognizer: VALUWriteSGPRVALUReadWaitstates = 2).  The compiler inserts them for its own code but not
   around inline asm, where gl.h places `s_nop 1` by hand or relies on the data flow (`*_settled`).  This walks the emitted
   assembly of every function as a control-flow graph and reports every VALU read of an SGPR closer than two wait states to
   a VALU write of it on ANY path.  One instruction = one wait state, `s_nop N` = N + 1.
"""
import re

# ----------------------------------------------------------------------------------------------------------- asm strings
_SALU_OK = {"s_nop"}


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", lambda m: " " * len(m.group(0)), src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def asm_templates(src):
    """(line number, template text) of every asm statement: the string literals before the first operand colon."""
    src = _strip_comments(src)
    out = []
    for m in re.finditer(r"\basm\s*(?:volatile\s*)?\(", src):
        i, depth, in_str, lits, cur = m.end(), 1, False, [], []
        while i < len(src) and depth:
            ch = src[i]
            if in_str:
                if ch == "\\":
                    cur.append(src[i:i + 2])
                    i += 2
                    continue
                if ch == '"':
                    in_str = False
                    lits.append("".join(cur))
                    cur = []
                else:
                    cur.append(ch)
            elif ch == '"':
                in_str = True
            elif ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif ch == ":" and depth == 1:
                break
            i += 1
        out.append((src.count("\n", 0, m.start()) + 1, "".join(lits).replace("\\n", "\n").replace("\\t", "\t")))
    return out


def salu_in_asm_strings(path):
    """[(line, mnemonic)] for every scalar-unit instruction other than s_nop inside an asm template of the file."""
    return salu_in_asm_strings_text(open(path).read())


def salu_in_asm_strings_text(src):
    bad = []
    for line, text in asm_templates(src):
        for ins in re.split(r"[\n;]", text):
            mn = ins.strip().split(" ")[0].split("\t")[0]
            if mn.startswith("s_") and mn not in _SALU_OK:
                bad.append((line, mn))
    return bad


# ------------------------------------------------------------------------------------------------------------ hazards
_SALU_NO_DST = ("s_cmp", "s_cbranch", "s_branch", "s_nop", "s_waitcnt", "s_barrier", "s_endpgm", "s_setpc", "s_sleep", "s_store", "s_bitcmp",
                "s_setprio", "s_sethalt", "s_setreg", "s_dcache", "s_icache", "s_trap", "s_sendmsg", "s_set_gpr", "s_setvskip")
NEED = 2  # wait states between a VALU write of an SGPR and a VALU read of it

_SREG = re.compile(r"(?<![\w.])(?:s\[(\d+):(\d+)\]|s(\d+)(?!\w)|(vcc_lo|vcc_hi|vcc)(?!\w))")


def _sregs(operand):
    regs = set()
    for m in _SREG.finditer(operand):
        if m.group(1) is not None:
            regs.update("s%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1))
        elif m.group(3) is not None:
            regs.add("s" + m.group(3))
        elif m.group(4) == "vcc":
            regs.update(("vcc_lo", "vcc_hi"))
        else:
            regs.add(m.group(4))
    return regs


def _split_operands(rest):
    ops, depth, cur = [], 0, []
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        ops.append("".join(cur).strip())
    return ops


def _is_valu(mn):
    return mn.startswith("v_") and not mn.startswith("v_mfma") and not mn.startswith("v_smfmac")


def _valu_sgpr_defs_uses(mn, ops):
    """SGPRs a VALU instruction writes and reads.  Destination layout of the opcodes that write SGPRs:
    v_mad_[ui]64_[ui]32 vdst, sdst, ...; v_{add,sub,subrev}{,c,b}_co_* vdst, sdst, ...; v_div_scale vdst, sdst, ...;
    v_cmp_* sdst, ... (v_cmpx: exec); v_readfirstlane / v_readlane sdst, ...  Every other VALU opcode has one (vector) dst."""
    defs, n_dst = set(), 1
    if re.match(r"v_mad_[ui]64_[ui]32", mn) or re.match(r"v_(add|sub|subrev)(c|b|brev)?_co_", mn) or mn.startswith("v_div_scale"):
        n_dst = 2
        if len(ops) > 1:
            defs = _sregs(ops[1])
    elif mn.startswith("v_cmpx"):
        n_dst = 1 if (ops and (ops[0].startswith("exec") or _sregs(ops[0]))) else 0
    elif mn.startswith("v_cmp"):
        defs = _sregs(ops[0]) if ops else set()
    elif mn.startswith("v_readfirstlane") or mn.startswith("v_readlane"):
        defs = _sregs(ops[0]) if ops else set()
    uses = set()
    for o in ops[n_dst:]:
        uses |= _sregs(o)
    return defs, uses


def parse_functions(asm_text):
    """{function name: [basic blocks]}, a block = (label or None, [(line no, mnemonic, operands)], branch targets, falls through)."""
    funcs, cur, name = {}, None, None
    for no, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L"):
                name, cur = lab, [[lab, [], [], True]]
                funcs[name] = cur
            elif lab.startswith(".Lfunc_end"):
                cur = None
            elif cur is not None:
                cur.append([lab, [], [], True])
            continue
        if cur is None or not line.startswith("\t") or line.strip().startswith("."):
            continue
        parts = line.strip().split(None, 1)
        mn, ops = parts[0], _split_operands(parts[1]) if len(parts) > 1 else []
        blk = cur[-1]
        if not blk[3]:  # code after an unconditional branch without a label of its own: unreachable filler
            cur.append([None, [], [], True])
            blk = cur[-1]
        blk[1].append((no, mn, ops))
        if mn.startswith("s_cbranch") or mn == "s_branch":
            blk[2].extend(o for o in ops if o.startswith(".L"))
            if mn == "s_branch":
                blk[3] = False
            else:
                cur.append([None, [], [], True])
        elif mn in ("s_endpgm", "s_setpc_b64"):
            blk[3] = False
    return funcs


def _walk(block, state, report):
    """state: {sgpr: wait states since its last VALU write} for the registers still inside the window."""
    state = dict(state)
    for no, mn, ops in block[1]:
        if _is_valu(mn):
            defs, uses = _valu_sgpr_defs_uses(mn, ops)
            if report is not None:
                for r in uses:
                    if r in state and state[r] < NEED:
                        report.append((no, mn, r, state[r]))
        else:
            defs = set()
            # A scalar-unit write ends the window: the scalar unit interlocks on VALU-written SGPRs it reads, and a VALU
            # read right behind a scalar write needs no wait state (the compiler's own `s_or_b64` + `v_cndmask` in gl::add).
            if mn.startswith("s_") and ops and not mn.startswith(_SALU_NO_DST):
                for r in _sregs(ops[0]):
                    state.pop(r, None)
        step = 1
        if mn == "s_nop":
            step = int(ops[0], 0) + 1
        state = {r: w + step for r, w in state.items() if w + step < NEED}
        for r in defs:
            state[r] = 0
    return state


def sgpr_hazards(asm_text, only=None):
    """[(function, asm line, mnemonic, sgpr, wait states seen)] for every VALU SGPR read inside the hazard window."""
    out = []
    for fn, blocks in parse_functions(asm_text).items():
        if only and not any(frag in fn for frag in only):
            continue
        index = {b[0]: i for i, b in enumerate(blocks) if b[0]}
        preds = [[] for _ in blocks]
        for i, b in enumerate(blocks):
            if b[3] and i + 1 < len(blocks):
                preds[i + 1].append(i)
            for t in b[2]:
                if t in index:
                    preds[index[t]].append(i)
        outs = [dict() for _ in blocks]
        ins = [dict() for _ in blocks]
        work = list(range(len(blocks)))
        while work:
            i = work.pop(0)
            merged = {}
            for p in preds[i]:
                for r, w in outs[p].items():
                    merged[r] = min(w, merged.get(r, NEED))
            ins[i] = merged
            o = _walk(blocks[i], merged, None)
            if o != outs[i]:
                outs[i] = o
                for j, ps in enumerate(preds):
                    if i in ps and j not in work:
                        work.append(j)
        for i, b in enumerate(blocks):
            rep = []
            _walk(b, ins[i], rep)
            out.extend((fn,) + r for r in rep)
    return out


def count_reads_at_minimum(asm_text):
    """How many VALU SGPR reads sit at exactly NEED wait states (no slack) -- reported, not asserted."""
    n = 0
    for blocks in parse_functions(asm_text).values():
        for b in blocks:
            state = {}
            for no, mn, ops in b[1]:
                step = int(ops[0], 0) + 1 if mn == "s_nop" else 1
                if _is_valu(mn):
                    defs, uses = _valu_sgpr_defs_uses(mn, ops)
                    n += sum(1 for r in uses if state.get(r) == NEED)
                else:
                    defs = set()
                state = {r: w + step for r, w in state.items() if w + step <= NEED}
                for r in defs:
                    state[r] = 0
    return n


def longest_load_batch(asm_text, fragment):
    """Most global loads a function whose name contains `fragment` has in flight at one point, walking its code in layout order
    (uniform branches between the loads of a step do not end a batch): loads are counted from one `s_waitcnt vmcnt(..)` to the
    next (a wait with a non-zero count leaves that many of them outstanding).
    Guards the kernels whose first step is written as 'every load issued, then the arithmetic' (loads_issued in kernels.h)
    against a compiler that goes back to loading one value, waiting, multiplying, loading the next."""
    best = 0
    for name, blocks in parse_functions(asm_text).items():
        if fragment not in name:
            continue
        outstanding = 0
        for _, instrs, _, _ in blocks:
            for _, mn, ops in instrs:
                if mn.startswith("global_load") or mn.startswith("buffer_load"):
                    outstanding += 1
                    best = max(best, outstanding)
                elif mn == "s_waitcnt":
                    m = re.search(r"vmcnt\((\d+)\)", " ".join(ops))
                    if m:
                        outstanding = min(outstanding, int(m.group(1)))
    return best


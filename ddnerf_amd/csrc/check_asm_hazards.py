"""Build-time check of the hand-placed MFMA streams (mlp_bf16.hip).

The matrix instructions of that kernel are written as inline asm (so that the accumulators stay in place), and hipcc pads
nothing around an asm statement: the wait states between a VALU write and an MFMA that reads the register, between an
MFMA and a VALU write to one of its sources, and between an MFMA and a reader of its result exist only because the
written instruction order keeps such pairs far apart.  A compiler-inserted register copy next to an MFMA would break that
silently (it did once: outputs off by 1e-3).  This script scans the compiler's assembly of the kernel and fails the build
when a dependent pair comes closer than the distances below (counted in wait states: one per instruction, n + 1 for
`s_nop n`, 2 for an MFMA in between, which holds the issue port for 8 cycles).

usage: check_asm_hazards.py file.s [kernel-name-substring]
       check_asm_hazards.py file.s kernel-name-substring --sload      (check_hidden_sloads below)"""
import re
import sys

# Distances (each one above what was measured / what hipcc itself pads for the builtin form of v_mfma_f32_16x16x32_bf16, a
# 4-pass XDL op): VALU write -> MFMA read needs 2 wait states (an `s_nop 1` in front of every MFMA made a build with
# compiler-inserted copies bit-exact again, `s_nop 0` did not); hipcc pads `s_nop 2` between such an MFMA and a write to
# its srcC registers (A / B operands are read at issue: it pads nothing for those); XDL 4-pass result -> VALU read: 7.
RAW_VALU_TO_MFMA = 3   # VALU (incl. v_accvgpr_write) writes a register an MFMA reads
WAR_MFMA_SRCC = 4      # a VALU instruction writes a register an MFMA issued just before reads as srcC
RAW_MFMA_TO_ANY = 8    # anything but an MFMA reads an MFMA result

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out.update((m.group(1), i) for i in range(lo, hi + 1))
    return out


def parse(path, kernel, with_labels=False):
    """instructions of every instantiation of `kernel`; with_labels: also (label -> index of the instruction behind it)"""
    ins, on, labels = [], False, {}
    for line in open(path):
        if re.match(r"^_Z\w*%s\w*:" % kernel, line):
            on = True
            continue
        if not on:
            continue
        s = line.strip()
        if s.startswith("s_endpgm"):
            on = False
            continue
        lab = s.split(";")[0].strip()       # (".LBB0_6:      ; =>This Inner Loop Header")
        if lab.endswith(":") and lab.startswith(".L"):
            labels[lab[:-1]] = len(ins)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        mnem, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(";")[0].split(",")] if rest else []
        ins.append((mnem, ops, s))
    return (ins, labels) if with_labels else ins


def defs_uses(mnem, ops):
    """(written registers, read registers) of the instruction forms this kernel contains"""
    if mnem.startswith(("v_mfma", "v_")) or mnem.startswith(("ds_read", "global_load", "buffer_load", "scratch_load")):
        d = regs(ops[0]) if ops else set()
        u = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        return d, u
    if mnem.startswith(("ds_write", "global_store", "buffer_store", "scratch_store", "global_atomic")):
        return set(), set().union(*[regs(o) for o in ops]) if ops else set()
    return set(), set().union(*[regs(o) for o in ops]) if ops else set()


WINDOW = 48  # instructions on either side of a loop's back edge that are scanned as one straight line


def check(path, kernel="mlp_bf16_fwd_kernel"):
    """Limits (the bit-exact GPU tests remain the real gate): the scan is a straight line over the listing plus, for every BACKWARD
    branch, the last WINDOW instructions in front of it followed by the first WINDOW behind its target (the tile loop's wrap-around);
    forward branches are not followed; operand 0 of an instruction is taken as its only definition (instructions that also
    write an SGPR / VCC are not modelled: none of them feeds an MFMA here); the distances are empirical (see above)."""
    ins, labels = parse(path, kernel, with_labels=True)
    if not any(m.startswith("v_mfma") for m, _, _ in ins):
        raise SystemExit("check_asm_hazards: no MFMA found in %s (kernel %s)" % (path, kernel))
    n, bad = _scan(ins)
    far = None  # target of a long branch being assembled: s_getpc_b64 / s_add_u32 x, x, (.LBBn_m-.Lpost_getpck)&... / s_setpc_b64
    seams = 0
    for p, (m, o, text) in enumerate(ins):  # loop back edges: scan the seam
        q = None
        if m.startswith("s_cbranch") or m == "s_branch":
            q = labels.get(o[0]) if o else None
        elif m == "s_add_u32":
            t = re.search(r"\((\.LBB\w+)-\.Lpost_getpc\d+\)&", text)
            far = t.group(1) if t else far
        elif m == "s_setpc_b64":
            q, far = labels.get(far), None
        if q is not None and q <= p:
            seams += 1
            seam = ins[max(q, p - WINDOW):p] + ins[q:q + WINDOW]
            bad += [b for b in _scan(seam)[1] if b not in bad]
    check.seams = seams
    return n, bad


def _scan(ins):
    du = [defs_uses(m, o) for m, o, _ in ins]
    # wait states an instruction contributes to a distance: s_nop n -> n + 1; an MFMA holds the issue port for 8 cycles -> 2
    slots = [int(o[0]) + 1 if m == "s_nop" and o else (2 if m.startswith("v_mfma") else 1) for m, o, _ in ins]
    bad = []
    for i, (m, o, text) in enumerate(ins):
        if not m.startswith("v_mfma"):
            continue
        d_i, u_i = du[i]
        # backwards: a VALU result read by this MFMA
        dist, j = 0, i - 1
        while j >= 0 and dist < RAW_VALU_TO_MFMA:
            mj = ins[j][0]
            if mj.startswith("v_") and not mj.startswith("v_mfma") and du[j][0] & u_i:
                bad.append("VALU write -> MFMA read, %d apart:\n    %s\n    %s" % (dist + 1, ins[j][2], text))
            dist += slots[j]
            j -= 1
        # forwards: writers of this MFMA's sources, readers of its result
        dist, j = 0, i + 1
        while j < len(ins) and dist < RAW_MFMA_TO_ANY:
            mj = ins[j][0]
            if not mj.startswith("v_mfma"):
                # (LDS / VMEM loads into such a register return tens of cycles later: only VALU writes can land in time)
                if dist < WAR_MFMA_SRCC and len(o) > 3 and mj.startswith("v_") and du[j][0] & regs(o[3]):
                    bad.append("MFMA srcC read -> write, %d apart:\n    %s\n    %s" % (dist + 1, text, ins[j][2]))
                if du[j][1] & d_i:
                    bad.append("MFMA write -> read, %d apart:\n    %s\n    %s" % (dist + 1, text, ins[j][2]))
            dist += slots[j]
            j += 1
    # The fused body (gen_bf16_g2.py, Gen(fused=True)) carries an encoder: transcendentals and IEEE divisions.  Two gfx940-family
    # rules hipcc pads for and nothing pads inside asm: a transcendental's result may not be read by the VALU instruction issued right
    # behind it (one wait state), and v_div_fmas reads the VCC a VALU instruction wrote no less than four wait states earlier.
    TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
    for i, (m, o, text) in enumerate(ins):
        if m in TRANS and i + 1 < len(ins):
            mj = ins[i + 1][0]
            if mj.startswith("v_") and mj not in TRANS and du[i + 1][1] & du[i][0]:
                bad.append("transcendental result read by the next instruction:\n    %s\n    %s" % (text, ins[i + 1][2]))
        if m.startswith("v_div_fmas"):
            dist, j = 0, i - 1
            while j >= 0 and dist < 4:
                mj, oj = ins[j][0], ins[j][1]
                if mj.startswith("v_") and any(x.strip() == "vcc" for x in oj[:2]) and not mj.startswith("v_div_fmas"):
                    bad.append("VALU write of vcc -> v_div_fmas, %d apart:\n    %s\n    %s" % (dist + 1, ins[j][2], text))
                dist += slots[j]
                j -= 1
    return len([1 for m, _, _ in ins if m.startswith("v_mfma")]), bad


def check_scalar_shell(path, kernel):
    """A kernel whose tile body is ONE inline-asm block that keeps its state in fixed vector registers across the iterations of a
    compiler-written loop (mlp_bf16_g2.hip): every instruction of that kernel outside the asm block, from the first block on, must be
    scalar.  Returns the offending lines."""
    bad, on, inside, seen = [], False, False, False
    for line in open(path):
        if re.match(r"^_Z\w*%s\w*:" % kernel, line):
            on, inside, seen = True, False, False
            continue
        if not on:
            continue
        t = line.strip()
        if ";#ASMSTART" in t:
            inside = seen = True
        elif ";#ASMEND" in t:
            inside = False
        elif t.startswith("s_endpgm"):
            on = False
        elif seen and not inside and re.match(r"^(v_|ds_|buffer_|global_|flat_|scratch_)", t):
            bad.append(t)
    return bad


SREG = re.compile(r"\bs(?:(\d+)|\[(\d+):(\d+)\])")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        lo = int(m.group(1) if m.group(1) is not None else m.group(2))
        hi = int(m.group(1) if m.group(1) is not None else m.group(3))
        out.update(range(lo, hi + 1))
    return out


def check_hidden_sloads(path, kernel):
    """Scalar loads issued from inline asm (mlp_f32_train.hip, SignLoader: the compiler's s_waitcnt insertion does not know them) deliver
    into registers the compiler believes are already written.  From such a load to the next `s_waitcnt ... lgkmcnt(0)` in program order
    NO instruction may read or write its destination registers -- a register-allocator copy or spill placed there would move stale
    bits.  Straight-line scan in text order (the kernel has no loop around these loads).  Returns (number of hidden loads, offending lines)."""
    bad, on, inside, pending, n = [], False, False, {}, 0
    for line in open(path):
        if re.match(r"^_Z\w*%s\w*:" % kernel, line):
            on, inside, pending = True, False, {}
            continue
        if not on:
            continue
        t = line.strip()
        if ";#ASMSTART" in t:
            inside = True
            continue
        if ";#ASMEND" in t:
            inside = False
            continue
        if t.startswith("s_endpgm"):
            if pending:
                bad.append("s_endpgm with scalar loads never waited for: s%s" % sorted(pending))
            on = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0]
        if code.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
            pending = {}
            continue
        touched = sregs(code.partition(" ")[2])
        hit = touched & set(pending)
        if hit:
            bad.append("%s   <- touches s%s, requested by `%s` and not yet waited for" % (t, sorted(hit), pending[min(hit)]))
        if inside and code.startswith("s_load_dword"):
            n += 1
            for r in sregs(code.partition(" ")[2].split(",")[0]):
                pending[r] = code
    return n, bad


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "--sload":
        n, bad = check_hidden_sloads(sys.argv[1], sys.argv[2])
        for b in bad[:20]:
            print(b)
        print("check_asm_hazards --sload: %d scalar loads issued from inline asm, %d instructions touch a destination before its wait" % (n, len(bad)))
        sys.exit(1 if bad or not n else 0)
    n, bad = check(sys.argv[1], *(sys.argv[2:3]))
    for b in bad[:20]:
        print(b)
    print("check_asm_hazards: %d MFMAs, %d too-close dependent pairs (straight-line scan + %d loop back-edge seams; limits: check.__doc__)"
          % (n, len(bad), check.seams))
    sys.exit(1 if bad else 0)

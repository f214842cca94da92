"""Static checks of the generated assembly tile body of the two-group bf16 MLP kernel (ddnerf_amd/csrc/gen_bf16_g2.py): what the
generator promises by construction is re-derived here from its OUTPUT -- operand ranges, the matrix-instruction count, register
budgets, and an independent replay of the memory counters (every s_waitcnt must be satisfiable and every register that a load fills
must have been waited for before a matrix instruction reads it).  No GPU: the bit-exact comparisons are tests/test_hip_bf16_g2.py."""
import importlib.util
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "ddnerf_amd", "csrc", "gen_bf16_g2.py")


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_bf16_g2", GEN)
    argv, sys.argv = sys.argv, ["gen_bf16_g2"]
    try:
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


@pytest.fixture(scope="module", params=[(0, 0, False), (1, 0, False), (0, 1, False), (0, 0, True), (1, 0, True)],
                ids=["rgb", "depth_head", "stamped", "fused_rgb", "fused_depth_head"])
def body(gen, request):
    """(depth head, stamped, fused: the body with the encoder inside, mlp_bf16_g2e.hip)"""
    gen.STAMP_PERIODS = bool(request.param[1])       # (the per-period stamps of the diagnostic build: opt-in in the generator)
    g, blocks, nk = gen.generate(*request.param)     # (asserts that the counters' steady state is a fixed point)
    gen.STAMP_PERIODS = False
    return [x.strip() for x in g.out if x.strip() and not x.startswith(";")], request.param


def _regs(tok):
    out = []
    for m in re.finditer(r"\b([vas])(?:(\d+)|\[(\d+):(\d+)\])", tok):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out += [(m.group(1), i) for i in range(lo, hi + 1)]
    return out


def test_plan(gen):
    assert gen.NPER == 82 and len(gen.real) == 47 and gen.IMG_BYTES % 4096 == 0
    assert sum(gen.npw_of(d["chunk"]) for d in gen.real) == 389            # pieces per wave and 512-sample tile (one-group kernel: 714)
    for d in gen.real:
        assert d["uses"][0] - d["issue"] >= 2
    # three files, roles rotate with period three; the spare file of a layer is the one neither group reads
    for l in range(1, 10):
        assert sorted(gen.files(l)) == [0, 1, 2] and gen.files(l) == gen.files(l + 3)


def test_operand_ranges_and_counts(gen, body):
    lines, (depth_head, stamp, fused) = body
    n_mfma = n_dma = 0
    for t in lines:
        op = t.split()[0]
        for kind, i in _regs(t.replace("%", " ")):
            assert 0 <= i < (256 if kind in "va" else 102), t
        if op == "v_mfma_f32_16x16x32_bf16":
            n_mfma += 1
        elif op == "ds_read_b128":
            off = int(re.search(r"offset:(\d+)", t).group(1))
            assert off + 16 <= gen.SLOT_BYTES, t                             # base register = slot + lane part; the immediate stays inside the slot
        elif op == "global_load_lds_dwordx4":
            n_dma += 1
            off = int(re.search(r"offset:(-?\d+)", t).group(1))
            assert -4096 <= off <= 4095 and off % 1024 == 0, t
        elif op.startswith("buffer_"):
            m = re.search(r"offset:(\d+)", t)
            assert m is None or int(m.group(1)) < 4096, t
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            assert m is None or int(m.group(1)) <= 63, t
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            assert m is None or int(m.group(1)) <= 15, t
    for a, b in zip(lines, lines[1:]):                                       # an SALU write of M0 needs a wait state before the LDS-DMA that uses it
        assert not (re.match(r"s_\w+ m0,", a) and b.startswith("global_load_lds")), (a, b)
    assert n_mfma == 9640                                                    # 2 groups x 4820: exactly the one-group kernel's per 256 samples
    pro = sum(gen.npw_of(d["chunk"]) for d in gen.real if d["for_next_tile"])
    assert n_dma == 389 + pro                                                # the steady state + the first tile's prologue
    assert sum(1 for t in lines if t == "s_barrier") == gen.NPER + 1
    n_store = sum(1 for t in lines if t.startswith("buffer_store") and "%1," in t)
    assert n_store == (16 if depth_head else 8)                              # the outputs
    assert sum(1 for t in lines if t.startswith("buffer_store")) == n_store + (96 if fused else 0)
    if fused:
        enc = [t for t in lines if t.startswith("buffer_store_dwordx2") and ("%16" in t or "%0" in t)]
        # the encoder: 2 groups x 24 four-column pieces per tile, and the same once more in the first tile's prologue; every piece of a
        # group's area ([piece][sample][8 bytes]) exactly once per group, the store's soffset set by the instruction in front of it
        assert len(enc) == 96
        start = lines.index(".Lsteady%=:") + 1
        # (the first tile's own rows go to the scratch operand %0; the NEXT tile's through %16, which the shell empties in a workgroup's last tile)
        for part, res in ((lines[:start], "%0,"), (lines[start:], "%16,")):
            offs = {}
            for a, b in zip(part, part[1:]):
                if b.startswith("buffer_store_dwordx2") and res in b:
                    m = re.fullmatch(r"s_add_u32 s76, (s4[67]), (\d+)", a)
                    assert m and b.split(",")[3].split()[0] == "s76" and "offset" not in b, (a, b)
                    offs.setdefault(m.group(1), []).append(int(m.group(2)))
            assert sorted(offs) == ["s46", "s47"] and all(sorted(v) == list(range(0, 24 * 512, 512)) for v in offs.values())
        assert sum(1 for t in lines if t.startswith("v_sin_f32")) == 4 * 96 and sum(1 for t in lines if t.startswith("v_exp_f32")) == 4 * 48
        for t in lines:       # packed fp32 instructions: even-aligned register pairs, scalar constants as pairs
            if t.startswith("v_pk_"):
                for m in re.finditer(r"[vs]\[(\d+):(\d+)\]", t):
                    assert int(m.group(1)) % 2 == 0 and int(m.group(2)) == int(m.group(1)) + 1, t
        assert sum(1 for t in lines if t.startswith("v_div_fmas_f32")) == 4 * 4


def test_memory_counter_replay(gen, body):
    """Replay of the steady-state path with a model of the hardware counters that knows nothing of the generator's bookkeeping: loads
    retire in order, a wait vmcnt(n) leaves at most n of them in flight.  Every register written by a load must be out of flight when a
    matrix instruction, a store or a VALU instruction reads it; every LDS read likewise (lgkmcnt).  The tile body is replayed three
    times in a row (the second and third start with what the previous one left in flight)."""
    lines, (_, _, fused) = body
    start = lines.index(".Lsteady%=:") + 1
    steady = lines[start:]
    vm, lg, sm = [], [], []  # in-flight loads: lists of (destination registers)
    for rep in range(3):
        for t in steady:
            op = t.split()[0]
            toks = t.replace("%", " ")
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    vm = vm[len(vm) - n:] if n else []
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    lg = lg[len(lg) - n:] if n else []
                    if not n:
                        sm = []             # (scalar memory returns out of order: only lgkmcnt(0) says that a scalar load has landed)
                continue
            args = toks.split(None, 1)[1] if " " in toks else ""
            parts = [p.strip() for p in args.split(",")]
            if op.startswith("buffer_load"):
                vm.append(set(_regs(parts[0])))
                reads = _regs(parts[1])
            elif op == "global_load_lds_dwordx4":
                vm.append(set())
                reads = _regs(parts[0])
            elif op == "ds_read_b128":
                lg.append(set(_regs(parts[0])))
                reads = _regs(parts[1])
            elif op == "s_memtime":
                lg.append(set(_regs(parts[0])))
                reads = []
            elif op.startswith("s_load"):
                sm.append(set(_regs(parts[0])))
                reads = _regs(parts[2]) if len(parts) > 2 else []
            elif op.startswith(("buffer_store", "global_store")):
                reads = _regs(parts[0]) + _regs(parts[1])
                if fused:       # (the fused body books its stores into the in-order queue: gen_bf16_g2.py, vm_inorder)
                    vm.append(set())
            elif op.startswith(("v_", "s_")) and parts and op not in ("s_barrier", "s_nop", "s_cbranch_scc1"):
                reads = [r for p in parts[1:] for r in _regs(p)]
                if op.startswith("v_mfma") or op.startswith("s_add") or op.startswith("s_mul"):
                    pass
            else:
                reads = []
            inflight = set().union(*vm) if vm else set()
            inflight_l = set().union(*(lg + sm)) if lg or sm else set()
            for r in reads:
                assert r not in inflight, ("read of a register a load is still filling", t)
                assert r not in inflight_l, ("read of a register an LDS read is still filling", t)
        assert not lg and not sm                            # (the body ends with lgkmcnt(0))
    assert len(vm) < 64


def test_weight_stream_replay(gen, body):
    """The LDS side of the same replay: a chunk slot may be read only when every LDS-DMA piece issued into it has (a) retired in the
    issuing wave's vmcnt order and (b) been followed by an s_barrier (every wave's pieces then have); and a piece may be issued into a
    slot only when an s_barrier separates it from the last read of the slot's previous content (all waves run this code in step with
    the barriers).  Slots are recovered from the scalar arithmetic in front of each `s_add_u32 m0` and of each base-register update."""
    lines, (_, _, fused) = body
    start = lines.index(".Lsteady%=:") + 1
    head, steady = lines[:start - 1], lines[start:]
    S = gen.SLOT_BYTES
    sg = {}                                   # scalar registers with known values

    def val(tok):
        tok = tok.strip()
        if tok.startswith("%"):
            return {"%4": 0, "%5": 3, "%6": 7, "%7": 256, "%8": 7, "%2": 0, "%3": 0}.get(tok)
        if re.fullmatch(r"-?\d+", tok):
            return int(tok)
        if re.fullmatch(r"0x[0-9a-f]+", tok):
            return int(tok, 16)
        return sg.get(tok)

    def salu(t):
        op, args = t.split(None, 1)
        a = [x.strip() for x in args.split(",")]
        if op in ("s_add_u32", "s_mul_i32", "s_lshl_b32") and len(a) == 3:
            x, y = val(a[1]), val(a[2])
            sg[a[0]] = None if x is None or y is None else ((x + y) if op == "s_add_u32" else (x * y) if op == "s_mul_i32" else (x << y)) & 0xffffffff
        elif op.startswith("s_") and a and re.fullmatch(r"s\d+|m0", a[0]):
            sg[a[0]] = None

    for t in head:                            # (the per-tile scalars: tile / wave bases)
        if t.startswith("s_") and " " in t and not t.startswith(("s_cmp", "s_cbranch", "s_cselect")):
            salu(t)
    base_slot, flight, barrier_no = {}, [], 0
    fills = {}
    uncertified = {s: [] for s in range(gen.NSLOT)}      # slot -> [retired?] flags of pieces not yet followed by a barrier
    last_read = {s: -1 for s in range(gen.NSLOT)}        # slot -> barrier count at its last read
    nread = 0
    for rep in range(3):
        for t in steady:
            op = t.split()[0]
            if op == "s_barrier":
                barrier_no += 1
                for s in uncertified:
                    uncertified[s] = [p for p in uncertified[s] if not p["retired"]]
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", t)
                if m:
                    n = int(m.group(1))
                    for p in flight[:len(flight) - n] if n else flight:
                        p["retired"] = True
                    flight = flight[len(flight) - n:] if n else []
            elif op == "global_load_lds_dwordx4":
                off = int(re.search(r"offset:(-?\d+)", t).group(1))
                assert sg.get("m0") is not None, t
                slot, rem = divmod(sg["m0"] + off, S)
                assert 0 <= slot < gen.NSLOT and rem + 1024 <= S, t
                assert last_read[slot] < barrier_no, ("LDS-DMA into a slot that was read since the last barrier", t)
                # source and destination belong together: (image offset of the piece) - (its offset in the slot) is the image offset
                # of ONE chunk for every piece of a fill, and this wave's pieces tile its share of the chunk exactly once
                assert sg.get("s84") is not None, t
                chunk_off = (sg["s84"] + off) - rem
                assert chunk_off in gen.IMG_OFF, ("a piece whose source and destination belong to different chunks", t)
                fills.setdefault((slot, chunk_off, barrier_no // 10**9), []).append(rem)
                p = {"retired": False}
                flight.append(p)
                uncertified[slot].append(p)
            elif op.startswith("buffer_load") or (fused and op.startswith("buffer_store")):
                flight.append({"retired": False})
            elif op == "ds_read_b128":
                b = t.split(",")[1].split()[0].strip()
                assert b in base_slot, t
                slot = base_slot[b]
                assert not uncertified[slot], ("read of a chunk slot with LDS-DMA pieces not yet certified by a barrier", t, barrier_no)
                last_read[slot] = barrier_no
                nread += 1
            elif op in ("v_add_u32", "v_add3_u32") and re.match(r"v24[0-4]\b", t.split(None, 1)[1]):
                a = [x.strip() for x in t.split(None, 1)[1].split(",")]
                src = [x for x in a[1:] if re.fullmatch(r"s\d+", x)]
                if src:
                    assert sg.get(src[0]) is not None and sg[src[0]] % S == 0, t
                    base_slot[a[0]] = sg[src[0]] // S
                else:
                    base_slot[a[0]] = next(base_slot[x] for x in a[1:] if x in base_slot)
            elif op.startswith("s_") and " " in t and not op.startswith(("s_cmp", "s_cbranch", "s_cselect", "s_nop", "s_memtime", "s_mov_b64")):
                salu(t)
    assert nread == 3 * (2410 + 308)                      # one A fragment per k-step, one bias tile per block
    for (slot, chunk_off, _), rems in fills.items():
        ci = gen.IMG_OFF.index(chunk_off)
        npw = gen.npw_of(ci)
        per_tile = sorted(rems)
        want = sorted([(3 * npw + i) * 1024 for i in range(npw)] * (len(rems) // npw))   # (the replay is wave 3: %5 = 3)
        assert len(rems) % npw == 0 and per_tile == want, ("a chunk's pieces do not tile this wave's share", slot, ci)

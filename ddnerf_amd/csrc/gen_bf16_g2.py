#!/usr/bin/env python3
"""Generator of the bf16 MLP kernel "two groups per weight pass" (mlp_bf16_g2.hip): the whole tile body as ONE block of gfx950
assembly with every register assigned here (`python gen_bf16_g2.py outdir`; ddnerf_amd/build.py runs it).

Why a second kernel: the one-group kernel (mlp_bf16.hip) moves the complete weight image L2 -> LDS once per 256 samples (357 LDS-DMA
pieces per wave and tile, 2.9 GB per launch at BASELINE size); on this power-limited part that stream is paid for in clock, not only
in issue slots.  Here a workgroup owns 512 samples: every wave two GROUPS of 64.  A layer's weights are staged into LDS once and used
by both groups, one after the other (pass g0, then pass g1): 389 pieces per wave and 512 samples, 54 % of the stream.

Why assembly: the plan needs the register file to the last register -- three activation files of 128 registers (F0, F1 in the
accumulator half, F2 in arch VGPRs: at layer l group 0 reads file in0 and writes the spare file, group 1 then reads in1 and writes
in0, dead by then; the roles rotate with period three), a 48-register transient, two accumulator tile sets, the fragment ring.  hipcc's
allocator at that pressure copies tuples around next to the MFMAs (wrong results: nothing pads VALU-write -> MFMA-read wait states
around inline asm) and spills.  With fixed registers there is no allocator; the generator also keeps the books of the memory counters
(every s_waitcnt below is the exact number of younger operations) and the hazard scan of the build (check_asm_hazards.py) reads the
result like any other kernel.

LDS: four slots of 36 KiB.  A 256-wide layer is four chunks of four 16-row slices; chunk j of layer l+1 replaces chunk j of layer l as
soon as group 1 is done with it (DMA issued in the next period, first read three periods later); the skip layer (K = 352, six chunks of
three slices) streams through the slots once per group.  One s_barrier per period (a chunk's blocks of one pass), preceded by
s_waitcnt vmcnt(loads issued during this period): everything issued in earlier periods has then landed, for every wave.

Encoded features (never kept): layer 0 reads them from where the previous tile's last pass parked them (group 0: the upper fragments
of F1, free while the 128-wide dir layer is written; group 1: the transient T); the skip layer's xyz columns and the view directions are
fetched again (L2 / MALL hits) into registers that are idle at that point, one or two passes ahead of their use.  The only fetch with
less lead is group 1's xyz block: T is busy with group 0's until the last k-steps of pass (5, g0), ten k-steps before group 1's first
block wants it (the k-steps of a block keep the one-group kernel's order: the two kernels are bit-identical)."""
import os
import struct
import sys
from collections import defaultdict

# ---- network -------------------------------------------------------------------------------------------------------
K = [96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128]      # packed layers (mlp_mfma16.inc: kK)
SLOT_BYTES = 36 * 1024
NSLOT = 4
DEPTH = 4        # A fragments are read this many k-steps ahead
# Diagnostic (-DBF16_STAMP) builds: the C++ shell stamps the tile loop of every workgroup (in-kernel clock, cycles per tile: what
# bench.py reports).
#
# EXPERIMENT SWITCHES.  They come from `--experiment key=value` on the generator's command line and from nowhere else (never the
# environment: a variable left over in a shell must not change the product kernel); `python gen_bf16_g2.py outdir` with no switch is
# the product body.  The switches in force are recorded in the generated tables (G2_GENERATOR_OPTIONS -> ddnerf_build_info()).
#   stamp_periods=1   (stamp builds) a stamp at the end of every period and per block inside the window STAMP_BLOCKS (tools/g2_clock.py
#                     reads them); each costs a scalar-memory round trip and an LDS drain
#   stamp_ksteps=p:b,...  (with stamp_periods) a stamp per k-step of block b of period p
#   sgpr_stamps=b0:b1 (stamp builds) ONE s_memtime per block boundary of blocks b0..b1 (at most 20 values) into SGPRs of their own,
#                     written out at the end of the tile: no wait, no store inside the window (8 bytes of code per stamp)
#   dummy_valu=v,t    v plain + t transcendental VALU instructions more per block on a register nothing reads
#   skip_fill=n / skip_every=m   every m-th block ends with a branch over n bytes of filler
#   head_pad=n        the whole body shifted by n bytes of skipped filler
#   align=n           the body's first instruction aligned to 2^n bytes (.p2align)
#   dma_span=a/b      the part of its issue period over which a chunk's LDS-DMA pieces are spread
#   touch=<file>      code-page touch: every block starts with an s_load_dword from the code address listed for it in <file> (one signed
#                     byte offset from the s_getpc at the body's top per block; `nop` = a filler of the same size) -- see tools/g2_touch.py
#   nodma=1 / nox=1   TIMING ONLY, results wrong: no LDS-DMA pieces / no feature fetches are issued
#   kmax=n            the re-pack of the previous block may use the gaps of the first n k-steps of an 8-k-step block (product: 7)
#   nobarrier=1       TIMING ONLY, results wrong (races): the period barriers are not issued -- what do the 82 barriers of a tile cost?
#   pf=1              the L2 prefetch touches of feature rows of rounds 3's body are issued again (a load of one word of each row of a group into
#                     a register nothing reads, one or two passes ahead of the group's feature fetch; results unchanged).  Round 4
#                     measured them as a net loss (profiles/r04_g2_experiments.md): the body without them runs 0.4 - 1.3 % faster in
#                     interleaved A/B, and a code-page crossing costs ~200 cycles instead of ~400 without them
#   pfpolicy=a_b      cache-policy bits of the L2 prefetch touches (default: none)
#   xsame=1           TIMING ONLY, results wrong: every tile fetches tile 0's feature rows (the fetches stay, their footprint goes)
#   touchglc=1        the code-page touches bypass the scalar cache (glc)
#   xpolicy=a_b       cache-policy bits of the feature fetches instead of `nt` (underscores for spaces; `none` = default policy)
#   enc_drain=1       (fused body) a full vmcnt drain between the encoder's last store and pass 9's first fetch of a stored row; the product relies on the
#                     memory pipeline keeping a wave's store and its later load of the same address in order
#   enc_store=0 / enc_valu=0 / enc_prologue=0   (fused body) TIMING ONLY, results wrong: the encoder's stores / its arithmetic / the first
#                     tile's straight-line encoder are not issued
#   enc_packed=1      (fused body) the column loop on the packed fp32 instructions (two columns per instruction; measured SLOWER: a
#                     v_pk_*_f32 holds the vector ALU about four times as long as the scalar instruction)
#   vm_inorder=0 / 1  stores are booked into the in-order queue of vmcnt like loads (a wait then allows the younger stores outstanding
#                     too) or not (every wait is then for "no younger LOAD outstanding": stricter than necessary while a store is in
#                     flight).  gfx9-family parts retire vector-memory loads AND stores of a wave in issue order (hipcc's own wait
#                     insertion books both as one event type on parts without a separate store counter); auto = 1 in the fused body
#                     (48 encoder stores per tile; measured 0.9 % of the launch), 0 in the unfused one (stores at the tile end only)
EXP_DEFAULT = dict(enc_packed="0", vm_inorder="auto", enc_drain="0", enc_store="1", enc_valu="1", enc_prologue="1", stamp_periods="0", stamp_ksteps="", sgpr_stamps="", dummy_valu="0,0", skip_fill="", skip_every="1", head_pad="0", align="0",
                   dma_span="1/1", touch="", nodma="0", nox="0", pf="0", xpolicy="nt", pfpolicy="none", xsame="0", touchglc="0", nobarrier="0", kmax="7")
EXP = dict(EXP_DEFAULT)


def set_experiment(opts):
    """install experiment switches (dict key -> string); unknown keys are an error"""
    global ENC_DRAIN, ENC_STORE, ENC_VALU, ENC_PROLOGUE, ENC_PACKED, VM_INORDER
    global STAMP_PERIODS, DUMMY_VALU, SKIP_FILL, SKIP_EVERY, HEAD_PAD, ALIGN, STAMP_KSTEPS, DMA_SPAN_NUM, DMA_SPAN_DEN, SGPR_STAMPS, TOUCH, NODMA, NOX, NOPF, XPOLICY, PFPOLICY, XSAME, TOUCHGLC, NOBARRIER, KMAX
    for k in opts:
        if k not in EXP_DEFAULT:
            raise SystemExit("gen_bf16_g2: unknown experiment switch %r (known: %s)" % (k, ", ".join(sorted(EXP_DEFAULT))))
    EXP.clear()
    EXP.update(EXP_DEFAULT)
    EXP.update(opts)
    STAMP_PERIODS = EXP["stamp_periods"] == "1"
    ENC_DRAIN, ENC_STORE, ENC_VALU, ENC_PROLOGUE = (EXP[k] == "1" for k in ("enc_drain", "enc_store", "enc_valu", "enc_prologue"))
    ENC_PACKED, VM_INORDER = EXP["enc_packed"] == "1", EXP["vm_inorder"]
    DUMMY_VALU = [int(x) for x in EXP["dummy_valu"].split(",")]
    SKIP_FILL = int(EXP["skip_fill"]) if EXP["skip_fill"] != "" else None
    SKIP_EVERY = int(EXP["skip_every"])
    HEAD_PAD = int(EXP["head_pad"])
    ALIGN = int(EXP["align"])
    STAMP_KSTEPS = [tuple(int(v) for v in x.split(":")) for x in EXP["stamp_ksteps"].split(",") if x]   # (period, block in period)
    DMA_SPAN_NUM, DMA_SPAN_DEN = [int(x) for x in EXP["dma_span"].split("/")]
    SGPR_STAMPS = tuple(int(v) for v in EXP["sgpr_stamps"].split(":")) if EXP["sgpr_stamps"] else None
    if SGPR_STAMPS:
        assert 0 <= SGPR_STAMPS[0] < SGPR_STAMPS[1] and SGPR_STAMPS[1] - SGPR_STAMPS[0] <= NSTAMP_SGPR - 1, "sgpr_stamps: at most %d blocks" % (NSTAMP_SGPR - 1)
    TOUCH = [x for x in open(EXP["touch"]).read().split()] if EXP["touch"] else None
    NODMA, NOX, NOPF = EXP["nodma"] == "1", EXP["nox"] == "1", EXP["pf"] != "1"
    XPOLICY = "" if EXP["xpolicy"] == "none" else " " + EXP["xpolicy"].replace("_", " ")
    PFPOLICY = "" if EXP["pfpolicy"] == "none" else " " + EXP["pfpolicy"].replace("_", " ")
    XSAME, TOUCHGLC = EXP["xsame"] == "1", EXP["touchglc"] == "1"
    NOBARRIER = EXP["nobarrier"] == "1"
    KMAX = int(EXP["kmax"])


def options_string():
    """the switches that differ from the product's, "key=value ..." ("" = product body)"""
    return " ".join("%s=%s" % (k, EXP[k]) for k in sorted(EXP) if EXP[k] != EXP_DEFAULT[k])


NSTAMP_SGPR = 20     # s[44:83]: 20 s_memtime values of the sgpr_stamps experiment
S_STAMP0 = 44
S_PC, S_JUNK = 98, 92    # touch experiment: s[98:99] = the address behind the body's s_getpc, s92 = where the touched word goes
STAMP_BLOCKS = (8, 20)
set_experiment({})
FEAT_ROW = 256   # bytes of one encoded sample (DDNERF_FEAT_LD bf16)
TILE = 512
# ---- the FUSED body (Gen(fused=True), mlp_bf16_g2e.hip): cast_rays + integrated_pos_enc inside the MLP kernel ------------------------------
# (reference stage: models/models.py:117-142 -- cast_rays, integrated_pos_enc, the view-direction columns and the MLP are ONE step there).
# The encoded feature rows never exist in HBM.  Inputs are the fenceposts t_vals [n, S+1] and a per-RAY table (128 B: o, d, radius^2,
# d^2, 1 - d^2/|d|^2 as fp32, then the ray's 32 view-direction columns as one k-order 16-bit row), S a multiple of 64 so that a group
# of 64 samples (one sample per lane) lies on one ray.  While layers 6 - 8 run -- T[c][1..2], 32 registers, are idle there -- the VALU
# gaps of the MFMA stream encode the NEXT tile's two groups, lane = sample: the Gaussian of the sample's interval (the arithmetic of
# rays_encode.hip's gaussian_of_interval / encode_kernel<1>, operation for operation: IEEE divisions, no contraction), then two columns
# at a time on the PACKED fp32 instructions (v_pk_mul / v_pk_fma / v_pk_add_f32: each half rounds like the scalar instruction) the
# damping 2^(-0.5 cov 4^d log2 e) and the sine and cosine on the hardware transcendentals -- the scaled means / covariances of a column
# pair live in a register pair that is multiplied by 4 / 16 from one pair of octaves to the next (exact) --, 16-bit pairs, and every
# four columns one 8-byte store into the workgroup's private scratch (96 KiB, written and re-read every tile: L2-resident), laid out
# [piece of four columns][sample] so that a store and the fetches are contiguous runs.
# Everything that used to fetch a feature row fetches a scratch row instead -- the next tile's layer-0 inputs during pass 9, the skip
# layer's xyz columns at layers 4 / 5 --; the view-direction k-step comes from the ray table.  Bit-identical to encode_kernel<1>
# followed by the unfused body (tests/test_hip_fused_mlp.py).
ENC_ROW = 192                                                        # bytes of one encoded sample
ENC_GROUP = 64 * ENC_ROW                                             # a group's area: [24 four-column pieces][64 samples][8 bytes]
V_EP = [128 + 12 * c + 4 + i for c in range(4) for i in range(8)]    # the encoder's 32 temporaries: T[c][1..2]
V_ET = {0: (254, 255), 1: (247, 253)}                                # fenceposts t0, t1 of the next tile's groups (fetched in pass (5, g1))
S_RAY = 48                                                           # s[48:63]: the ray-table row of the group being encoded
# equal-halved constant pairs of the packed fp32 instructions (s_mov'ed by the head of every tile)
S_C1, S_CT2, S_C2, S_HP, S_4, S_16 = 64, 66, 68, 70, 72, 74
S_E0, S_E1, S_E2, S_E3 = 80, 81, 82, 83
S_GB = 46                                                            # s46, s47: scratch byte offsets of this wave's two groups
S_ST = 76                                                            # soffset of the encoder's store being issued
S_GC = 93                                                            # group index of this tile's first group
# asm operands of the fused body: %0 scratch rows of this workgroup (buffer resource)  %1 outputs  %2 %3 weight image  %4 LDS base  %5 wave
#   %6 tile  %7 grid  %8 first tile  %9 t_vals (buffer resource)  %10 ray table (buffer resource)  %11 S + 1  %12 ceil(2^31 / (S / 64))
#   %13 n - 1  %14 S / 64  %15 ray table (64-bit address)  %16 the scratch rows as the encoder's STORES see them: operand %0, or -- in a
#   workgroup's last tile, whose "next tile" does not exist -- a zero-length copy of it (the bounds check drops the stores: 25 MB per
#   launch that nobody would read)


def lit(x):
    """a float as the 32-bit literal of a VOP2 / SOP1 instruction"""
    return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


def f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def rowb(k):
    return 2 * k + 32


def slice_bytes(k):
    return 16 * rowb(k) + 64


# chunks in consumption order of ONE group's walk through the network: lists of (layer, block)
CHUNKS = []
for l in range(9):
    if l == 5:
        for j in range(0, 16, 3):
            CHUNKS.append([(5, b) for b in range(j, min(16, j + 3))])
    else:
        for j in range(0, 16, 4):
            CHUNKS.append([(l, b) for b in range(j, j + 4)])
CHUNKS.append([(9, 0), (9, 1), (9, 2)])
CHUNKS.append([(9, 3), (9, 4), (9, 5)])
CHUNKS.append([(9, 6), (9, 7), (9, 8), (10, 0)])


def chunk_layout(ch):
    off, offs = 0, []
    for (l, b) in ch:
        offs.append(off)
        off += slice_bytes(K[l])
    size = (off + 4095) // 4096 * 4096
    assert size <= SLOT_BYTES, (ch, size)
    return offs, off, size


def npw_of(ci):
    return chunk_layout(CHUNKS[ci])[2] // 4096   # KiB pieces per wave


IMG_OFF, _off = [], 0
for ch in CHUNKS:
    IMG_OFF.append(_off)
    _off += chunk_layout(ch)[2]
IMG_BYTES = _off

# ---- passes and periods ----------------------------------------------------------------------------------------------
# a PASS = one group walking the chunks of one layer (layers 9 + 10 together: key 9); a PERIOD = one chunk of one pass
chunks_of_layer = {}
for ci, ch in enumerate(CHUNKS):
    chunks_of_layer.setdefault(min(l for l, _ in ch) if ch[0][0] != 10 else 9, []).append(ci)
PASSES = [(l, g) for l in range(10) for g in (0, 1)]
PERIODS = [(pi, ci) for pi, (l, g) in enumerate(PASSES) for ci in chunks_of_layer[l]]
NPER = len(PERIODS)

# instances: one load of a chunk; resident layers: one instance serves both groups' periods; layer 5: one per group
INST = []
for l in range(10):
    cis = chunks_of_layer[l]
    if l == 5:
        for g in (0, 1):
            for ci in cis:
                INST.append(dict(chunk=ci, uses=[PERIODS.index((PASSES.index((5, g)), ci))]))
    else:
        for ci in cis:
            INST.append(dict(chunk=ci, uses=[PERIODS.index((PASSES.index((l, 0)), ci)), PERIODS.index((PASSES.index((l, 1)), ci))]))
INST.sort(key=lambda d: d["uses"][0])
while len(INST) % NSLOT:   # (a dummy instance keeps the round-robin slot assignment the same in every tile)
    INST.append(dict(chunk=None, uses=[]))
for i, d in enumerate(INST):
    d["slot"] = i % NSLOT
real = [d for d in INST if d["chunk"] is not None]
for i, d in enumerate(INST):
    if d["chunk"] is None:
        continue
    j = i - NSLOT
    while INST[j % len(INST)]["chunk"] is None:
        j -= NSLOT
    prev = INST[j % len(INST)]
    free_after = prev["uses"][-1] - (NPER if j < 0 else 0)
    d["issue"] = free_after + 1          # the period after the slot's previous occupant was read for the last time (negative: previous tile)
    assert d["uses"][0] - d["issue"] >= 2, ("lead", i, d)
ISSUE_IN = {}
for d in real:
    ISSUE_IN.setdefault(d["issue"] % NPER, []).append(d)
    d["for_next_tile"] = d["issue"] < 0
for p, lst in ISSUE_IN.items():
    assert len(lst) == 1, ("two instances in one period", p)


def inst_of(period):
    for d in real:
        if period in d["uses"]:
            return d
    raise KeyError(period)


def files(l):
    """(input of group 0, input of group 1, spare) at layer l >= 1"""
    t = (0, 1, 2)
    for _ in range(l - 1):
        t = (t[2], t[0], t[1])
    return t


def out_file(l, g):
    if l == 0:
        return g           # layer 0: group 0 writes F0, group 1 F1
    a, b, c = files(l)
    return c if g == 0 else a


# ---- registers ----------------------------------------------------------------------------------------------------------
F_BASE = {0: ("a", 0), 1: ("a", 128), 2: ("v", 0)}
V_T, V_ACC, V_RING, V_BIAS, V_RP = 128, 176, 208, 224, 232
V_ABP, V_BBP, V_ABH, V_LANE16, V_VX, V_PF, V_LG16, V_VST, V_VST1, V_TMP0, V_TMP1, V_ZERO, V_STAMP = 240, 242, 244, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254
S_CUR, S_TB, S_TBN, S_RB, S_SWAVE, S_T0, S_T1, S_TIME, S_SEXEC = 84, 86, 87, 88, 89, 90, 91, 94, 96
S_CLOBBER = list(range(84, 98))
# asm operands: %0 feature rows (buffer resource)  %1 outputs (buffer resource)  %2 %3 weight image (address lo, hi)  %4 LDS base  %5 wave
#               %6 tile  %7 tiles per round (grid size)  %8 first tile of this workgroup  %9 stamps of this workgroup (diagnostic)


def reg(kind, r, n=1):
    return "%s%d" % (kind, r) if n == 1 else "%s[%d:%d]" % (kind, r, r + n - 1)


def frag(f, c, t):
    kind, base = F_BASE[f]
    return kind, base + 4 * (8 * c + t)


def treg(c, q):
    return "v", V_T + 4 * (3 * c + q)


def acc(par, c):
    return V_ACC + 16 * par + 4 * c


class Gen:
    def __init__(self, depth_head, stamp, carry=None, fused=False):
        self.depth_head, self.stamp, self.fused = depth_head, stamp, fused
        self.rs = 24 if depth_head else 16          # bytes of one output row
        self.out = []
        self.vm = list(carry["vm"]) if carry else []        # outstanding vector-memory loads: (serial, "dma" | "x")
        self.vm_serial = carry["serial"] if carry else 0
        self.pending = dict(carry["pending"]) if carry else {}   # register key -> serial of the load that fills it
        self.last_piece = dict(carry["last_piece"]) if carry else {}   # chunk load (index in `real`) -> serial of its last LDS-DMA piece
        self.lg = []                                        # outstanding LDS reads (keys)
        self.nmfma = 0
        self.enc_held, self.enc_stream = [], []             # fused body: encoder stores waiting for the first half of a period

    def e(self, s):
        self.out.append("\t" + s)

    def comment(self, s):
        self.out.append("; " + s)

    # ---- counters
    # vmcnt: vector-memory LOADS retire in order on this architecture -- loads into registers and LDS-DMA pieces alike (hipcc's own wait
    # insertion counts on exactly that for gfx9-family parts) -- while a store may retire at any time relative to them.  The queue below
    # therefore holds the loads only; a wait for a load allows as many operations outstanding as there are YOUNGER LOADS, which is exact
    # when no store is in flight and stricter than necessary when one is.
    def vm_issue(self, kind, key=None):
        self.vm_serial += 1
        if kind != "store" or VM_INORDER == "1" or (VM_INORDER == "auto" and self.fused):
            self.vm.append((self.vm_serial, kind))
        if key is not None:
            self.pending[key] = self.vm_serial
        return self.vm_serial

    def vm_wait_serial(self, serial, kind=None):
        """every load up to `serial` has completed"""
        if not self.vm or self.vm[0][0] > serial:
            return
        n = sum(1 for s_, _ in self.vm if s_ > serial)
        self.e("s_waitcnt vmcnt(%d)" % min(n, 63))
        cut = serial if n <= 63 else self.vm[len(self.vm) - 64][0]
        self.vm = [x for x in self.vm if x[0] > cut]
        self.pending = {k: s_ for k, s_ in self.pending.items() if s_ > cut}

    def vm_need(self, key):
        if key in self.pending:
            self.vm_wait_serial(self.pending[key], "x")
            self.pending.pop(key, None)

    def lg_issue(self, key):
        self.lg.append(key)

    def lg_need(self, key):
        if key in self.lg:
            n = len(self.lg) - 1 - self.lg.index(key)
            assert n <= 15
            self.e("s_waitcnt lgkmcnt(%d)" % n)
            self.lg = self.lg[len(self.lg) - n:] if n else []

    def lg_flush(self):
        self.e("s_waitcnt lgkmcnt(0)")
        self.lg = []

    # ---- building blocks
    def set_bases(self, par, rb, slot):
        self.e("s_add_u32 s%d, %%4, %d" % (S_T0, slot))
        self.e("v_add_u32 v%d, s%d, v%d" % (V_BBP + par, S_T0, V_LG16))
        self.e("v_bfe_u32 v%d, v%d, 4, 4" % (V_TMP0, V_LANE16))       # lane & 15
        self.e("v_mul_u32_u24 v%d, %d, v%d" % (V_TMP0, rb, V_TMP0))
        self.e("v_add_u32 v%d, v%d, v%d" % (V_ABP + par, V_TMP0, V_BBP + par))

    def set_base_h(self, rb, slot):
        self.e("s_add_u32 s%d, %%4, %d" % (S_T0, slot))
        self.e("v_bfe_u32 v%d, v%d, 4, 4" % (V_TMP0, V_LANE16))
        self.e("v_mul_u32_u24 v%d, %d, v%d" % (V_TMP0, rb, V_TMP0))
        self.e("v_add3_u32 v%d, v%d, v%d, s%d" % (V_ABH, V_TMP0, V_LG16, S_T0))

    def x_load(self, dst, nxt, g, c, q, key):
        """16 bytes of the feature row of sample (this / next tile, group g, column block c, this lane's row): 32-column group q.
        The row offset goes through the bounds-checked voffset: rows past the end read as zero (their outputs are never stored)."""
        kind, r = dst
        if self.fused:      # (the workgroup's scratch rows: one buffer, this tile's or the next one's according to WHEN the fetch runs)
            if q == 3:
                return self.dir_load(dst, g, key)
            # positions 8 lg .. 8 lg + 7 of 32-column group q = the four-column pieces 8 q + lg and 8 q + 4 + lg of sample 16 c + s
            self.e("s_add_u32 s%d, s%d, %d" % (S_T0, S_TB, g * ENC_GROUP + 8 * q * 512 + c * 128))
            self.e("v_add_u32 v%d, s%d, v%d" % (V_TMP1, S_T0, V_VX))
            self.e("buffer_load_dwordx2 %s, v%d, %%0, 0 offen sc1" % (reg(kind, r, 2), V_TMP1))
            self.vm_issue("x")
            self.e("buffer_load_dwordx2 %s, v%d, %%0, 0 offen offset:2048 sc1" % (reg(kind, r + 2, 2), V_TMP1))
            self.vm_issue("x", key)
            return
        self.e("s_add_u32 s%d, s%d, %d" % (S_T0, S_TBN if nxt else S_TB, (g * 64 + c * 16) * FEAT_ROW))
        self.e("v_add_u32 v%d, s%d, v%d" % (V_TMP1, S_T0, V_VX))
        if not NOX:
            self.e("buffer_load_dwordx4 %s, v%d, %%0, 0 offen offset:%d%s" % (reg(kind, r, 4), V_TMP1, 64 * q, XPOLICY))
        self.vm_issue("x", key)

    def ray_of_group(self, sreg, cur, g, emit=None):
        """sreg <- ray index of group g of this / the next tile (groups of 64 samples; S / 64 groups per ray), unclamped"""
        e = emit or self.e
        e("s_add_u32 s%d, s%d, %d" % (sreg, S_GC if cur else S_TBN, g))
        e("s_lshl_b32 s%d, s%d, 1" % (sreg, sreg))
        e("s_mul_hi_u32 s%d, s%d, %%12" % (sreg, sreg))

    def dir_load(self, dst, g, key):
        """the view-direction k-step of group g of THIS tile: 16 bytes of the ray's k-order row per lane (lane group lg: bytes 16 lg ..)"""
        kind, r = dst
        self.ray_of_group(S_T0, True, g)
        self.e("s_min_u32 s%d, s%d, %%13" % (S_T0, S_T0))
        self.e("s_lshl_b32 s%d, s%d, 7" % (S_T0, S_T0))
        self.e("buffer_load_dwordx4 %s, v%d, %%10, s%d offen offset:64" % (reg(kind, r, 4), V_LG16, S_T0))
        self.vm_issue("x", key)

    # ---- the encoder of the fused body
    def enc_fetch_ops(self, cur, g, what="rt"):
        """what a group's encoder reads: "r" its ray's table row -> s[S_RAY..+15], "t" its samples' fenceposts -> V_ET[g] (lane = sample)"""
        t0, t1 = V_ET[g]
        ops = []
        add = lambda s_: ops.append(("i", s_))
        self.ray_of_group(S_E1, cur, g, add)
        if "t" in what:
            add("s_add_u32 s%d, s%d, %d" % (S_E0, S_GC if cur else S_TBN, g))
            add("s_mul_i32 s%d, s%d, %%14" % (S_E2, S_E1))
            add("s_sub_u32 s%d, s%d, s%d" % (S_E2, S_E0, S_E2))
            add("s_lshl_b32 s%d, s%d, 6" % (S_E2, S_E2))                       # first sample of the group on its ray
            add("s_mul_i32 s%d, s%d, %%11" % (S_E3, S_E1))
            add("s_add_u32 s%d, s%d, s%d" % (S_E3, S_E3, S_E2))
            add("s_lshl_b32 s%d, s%d, 2" % (S_E3, S_E3))                       # byte offset of t_vals[ray][j0] (past the end: reads as zero)
            add("v_lshrrev_b32 v%d, 2, v%d" % (V_TMP1, V_LANE16))
            ops.append(("tload", "buffer_load_dword v%d, v%d, %%9, s%d offen" % (t0, V_TMP1, S_E3), ("t", g, 0)))
            ops.append(("tload", "buffer_load_dword v%d, v%d, %%9, s%d offen offset:4" % (t1, V_TMP1, S_E3), ("t", g, 1)))
        if "r" in what:
            add("s_min_u32 s%d, s%d, %%13" % (S_E1, S_E1))
            add("s_lshl_b32 s%d, s%d, 7" % (S_E1, S_E1))
            ops.append(("sload", "s_load_dwordx16 s[%d:%d], %%15, s%d" % (S_RAY, S_RAY + 15, S_E1)))
        return ops

    def enc_unit_ops(self, g, cur=False):
        """the encoder of one group of 64 samples (lane = sample) as a list of single instructions: ("i", text) plain, ("tneed", g)
        wait for the fenceposts, ("sflush",) wait for the ray row, ("store", text, pair) an 8-byte store of four columns, and -- in the
        stream of group 0, once its ray row has been consumed -- the fetch of group 1's row into the same scalar registers.
        Arithmetic: rays_encode.hip gaussian_of_interval (cone) and encode_kernel<1> phase 2, operation for operation."""
        t0, t1 = V_ET[g]
        R = S_RAY
        P = list(V_EP)
        ops = []
        I = lambda s_: ops.append(("i", s_))
        v = lambda r: "v%d" % r
        v2 = lambda r: "v[%d:%d]" % (r, r + 1)
        s2 = lambda r: "s[%d:%d]" % (r, r + 1)
        RR = [P[0], P[2], P[4]]          # scaled means of the three column pairs of an octave pair (register pairs)
        QQ = [P[6], P[8], P[10]]         # their scaled covariances times -0.5 log2 e
        U, F, YC, A = P[12], P[14], P[16], P[18]
        packs = [(P[20], P[22]), (P[24], P[26])]       # (sine piece, cosine piece) of a four-column step, two steps in flight
        vrow = P[28]
        spare = P[29:32]
        for x in RR + QQ + [U, F, YC, A] + [q for pr in packs for q in pr]:
            assert x % 2 == 0
        # the Gaussian's temporaries (all dead before the column loop starts)
        w = [U, U + 1, F, F + 1, YC, YC + 1, A, A + 1, P[20], P[21], P[22], P[23], P[24]]
        cov = [P[25], P[26], P[27]]

        def div(q, a, b, tmp):
            """q = a / b, correctly rounded (the sequence the compiler emits for an fp32 division with denormals on); tmp: 4 registers"""
            ds, ns, r, e = tmp
            I("v_div_scale_f32 %s, vcc, %s, %s, %s" % (v(ds), v(b), v(b), v(a)))
            I("v_rcp_f32 %s, %s" % (v(r), v(ds)))
            I("v_div_scale_f32 %s, vcc, %s, %s, %s" % (v(ns), v(a), v(b), v(a)))     # (the LAST writer of vcc in front of v_div_fmas)
            I("v_fma_f32 %s, -%s, %s, 1.0" % (v(e), v(ds), v(r)))
            I("v_fma_f32 %s, %s, %s, %s" % (v(r), v(e), v(r), v(r)))
            I("v_mul_f32 %s, %s, %s" % (v(q), v(ns), v(r)))
            I("v_fma_f32 %s, -%s, %s, %s" % (v(e), v(ds), v(q), v(ns)))
            I("v_fma_f32 %s, %s, %s, %s" % (v(q), v(e), v(r), v(q)))
            I("v_fma_f32 %s, -%s, %s, %s" % (v(e), v(ds), v(q), v(ns)))
            I("v_div_fmas_f32 %s, %s, %s, %s" % (v(q), v(e), v(r), v(q)))
            I("v_div_fixup_f32 %s, %s, %s, %s" % (v(q), v(q), v(b), v(a)))

        mu, hw, mu2, hw2, hw4, den, a_, q1, three = w[0:9]
        tmp = w[9:13]
        tm, tv, rv = spare                            # t_mean, t_var, r_var
        ops.append(("tneed", g))
        I("v_lshrrev_b32 %s, 1, v%d" % (v(vrow), V_LANE16))                          # lane * 8: the sample's slot inside a piece
        I("v_add_f32 %s, %s, %s" % (v(mu), v(t0), v(t1)))
        I("v_sub_f32 %s, %s, %s" % (v(hw), v(t1), v(t0)))
        I("v_mul_f32 %s, 0.5, %s" % (v(mu), v(mu)))                                  # (t0 + t1) / 2
        I("v_mul_f32 %s, 0.5, %s" % (v(hw), v(hw)))                                  # (t1 - t0) / 2
        I("v_mul_f32 %s, %s, %s" % (v(mu2), v(mu), v(mu)))
        I("v_mul_f32 %s, %s, %s" % (v(hw2), v(hw), v(hw)))
        I("v_mul_f32 %s, %s, %s" % (v(hw4), v(hw2), v(hw2)))
        I("v_mul_f32 %s, %s, %s" % (v(den), lit(3.0), v(mu2)))
        I("v_add_f32 %s, %s, %s" % (v(den), v(den), v(hw2)))                         # 3 mu^2 + hw^2
        I("v_add_f32 %s, %s, %s" % (v(a_), v(mu), v(mu)))
        I("v_mul_f32 %s, %s, %s" % (v(a_), v(a_), v(hw2)))                           # 2 mu hw^2
        div(q1, a_, den, tmp)
        I("v_add_f32 %s, %s, %s" % (v(tm), v(mu), v(q1)))                            # t_mean
        I("v_mov_b32 %s, %s" % (v(three), lit(3.0)))
        div(q1, hw2, three, tmp)                                                     # hw^2 / 3
        I("v_mul_f32 %s, %s, %s" % (v(a_), lit(12.0), v(mu2)))
        I("v_sub_f32 %s, %s, %s" % (v(a_), v(a_), v(hw2)))
        I("v_mul_f32 %s, %s, %s" % (v(a_), v(hw4), v(a_)))                           # hw^4 (12 mu^2 - hw^2)
        I("v_mul_f32 %s, %s, %s" % (v(three), v(den), v(den)))
        div(mu, a_, three, tmp)                                                      # (mu is dead: mu^2 carries on)
        I("v_mul_f32 %s, %s, %s" % (v(mu), lit(0.266666681), v(mu)))
        I("v_sub_f32 %s, %s, %s" % (v(tv), v(q1), v(mu)))                            # t_var
        I("v_mul_f32 %s, %s, %s" % (v(a_), lit(0.25), v(mu2)))                       # mu^2 / 4
        I("v_mul_f32 %s, %s, %s" % (v(q1), lit(0.416666657), v(hw2)))
        I("v_add_f32 %s, %s, %s" % (v(a_), v(a_), v(q1)))
        I("v_mul_f32 %s, %s, %s" % (v(q1), lit(0.266666681), v(hw4)))
        div(mu, q1, den, tmp)
        I("v_sub_f32 %s, %s, %s" % (v(a_), v(a_), v(mu)))
        ops.append(("sflush",))                                                      # the ray row (scalar memory returns out of order)
        I("v_mul_f32 %s, s%d, %s" % (v(rv), R + 6, v(a_)))                           # r_var = radius^2 (...)
        # cov_a = t_var d_a^2 + r_var (1 - d_a^2 / |d|^2);  mean_a = d_a t_mean + o_a   (two roundings each: no fma)
        for a in range(3):
            I("v_mul_f32 %s, s%d, %s" % (v(cov[a]), R + 7 + a, v(tv)))
            I("v_mul_f32 %s, s%d, %s" % (v(w[a]), R + 10 + a, v(rv)))
        for a in range(3):
            I("v_add_f32 %s, %s, %s" % (v(cov[a]), v(cov[a]), v(w[a])))
        for a in range(3):
            I("v_mul_f32 %s, s%d, %s" % (v(w[a]), R + 3 + a, v(tm)))
        # the column pairs of octaves (0, 1): (x, y) | (z, x') | (y', z'), ' = the second octave: means into RR, -0.5 log2 e cov into QQ
        m_dst = [RR[0], RR[0] + 1, RR[1]]
        for a in range(3):
            I("v_add_f32 %s, s%d, %s" % (v(m_dst[a]), R + a, v(w[a])))
        I("v_add_f32 %s, %s, %s" % (v(RR[1] + 1), v(m_dst[0]), v(m_dst[0])))
        I("v_add_f32 %s, %s, %s" % (v(RR[2]), v(m_dst[1]), v(m_dst[1])))
        I("v_add_f32 %s, %s, %s" % (v(RR[2] + 1), v(m_dst[2]), v(m_dst[2])))
        k0 = -0.5 * f32(1.44269502)
        for dst, a, k in ((QQ[0], 0, k0), (QQ[0] + 1, 1, k0), (QQ[1], 2, k0), (QQ[1] + 1, 0, 4 * k0), (QQ[2], 1, 4 * k0), (QQ[2] + 1, 2, 4 * k0)):
            I("v_mul_f32 %s, %s, %s" % (v(dst), lit(k), v(cov[a])))
        if g == 0:       # the ray row is consumed: group 1's may take its place
            ops += self.enc_fetch_ops(cur, 1, "r")
        if not ENC_PACKED:
            # 48 columns in order: column 3 d + a of the sine block, 48 + 3 d + a of the cosine block (octave d, axis a); the scaled
            # covariance -0.5 log2 e cov_a (QQ) times 4^d and the mean (RR) times 2^d are exact scalings
            mean = m_dst
            covk = [QQ[0], QQ[0] + 1, QQ[1]]
            y, u, ts, tc, yc, hs, hc = U, U + 1, F, F + 1, YC, A, A + 1
            for col in range(48):
                d, a = divmod(col, 3)
                if d == 0:
                    ysrc = v(mean[a])
                else:
                    I("v_mul_f32 %s, %s, %s" % (v(y), lit(float(1 << d)), v(mean[a])))
                    ysrc = v(y)
                if d == 0:
                    I("v_exp_f32 %s, %s" % (v(u), v(covk[a])))
                else:
                    I("v_mul_f32 %s, %s, %s" % (v(u), lit(float(4 ** d)), v(covk[a])))
                    I("v_exp_f32 %s, %s" % (v(u), v(u)))
                I("v_mul_f32 %s, %s, %s" % (v(ts), lit(0.0031830988), ysrc))
                I("v_add_f32 %s, %s, %s" % (v(yc), lit(1.57079637), ysrc))
                I("v_floor_f32 %s, %s" % (v(ts), v(ts)))
                I("v_mul_f32 %s, %s, %s" % (v(tc), lit(0.0031830988), v(yc)))
                I("v_fma_f32 %s, -%s, s%d, %s" % (v(ts), v(ts), S_CT2, ysrc))
                I("v_floor_f32 %s, %s" % (v(tc), v(tc)))
                I("v_mul_f32 %s, %s, %s" % (v(ts), lit(0.15915494), v(ts)))
                I("v_fma_f32 %s, -%s, s%d, %s" % (v(tc), v(tc), S_CT2, v(yc)))
                I("v_sin_f32 %s, %s" % (v(ts), v(ts)))
                I("v_mul_f32 %s, %s, %s" % (v(tc), lit(0.15915494), v(tc)))
                I("v_sin_f32 %s, %s" % (v(tc), v(tc)))
                ps, pc = packs[(col // 4) % 2]
                if col % 2 == 0:
                    I("v_mul_f32 %s, %s, %s" % (v(hs), v(u), v(ts)))
                    I("v_mul_f32 %s, %s, %s" % (v(hc), v(u), v(tc)))
                else:
                    half = (col % 4) // 2
                    I("v_mul_f32 %s, %s, %s" % (v(ts), v(u), v(ts)))
                    I("v_mul_f32 %s, %s, %s" % (v(tc), v(u), v(tc)))
                    if half == 0:
                        ops.append(("palloc", ps))
                        ops.append(("palloc", pc))
                    I("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(ps + half), v(hs), v(ts)))
                    I("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(pc + half), v(hc), v(tc)))
                    if half == 1:
                        m = col // 4
                        for pair, piece in ((ps, m), (pc, 12 + m)):
                            ops.append(("store", "s_add_u32 s%d, s%d, %d\n\tbuffer_store_dwordx2 v[%d:%d], %s, %%16, s%d offen"
                                        % (S_ST, S_GB + g, piece * 512, pair, pair + 1, v(vrow), S_ST), pair))
            return ops
        # 24 column pairs: pair i = columns (2 i, 2 i + 1) of the sine block and of the cosine block; i = 3 j + r: octave pair j, slot r
        for i in range(24):
            r = i % 3
            RRr, QQr = RR[r], QQ[r]
            ps, pc = packs[(i // 2) % 2]
            half = i % 2
            if half == 0:
                ops.append(("palloc", ps))
                ops.append(("palloc", pc))
            I("v_exp_f32 %s, %s" % (v(U), v(QQr)))
            I("v_exp_f32 %s, %s" % (v(U + 1), v(QQr + 1)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(F), v2(RRr), s2(S_C1)))
            I("v_pk_add_f32 %s, %s, %s" % (v2(YC), v2(RRr), s2(S_HP)))
            I("v_floor_f32 %s, %s" % (v(F), v(F)))
            I("v_floor_f32 %s, %s" % (v(F + 1), v(F + 1)))
            I("v_pk_fma_f32 %s, %s, %s, %s neg_lo:[1,0,0] neg_hi:[1,0,0]" % (v2(A), v2(F), s2(S_CT2), v2(RRr)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(F), v2(YC), s2(S_C1)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(A), v2(A), s2(S_C2)))
            I("v_floor_f32 %s, %s" % (v(F), v(F)))
            I("v_floor_f32 %s, %s" % (v(F + 1), v(F + 1)))
            I("v_sin_f32 %s, %s" % (v(A), v(A)))
            I("v_sin_f32 %s, %s" % (v(A + 1), v(A + 1)))
            I("v_pk_fma_f32 %s, %s, %s, %s neg_lo:[1,0,0] neg_hi:[1,0,0]" % (v2(YC), v2(F), s2(S_CT2), v2(YC)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(RRr), v2(RRr), s2(S_4)))              # the next octave pair's means
            I("v_pk_mul_f32 %s, %s, %s" % (v2(A), v2(U), v2(A)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(YC), v2(YC), s2(S_C2)))
            I("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(ps + half), v(A), v(A + 1)))
            I("v_sin_f32 %s, %s" % (v(YC), v(YC)))
            I("v_sin_f32 %s, %s" % (v(YC + 1), v(YC + 1)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(QQr), v2(QQr), s2(S_16)))
            I("v_pk_mul_f32 %s, %s, %s" % (v2(YC), v2(U), v2(YC)))
            I("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(pc + half), v(YC), v(YC + 1)))
            if half == 1:
                m = i // 2
                for pair, piece in ((ps, m), (pc, 12 + m)):
                    ops.append(("store", "s_add_u32 s%d, s%d, %d\n\tbuffer_store_dwordx2 v[%d:%d], %s, %%16, s%d offen"
                                % (S_ST, S_GB + g, piece * 512, pair, pair + 1, v(vrow), S_ST), pair))
        return ops

    def enc_group_base(self, g, emit=None):
        """s[S_GB + g] <- byte offset of group g's area in the workgroup's scratch (the soffset of the unit's stores)"""
        (emit or self.e)("s_add_u32 s%d, s%d, %d" % (S_GB + g, S_TB, g * ENC_GROUP))

    def enc_constants(self):
        for sreg, val in ((S_C1, 0.0031830988), (S_CT2, 314.159271), (S_C2, 0.15915494), (S_HP, 1.57079637), (S_4, 4.0), (S_16, 16.0)):
            self.e("s_mov_b32 s%d, %s" % (sreg, lit(val)))
            self.e("s_mov_b32 s%d, %s" % (sreg + 1, lit(val)))

    def emit_enc(self, op, early):
        """one instruction of the encoder's stream into the tile body, with the books of vmcnt / lgkmcnt kept"""
        if early:
            for h in self.enc_held:
                for ln in h[1].split("\n\t"):
                    self.e(ln)
                self.vm_issue("store")
            self.enc_held = []
        kind = op[0]
        if kind == "i":
            if ENC_VALU or not op[1].startswith("v_"):
                self.e(op[1])
        elif kind == "store":
            if not ENC_STORE:
                pass
            elif early:
                for ln in op[1].split("\n\t"):
                    self.e(ln)
                self.vm_issue("store")
            else:
                self.enc_held.append(op)
        elif kind == "palloc":
            assert all(h[2] != op[1] for h in self.enc_held), "a pack register pair is reused before its store was issued"
        elif kind == "tneed":
            self.vm_need(("t", op[1], 0))
            self.vm_need(("t", op[1], 1))
        elif kind == "sflush":
            self.lg_flush()
        elif kind == "sload":
            self.e(op[1])
        elif kind == "tload":
            self.e(op[1])
            self.vm_issue("x", op[2])
        else:
            raise KeyError(kind)

    def enc_group_base(self, g, emit=None):
        """s[S_GB + g] <- byte offset of group g's rows in the workgroup's scratch (the soffset of the unit's stores)"""
        (emit or self.e)("s_add_u32 s%d, s%d, %d" % (S_GB + g, S_TB, g * 64 * ENC_ROW))

    def prefetch(self, nxt, g, line):
        """one 128-byte line of each feature row of group g (this / next tile) into the L2 / MALL: every lane touches one row; the loaded
        word goes to a register nothing reads"""
        if NOPF:     # (nothing is issued, so nothing enters the books of vmcnt either)
            return
        self.e("s_add_u32 s%d, s%d, %d" % (S_T0, S_TBN if nxt else S_TB, g * 64 * FEAT_ROW))
        self.e("v_lshlrev_b32 v%d, 4, v%d" % (V_TMP1, V_LANE16))
        self.e("v_add_u32 v%d, s%d, v%d" % (V_TMP1, S_T0, V_TMP1))
        if not NOX:
            self.e("buffer_load_dword v%d, v%d, %%0, 0 offen offset:%d%s" % (V_PF, V_TMP1, 128 * line, PFPOLICY))
        self.vm_issue("x")

    def dma_setup(self, img, lds, npw):
        self.e("s_mul_i32 s%d, s%d, %d" % (S_T0, S_SWAVE, npw))
        self.e("s_add_u32 s%d, s%d, %d" % (S_T1, S_T0, img))
        self.e("s_add_u32 s%d, %%2, s%d" % (S_CUR, S_T1))
        self.e("s_addc_u32 s%d, %%3, 0" % (S_CUR + 1))
        self.e("s_add_u32 s%d, s%d, %d" % (S_T1, S_T0, lds))
        self.e("s_add_u32 m0, s%d, %%4" % S_T1)

    def dma_piece(self, imm, key=None, last=False):
        if not NODMA:
            self.e("global_load_lds_dwordx4 v%d, s[%d:%d] offset:%d" % (V_LANE16, S_CUR, S_CUR + 1, imm))
        serial = self.vm_issue("dma")
        if last:
            self.last_piece[key] = serial

    def stamp_pass(self, p):
        if not (self.stamp and STAMP_PERIODS):
            return
        self.e("s_memtime s[%d:%d]" % (S_TIME, S_TIME + 1))
        self.lg_flush()
        self.e("v_mov_b32 v%d, s%d" % (V_STAMP, S_TIME))
        self.e("v_mov_b32 v%d, s%d" % (V_STAMP + 1, S_TIME + 1))
        self.e("s_mov_b64 exec, s[%d:%d]" % (S_SEXEC, S_SEXEC + 1))
        self.e("global_store_dwordx2 v%d, v[%d:%d], %%9 offset:%d" % (V_ZERO, V_STAMP, V_STAMP + 1, 8 * p))
        self.vm_issue("store")
        self.e("s_mov_b64 exec, -1")

    def store_raw(self, g, parh, para):
        """outputs of group g: heads tile parh (rows 0-2 rgb on lane group 0, rows 4-5 mu / sigma on lane group 1), alpha = row 128 of the
        dir layer = its ninth block (tile para), register 0, lane group 0"""
        self.e("s_nop 7")
        self.e("s_nop 7")   # (16 wait states before anything but an MFMA reads an MFMA result)
        for c in range(4):
            self.e("v_mov_b32 v%d, v%d" % (acc(parh, c) + 3, acc(para, c)))
        self.e("s_mov_b64 exec, 0xffff")
        for c in range(4):
            self.e("s_add_u32 s%d, s%d, %d" % (S_T0, S_RB, (g * 64 + c * 16) * self.rs))
            self.e("v_add_u32 v%d, s%d, v%d" % (V_TMP0, S_T0, V_VST))
            self.e("buffer_store_dwordx4 %s, v%d, %%1, 0 offen" % (reg("v", acc(parh, c), 4), V_TMP0))
            self.vm_issue("store")
        if self.depth_head:
            self.e("s_mov_b64 exec, 0xffff0000")
            for c in range(4):
                self.e("s_add_u32 s%d, s%d, %d" % (S_T0, S_RB, (g * 64 + c * 16) * self.rs))
                self.e("v_add_u32 v%d, s%d, v%d" % (V_TMP0, S_T0, V_VST1))
                self.e("buffer_store_dwordx2 %s, v%d, %%1, 0 offen" % (reg("v", acc(parh, c), 2), V_TMP0))
                self.vm_issue("store")
        self.e("s_mov_b64 exec, -1")

    # ---- the tile body
    def build_blocks(self):
        blocks, k0 = [], 0
        for per, (pi, ci) in enumerate(PERIODS):
            lk, g = PASSES[pi]
            d = inst_of(per)
            offs = chunk_layout(CHUNKS[ci])[0]
            for bi, (l, b) in enumerate(CHUNKS[ci]):
                nks = K[l] // 32
                order = list(range(nks))
                # (the k-steps of a block run in the one-group kernel's order: fp32 accumulation order is part of the bit-exact contract)
                blocks.append(dict(l=l, b=b, g=g, lk=lk, period=per, K0=k0, order=order, lds=d["slot"] * SLOT_BYTES + offs[bi],
                                   first=bi == 0, last=bi == len(CHUNKS[ci]) - 1, outf=out_file(min(l, 9), g) if l < 10 else None))
                k0 += nks
        for i, blk in enumerate(blocks):
            blk["par"] = i & 1
        assert len(blocks) % 2 == 0
        return blocks, k0

    def bsrc(self, blk, ks, c):
        """(register kind, first register, key) of the B fragment of k-step ks, column block c"""
        l, g = blk["l"], blk["g"]
        if l == 0:
            if g == 0:
                return frag(1, c, 4 + ks) + (("F", 1, c, 4 + ks),)
            return treg(c, ks) + (("T", c, ks),)
        if l == 5 and ks >= 8:
            return treg(c, ks - 8) + (("T", c, ks - 8),)
        if l == 9 and ks == 8:
            if g == 0:
                return treg(c, 0) + (("T", c, 0),)
            return frag(0, c, 4) + (("F", 0, c, 4),)
        if l == 10:
            f = out_file(9, g)
        else:
            a, b_, _ = files(l)
            f = a if g == 0 else b_
        return frag(f, c, ks) + (("F", f, c, ks),)

    def repack_ops(self, p):
        """the re-pack of block p's four tiles as single instructions, step-major (a column's dependent steps are four gaps apart)"""
        l, b, par, outf = p["l"], p["b"], p["par"], p["outf"]
        if not (l < 9 or (l == 9 and b < 8)):
            return []
        relu = l != 8            # fc_feat has no activation
        kind, _ = F_BASE[outf]
        ops = []

        def dst(c, h):
            k, r = frag(outf, c, b // 2)
            return r + 2 * (b & 1) + h
        steps = []
        if kind == "a":
            steps.append(lambda c, h: "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (V_RP + 2 * c + h, acc(par, c) + 2 * h, acc(par, c) + 2 * h + 1))
            if relu:
                steps.append(lambda c, h: "v_pk_max_i16 v%d, v%d, 0" % (V_RP + 2 * c + h, V_RP + 2 * c + h))
            steps.append(lambda c, h: "v_accvgpr_write_b32 a%d, v%d" % (dst(c, h), V_RP + 2 * c + h))
        elif relu:
            steps.append(lambda c, h: "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (V_RP + 2 * c + h, acc(par, c) + 2 * h, acc(par, c) + 2 * h + 1))
            steps.append(lambda c, h: "v_pk_max_i16 v%d, v%d, 0" % (dst(c, h), V_RP + 2 * c + h))
        else:
            steps.append(lambda c, h: "v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (dst(c, h), acc(par, c) + 2 * h, acc(par, c) + 2 * h + 1))
        for st in steps:
            for h in range(2):
                for c in range(4):
                    ops.append(st(c, h))
        return ops

    def x_events(self, blk):
        """feature fetches hosted by this block: ((k-step position, gap), "x", dst, next tile?, group, column block, feature group, key) and
        ((position, gap), "pf", next tile?, group, line).  Every workgroup of the launch runs this schedule in step, so a fetch is a burst of
        the whole chip: each is issued a pass or more ahead of its use, and the one that cannot be (group 1's xyz block: its registers are
        busy until ten k-steps before) is preceded by a prefetch of its lines into the L2."""
        lk, g, l, b = blk["lk"], blk["g"], blk["l"], blk["b"]
        ev = []

        def x(pos, dst, nxt, gg, c, q, key):
            ev.append((pos, "x", dst, nxt, gg, c, q, key))
        if lk == 4 and g == 0 and b < 12:        # xyz of group 0 for the skip layer (T idles through layers 1-4)
            c, q = divmod(b, 3)
            x((1, 0), treg(c, q), 0, 0, c, q, ("T", c, q))
        if lk == 5 and g == 0 and b < 2 and not self.fused:         # (lines of group 1's rows -> L2, see above)
            ev.append(((1, 0), "pf", 0, 1, b))
        # xyz of group 1: fragment q of T is free once the LAST block of pass (5, g0) has issued its k-step 8 + q
        if lk == 5 and g == 0 and b == 15:
            for q in range(2):
                for c in range(4):
                    x((9 + q, c), treg(c, q), 0, 1, c, q, ("T", c, q))
        if lk == 5 and g == 1 and b == 0:
            for c in range(4):
                x((0, c), treg(c, 2), 0, 1, c, 2, ("T", c, 2))
        # the NEXT tile's layer-0 input of group 1 -> T: fragments 1, 2 are free from here on, fragment 0 carries group 0's view directions
        # through pass (9, g0)
        if lk in (6, 7) and g == 0 and b % 4 == 0 and not self.fused:
            j = (lk - 6) * 4 + b // 4
            c, q = j // 2, 1 + j % 2
            x((1, 0), treg(c, q), 1, 1, c, q, ("T", c, q))
        # (fused body: T[c][1..2] are the encoder's temporaries through layers 6 - 8; the rows it wrote are fetched in pass (9, g0))
        if self.fused and lk == 9 and g == 0 and l == 9 and b < 8:
            c, q = b // 2, 1 + b % 2
            x((2, 0), treg(c, q), 1, 1, c, q, ("T", c, q))
        if lk == 7 and g == 1 and b in (0, 8) and not self.fused:   # (lines of the next tile's group-0 rows -> L2 / MALL; group 1's were touched above)
            ev.append(((1, 0), "pf", 1, 0, b // 8))
        if self.fused and lk == 5 and g == 1 and b in (2, 6):      # the encoder's inputs of the next tile's groups (ray row, fenceposts)
            ev.append(((3, 0), "encfetch", (b - 2) // 4))
        if lk == 8 and g == 0 and b < 4:         # view directions of group 0 for layer 9
            x((1, 0), treg(b, 0), 0, 0, b, 3, ("T", b, 0))
        if lk == 9 and g == 0 and l == 9 and b < 4:   # view directions of group 1: fragment 4 of F0 (layer 9 writes fragments 0-3 only)
            x((1, 0), frag(0, b, 4), 0, 1, b, 3, ("F", 0, b, 4))
        # the next tile's layer-0 input of group 0 -> F1[c][4..6], as early in the last pass as F1 is free (loads retire in order: one
        # issued late in the tile would wait behind the LDS-DMA pieces of the next tile's first chunks, 2.5 - 3 k cycles each)
        if lk == 9 and g == 1 and l == 9 and b < 3:
            for j in range(4):
                c, q = divmod(4 * b + j, 3)
                x((1 + j, 0), frag(1, c, 4 + q), 1, 0, c, q, ("F", 1, c, 4 + q))
        if lk == 9 and g == 1 and l == 9 and 3 <= b < 7:
            x((1, 0), treg(b - 3, 0), 1, 1, b - 3, 0, ("T", b - 3, 0))
        return ev

    def tile(self):
        blocks, NK = self.build_blocks()
        kstep_blk = []
        for i, blk in enumerate(blocks):
            kstep_blk += [(i, ks) for ks in blk["order"]]
        per_ksteps = defaultdict(list)
        for n, (i, ks) in enumerate(kstep_blk):
            per_ksteps[blocks[i]["period"]].append(n)
        # DMA pieces: the instance issued in a period, spread evenly over the period's k-steps
        dma_at = defaultdict(list)
        # (the chunk whose turn is the one-block period that ends pass (5, g0) is issued a period later, behind group 1's xyz loads of
        # that block: loads retire in order, and behind nine fresh LDS-DMA pieces those loads cost group 1's first block ~2 k cycles)
        e2_period = next(x["period"] for x in blocks if x["lk"] == 5 and x["g"] == 0 and x["b"] == 15)
        eff_issue = {}
        for per in range(NPER):
            for d in ISSUE_IN.get(per, []):
                npw = npw_of(d["chunk"])
                shifted = per == e2_period
                eff_issue[id(d)] = per + 1 if shifted else per
                # (a group of pieces shares M0 and the source base with its setup: two chunks' pieces must not interleave, so the period
                # that takes the shifted chunk is split -- first half the shifted one, second half its own)
                if shifted:
                    ks_list = per_ksteps[per + 1][2:len(per_ksteps[per + 1]) // 2]
                elif per == e2_period + 1:
                    ks_list = per_ksteps[per][len(per_ksteps[per]) // 2:]
                else:
                    ks_list = per_ksteps[per]
                # (measured, bf16 fine pass: spread over the whole period 0.6726 of peak, over 3/4 0.6704, 1/2 0.6686, 1/3 0.6678 -- the
                # workgroups of a launch run in step, so a denser issue is a burst on the L2 of every XCD)
                span = len(ks_list) * DMA_SPAN_NUM // DMA_SPAN_DEN
                for i in range(npw):
                    dma_at[ks_list[(i * span) // npw]].append((d, i, npw))

        # When must a chunk have landed (for every wave: a wait for this wave's pieces, then a barrier)?  Its first fragments are read
        # AHEAD, in the last k-steps of the period before its first use q: by the barrier that ends period q - 2.  A chunk that this
        # would give too little time -- issued only two periods before q, or followed by a short period (an LDS-DMA piece takes 2.5 - 3 k
        # cycles to land when every CU streams; the one-block periods that end the skip layer's passes are 700) -- is "late": nothing of
        # it is read ahead (the reads wait in `deferred` and follow the barrier that ends period q - 1, which certifies it).
        # The wait at the end of a period is for exactly the chunks due there, not for everything issued so far.
        late, due = set(), defaultdict(list)
        for key, d in enumerate(real):
            eff = eff_issue[id(d)] if d["issue"] >= 0 else d["issue"]
            is_late = d["uses"][0] - eff < 3 or len(per_ksteps[(eff + 1) % NPER]) < 20
            if is_late:
                late.add(d["uses"][0])
            due[(d["uses"][0] - (1 if is_late else 2)) % NPER].append(key)
            assert (d["uses"][0] - (1 if is_late else 2)) >= eff, ("a chunk due before it is issued", key)
        state = dict(period=0, deferred=[])

        def rd_a(n, now=False):
            if n >= NK:
                return
            i, ks = kstep_blk[n]
            blk = blocks[i]
            if not now and blk["period"] in late and blk["period"] != state["period"]:
                state["deferred"].append(("a", n))
                return
            base = V_ABH if blk["l"] == 10 else V_ABP + (blk["period"] & 1)
            self.e("ds_read_b128 %s, v%d offset:%d" % (reg("v", V_RING + 4 * (n % DEPTH), 4), base, blk["lds"] % SLOT_BYTES + 64 * ks))
            self.lg_issue(("ring", n))

        def rd_bias(i, now=False):
            blk = blocks[i]
            if not now and blk["period"] in late and blk["period"] != state["period"]:
                state["deferred"].append(("bias", i))
                return
            self.e("ds_read_b128 %s, v%d offset:%d" % (reg("v", V_BIAS + 4 * blk["par"], 4), V_BBP + (blk["period"] & 1),
                                                       blk["lds"] % SLOT_BYTES + 16 * rowb(K[blk["l"]])))
            self.lg_issue(("bias", i))

        def centre(i, npw):
            first = i // 8 * 8
            return first + (4 if min(8, npw - first) > 4 else 0)

        # ---- tile begin: the bases of period 0, the first fragments, block 0's bias tile
        b0 = blocks[0]
        self.set_bases(0, rowb(K[b0["l"]]), b0["lds"] // SLOT_BYTES * SLOT_BYTES)
        for n in range(DEPTH - 1):
            rd_a(n)
        rd_bias(0)
        period_mark = self.vm_serial
        self.stamp_pass(0)
        for i, blk in enumerate(blocks):
            l, b, g, par, K0, order = blk["l"], blk["b"], blk["g"], blk["par"], blk["K0"], blk["order"]
            nks = len(order)
            kmax = nks if nks <= 4 else min(nks - 1 if KMAX == 7 else nks, KMAX)
            self.comment("---- block %d: layer %d block %d group %d (period %d)" % (i, l, b, g, blk["period"]))
            if TOUCH is not None:       # (experiment: 8 bytes per block either way, so that a second pass over the addresses changes no address)
                if TOUCH[i] == "nop":
                    self.e("s_nop 0")
                    self.e("s_nop 0")
                else:
                    self.e("s_load_dword s%d, s[%d:%d], 0x%x%s" % (S_JUNK, S_PC, S_PC + 1, int(TOUCH[i]), " glc" if TOUCHGLC else ""))
            if self.stamp and SGPR_STAMPS and SGPR_STAMPS[0] <= i <= SGPR_STAMPS[1]:
                j = i - SGPR_STAMPS[0]
                self.e("s_memtime s[%d:%d]" % (S_STAMP0 + 2 * j, S_STAMP0 + 2 * j + 1))
            gaps = defaultdict(list)
            if blk["first"]:
                period_mark = self.vm_serial
                assert state["period"] == blk["period"]
                nper = blk["period"] + 1
                if nper < NPER:
                    nb = next(x for x in blocks if x["period"] == nper)
                    gaps[(0, 0)].append(("bases", nper & 1, rowb(K[nb["l"]]), nb["lds"] // SLOT_BYTES * SLOT_BYTES))
                if any(x["l"] == 10 and x["period"] == blk["period"] for x in blocks):
                    gaps[(0, 0)].append(("base_h", rowb(K[10]), blk["lds"] // SLOT_BYTES * SLOT_BYTES))
            # fused body: the encoder of the next tile's groups rides in the gaps of layers 6 - 8 (96 blocks: 48 per group)
            in_period = [x for x in blocks if x["period"] == blk["period"]].index(blk)
            enc_early = True
            if self.fused and blk["lk"] in (6, 7, 8):
                wdx = (blk["lk"] - 6) * 32 + g * 16 + b
                ug, wi = divmod(wdx, 48)
                if wi == 0:
                    if ug == 0:
                        gaps[(0, 0)].append(("encbase",))
                    self.enc_stream = self.enc_unit_ops(ug)
                ne = len(self.enc_stream)
                mine = self.enc_stream[wi * ne // 48:(wi + 1) * ne // 48]
                for t, op in enumerate(mine):
                    gaps[divmod((t * nks * 4) // len(mine), 4)].append(("enc", op))
                # (stores count into vmcnt: none in the second half of a period, where the wait for the period's weight chunk would sit
                # behind them; the window's last period takes what is left)
                enc_early = in_period < 2 or wdx >= 92
            # re-pack of the previous block: one instruction per gap (gaps 1-3 of the first kmax k-steps)
            ops = self.repack_ops(blocks[i - 1]) if i > 0 else []
            # (experiment, G2_DUMMY_VALU="v,t": v plain + t transcendental VALU instructions more per block on a register nothing reads --
            # what an encoder inside this kernel would add to the MFMA gaps)
            for j in range(DUMMY_VALU[0] + DUMMY_VALU[1]):
                ops.insert((j * (len(ops) + 1)) // (DUMMY_VALU[0] + DUMMY_VALU[1]) + j if ops else j,
                           ("v_exp_f32 v%d, v%d" if j < DUMMY_VALU[1] else "v_mul_f32 v%d, 0x3fb8aa3b, v%d") % (V_PF, V_PF))
            slots = [(u, mi) for u in range(kmax) for mi in (1, 2, 3)]
            for t, op in enumerate(ops):
                gaps[slots[(t * len(slots)) // len(ops)]].append(("op", op))
            # weight stream
            for u in range(nks):
                for j, (d, pi_, npw) in enumerate(dma_at.get(K0 + u, [])):
                    assert j < 3
                    if pi_ % 8 == 0:
                        c0 = centre(pi_, npw)
                        gaps[(u, j)].append(("dma_setup", IMG_OFF[d["chunk"]] + c0 * 1024, d["slot"] * SLOT_BYTES + c0 * 1024, npw))
                    gaps[(u, j + 1)].append(("dma", (pi_ - centre(pi_, npw)) * 1024, real.index(d), pi_ == npw - 1))
            for it in self.x_events(blk):
                gaps[it[0]].append(it[1:])
            kst = (blk["period"], [x for x in blocks if x["period"] == blk["period"]].index(blk))
            for u, ks in enumerate(order):
                n = K0 + u
                if STAMP_PERIODS and kst in STAMP_KSTEPS:
                    self.stamp_pass(130 + 12 * STAMP_KSTEPS.index(kst) + u)
                for mi in range(4):
                    c = mi
                    kind, r, key = self.bsrc(blk, ks, c)
                    self.vm_need(key)
                    if mi == 0:
                        if u == 0:
                            self.lg_need(("bias", i))
                        self.lg_need(("ring", n))
                    a_ = reg("v", V_RING + 4 * (n % DEPTH), 4)
                    cc = reg("v", V_BIAS + 4 * par, 4) if u == 0 else reg("v", acc(par, c), 4)
                    self.e("v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s" % (reg("v", acc(par, c), 4), a_, reg(kind, r, 4), cc))
                    self.nmfma += 1
                    if mi == 0:
                        rd_a(n - 1 + DEPTH)
                        if u == 1 and i + 1 < len(blocks):
                            rd_bias(i + 1)
                    for it in gaps.get((u, mi), []):
                        if it[0] == "op":
                            self.e(it[1])
                        elif it[0] == "dma_setup":
                            self.dma_setup(*it[1:])
                        elif it[0] == "dma":
                            self.dma_piece(*it[1:])
                        elif it[0] == "x":
                            self.x_load(*it[1:])
                        elif it[0] == "pf":
                            self.prefetch(*it[1:])
                        elif it[0] == "bases":
                            self.set_bases(*it[1:])
                        elif it[0] == "base_h":
                            self.set_base_h(*it[1:])
                        elif it[0] == "enc":
                            self.emit_enc(it[1], enc_early)
                        elif it[0] == "encfetch":
                            for op in self.enc_fetch_ops(False, it[1], "rt" if it[1] == 0 else "t"):
                                self.emit_enc(op, True)
                        elif it[0] == "encbase":
                            self.enc_group_base(0)
                            self.enc_group_base(1)
            if SKIP_FILL is not None and i % SKIP_EVERY == SKIP_EVERY - 1:
                self.out.append("\ts_branch .Lskip%%=_%d" % i)
                if SKIP_FILL:
                    self.out.append("\t.fill %d, 4, 0xbf800000" % (SKIP_FILL // 4))      # (s_nop 0: never executed)
                self.out.append(".Lskip%%=_%d:" % i)
            if blk["last"]:
                serials = [self.last_piece[k] for k in due.get(blk["period"], []) if k in self.last_piece]
                if serials:
                    self.vm_wait_serial(max(serials))
                if self.fused and blk["lk"] == 8 and g == 1 and b == 15:
                    # every row the encoder stored is in the L2 before pass 9 fetches the first of them (a store retires at any time
                    # relative to the loads: only a full drain says so)
                    assert not self.enc_held, "encoder stores left over at the end of the window"
                    if ENC_DRAIN:
                        self.e("s_waitcnt vmcnt(0)")
                        self.vm, self.pending = [], {}
                if not NOBARRIER:
                    self.e("s_barrier")
                state["period"] = blk["period"] + 1
                for kind_, x in state["deferred"]:
                    (rd_a if kind_ == "a" else rd_bias)(x, True)
                state["deferred"] = []
            if l == 10:
                self.store_raw(g, par, blocks[i - 1]["par"])
            if STAMP_BLOCKS[0] <= blk["period"] < STAMP_BLOCKS[1]:
                self.stamp_pass(NPER + 1 + sum(len(CHUNKS[PERIODS[p_][1]]) for p_ in range(STAMP_BLOCKS[0], blk["period"])) + [x for x in blocks if x["period"] == blk["period"]].index(blk))
            if blk["last"]:
                self.stamp_pass(blk["period"] + 1)          # (diagnostic builds: the clock at the end of every period; slot 0: tile begin)
        self.lg_flush()
        if self.stamp and SGPR_STAMPS:      # (the window's block-boundary clocks leave the SGPRs here, behind the tile's last wait)
            self.e("s_mov_b64 exec, s[%d:%d]" % (S_SEXEC, S_SEXEC + 1))
            for j in range(SGPR_STAMPS[1] - SGPR_STAMPS[0] + 1):
                self.e("v_mov_b32 v%d, s%d" % (V_STAMP, S_STAMP0 + 2 * j))
                self.e("v_mov_b32 v%d, s%d" % (V_STAMP + 1, S_STAMP0 + 2 * j + 1))
                self.e("s_nop 0")
                self.e("global_store_dwordx2 v%d, v[%d:%d], %%9 offset:%d" % (V_ZERO, V_STAMP, V_STAMP + 1, 8 * (150 + j)))
                self.vm_issue("store")
            self.e("s_mov_b64 exec, -1")
        return blocks, NK

    def prologue(self):
        """first tile of a workgroup: lane constants, the chunks the steady state expects from "the previous tile", this tile's inputs"""
        e = self.e
        e("v_mbcnt_lo_u32_b32 v%d, -1, 0" % V_TMP0)
        e("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (V_TMP0, V_TMP0))
        e("v_lshlrev_b32 v%d, 4, v%d" % (V_LANE16, V_TMP0))
        e("v_and_b32 v%d, 15, v%d" % (V_TMP1, V_TMP0))
        e("v_lshrrev_b32 v%d, 4, v%d" % (V_LG16, V_TMP0))
        e("v_lshlrev_b32 v%d, 4, v%d" % (V_LG16, V_LG16))
        if self.fused:      # lane group lg reads the pieces lg, 4 + lg of a 32-column group: 512 bytes apart; sample s: 8 bytes apart
            e("v_lshlrev_b32 v%d, 3, v%d" % (V_VX, V_TMP1))
            e("v_lshlrev_b32 v%d, 5, v%d" % (V_TMP0, V_LG16))      # (the lane number in V_TMP0 has served; V_TMP1 = lane & 15 is still needed below)
            e("v_add_u32 v%d, v%d, v%d" % (V_VX, V_VX, V_TMP0))
        else:
            e("v_lshlrev_b32 v%d, 8, v%d" % (V_VX, V_TMP1))
            e("v_add_u32 v%d, v%d, v%d" % (V_VX, V_VX, V_LG16))
        e("v_mul_u32_u24 v%d, %d, v%d" % (V_VST, self.rs, V_TMP1))
        e("v_add_u32 v%d, 16, v%d" % (V_VST1, V_VST))
        e("v_mov_b32 v%d, 0" % V_ZERO)
        pro = sorted([d for d in real if d["for_next_tile"]], key=lambda d: d["issue"])
        if self.fused:      # the encoder's inputs of this tile's groups first: the wait for them leaves the LDS-DMA pieces behind them in flight
            for g in (0, 1):
                for op in self.enc_fetch_ops(True, g, "rt" if g == 0 else "t"):
                    e(op[1])
            self.pro_ndma = sum(npw_of(d["chunk"]) for d in pro)
        for d in pro:
            npw = npw_of(d["chunk"])
            for i in range(npw):
                c0 = i // 8 * 8 + (4 if min(8, npw - i // 8 * 8) > 4 else 0)
                if i % 8 == 0:
                    self.dma_setup(IMG_OFF[d["chunk"]] + c0 * 1024, d["slot"] * SLOT_BYTES + c0 * 1024, npw)
                    e("s_nop 0")
                e("global_load_lds_dwordx4 v%d, s[%d:%d] offset:%d" % (V_LANE16, S_CUR, S_CUR + 1, (i - c0) * 1024))
        if self.fused:      # the first tile's rows: both groups encoded here, nothing beside them (the weight chunks land meanwhile)
            self.enc_group_base(0)
            self.enc_group_base(1)
            for g in (0, 1):
                for op in self.enc_unit_ops(g, cur=True):
                    if op[0] == "sflush":
                        e("s_waitcnt lgkmcnt(0)")
                    elif op[0] == "tneed":
                        # issued so far, in this order: fenceposts of group 0 (2 loads), of group 1 (2), the LDS-DMA pieces; loads retire
                        # in order, so "at most this many in flight" says that the group's fenceposts have landed
                        e("s_waitcnt vmcnt(%d)" % min(63, self.pro_ndma + (2 if g == 0 else 0)))
                    elif op[0] == "sload" or (op[0] == "i" and (ENC_PROLOGUE or not op[1].startswith("v_"))) or (op[0] == "store" and ENC_PROLOGUE):
                        for ln in op[1].split("\n\t"):
                            e(ln.replace("%16", "%0"))       # (THIS tile's rows: always stored, also when the first tile is the last one)
            # (a wave's store and its later load of the same address stay in order in the memory pipeline: no wait between them)
        for c in range(4):
            for q in range(3):
                for (dst, g) in ((frag(1, c, 4 + q), 0), (treg(c, q), 1)):
                    if self.fused:
                        e("s_add_u32 s%d, s%d, %d" % (S_T0, S_TB, g * ENC_GROUP + 8 * q * 512 + c * 128))
                        e("v_add_u32 v%d, s%d, v%d" % (V_TMP1, S_T0, V_VX))
                        e("buffer_load_dwordx2 %s, v%d, %%0, 0 offen sc1" % (reg(dst[0], dst[1], 2), V_TMP1))
                        e("buffer_load_dwordx2 %s, v%d, %%0, 0 offen offset:2048 sc1" % (reg(dst[0], dst[1] + 2, 2), V_TMP1))
                        continue
                    e("s_add_u32 s%d, s%d, %d" % (S_T0, S_TB, (g * 64 + c * 16) * FEAT_ROW))
                    e("v_add_u32 v%d, s%d, v%d" % (V_TMP1, S_T0, V_VX))
                    e("buffer_load_dwordx4 %s, v%d, %%0, 0 offen offset:%d%s" % (reg(dst[0], dst[1], 4), V_TMP1, 64 * q, XPOLICY))
        e("s_waitcnt vmcnt(0)")
        e("s_barrier")

    def head(self):
        e = self.e
        if TOUCH is not None:
            e("s_getpc_b64 s[%d:%d]" % (S_PC, S_PC + 1))
        if ALIGN:
            self.out.append("\ts_branch .Lal%=")
            self.out.append("\t.p2align %d" % ALIGN)
            self.out.append(".Lal%=:")
        if HEAD_PAD:
            self.out.append("\ts_branch .Lpad%=")
            self.out.append("\t.fill %d, 4, 0xbf800000" % (HEAD_PAD // 4))
            self.out.append(".Lpad%=:")
        e("s_lshl_b32 s%d, %%5, 10" % S_SWAVE)
        if self.fused:
            e("s_mul_i32 s%d, %%5, %d" % (S_TB, 2 * ENC_GROUP))     # this wave's two group areas in the workgroup's scratch
            e("s_lshl_b32 s%d, %%6, 3" % S_GC)                        # groups of 64 samples: tile * 8 + wave * 2 + g
            e("s_lshl_b32 s%d, %%5, 1" % S_T0)
            e("s_add_u32 s%d, s%d, s%d" % (S_GC, S_GC, S_T0))
            e("s_lshl_b32 s%d, %%7, 3" % S_TBN)
            e("s_add_u32 s%d, s%d, s%d" % (S_TBN, S_TBN, S_GC))       # (S_TBN: the next tile's first group)
            self.enc_constants()
        elif XSAME:
            e("s_mov_b32 s%d, 0" % S_TB)
        else:
            e("s_lshl_b32 s%d, %%6, 17" % S_TB)           # tile * 512 rows * 256 bytes
        if not self.fused:
            e("s_lshl_b32 s%d, %%5, 15" % S_T0)           # wave * 128 rows
            e("s_add_u32 s%d, s%d, s%d" % (S_TB, S_TB, S_T0))
        if self.fused:
            pass
        elif XSAME:
            e("s_mov_b32 s%d, s%d" % (S_TBN, S_TB))
        else:
            e("s_lshl_b32 s%d, %%7, 17" % S_TBN)
            e("s_add_u32 s%d, s%d, s%d" % (S_TBN, S_TBN, S_TB))
        e("s_lshl_b32 s%d, %%6, 9" % S_RB)
        e("s_lshl_b32 s%d, %%5, 7" % S_T0)
        e("s_add_u32 s%d, s%d, s%d" % (S_RB, S_RB, S_T0))
        e("s_mul_i32 s%d, s%d, %d" % (S_RB, S_RB, self.rs))
        if self.stamp and (STAMP_PERIODS or SGPR_STAMPS):
            e("s_cmp_eq_u32 %5, 0")
            e("s_cselect_b64 s[%d:%d], 1, 0" % (S_SEXEC, S_SEXEC + 1))
        e("s_cmp_lg_u32 %6, %8")
        e("s_cbranch_scc1 .Lsteady%=")
        self.prologue()
        self.out.append(".Lsteady%=:")


def generate(depth_head, stamp, fused=False):
    g1 = Gen(depth_head, stamp, fused=fused)
    g1.tile()                                                  # pass 1: what a tile leaves outstanding for the next one
    carry = dict(vm=g1.vm, serial=g1.vm_serial, pending=g1.pending, last_piece=g1.last_piece)
    g2 = Gen(depth_head, stamp, carry, fused=fused)
    g2.head()
    blocks, NK = g2.tile()
    g3 = Gen(depth_head, stamp, dict(vm=g2.vm, serial=g2.vm_serial, pending=g2.pending, last_piece=g2.last_piece), fused=fused)   # (fixed point: pass 3 must repeat pass 2)
    g3.head()
    g3.tile()
    strip = lambda out: [x for x in out]
    assert [x for x in g3.out] == [x for x in g2.out], "the steady state of the counters is not a fixed point"
    return g2, blocks, NK


def tables(blocks):
    t = []
    slices = []
    for ci, ch in enumerate(CHUNKS):
        offs = chunk_layout(ch)[0]
        for (l, b), o in zip(ch, offs):
            slices.append((l, b, IMG_OFF[ci] + o))
    t.append("#define G2_IMG_BYTES %d" % IMG_BYTES)
    t.append("#define G2_NSLICE %d" % len(slices))
    t.append("static constexpr int kG2Slice[G2_NSLICE][3] = {%s};" % ", ".join("{%d, %d, %d}" % s for s in slices))
    sregs = list(S_CLOBBER) + (list(range(S_STAMP0, S_STAMP0 + 2 * NSTAMP_SGPR)) if SGPR_STAMPS else []) + ([S_PC, S_PC + 1] if TOUCH is not None else [])
    t.append("#define G2_GENERATOR_OPTIONS \"%s\"" % options_string())
    regs = ['"v%d"' % i for i in range(256)] + ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in sregs]
    t.append("#define G2_CLOBBERS %s, \"vcc\", \"scc\", \"memory\"" % ", ".join(regs))
    regs = ['"v%d"' % i for i in range(256)] + ['"a%d"' % i for i in range(256)] + ['"s%d"' % i for i in list(range(S_GB, S_E3 + 1)) + list(S_CLOBBER)]
    t.append("#define G2E_CLOBBERS %s, \"vcc\", \"scc\", \"memory\"" % ", ".join(regs))
    t.append("#define G2E_ROW_BYTES %d" % ENC_ROW)
    return t


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="generate the tile bodies of mlp_bf16_g2.hip")
    ap.add_argument("outdir")
    ap.add_argument("--experiment", action="append", default=[], metavar="key=value", help="experiment switch (see the header); none = the product body")
    args = ap.parse_args()
    set_experiment(dict(x.split("=", 1) for x in args.experiment))
    outdir = args.outdir
    nlines = 0
    for depth_head in (0, 1):
        for stamp in (0, 1):
            g, blocks, NK = generate(depth_head, stamp)
            text = "".join('"%s\\n"\n' % x for x in g.out)
            with open(os.path.join(outdir, "mlp_bf16_g2_body_d%d%s.gen.inc" % (depth_head, "s" if stamp else "")), "w") as f:
                # (one string literal per line: clang locates every line of an asm string by re-lexing its token from the start, which
                # on one megabyte-long literal takes minutes)
                f.write(text)
            if not stamp:
                # the fp16 tier (mlp_f16_g2.hip): the same body on v_mfma_f32_16x16x32_f16 with v_cvt_pk_f16_f32 re-packs -- same registers,
                # same LDS image, same schedule; the ReLU on the packed bit patterns (v_pk_max_i16) holds for any sign-magnitude format
                with open(os.path.join(outdir, "mlp_f16_g2_body_d%d.gen.inc" % depth_head), "w") as f:
                    f.write(text.replace("v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16").replace("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32"))
            nlines = len(g.out)
        # the fused body (encoder inside the kernel)
        g, blocks, NK = generate(depth_head, 0, fused=True)
        text = "".join('"%s\\n"\n' % x for x in g.out)
        with open(os.path.join(outdir, "mlp_bf16_g2e_body_d%d.gen.inc" % depth_head), "w") as f:
            f.write(text)
        # ... and its fp16 twin (mlp_f16_g2e.hip): the MFMA, the re-pack AND the encoder's packing on the f16 forms (since round 5 the fp16
        # rows take the bf16 rows' one-fma remainder: rays_encode.hip encode_kernel, KIND != 0)
        with open(os.path.join(outdir, "mlp_f16_g2e_body_d%d.gen.inc" % depth_head), "w") as f:
            f.write(text.replace("v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16").replace("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32"))
    with open(os.path.join(outdir, "mlp_bf16_g2_tables.gen.inc"), "w") as f:
        f.write("\n".join(tables(blocks)) + "\n")
    total_pieces = sum(npw_of(d["chunk"]) for d in real)
    print("gen_bf16_g2: %d blocks, %d k-steps (%d MFMAs), %d periods, %d chunk loads, %d pieces per wave per 512-sample tile (one-group kernel: 714), "
          "image %d bytes, %d lines per body" % (len(blocks), NK, g.nmfma, NPER, len(real), total_pieces, IMG_BYTES, nlines))

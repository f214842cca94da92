"""Model of the bf16 MLP kernel's static schedule (mirrors the constexpr plan in ddnerf_amd/csrc/mlp_bf16.hip):
blocks, stages, the LDS ring of 4 stage buffers and the constant-rate weight piece stream; checks the certification
windows.  Exploration tool, not part of the product."""
import sys
NL = 11
K = [96, 256, 256, 256, 256, 352, 256, 256, 256, 288, 128]
NBLK = [16, 16, 16, 16, 16, 16, 16, 16, 16, 9, 1]
# stages as lists of (layer, first block, nblocks)
def stages():
    st = []
    st += [[(0, 0, 6)], [(0, 6, 5)], [(0, 11, 5)]]
    for l in (1, 2, 3, 4):
        st += [[(l, 4 * i, 4)] for i in range(4)]
    st += [[(5, 0, 3)], [(5, 3, 3)], [(5, 6, 3)], [(5, 9, 3)], [(5, 12, 2)], [(5, 14, 2)]]
    for l in (6, 7, 8):
        st += [[(l, 4 * i, 4)] for i in range(4)]
    st += [[(9, 0, 3)], [(9, 3, 3)], [(9, 6, 3), (10, 0, 1)]]
    return st
def slice_bytes(k): return 16 * (2 * k + 32) + 64
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 5
PFD = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ST = stages()
NS = len(ST)
s_bytes = [sum(n * slice_bytes(K[l]) for l, f, n in s) for s in ST]
s_pieces = [(b + 1023) // 1024 for b in s_bytes]
s_npw = [(p + 3) // 4 for p in s_pieces]
s_ks = [sum(n * K[l] // 32 for l, f, n in s) for s in ST]
print("stages", NS, "max bytes", max(s_bytes), "total packed", sum(p * 1024 for p in s_pieces), "ksteps", sum(s_ks))
k0 = [0]
for x in s_ks: k0.append(k0[-1] + x)
NK = k0[-1]
stage_of = []
for s in range(NS): stage_of += [s] * s_ks[s]
usable = [n for n in range(NK) if k0[stage_of[n] + 1] - n > DEPTH + 1]
# per-wave piece list in stage order
pieces = []
for s in range(NS): pieces += [s] * s_npw[s]
NPW = len(pieces)
print("pieces per wave per tile", NPW, "items", 2 * NPW, "usable gaps", len(usable), "of", NK)
def check(offset, verbose=False):
    ok = True
    worst_lo, worst_hi = 99, -99
    for k in range(2 * NPW):
        g = usable[k * len(usable) // (2 * NPW)]
        S = stage_of[g]
        if k % 2 == 0:  # park of piece (k/2 - PFD + offset)
            i = (k // 2 - PFD + offset) % NPW
            T = pieces[i]
            d = (T - S) % NS
            # fraction position inside stage S
            if d not in (2, 3):
                ok = False
                if verbose: print("park piece", i, "of stage", T, "in stage", S, "d", d)
    return ok
good = [o for o in range(NPW) if check(o)]
print("valid offsets:", good[:5], "...", good[-5:] if good else None, len(good))
# diagnose: for each offset count violations
best = None
for o in range(NPW):
    bad = 0
    for k in range(0, 2 * NPW, 2):
        g = usable[k * len(usable) // (2 * NPW)]
        S = stage_of[g]
        T = pieces[(k // 2 - PFD + o) % NPW]
        if (T - S) % NS not in (2, 3): bad += 1
    if best is None or bad < best[0]: best = (bad, o)
print("best", best)
o = best[1]
for k in range(0, 2 * NPW, 2):
    g = usable[k * len(usable) // (2 * NPW)]
    S = stage_of[g]
    T = pieces[(k // 2 - PFD + o) % NPW]
    if (T - S) % NS not in (2, 3): print("  violation: park of stage", T, "piece in compute stage", S, "kstep", g, "stage ksteps", k0[S], k0[S+1])

"""Second model: the piece stream is assigned per compute stage (stage S parks the 2nd half of stage S+2 and the 1st half of
stage S+3), 4 LDS buffers.  Prints per-stage item / gap counts."""
import sys
sys.argv = sys.argv[:1] + sys.argv[1:]
exec(open(__file__.replace("bf16_plan2", "bf16_plan")).read().split("def check")[0])
start = [0]
for s in range(NS): start.append(start[-1] + s_npw[s])
mid = [start[s] + s_npw[s] // 2 for s in range(NS)]
worst = 0
for S in range(NS):
    a = mid[(S + 2) % NS]; b = mid[(S + 3) % NS]
    n = (b - a) % NPW
    U = [g for g in usable if stage_of[g] == S]
    worst = max(worst, 2 * n / len(U))
    print("stage %2d: ksteps %3d usable %3d  parks %2d items %2d  items/gap %.2f" % (S, s_ks[S], len(U), n, 2 * n, 2 * n / len(U)))
print("worst items/gap", worst)

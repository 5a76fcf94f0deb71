"""gap analysis of a rocprofv3 --kernel-trace sqlite database: per bench step (estdt launch to estdt launch) the sum of kernel
durations, the idle time between kernels, and the largest gaps with the kernels around them.  usage: trace_gaps.py <db> [step]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start").fetchall()
est = [i for i, r in enumerate(rows) if r[0].startswith("kk_estdt(")]
if len(est) < 2: est = [i for i, r in enumerate(rows) if r[0].startswith("kk_estdt_b")]      # (every level in boxes: the batched form only)
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(est) - 2
a, b = est[which], est[which + 1]
seg = rows[a:b]
wall = (seg[-1][2] - seg[0][1]) / 1e6
busy = sum(r[2] - r[1] for r in seg) / 1e6
print("step %d: %d kernels, wall %.3f ms, sum of kernel durations %.3f ms, idle %.3f ms" % (which, len(seg), wall, busy, wall - busy))
gaps = []
for i in range(1, len(seg)):
    g = seg[i][1] - seg[i - 1][2]
    gaps.append((g, i))
hist = collections.Counter()
for g, i in gaps:
    hist[min(int(max(g, 0) / 1000), 50)] += 1
print("gap histogram (us: count, total ms):", [(k, v, round(sum(g for g, _ in gaps if min(int(max(g, 0) / 1000), 50) == k) / 1e6, 3)) for k, v in sorted(hist.items())])
print("largest gaps:")
for g, i in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us after %-50s before %-50s" % (g / 1e3, seg[i - 1][0][:50], seg[i][0][:50]))
print("around the four largest gaps (five kernels before, five after):")
for g, i in sorted(gaps, reverse=True)[:4]:
    print("  gap %.1f us: ... %s  ||  %s ..." % (g / 1e3, " > ".join(r[0].split("(")[0][-40:] for r in seg[max(0, i - 5):i]), " > ".join(r[0].split("(")[0][-40:] for r in seg[i:i + 5])))
grp = collections.defaultdict(lambda: [0, 0.0])
for g, i in gaps:
    if g >= 50e3: k = (seg[i - 1][0].split("(")[0][:44], seg[i][0].split("(")[0][:44]); grp[k][0] += 1; grp[k][1] += g / 1e6
print("gaps of 50 us and more by the kernels around them (count, total ms):")
for k, v in sorted(grp.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-46s -> %-46s %4d  %8.3f" % (k[0], k[1], v[0], v[1]))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    key = r[0].split("(")[0][:60] + (" g=%dx%dx%d" % (r[3], r[4], r[5]) if len(sys.argv) > 3 else "")
    agg[key][0] += 1; agg[key][1] += (r[2] - r[1]) / 1e6
print("kernels of the step:")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("  %-90s %5d  %8.3f ms  %8.1f us avg" % (k, v[0], v[1], 1e3 * v[1] / v[0]))

"""per-step sums of the library's roctx ranges from a rocprofv3 --marker-trace csv:  python tools/probes/marker_steps.py <m_marker_api_trace.csv> [last N steps]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
adv = [r for r in rows if r["Function"] == "advance"]
names = ["advance_premac", "MAC_Project", "Scalar_update", "Velocity_update", "HG_Project", "xplan_build", "mlcc_build_sets", "cc_build", "nd_build", "mac_multigrid", "hg_multigrid"]
print("step   total   gap-before  " + "  ".join("%-12s" % n[:12] for n in names))
for i, a in enumerate(adv):
    if i < len(adv) - last:
        continue
    s, e = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
    inner = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        rs, re_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if rs >= s and re_ <= e and r is not a:
            inner[r["Function"]][0] += 1; inner[r["Function"]][1] += (re_ - rs) / 1e6
    gap = (s - int(adv[i - 1]["End_Timestamp"])) / 1e6 if i else 0.0
    print("%3d %8.1f %9.1f    " % (i, (e - s) / 1e6, gap) + "  ".join("%6.1f(%3d) " % (inner[n][1], inner[n][0]) for n in names))

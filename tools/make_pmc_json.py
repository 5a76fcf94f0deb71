"""profiles/r05_smoother_split_pmc.json (or, with a third argument `pair`, r05_smoother_rho_pmc.json: VDN_MAC_SPLIT=0 runs) from the summary tools/pmc_summary.py wrote for `tools/smoother_probe.py 256 20` (tools/final_profiles_r05.sh):
HBM bytes per launch of the roofline kernel = 2 x FETCH_SIZE (gfx950 reports half of a coalesced streaming read, MI355X_MICROARCH.md, checked on
the k_copy line of the same run) + WRITE_SIZE.  usage: make_pmc_json.py <smoother_pmc_summary.txt> <out.json>"""
import json, sys
lines = open(sys.argv[1]).read().splitlines()
hdr = lines[0].split()
names = hdr[3:]
def row(prefix):
    for ln in lines[1:]:
        if ln.startswith(prefix):
            f = ln.split()[-(len(names) + 2):]                      # (a kernel name may hold blanks)
            return {"grid": f[0], "calls": int(f[1]), **{n: float(v) for n, v in zip(names, f[2:])}}
    raise SystemExit("no line for " + prefix)
pair = len(sys.argv) > 3 and sys.argv[3] == "pair"
kname = "kk_cc_gsrb_rho_pair" if pair else "void kk_cc_gsrb_rho_split<0, false>"
k, c = row(kname), row("k_copy")
n = 256
out = {
    "_comment": "rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TA_BUSY_avr TA_BUSY_max / TCC_HIT_sum TCC_MISS_sum / VALUBusy MemUnitBusy, separate runs, "
                "--kernel-trace, csv; tools/final_profiles_r05.sh) of `python3 tools/smoother_probe.py 256 20` on MI355X, round 5; per-launch means over %d dispatches of "
                "the kernel at 256^3, the colour pass macproject runs on its finest level (kk_cc_gsrb_rho_split: the level stored by colour, round 5; "
                "kk_cc_gsrb_rho_pair: the interleaved level, VDN_MAC_SPLIT=0 and every multi-box run).  Sizes in KB.  FETCH_SIZE is doubled per MI355X_MICROARCH.md; "
                "the k_copy calibration of the same run (134217728 B read and written per launch) is alongside." % k["calls"],
    "kernel": kname.replace("void ", ""), "n": n,
    "fetch_size_kb_raw": k["FETCH_SIZE"], "write_size_kb": k["WRITE_SIZE"],
    "hbm_bytes_per_launch": int(round((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024)),
    "algorithmic_bytes_per_launch": 48 * n ** 3,
    "bytes_of_the_entries_the_pass_touches": (24 if not pair else 34) * n ** 3,
    "tcc_hit_per_launch": k.get("TCC_HIT_sum"), "tcc_miss_per_launch": k.get("TCC_MISS_sum"),
    "ta_busy_avr_cycles": k.get("TA_BUSY_avr"), "ta_busy_max_cycles": k.get("TA_BUSY_max"), "valu_busy_percent": k.get("VALUBusy"),
    "calibration_k_copy": {"fetch_size_kb_raw": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "bytes_read": 8 * n ** 3, "bytes_written": 8 * n ** 3},
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))

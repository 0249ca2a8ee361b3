"""Fold the passes of scripts/pmc_r2.sh into profiles/: the per-launch counter means of the fused correlation as
profiles/<tag>_corr_pmc_<config>.txt and the figures bench.py reports (HBM-side bytes with the guide's gfx950 correction,
VALU wave-instructions per launch) in profiles/corr_traffic.json.

    python scripts/pmc_to_json.py gpurun_out/<pmc dir> <config> <tag>
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def means(root, sub):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    d, config, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    m = means(d, "corr_fused")
    known = means(d, "nchw_to_nhwc")
    lines = ["# corr_fused2_kernel<24, 2>, %s workload: PMC counters, mean per dispatch; rocprofv3 --pmc, one pass per "
             "line group (scripts/pmc_r2.sh)" % config]
    for k in sorted(m):
        lines.append("%-36s n=%4d mean=%16.1f" % (k, m[k][1], m[k][0]))
    if "FETCH_SIZE" in known and "WRITE_SIZE" in known:
        lines.append("# check on a kernel of known byte count (planar -> channels-last of a whole ring): "
                     "2 x FETCH_SIZE x 1024 = %.0f B read, WRITE_SIZE x 1024 = %.0f B written"
                     % (2 * known["FETCH_SIZE"][0] * 1024, known["WRITE_SIZE"][0] * 1024))
    open(os.path.join(ROOT, "profiles", "%s_corr_pmc_%s.txt" % (tag, config)), "w").write("\n".join(lines) + "\n")
    path = os.path.join(ROOT, "profiles", "corr_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    ent = {"round": tag,
           "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        ent["read_bytes"] = 2.0 * m["FETCH_SIZE"][0] * 1024
        ent["write_bytes"] = m["WRITE_SIZE"][0] * 1024
        ent["hbm_bytes_per_launch"] = ent["read_bytes"] + ent["write_bytes"]
    for k_json, k_pmc in (("valu_insts_per_launch", "SQ_INSTS_VALU"), ("salu_insts_per_launch", "SQ_INSTS_SALU"),
                          ("mfma_insts_per_launch", "SQ_INSTS_MFMA"), ("valu_active_quadcycles", "SQ_ACTIVE_INST_VALU"),
                          ("waves", "SQ_WAVES"), ("l2_requests", "TCC_REQ_sum"), ("l2_hits", "TCC_HIT_sum"),
                          ("l1_accesses", "TCP_TOTAL_CACHE_ACCESSES_sum"), ("ta_busy_cycles_avg", "TA_BUSY_avr")):
        if k_pmc in m:
            ent[k_json] = m[k_pmc][0]
    doc[config] = ent
    json.dump(doc, open(path, "w"), indent=1)
    print("\n".join(lines))
    print(json.dumps(ent, indent=1))


if __name__ == "__main__":
    main()

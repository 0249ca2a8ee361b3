"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately: they do not fit one TCC pass)
into profiles/corr_traffic.json, the per-launch HBM traffic bench.py reports as roofline.traffic.

    python scripts/pmc_traffic.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <config> [round tag]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KB (x1024);
on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B per lane) reads -> doubled; WRITE_SIZE is exact for
16 B per lane stores.  The same correction is checked on a kernel of known byte count (cdv_fmap_to_nhwc over a whole
ring: reads and writes mem*C*H*W*2 bytes) whose dispatches are in the same passes."""
import collections, csv, glob, json, os, sys


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    d_fetch, d_write, config = sys.argv[1], sys.argv[2], sys.argv[3]
    tag = sys.argv[4] if len(sys.argv) > 4 else "r1"
    fe, wr = per_kernel(d_fetch, "FETCH_SIZE"), per_kernel(d_write, "WRITE_SIZE")
    out = {}
    for name in fe:
        short = ("corr_fused" if "corr_fused" in name else "nchw_to_nhwc" if "nchw_to_nhwc" in name else None)
        if short is None:
            continue
        f = sorted(fe[name])[len(fe[name]) // 2]
        w_list = wr.get(name, [0.0])
        w = sorted(w_list)[len(w_list) // 2]
        out[short] = {"kernel": name[:80], "dispatches": len(fe[name]), "FETCH_SIZE_KB_median": f,
                      "WRITE_SIZE_KB_median": w, "read_bytes": 2.0 * f * 1024, "write_bytes": w * 1024,
                      "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "corr_traffic.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    c = out.get("corr_fused")
    doc[config] = {"hbm_bytes_per_launch": c["hbm_bytes_per_launch"] if c else None, "round": tag,
                   "correction": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts 128-B "
                                 "requests as 64 B)", "kernels": out}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc[config], indent=1))


if __name__ == "__main__":
    main()

"""print the kernel timeline of the last full update step from a rocprofv3 --kernel-trace CSV (start/end in us
relative to the step's first kernel; queue id shows which HIP stream) -- to check overlap between streams"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the ingest kernel that is followed by at least 20 kernels
idx = [i for i, r in enumerate(rows) if "fmap_ingest" in r["Kernel_Name"]]
i0 = idx[-3] if len(idx) > 3 else idx[0]
i0 = max(0, i0 - 8)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 30]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:28]
    print("%-28s q=%s  %8.1f -> %8.1f  (%.1f us)" % (n, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))

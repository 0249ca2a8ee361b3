"""print a rocprofv3 kernel_stats.csv as a short table"""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:>9.1f} min_us={float(r['MinNs'])/1e3:>8.1f} pct={float(r['Percentage']):5.1f}")

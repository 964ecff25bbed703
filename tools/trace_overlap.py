"""Start / end of consecutive fused launches from a rocprofv3 kernel trace CSV: period, duration, overlap with the predecessor."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "mcep_mfma_fwd" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None; prev_start = None
out = []
for r in rows[10:30]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    out.append("start %9.1f us  dur %7.1f us  %s  %s" % (s / 1e3, (e - s) / 1e3, "period %.1f" % ((s - prev_start) / 1e3) if prev_start is not None else "", "overlap with predecessor %.1f us" % ((prev_end - s) / 1e3) if prev_end is not None else ""))
    prev_end, prev_start = e, s
print("\n".join(out))

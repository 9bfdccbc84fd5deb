"""Sum of kernel durations in a rocprofv3 kernel-trace CSV (ms), number of dispatches, and the span first start .. last end."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
tot = n = 0; lo = None; hi = 0
for row in csv.DictReader(open(path)):
    s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
    tot += e - s; n += 1; lo = s if lo is None else min(lo, s); hi = max(hi, e)
print(f"dispatches {n}  kernel time {tot / 1e6:.3f} ms  span {(hi - lo) / 1e6:.3f} ms")

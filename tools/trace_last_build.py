"""Per-launch durations of the LAST BLAS build in a rocprofv3 kernel trace csv (argv[1])."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_tri_prepare" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    name = r["Kernel_Name"]
    short = name.split("(")[0][-60:]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{dur:8.1f}  grid {r['Grid_Size_X']:>9}  wg {r['Workgroup_Size_X']:>5}  {short}")
print("sum of kernel time", tot, "us; span", (int(last[-1]["End_Timestamp"]) - t0) / 1e3)

"""Counts of HIP API calls in a rocprofv3 --hip-trace csv (argv[1]); argv[2:] = names to show (default: synchronising calls)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
c = collections.Counter(r["Function"] for r in rows)
show = sys.argv[2:] or ["hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyAsync", "hipMemsetAsync",
                        "hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel", "hipMalloc", "hipFree"]
for k in show:
    print(f"{k:28s} {c.get(k, 0)}")

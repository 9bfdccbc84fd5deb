"""hipcc -Rpass-analysis=kernel-resource-usage of one csrc/*.hip file as a table: kernel, VGPRs, AGPRs, scratch bytes, waves/SIMD.
usage: python tools/kernel_resources.py pipeline.hip [filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "low-precision-raytracer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.join(csrc, src), "-o", "/tmp/kr.o"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (?:Function Name: (\S+)|\s+([\w \[\]/]+?): (\d+))", line)
    if not m:
        continue
    if m.group(1):
        cur = m.group(1)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(2).strip()] = int(m.group(3))
for k, d in sorted(rows.items(), key=lambda kv: short(kv[0])):
    name = short(k) + (" " + "".join(re.findall(r"Lb([01])E", k.split("EEv")[0])) if "Lb" in k else "")
    if flt in name:
        print(f"{name:44s} VGPR {d.get('VGPRs', 0):3d} AGPR {d.get('AGPRs', 0):3d} scratch {d.get('ScratchSize [bytes/lane]', 0):4d} B  spills v{d.get('VGPRs Spill', 0)} s{d.get('SGPRs Spill', 0)}  waves/SIMD {d.get('Occupancy [waves/SIMD]', 0)}  LDS {d.get('LDS Size [bytes/block]', 0)}")

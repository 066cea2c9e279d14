"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
usage: python tools/kernel_resources.py fish_speech_amd/csrc/dualar_kernels.hip [name-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "Occupancy [waves/SIMD]": "occ", "VGPRs Spill": "spill", "LDS Size [bytes/block]": "lds",
        "ScratchSize [bytes/lane]": "scratch", "TotalSGPRs": "sgpr"}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, short in keys.items():
        m = re.search(r"    " + re.escape(k) + r": (\d+)", line)
        if m and cur is not None:
            cur[short] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, names):
    if flt in d:
        d = d.replace("void fmi::", "").replace("(fmi::LinearArgs)", "")
        print(f"{d[:110]:110s} vgpr {r.get('vgpr')} agpr {r.get('agpr')} sgpr {r.get('sgpr')} occ {r.get('occ')} lds {r.get('lds')} spill {r.get('spill')} scratch {r.get('scratch')}")

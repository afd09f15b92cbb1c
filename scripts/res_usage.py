"""Per-kernel registers / scratch / occupancy from a `hipcc -Rpass-analysis=kernel-resource-usage` log:
  python scripts/res_usage.py /tmp/resusage.txt [name filter regex]"""
import re
import sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else "."
K = {"VGPR": r"VGPRs", "AGPR": r"AGPRs", "scratch": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]", "SGPR": r"TotalSGPRs"}
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split(" ")[0]
    if not re.search(flt, name):
        continue
    vals = {k: (re.search(v + r": (\d+)", b).group(1) if re.search(v + r": (\d+)", b) else "?") for k, v in K.items()}
    print(f"{name[:64]:64s} " + " ".join(f"{k} {v}" for k, v in vals.items()))

"""Registers / scratch / LDS of every kernel in a built library, from the code-object metadata:
  python scripts/kernel_meta.py [path/to/lib.so] [name filter regex]
(private_segment_fixed_size = scratch bytes per lane; vgpr_spill_count = spilled registers)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_meta(lib):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        # the device code objects sit in the .hip_fatbin section as a clang offload bundle
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        # one bundle per translation unit, concatenated in the section
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(magic, blob)]
        notes = ""
        for i, st in enumerate(starts):
            part = os.path.join(td, f"fat{i}.bin")
            open(part, "wb").write(blob[st:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            co = os.path.join(td, f"gfx950_{i}.co")
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
            notes += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        out[name] = {k: g(k) for k in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count",
                                       "sgpr_spill_count", "group_segment_fixed_size")}
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "refnerf-pl_amd", "csrc", "librefnerf_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else "."
    for name, m in sorted(kernel_meta(lib).items()):
        if re.search(flt, name):
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"{short[:48]:48s} vgpr {m['vgpr_count']:>3} agpr {m['agpr_count']:>3} sgpr {m['sgpr_count']:>3} "
                  f"scratch {m['private_segment_fixed_size']:>4} B/lane  vgpr spills {m['vgpr_spill_count']:>3}")

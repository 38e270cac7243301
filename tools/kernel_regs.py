#!/usr/bin/env python
"""Register / spill / LDS figures per kernel from the metadata of a `hipcc -S --cuda-device-only` listing.
usage: python tools/kernel_regs.py file.s [name-filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
    name = f.get("name", "?")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    if flt in name:
        print("%-60s vgpr %3s agpr %3s vspill %3s sspill %3s lds %6s" % (name[:60], f.get("vgpr_count"), f.get("agpr_count"),
              f.get("vgpr_spill_count"), f.get("sgpr_spill_count"), f.get("group_segment_fixed_size")))

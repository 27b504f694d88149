#!/usr/bin/env python3
"""Register / spill / scratch metadata of the kernels in libdust_amd.so whose mangled name matches a pattern.
   python tools/kernel_regs.py [pattern]"""
import os, re, shutil, subprocess, sys, tempfile

llvm = "/opt/rocm/lib/llvm/bin"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
with tempfile.TemporaryDirectory() as d:
    shutil.copy(os.path.join(root, "dust_amd", "libdust_amd.so"), os.path.join(d, "l.so"))
    subprocess.run([llvm + "/llvm-objdump", "--offloading", "l.so"], cwd=d, check=True, capture_output=True)
    notes = "".join(subprocess.run([llvm + "/llvm-readelf", "--notes", f], cwd=d, check=True, capture_output=True, text=True).stdout
                    for f in os.listdir(d) if "gfx950" in f)
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, re.S):
    if pat.search(m.group(1)):
        blk = m.group(2)
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        print("%-90s vgpr %3s spill %3s sgpr %3s sspill %3s scratch %4s lds %6s" % (m.group(1)[:90], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"),
              g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))

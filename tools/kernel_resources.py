#!/usr/bin/env python3
"""Register and LDS use of every kernel as the COMPILER reports it (the code object's metadata), which is what occupancy
follows: `.vgpr_count` is the unified count — architectural VGPRs plus the accumulation registers the compiler spills
into — where rocprofv3's kernel trace shows the architectural ones only.

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt      (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "3bz_amd", "csrc", "tbz_amd.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "tbz.s")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                           "--cuda-device-only", "-Wno-unused-value", src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
recs = []
for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", text, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
    recs.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                 g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
print("%-28s %6s %6s %6s %7s %7s %8s %8s %6s  %s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "lds_B", "scratch", "wg", "waves/SIMD by registers"))
for r in sorted(recs):
    v = int(r[1]) if r[1].isdigit() else 0
    occ = min(8, 512 // max(8, (v + 7) // 8 * 8)) if v else "?"
    print("%-28s %6s %6s %6s %7s %7s %8s %8s %6s  %s" % (r + (occ,)))

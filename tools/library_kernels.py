#!/usr/bin/env python3
"""Every kernel that libmatinv_hip.so ships, with its register budget: python tools/library_kernels.py [path/to/libmatinv_hip.so]
(llvm-objdump --offloading on a copy of the library in a temporary directory, llvm-readelf --notes on the gfx950 code objects.)
regs = VGPRs + AGPRs of one lane (unified file: 512 / waves per SIMD), agpr = the AGPR part, scratch in bytes per lane."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-matrix-inversion_amd", "libmatinv_hip.so")
with tempfile.TemporaryDirectory() as tmp:
    lib = shutil.copy(so, os.path.join(tmp, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", lib], capture_output=True, text=True, cwd=tmp)
    rows = []
    for f in sorted(glob.glob(os.path.join(tmp, "*gfx950"))):
        t = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
        for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                             r"\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", t, re.S):
            rows.append((m.group(2), int(m.group(5)), int(m.group(1)), int(m.group(3)), int(m.group(6))))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
out = []
for r, nm in zip(rows, names):
    nm = re.sub(r"\(.*", "", nm).replace("void matinv::", "")
    out.append(f"{nm:66s} regs {r[1]:4d}  agpr {r[2]:3d}  scratch {r[3]:5d}  spilled {r[4]:4d}")
print(f"# {len(out)} kernels in {os.path.basename(so)} ({os.path.getsize(so)} bytes)")
print("\n".join(sorted(out)))

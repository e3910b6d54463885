#!/usr/bin/env python3
"""Scan the ISA of every kernel under learning3d_amd/csrc for loads that are waited for at once: a `global_load` whose next
instruction (NOPs and further loads aside) is `s_waitcnt vmcnt(0)`.  That is what a load under a per-lane or wave-uniform condition,
or in a rolled `load; use` loop, compiles to -- and a prefetch built from such loads waits for itself (LABLOG R4.11, R4.12).
usage: tools/isa_audit.py [threshold]   (needs hipcc; no GPU)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "learning3d_amd", "csrc")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tmp = tempfile.mkdtemp(prefix="isa_")
procs = []
for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
    out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
    procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "-o", out,
                                         "--cuda-device-only", f"-I{CSRC}", f], stderr=subprocess.DEVNULL)))
rows = []
for out, p in procs:
    p.wait()
    if not os.path.exists(out):
        continue
    txt = open(out).read()
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, flags=re.S | re.M):
        body = [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
        n = 0
        for i, l in enumerate(body):
            if l.startswith(("global_load", "buffer_load")):
                for j in range(i + 1, min(i + 3, len(body))):
                    if body[j].startswith("s_waitcnt vmcnt(0)"):
                        n += 1
                        break
                    if not body[j].startswith(("global_load", "s_nop")):
                        break
        if n >= thr:
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            rows.append((n, os.path.basename(out), dem[:90]))
for r in sorted(rows, reverse=True):
    print(f"{r[0]:4d}  {r[1]:22s} {r[2]}")

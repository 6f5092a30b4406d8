#!/usr/bin/env python3
"""Dev helper: registers / scratch / LDS of every render_kernel instantiation, from the ISA hipcc emits for gfx950
(the same compile scripts/check_isa.py does).  A non-zero scratch size = spilled registers.
    python scripts/kernel_resources.py [--all]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from check_isa import DEFAULT_FLAGS, SRC
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + DEFAULT_FLAGS.split() + ["--cuda-device-only", "-S", "-o", out, SRC])
    txt = open(out).read()
names = {"COUNT": 0, "PILOT": 1, "CTR": 2, "SMALL": 3, "MFMA": 4, "DBG": 5}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    sym, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    if "render_kernel" not in dem and "--all" not in sys.argv:
        continue
    get = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    # the directive values are granulated; the comment block after the kernel has the exact numbers
    tail = txt[m.end():m.end() + 3000]
    ex = lambda k: (re.search(r";\s*" + k + r":\s*(\d+)", tail) or [None, "?"])[1]
    print(f"{dem[:95]:95s} vgpr {ex('NumVgprs'):>3s} agpr {ex('NumAgprs'):>2s} sgpr {ex('NumSgprs'):>3s} scratch {ex('ScratchSize'):>4s} occupancy {ex('Occupancy')}")

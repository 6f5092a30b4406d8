#!/usr/bin/env python3
"""Build-time check of the hand-issued scalar loads in mrt::render_kernel.

The discriminant sweep issues `s_load_dwordx16` in inline asm and waits for them one group
later.  hipcc does not know the destinations are in flight between those two statements, so
a spill (v_writelane), copy (s_mov) or reuse of those SGPRs in between would read or clobber
data that has not landed.  This script compiles kernels.hip to ISA, rebuilds the control-flow
graph of every render_kernel instantiation and verifies that on every path from a hand-issued
load to the first `s_waitcnt ... lgkmcnt(0)` (ours or hipcc's: either lands all outstanding
scalar loads) no instruction reads or writes the load's destination SGPRs.  It also pins two compiler
accidents found in round 4 (no flat memory instructions; the large-scene walk's queue scheduling on the
scalar side): see check_kernel.  Exit code 0 = verified.

    python scripts/check_isa.py [--flags "<hipcc flags>"]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "myraytracer_amd", "csrc", "kernels.hip")
DEFAULT_FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-vectorize -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"


def sregs(text):
    """All SGPR indices mentioned in an operand string."""
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bs(\d+)\b", text):
        out.add(int(a))
    return out


def check_kernel(name, lines):
    """Forward reachability from every hand-issued s_load_dwordx16 to the first lgkmcnt(0) wait on each
    path (any such wait -- ours or hipcc's -- lands every outstanding scalar load): nothing on the way may
    read or write the load's destination SGPRs."""
    n = len(lines)
    label_at = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = k

    def instr(k):
        return lines[k].split(";")[0].strip()

    def successors(k):
        l = instr(k)
        parts = l.split()
        if parts and parts[0] == "s_branch":
            return [label_at[parts[1]]]
        if parts and parts[0].startswith("s_cbranch"):
            return [label_at[parts[1]], k + 1]
        if parts and parts[0] == "s_endpgm":
            return []
        return [k + 1] if k + 1 < n else []

    errors, loads = [], 0
    for k0 in range(n):
        m = re.search(r"^\s*s_load_dwordx16\s+(s\[\d+:\d+\])", lines[k0])
        if not m:
            continue
        # only hand-issued loads (inside an asm statement); hipcc's own are tracked by its waitcnt pass
        j = k0
        while j >= 0 and "#ASMSTART" not in lines[j] and "#ASMEND" not in lines[j]:
            j -= 1
        if j < 0 or "#ASMSTART" not in lines[j]:
            continue
        loads += 1
        dest = sregs(m.group(1))
        seen, stack = set(), list(successors(k0))
        while stack:
            k = stack.pop()
            if k in seen or k >= n:
                continue
            seen.add(k)
            l = instr(k)
            if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l:
                continue                                   # everything has landed on this path
            if l and not l.endswith(":") and not l.startswith((".", "s_load_dwordx16")):
                ops = l.split(None, 1)[1] if " " in l or "\t" in l else ""
                if sregs(ops) & dest:
                    errors.append(f"{name}: `{l}` (line {k}) touches {m.group(1)} while its load from line {k0} "
                                  f"may still be in flight")
            elif l.startswith("s_load_dwordx16"):
                ops = l.split(None, 1)[1]
                if sregs(ops.split(",", 1)[1]) & dest:
                    errors.append(f"{name}: `{l}` (line {k}) reads {m.group(1)} while in flight")
            stack.extend(successors(k))
    # render_kernel<COUNT, PILOT, CTR, SC, MFMA, DBG>: the matrix-core variant of the sweep (fifth argument
    # true, mangled render_kernelILb?ELb?ELb?ELi?ELb1ELb?EEEv...) has no hand-issued scalar loads; every other one must have them
    targs = re.search(r"render_kernelI((?:L[bi][0-9]+E)+)E", name)
    flags = re.findall(r"L[bi]([0-9]+)E", targs.group(1)) if targs else []
    mfma_variant = len(flags) >= 5 and flags[4] == "1"
    if loads == 0 and not mfma_variant:
        errors.append(f"{name}: no hand-issued s_load_dwordx16 found (sweep not recognised)")
    if mfma_variant and not any("v_mfma_f32_32x32x16_bf16" in l for l in lines):
        errors.append(f"{name}: matrix-core sweep variant without v_mfma_f32_32x32x16_bf16")
    # Two compiler accidents of round 4, pinned (DESIGN_HISTORY.md): (1) no flat memory instruction anywhere -- two sources of
    # one value (LDS / global copies of the boxes) as generic pointers were merged into a flat load of a selected pointer,
    # which goes down the texture path whatever it reads; (2) large scenes (fourth argument 1 / 2): which queue the walk's
    # next round takes is decided on the scalar side -- a reference to one of the queue counters had put the counters, and
    # with them the scheduling, into VGPRs (`v_cmp_lt_u32 vcc, 63, v..` in every round)
    flat = [instr(k) for k in range(n) if re.match(r"flat_(load|store|atomic)", instr(k))]
    if flat:
        errors.append(f"{name}: {len(flat)} flat memory instruction(s), first `{flat[0]}`")
    large = len(flags) >= 4 and flags[3] in ("1", "2")
    if large and any(re.match(r"v_cmp_\w+_u32\w*\s+(vcc|s\[\d+:\d+\]),\s*6[34],\s*v\d+", instr(k)) for k in range(n)):
        errors.append(f"{name}: a queue counter is compared with 63 / 64 on the vector side (the walk's scheduling left the SGPRs)")
    return errors


def main():
    flags = DEFAULT_FLAGS
    if "--flags" in sys.argv:
        flags = sys.argv[sys.argv.index("--flags") + 1]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags.split(), "--cuda-device-only", "-S", "-o", out, SRC],
                              stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    errors, checked = [], 0
    i = 0
    while i < len(text):
        m = re.match(r"^(_ZN3mrt\S*render_kernel\S*):", text[i])
        if m:
            j = i
            while "s_endpgm" not in text[j]:
                j += 1
            errors += check_kernel(m.group(1), text[i:j + 1])
            checked += 1
            i = j
        i += 1
    if checked == 0:
        errors.append("no render_kernel instantiation found in the ISA")
    for e in errors:
        print("check_isa:", e)
    print(f"check_isa: {checked} render_kernel instantiation(s) checked, {len(errors)} problem(s)")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""ISA check of the "operand fragment consumed while it arrives" protocol (poweflownet_amd/csrc/seg_tile.hpp, seg_load_a_async):
the 17 loads of a fragment are inline asm, hidden from hipcc, and their destination registers hold garbage until the hand-placed
`s_waitcnt vmcnt(16 - m)` of chunk m.  Nothing may read or write those registers in between -- a copy, a spill, a register handed
to something else.  hipcc does not know that, so this script reads the ISA it produced:

    python tools/check_async_fragments.py            # checks the assembly of the objects csrc/Makefile links (csrc/isa/*.s)
    python tools/check_async_fragments.py file.s ... # checks assembly that is already there (the Makefile calls it this way
                                                     # on every build of ea_seg.o / seg_lin_hops.o: a problem fails the build)

For every kernel with async fragment loads: each group of 17 loads must be followed by at least one complete wait sequence
(vmcnt 16, 15, .. 0: one per exclusive multiply block), and for every such sequence no instruction between a load and the wait of
its chunk may name a destination register of that load -- scanning the text from the load to the wait but skipping the OTHER
sequences' blocks (they are exclusive branches).  Exit code 1 and a list of offending lines otherwise."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "poweflownet_amd", "csrc")
SOURCES = ("ea_seg.hip", "seg_lin_hops.hip")
NCH = 17


def regs_in(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        out.add(int(a))
    return out


def check_function(name, lines):
    """-> (number of load groups, list of problems)"""
    asm_line = lambda i: i > 0 and lines[i - 1].strip().startswith(";;#ASMSTART")
    loads = []
    for i, l in enumerate(lines):
        m = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off( offset:\S+)?$", l)
        if m and asm_line(i):
            loads.append((i, int(m.group(1)), int(m.group(2))))
    if not loads:
        return 0, []
    problems = []
    if len(loads) % NCH:
        return 0, [f"{name}: {len(loads)} async loads (not a multiple of {NCH})"]
    waits = [(i, int(re.search(r"vmcnt\((\d+)\)", l).group(1))) for i, l in enumerate(lines)
             if re.match(r"\s*s_waitcnt vmcnt\(\d+\)$", l) and asm_line(i)]
    label = lambda j: lines[j].startswith(".LBB") or lines[j].startswith("; %bb.")
    for g in range(0, len(loads), NCH):
        grp = loads[g:g + NCH]
        last = grp[-1][0]
        nxt = loads[g + NCH][0] if g + NCH < len(loads) else len(lines)
        seqs = []
        for idx, (w, c) in enumerate(waits):
            if last < w < nxt and c == NCH - 1:
                seq = waits[idx:idx + NCH]
                if [c2 for _, c2 in seq] != list(range(NCH - 1, -1, -1)):
                    problems.append(f"{name}: broken wait sequence at line {w + 1}")
                    continue
                start = next(j for j in range(w, 0, -1) if label(j))
                end = next((j for j in range(seq[-1][0], len(lines)) if label(j)), len(lines))
                seqs.append((seq, start, end))
        if not seqs:
            problems.append(f"{name}: no wait sequence behind the loads at line {grp[0][0] + 1}")
        for seq, start, end in seqs:
            others = [(s2, e2) for q2, s2, e2 in seqs if q2 is not seq]
            for k, (li, a, b) in enumerate(grp):
                for j in range(li + 1, seq[k][0]):
                    if any(s2 <= j < e2 for s2, e2 in others):
                        continue
                    t = lines[j].strip()
                    if not t or t[0] in ";." or t.startswith("s_"):
                        continue
                    if regs_in(t) & set(range(a, b + 1)):
                        problems.append(f"{name}: line {j + 1}: `{t}` names v[{a}:{b}] of the load at line {li + 1} "
                                        f"before its wait at line {seq[k][0] + 1}")
    return len(loads) // NCH, problems


def check_text(text):
    groups, problems = 0, []
    cur, name = [], None
    for l in text.split("\n"):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, cur = m.group(1), []
        if name is not None:
            cur.append(l)
            if l.strip() == "s_endpgm":
                n, p = check_function(name, cur)
                groups += n
                problems += p
                name = None
    return groups, problems


def linked_assembly():
    """The device assembly of the objects that are LINKED into libpfn_hip.so: csrc/Makefile compiles the async sources with
    -save-temps=obj (its own HIPCC / CXXFLAGS / ARCH, whatever they are set to) and keeps the .s under csrc/isa/."""
    objs = [s.replace(".hip", ".o") for s in SOURCES]
    asm = lambda s: glob.glob(os.path.join(CSRC, "isa", s.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-*.s"))
    subprocess.run(["make", "-s", "-C", CSRC] + objs, check=True, stdout=subprocess.DEVNULL)
    if not all(asm(s) for s in SOURCES):   # objects that arrived without their temporaries (a copied tree): rebuild those two
        subprocess.run(["make", "-s", "-B", "-C", CSRC] + objs, check=True, stdout=subprocess.DEVNULL)
    out = []
    for s in SOURCES:
        hits = asm(s)
        if len(hits) != 1:
            raise SystemExit(f"{s}: expected one device assembly file under csrc/isa/, found {hits}")
        out += hits
    return out


def main(argv):
    files = argv[1:] or linked_assembly()
    total, problems = 0, []
    for f in files:
        n, p = check_text(open(f).read())
        print(f"{os.path.basename(f)}: {n} async fragment(s) checked, {len(p)} problem(s)")
        total += n
        problems += p
    for p in problems:
        print("  " + p)
    if not problems and total == 0:
        print("no async fragment loads found: the protocol is not in use (or the pattern changed)")
        return 1
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))

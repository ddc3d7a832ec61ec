#!/usr/bin/env python3
"""Static screen of a compiled kernel's ISA for the hazard of asynchronous register loads (k_conv4r, csrc/sgo_conv4r.hpp):
between the issue of a `global_load_dwordx4` and the counted `s_waitcnt vmcnt(N)` that retires it, no instruction may read or
write its destination registers (the compiler believes the value is there from the asm statement on; the hardware delivers it
later).  Walks the loop body's text twice (wrap-around), replays the vmcnt queue (loads and LDS-DMA in issue order), and reports
every instruction that touches an in-flight destination.  Conditional wait chains laid out as cold blocks are not followed:
pass their vmcnt values to ignore them where they appear inline.

usage: hipcc ... -save-temps=obj -c sgo_conv.hip; extract one kernel's text into k.s;
       vmcnt_isa_check.py k.s <first line of the loop body> <last line> [vmcnt values to ignore ...]
round 3: k_conv4r<true / false, 1> (the shipped variant) and <true, 17>: 0 violations."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])          # loop body line range (1-based, inclusive)
skipwaits = set(int(a) for a in sys.argv[4:])          # vmcnt values to ignore (conditional chains)
body = lines[lo - 1:hi]
def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out
queue = []   # list of (set of regs or None for lds dma, line)
bad = 0
for rep in range(2):
    for i, l in enumerate(body):
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", s)
        if m:
            n = int(m.group(1))
            if n in skipwaits:
                continue
            while len(queue) > n:
                queue.pop(0)
            continue
        if "s_waitcnt" in s and "vmcnt" in s:
            m = re.search(r"vmcnt\((\d+)\)", s)
            n = int(m.group(1))
            while len(queue) > n:
                queue.pop(0)
            continue
        touched = regs(s.split(";")[0])
        for (rs, ln) in queue:
            if rs and touched & rs:
                print("rep", rep, "line", lo + i, s[:80], "touches in-flight", sorted(touched & rs)[:4], "loaded at", ln)
                bad += 1
        if s.startswith("global_load_dwordx4"):
            dst = regs(s.split(",")[0])
            queue.append((dst, lo + i))
        elif s.startswith("global_load_lds"):
            queue.append((None, lo + i))
print("violations", bad, "queue at end", len(queue))

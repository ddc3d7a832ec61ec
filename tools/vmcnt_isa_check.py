#!/usr/bin/env python3
"""Static screen of a compiled kernel's ISA for the hazard of asynchronous register loads (k_conv4r, csrc/sgo_conv4r.hpp):
between the issue of a `global_load_dwordx4` and the counted `s_waitcnt vmcnt(N)` that retires it, no instruction may read or
write its destination registers (the compiler believes the value is there from the asm statement on; the hardware delivers it
later -- a load that lands in registers the compiler has meanwhile given to an address corrupts it; the one timing ablation
that allowed this ended in a GPU memory access fault).  The checker walks the K loop's main flow twice (wrap-around), replays the
wave's vmcnt queue (register loads and LDS-DMA in issue order; a wait leaves the N youngest in flight) and reports every
instruction that touches an in-flight destination.  The conditional wait chains of the chunk boundaries are cold blocks behind the
main flow and are not followed (the main flow's own waits are the weaker ones, so this errs on the strict side).

usage: vmcnt_isa_check.py [device asm of sgo_conv.hip | nothing = compile it]   (hipcc --cuda-device-only -S)
round 3: k_conv4r<true / false, 1> (the shipped variant): 0 violations."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(extra_flags=()):
    """Device assembly of csrc/sgo_conv.hip for gfx950 (a few seconds); None without hipcc."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        return None
    out = os.path.join(tempfile.mkdtemp(prefix="sgo_isa_"), "sgo_conv.s")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"),
           "--cuda-device-only", "-S", "-o", out, os.path.join(ROOT, "sejonggo_amd", "csrc", "sgo_conv.hip")] + list(extra_flags)
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_text(asm, mangled_prefix):
    """Lines of the first kernel whose symbol starts with `mangled_prefix` (label to .end_amdhsa_kernel)."""
    lines = asm.split("\n")
    for i, l in enumerate(lines):
        if l.startswith(mangled_prefix) and l.split(":")[0].endswith("ii") and ":" in l and not l.startswith("\t"):
            for j in range(i, len(lines)):
                if ".end_amdhsa_kernel" in lines[j]:
                    return lines[i:j]
    return None


def _regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def k_loop(lines):
    """(first, last) line index of the K loop's main flow: from the depth-1 loop header that contains the MFMAs to the first
    branch back to the latch blocks laid out just before the header."""
    for h, l in enumerate(lines):
        if "Loop Header: Depth=1" in l and sum("v_mfma" in x for x in lines[h:h + 400]) > 16:
            latches = [x.split(":")[0] for x in lines[max(0, h - 14):h] if x.startswith(".LBB")]
            for e in range(h + 1, len(lines)):
                s = lines[e].strip()
                if s.startswith(("s_cbranch", "s_branch")) and s.split()[-1] in latches:
                    return h, e
    return None


def check(lines, first, last):
    """Violations [(line index, instruction, registers, line of the load)] and the number of register loads seen."""
    body = lines[first:last + 1]
    queue, bad, loads = [], [], 0
    for rep in range(2):
        for i, l in enumerate(body):
            s = l.strip()
            if not s or s.startswith(";") or s.startswith("."):
                continue
            m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
                continue
            touched = _regs(s.split(";")[0])
            for (rs, ln) in queue:
                if rs and touched & rs:
                    bad.append((first + i, s, sorted(touched & rs), ln))
            if s.startswith("global_load_dwordx4"):
                queue.append((_regs(s.split(",")[0]), first + i))
                loads += rep == 0
            elif s.startswith("global_load_lds"):
                queue.append((None, first + i))
    return bad, loads


def main():
    asm = open(sys.argv[1]).read() if len(sys.argv) > 1 else device_asm()
    rc = 0
    for skip in (1, 0):
        k = kernel_text(asm, "_ZN10sgo_conv4r8k_conv4rILb%dELi1E" % skip)
        first, last = k_loop(k)
        bad, loads = check(k, first, last)
        print("k_conv4r<%s, 1>: K loop lines %d..%d, %d register loads, %d violations" % ("true" if skip else "false", first, last, loads, len(bad)))
        for b in bad[:10]:
            print("   ", b)
        rc |= bool(bad)
    sys.exit(rc)


if __name__ == "__main__":
    main()

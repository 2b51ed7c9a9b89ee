#!/usr/bin/env python3
"""Where do a kernel's register spills live?  (developer tool, runs in the build container: needs only hipcc)

    python tools/loop_scratch.py [kernel-name-substring ...]        default: the register-resident persistent kernels

Compiles dcmrta_env.hip to gfx950 assembly (`make -C dcmrta_amd/csrc asm`), and for every matching kernel walks the compiler's own
loop annotations (`; in Loop: Header=BBx_y Depth=N`, `; Parent Loop BBx_y`) to find the DECISION loop -- the loop that contains the
distance sqrt (`v_rsq_f64`) of decide() -- and counts the scratch (spill) instructions inside that loop and its
child loops, next to the kernel's total.  A kernel whose compiler report shows scratch can still have a spill-free decision loop:
the spills then sit in the once-per-episode general code (reset, terminal metrics, first event)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dcmrta_amd", "csrc")
ASM = os.environ.get("LOOP_SCRATCH_ASM") or os.path.join(CSRC, "dcmrta_env.gfx950.s")      # (LOOP_SCRATCH_ASM + LOOP_SCRATCH_KEEP_ASM=1: another build)


def kernels(lines):
    """(name, first line, last line) of every kernel body in the assembly"""
    out, name, start = [], None, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):\s+; @", l)
        if m:
            name, start = m.group(1), i
        elif name and l.startswith("\t.amdhsa_kernel " + name):
            out.append((name, start, i))
            name = None
    return out


def analyse(body):
    """blocks -> loop header; returns (scratch in the decision loop incl. child loops, scratch total, decision-loop header, its depth)"""
    block_loop, parent, cur, pending = {}, {}, None, None
    block_of = []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)[2:]                       # BBx_y
            pending = cur
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            if h:
                block_loop[cur] = h.group(1)
            p = re.search(r"Parent Loop (BB\d+_\d+)", l)
            if p:
                parent[cur] = p.group(1)
                block_loop[cur] = cur
            if "Loop Header" in l:
                block_loop[cur] = cur
        elif pending and l.lstrip().startswith(";"):   # continuation lines of the block comment
            if "Loop Header" in l:
                block_loop[pending] = pending
            p = re.search(r"Parent Loop (BB\d+_\d+)", l)
            if p:                                      # (outermost first: the last line names the immediate parent)
                parent[pending] = p.group(1)
                block_loop[pending] = pending
        else:
            pending = None
        block_of.append(cur)

    def chain(b):                                      # loop headers enclosing block b, innermost first
        out, h = [], block_loop.get(b)
        while h and h not in out:
            out.append(h)
            h = parent.get(h)
        return out

    marker = [i for i, l in enumerate(body) if "v_rsq_f64" in l]      # the sqrt of decide()'s distance: the only one in these kernels
    total = sum(1 for l in body if re.match(r"\s+scratch_(load|store)", l))
    if not marker:
        return None, total, None, 0
    encl = chain(block_of[marker[0]])
    # the decision loop: the enclosing loop that has exactly one loop above it (the episode loop of the kernel)
    dec = encl[-2] if len(encl) >= 2 else encl[-1]
    size = sum(1 for i, l in enumerate(body) if re.match(r"\s+[a-z]", l) and dec in chain(block_of[i]))
    inside = sum(1 for i, l in enumerate(body) if re.match(r"\s+scratch_(load|store)", l) and dec in chain(block_of[i]))
    return inside, total, dec, size


def main():
    pats = sys.argv[1:] or ["k_rollout_fastI", "k_rollout_fast_mcI", "k_rollout_fast_gI"]
    if not os.environ.get("LOOP_SCRATCH_KEEP_ASM"):
        subprocess.run(["make", "-C", CSRC, "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(ASM).read().split("\n")
    for name, a, b in kernels(lines):
        if not any(p in name for p in pats):
            continue
        inside, total, dec, depth = analyse(lines[a:b])
        short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name).split("EEv")[0]
        if inside is None:
            print(f"{short:44s} scratch instructions: {total:4d} in the kernel; no decision loop found")
        else:
            print(f"{short:44s} scratch instructions: {total:4d} in the kernel, {inside:3d} in the decision loop ({dec}: {depth} instructions)")


if __name__ == "__main__":
    main()

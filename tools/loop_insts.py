#!/usr/bin/env python3
"""Static instruction census of a persistent kernel's DECISION loop (developer tool, build container: needs only hipcc).

    python tools/loop_insts.py [-D...] [--kernel 'k_rollout_fast<20, 50, false, true>'] [--keep out.s] [--blocks]

Compiles ONE explicit instantiation of the kernel (a scratch translation unit that includes dcmrta_env.hip with its host API
compiled out: seconds instead of the minute the whole library takes), finds the decision loop -- the loop around decide()'s
distance sqrt (v_rsq_f64) -- and counts its instructions by class: VALU (fp64 / DPP / readlane+mbcnt / v_mov / other), SALU, s_nop,
branches, exec-mask saves, LDS, VMEM, waitcnt.  --blocks lists the basic blocks of the loop with their sizes."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dcmrta_amd", "csrc")
SIG = ("(int, int, int, int, KP, unsigned char*, int, float*, float*, uint8_t*, int64_t*, double*, uint16_t*, const int32_t*, int64_t, "
       "const int64_t*, unsigned char*, double*, int)")


def classify(op):
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if "saveexec" in op:
        return "saveexec"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if "dpp" in op:
            return "valu:dpp"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_mbcnt", "v_writelane")):
            return "valu:lane"
        if op.startswith("v_mov"):
            return "valu:mov"
        if re.search(r"_f64|_b64|_u64|_i64", op) and not op.startswith("v_mov"):
            return "valu:64"
        return "valu:other"
    return "other"


def main():
    args = sys.argv[1:]
    kernel = "k_rollout_fast<20, 50, false, true>"
    keep, blocks, defs = None, False, []
    i = 0
    while i < len(args):
        if args[i] == "--kernel":
            kernel = args[i + 1]; i += 2
        elif args[i] == "--keep":
            keep = args[i + 1]; i += 2
        elif args[i] == "--blocks":
            blocks = True; i += 1
        else:
            defs.append(args[i]); i += 1
    with tempfile.TemporaryDirectory() as td:
        tu = os.path.join(td, "one.hip")
        with open(tu, "w") as f:
            f.write(f'#define DCM_DEVICE_ONLY_TU 1\n#include "{CSRC}/dcmrta_env.hip"\nnamespace {{ template __global__ void {kernel}{SIG}; }}\n')
        out = keep or os.path.join(td, "one.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-phi-elim-split-all-critical-edges=1"] + defs + [tu, "-o", out],
                              stderr=open(os.path.join(td, "remarks.txt"), "w"))
        rem = open(os.path.join(td, "remarks.txt")).read()
        body = open(out).read().splitlines()
    # the body of the instantiated kernel only (the TU also holds the non-template helper kernels)
    base = re.match(r"\w+", kernel).group(0)
    starts = [i for i, l in enumerate(body) if re.match(r"^_Z\w*" + re.escape(f"{len(base)}{base}") + r"\w*:\s+; @", l)]
    ends = [i for i, l in enumerate(body) if l.startswith("\t.amdhsa_kernel ") and base in l]
    body = body[starts[-1]:ends[-1]]
    rem = rem[rem.rfind("Function Name: " + body[0].split(":")[0]):]
    print("   " + "  ".join(m.group(1).strip() for key in ("VGPRs:", "SGPRs:", "ScratchSize", "Occupancy", "LDS Size")
                            for m in [re.search(r"remark: [^\n]*?(" + re.escape(key) + r"[^\n]*?) \[", rem)] if m))
    # loop structure from the compiler's annotations
    block_loop, parent, cur, pending, block_of = {}, {}, None, None, []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = pending = m.group(1)[2:]
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            if h:
                block_loop[cur] = h.group(1)
            if "Loop Header" in l:
                block_loop[cur] = cur
            p = re.search(r"Parent Loop (BB\d+_\d+)", l)
            if p:
                parent[cur] = p.group(1); block_loop[cur] = cur
        elif re.match(r"^; %bb\.\d+:", l):
            m2 = re.match(r"^; %bb\.(\d+):", l)
            cur = pending = "bb." + m2.group(1)
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            if h:
                block_loop[cur] = h.group(1)
        elif pending and l.lstrip().startswith(";") and not l.lstrip().startswith(";;"):
            if "Loop Header" in l:
                block_loop[pending] = pending
            p = re.search(r"Parent Loop (BB\d+_\d+)", l)
            if p:
                parent[pending] = p.group(1); block_loop[pending] = pending
        else:
            pending = None
        block_of.append(cur)

    def chain(b):
        o, h = [], block_loop.get(b)
        while h and h not in o:
            o.append(h); h = parent.get(h)
        return o
    marker = [i for i, l in enumerate(body) if "v_rsq_f64" in l]
    encl = chain(block_of[marker[0]])
    dec = encl[-2] if len(encl) >= 2 else encl[-1]
    tot, per_block = collections.Counter(), collections.OrderedDict()
    for i, l in enumerate(body):
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if not m or l.startswith("\t.") or dec not in chain(block_of[i]):
            continue
        c = classify(m.group(1))
        tot[c] += 1
        per_block.setdefault(block_of[i], collections.Counter())[c] += 1
    valu = sum(v for k, v in tot.items() if k.startswith("valu"))
    print(f"decision loop {dec} of {kernel}: {sum(tot.values())} instructions, {valu} VALU")
    print("   " + "  ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    if blocks:
        for b, c in per_block.items():
            print(f"   {b:12s} {sum(c.values()):4d}  " + " ".join(f"{k}:{v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()

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
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
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


def cheap(op):
    """VALU instructions the calibration (profiles/r06_calib) prices at 2.35 SIMD clocks: 32-bit VOP1 / VOP2 in their e32 encoding
    (moves, integer add / sub / logic / shifts, selects on VCC, fp32).  Everything else -- 64-bit operands or results, compares,
    VOP3 (e64: SGPR masks or destinations, three operands, v_bfe / v_mad / v_or3 / v_bitop3), SDWA, DPP, v_readlane / v_mbcnt --
    measured 4.2-4.3."""
    return op.endswith("_e32") and not re.search(r"_f64|_b64|_u64|_i64|^v_cmp|^v_readfirstlane|^v_mbcnt", op)


def main():
    args = sys.argv[1:]
    kernel = "k_rollout_fast<20, 50, false, true>"
    keep, blocks, defs, dump, phases, weights = None, False, [], None, False, None
    i = 0
    while i < len(args):
        if args[i] == "--kernel":
            kernel = args[i + 1]; i += 2
        elif args[i] == "--keep":
            keep = args[i + 1]; i += 2
        elif args[i] == "--blocks":
            blocks = True; i += 1
        elif args[i] == "--phases":                    # needs -DDCM_PHASE_MARKS: per-phase census between the FPHMARK comments
            phases = True; i += 1
        elif args[i] == "--weights":                   # --phases: JSON {"<phase>[ <sub-region>]": executions per decision}
            import json
            weights = json.load(open(args[i + 1])); i += 2
        elif args[i] == "--dump":                      # write the decision loop's lines (block labels + instructions) to a file
            dump = args[i + 1]; i += 2
        else:
            defs.append(args[i]); i += 1
    with tempfile.TemporaryDirectory() as td:
        tu = os.path.join(td, "one.hip")
        with open(tu, "w") as f:
            f.write(f'#define DCM_DEVICE_ONLY_TU 1\n#include "{CSRC}/dcmrta_env.hip"\nnamespace {{ template __global__ void {kernel}{SIG}; }}\n')
        out = keep or os.path.join(td, "one.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                               "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-mllvm", "-amdgpu-sched-strategy=max-ilp"] + ([] if "fast_g" in kernel else ["-mllvm", "-phi-elim-split-all-critical-edges=1", "-DDCM_SPLIT_G"]) + defs + [tu, "-o", out],
                              stderr=open(os.path.join(td, "remarks.txt"), "w"))
        rem = open(os.path.join(td, "remarks.txt")).read()
        body = open(out).read().splitlines()
    # the body of the instantiated kernel only (the TU also holds the non-template helper kernels)
    base = re.match(r"\w+", kernel).group(0)
    targs = re.search(r"<(.*)>", kernel).group(1).split(",")
    mangled = f"{len(base)}{base}I" + "".join(f"Lb{int(a.strip() == 'true')}E" if a.strip() in ("true", "false") else f"Li{int(a)}E"
                                               for a in targs) + "E"
    starts = [i for i, l in enumerate(body) if re.match(r"^_Z\w*" + re.escape(mangled) + r"\w*:\s+; @", l)]
    ends = [i for i, l in enumerate(body) if l.startswith("\t.amdhsa_kernel ") and mangled in l]
    body = body[starts[-1]:ends[-1]]
    rem = rem[rem.rfind("Function Name: " + body[0].split(":")[0]):]
    rem = "\n".join(rem.splitlines()[:12])                      # (the remarks of this function only)
    keys = ("VGPRs:", "AGPRs:", "SGPRs:", "ScratchSize", "Occupancy", "LDS Size")
    print("   " + "  ".join(l.split("remark:", 1)[1].split(" [-Rpass")[0].strip() for l in rem.splitlines()
                            if "remark:" in l and any(k in l for k in keys)))
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
    if dump:
        with open(dump, "w") as f:
            for i, l in enumerate(body):
                if dec in chain(block_of[i]):
                    f.write(l + "\n")
    if phases:
        # text order inside the loop: an instruction belongs to the phase whose mark FOLLOWS it (FPH(i) closes phase i); between the
        # sub-phase brackets 20..21 (member removal) it is counted separately; instructions of an inner loop (depth >= 3: the
        # follower draws, the group split, the walk over the dropping tasks) as well
        seq, pend, sub, stack = [], [], None, []
        res = collections.OrderedDict()
        depth = {}
        for b in set(block_of):
            if b:
                depth[b] = len(chain(b))
        base_depth = len(chain(dec))
        for i, l in enumerate(body):
            if dec not in chain(block_of[i]):
                continue
            m = re.match(r"^\s+; FPHMARK (\d+)", l)
            if m:
                k = int(m.group(1))
                if k == 20:
                    stack.append(sub)
                    sub = "removal"
                elif k == 21:
                    sub = stack.pop() if stack else None
                elif k in (22, 23):
                    pass
                else:
                    for cls, sb in pend:
                        res.setdefault((k, sb), collections.Counter())[cls] += 1
                    pend = []
                continue
            m = re.match(r"^\s+([a-z_0-9]+)", l)
            if not m or l.startswith("\t."):
                continue
            inner = len(chain(block_of[i])) > base_depth
            cls = classify(m.group(1))
            pend.append((cls, sub or ("inner loop" if inner else "")))
            if cls.startswith("valu") and cheap(m.group(1)):
                pend.append(("cheap", sub or ("inner loop" if inner else "")))
        for cls, sb in pend:
            res.setdefault((99, sb), collections.Counter())[cls] += 1
        print("phase (closing mark) / sub-region: VALU (64-bit, dpp, lane, mov, other) | SALU | s_nop | branch | saveexec | LDS | VMEM | waitcnt")
        for (k, sb), c in res.items():
            v = [c.get("valu:" + x, 0) for x in ("64", "dpp", "lane", "mov", "other")]
            print(f"   {k:3d} {sb:22s} VALU {sum(v):4d} ({v[0]:3d} {v[1]:3d} {v[2]:3d} {v[3]:3d} {v[4]:3d}) | SALU {c.get('salu', 0):4d} | nop {c.get('s_nop', 0):3d} | "
                  f"br {c.get('branch', 0):3d} | sx {c.get('saveexec', 0):3d} | lds {c.get('lds', 0):3d} | vmem {c.get('vmem', 0):2d} | wait {c.get('waitcnt', 0):2d}"
                  f" | 2.35-clock VALU {c.get('cheap', 0):3d}" + (f" | x {weights.get((str(k) + ' ' + sb).strip(), 0):.3f}" if weights else ""))
        if weights:
            dyn = collections.Counter()
            for (k, sb), c in res.items():
                w = weights.get((str(k) + " " + sb).strip(), 0.0)
                for cls, n in c.items():
                    dyn[cls] += w * n
            dv = sum(v for k, v in dyn.items() if k.startswith("valu"))
            print(f"weighted per decision: VALU {dv:.1f} (2.35-clock class {dyn['cheap']:.1f} = {dyn['cheap'] / dv:.3f}; 64-bit {dyn['valu:64']:.1f}, dpp {dyn['valu:dpp']:.1f}, "
                  f"lane {dyn['valu:lane']:.1f}, v_mov {dyn['valu:mov']:.1f})  SALU {dyn['salu'] + dyn['saveexec']:.1f}  s_nop {dyn['s_nop']:.1f}  "
                  f"branch {dyn['branch']:.1f}  LDS {dyn['lds']:.1f}  VMEM {dyn['vmem']:.1f}  waitcnt {dyn['waitcnt']:.1f}")
    valu = sum(v for k, v in tot.items() if k.startswith("valu"))
    print(f"decision loop {dec} of {kernel}: {sum(tot.values())} instructions, {valu} VALU")
    print("   " + "  ".join(f"{k} {v}" for k, v in sorted(tot.items())))
    if blocks:
        for b, c in per_block.items():
            print(f"   {b:12s} {sum(c.values()):4d}  " + " ".join(f"{k}:{v}" for k, v in sorted(c.items())))


if __name__ == "__main__":
    main()

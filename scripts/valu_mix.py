#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel, priced with the issue costs measured by scripts/issue_rate.hip
(profiles/r3_issue_rate.json): cycles per wave64 VALU instruction for THIS kernel's mix.

SQ_ACTIVE_INST_VALU ticks once per instruction whatever it costs (profiles/r3_issue_rate_sq_pmc.csv), so a
VALU-issue fraction needs the mix:  valu_issue_frac = SQ_INSTS_VALU x cycles_per_inst(mix) / (GRBM_GUI_ACTIVE x SIMDs).
The mix is static (every instruction of the kernel's text counted once): the plain projection is close to straight-line
code, its rarely-taken fallbacks (literal walks) are excluded with --until LABEL-free heuristics only by listing basic
blocks -- so treat the figure as +-10 %.

usage: valu_mix.py [--asm /tmp/kernels.s] [--kernel SUBSTRING] [--json out.json]
       (the asm comes from: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/kernels.s impg_amd/csrc/kernels.hip)
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# measured classes (profiles/r3_issue_rate.json): 2.07 cycles for these with VGPR / literal / inline-constant operands ...
FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32",
             "v_ashrrev_i32", "v_fma_f32", "v_add_f32", "v_mul_f32", "v_not_b32"}
FULL, HALF = 2.07, 4.13
PAIR_CMP_CND = 6.32  # v_cmp + the v_cndmask that reads it: 3.16 each


def cost(op, operands):
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if base.startswith("v_cmp") or base.startswith("v_cndmask"):
        return PAIR_CMP_CND / 2
    if base in FULL_RATE:
        # ... but 4.13 with an SGPR source operand, DPP or SDWA
        srcs = operands.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec|m0)", x) for x in srcs) or "_dpp" in op or "_sdwa" in op:
            return HALF
        return FULL
    return HALF


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default="/tmp/kernels.s")
    ap.add_argument("--kernel", default="project_kernelILb1ELi0E", help="substring of the mangled kernel name")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    if not os.path.exists(args.asm):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", args.asm,
                               os.path.join(ROOT, "impg_amd", "csrc", "kernels.hip")])
    inside = False
    ops = collections.Counter()
    cyc = collections.Counter()
    n_valu = n_salu = n_vmem = n_lds = 0
    for line in open(args.asm):
        if not inside:
            if re.match(r"^_Z\S*%s\S*:" % re.escape(args.kernel), line):
                inside = True
            continue
        if line.startswith(".Lfunc_end") or line.strip().startswith("s_endpgm") and False:
            break
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)(;.*)?$", line)
        if not m:
            continue
        op, operands = m.group(1), m.group(2)
        if op.startswith("v_"):
            n_valu += 1
            ops[op] += 1
            cyc[op] += cost(op, operands)
        elif op.startswith("s_"):
            n_salu += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            n_vmem += 1
        elif op.startswith("ds_"):
            n_lds += 1
    if not n_valu:
        sys.exit("kernel not found: " + args.kernel)
    total = sum(cyc.values())
    out = {"kernel": args.kernel, "static_valu": n_valu, "static_salu": n_salu, "static_vmem": n_vmem, "static_lds": n_lds,
           "cycles_per_valu_inst": total / n_valu,
           "top": [{"op": o, "count": c, "cycles_each": round(cyc[o] / c, 2)} for o, c in ops.most_common(25)]}
    print(json.dumps(out, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/<tag>_sq.json from the SQ pass of scripts/profile_r2.sh: the VALU-issue fraction of project_kernel.
SQ_ACTIVE_INST_VALU counts, in quad-cycles, the time waves spend executing VALU instructions; on this kernel (32-bit
integer ops, no packed / MFMA work) it equals SQ_INSTS_VALU to 1 %: one wave64 instruction keeps its SIMD's VALU
one quad-cycle = 4 clocks (the 2-clock figure of MI355X_MICROARCH.md is the packed / dual-issue rate, which integer
compare / select / add code does not reach).  So
valu_issue_frac = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs), counters summed over the kernel's
dispatches (SQ_INSTS_VALU x 4 when the pass did not collect SQ_ACTIVE_INST_VALU).
usage: make_sq_json.py <dir with sq_pmc.csv and sq_bench.json> <out.json>"""
import csv, json, os, sys

d, out = sys.argv[1], sys.argv[2]
vals, rows_of, disp = {}, {}, 0
for row in csv.DictReader(open(os.path.join(d, "sq_pmc.csv"))):
    if "project_kernel" in row["Name"]:
        vals[row["Counter"]] = vals.get(row["Counter"], 0.0) + float(row["Sum"])
        rows_of[row["Counter"]] = rows_of.get(row["Counter"], 0) + int(row["Dispatches"])
for row in csv.DictReader(open(os.path.join(d, "sq_kernel_stats.csv"))):
    if "project_kernel" in row["Name"]:
        disp += int(row["Calls"])
bench = json.loads(open(os.path.join(d, "sq_bench.json")).read().strip().splitlines()[-1])
passes = bench["steps"] + bench["warmup"]
pairs = bench["pairs_per_step_rank0"] * passes
# rocprofv3 reports GRBM_GUI_ACTIVE once per XCD and dispatch (8 rows per dispatch on MI355X): the kernel's cycles,
# summed over its dispatches, are the mean row x the number of dispatches
gui = vals.get("GRBM_GUI_ACTIVE", 0.0) / max(1, rows_of.get("GRBM_GUI_ACTIVE", 1)) * disp
valu = vals.get("SQ_INSTS_VALU", 0.0)
res = {"kernel": "project_kernel", "command": "bench.py " + " ".join(bench.get("argv", [])) + " (scripts/profile_r2.sh, --pmc pass)",
       "dispatches": disp, "pairs": pairs, "counters": vals, "counter_rows": rows_of, "kernel_cycles_total": gui,
       "valu_insts_per_wave": valu / vals["SQ_WAVES"] if vals.get("SQ_WAVES") else None,
       "valu_insts_per_pair": valu / pairs / 1.0 if pairs else None,  # wave-instructions per pair (x64 lanes / 64 pairs per wave)
       "salu_insts_per_wave": vals.get("SQ_INSTS_SALU", 0.0) / vals["SQ_WAVES"] if vals.get("SQ_WAVES") else None,
       "vmem_insts_per_wave": vals.get("SQ_INSTS_VMEM", 0.0) / vals["SQ_WAVES"] if vals.get("SQ_WAVES") else None,
       "valu_issue_frac": (vals.get("SQ_ACTIVE_INST_VALU", valu) * 4.0) / (gui * 1024.0) if gui else None,
       "note": "GRBM_GUI_ACTIVE = kernel cycles (summed over dispatches); 1024 SIMDs; SQ_ACTIVE_INST_VALU is in quad-cycles and "
               "equals SQ_INSTS_VALU here: a wave64 integer VALU instruction occupies its SIMD for 4 clocks"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

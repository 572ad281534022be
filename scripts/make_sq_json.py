#!/usr/bin/env python3
"""profiles/<tag>_sq.json from the SQ pass of scripts/profile_r2.sh: the VALU-issue fraction of project_kernel.
A wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md, wave scheduling), so
valu_issue_frac = SQ_INSTS_VALU x 2 / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs), counters summed over the kernel's dispatches.
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
       "valu_issue_frac": (valu * 2.0) / (gui * 1024.0) if gui else None,
       "note": "GRBM_GUI_ACTIVE = kernel cycles (summed over dispatches); 1024 SIMD-32 units; 2 cycles per wave64 VALU instruction "
               "(64-bit and transcendental ops cost more, so this is a lower bound on VALU-port occupancy)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

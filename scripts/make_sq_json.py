#!/usr/bin/env python3
"""profiles/<tag>_sq.json from the SQ pass of scripts/profile_r2.sh: the VALU-issue fraction of project_kernel.
SQ_ACTIVE_INST_VALU ticks ONCE PER INSTRUCTION whatever the instruction costs (scripts/issue_rate.hip,
profiles/r3_issue_rate_sq_pmc.csv: it equals SQ_INSTS_VALU for 2-cycle v_add_u32 and 4-cycle v_bfe_u32 alike), so it
cannot give the issue time by itself.  The time comes from the measured per-instruction costs (profiles/r3_issue_rate.json:
2.07 cycles for add / sub / logic / mov / right shifts on VGPRs, 4.13 for everything else incl. any SGPR operand, 6.3 per
v_cmp + v_cndmask pair) weighted with the kernel's static instruction mix (scripts/valu_mix.py):
valu_issue_frac = SQ_INSTS_VALU x cycles_per_inst(mix) / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs), counters summed over
the kernel's dispatches.  Rounds 1-2 used a flat 2, then a flat 4.
usage: make_sq_json.py <dir with sq_pmc.csv and sq_bench.json> <out.json> [cycles_per_valu_inst | valu_mix.json]"""
import csv, json, os, re, sys

# the projection of a step: project_kernel (sparse levels), project_staged_kernel (dense listed levels), project_entries_kernel (dense final level)
PROJ = re.compile(r"project_(kernel|staged_kernel|entries_kernel)")

d, out = sys.argv[1], sys.argv[2]
cpi, cpi_src = 4.13, "default: the half-rate class (no mix given)"
if len(sys.argv) > 3:
    if os.path.exists(sys.argv[3]):
        cpi = json.load(open(sys.argv[3]))["cycles_per_valu_inst"]
        cpi_src = "static mix of the kernel, " + os.path.basename(sys.argv[3])
    else:
        cpi, cpi_src = float(sys.argv[3]), "given on the command line"
vals, rows_of, disp = {}, {}, 0
for row in csv.DictReader(open(os.path.join(d, "sq_pmc.csv"))):
    if PROJ.search(row["Name"]):
        vals[row["Counter"]] = vals.get(row["Counter"], 0.0) + float(row["Sum"])
        rows_of[row["Counter"]] = rows_of.get(row["Counter"], 0) + int(row["Dispatches"])
for row in csv.DictReader(open(os.path.join(d, "sq_kernel_stats.csv"))):
    if PROJ.search(row["Name"]):
        disp += int(row["Calls"])
bench = json.loads(open(os.path.join(d, "sq_bench.json")).read().strip().splitlines()[-1])
passes = bench["steps"] + bench["warmup"]
pairs = bench["pairs_per_step_rank0"] * passes
# rocprofv3 reports GRBM_GUI_ACTIVE once per XCD and dispatch (8 rows per dispatch on MI355X): the kernel's cycles,
# summed over its dispatches, are the mean row x the number of dispatches
gui = vals.get("GRBM_GUI_ACTIVE", 0.0) / max(1, rows_of.get("GRBM_GUI_ACTIVE", 1)) * disp
valu = vals.get("SQ_INSTS_VALU", 0.0)
res = {"kernel": "project_kernel + project_staged_kernel + project_entries_kernel (the projections of a step's levels)", "command": "bench.py " + " ".join(bench.get("argv", [])) + " (scripts/profile_r2.sh, --pmc pass)",
       "dispatches": disp, "pairs": pairs, "counters": vals, "counter_rows": rows_of, "kernel_cycles_total": gui,
       "valu_insts_per_wave": valu / vals["SQ_WAVES"] if vals.get("SQ_WAVES") else None,
       "valu_insts_per_pair": valu / pairs / 1.0 if pairs else None,  # wave-instructions per pair (x64 lanes / 64 pairs per wave)
       "salu_insts_per_wave": vals.get("SQ_INSTS_SALU", 0.0) / vals["SQ_WAVES"] if vals.get("SQ_WAVES") else None,
       "vmem_insts_per_wave": vals.get("SQ_INSTS_VMEM", 0.0) / vals["SQ_WAVES"] if vals.get("SQ_WAVES") else None,
       "cycles_per_valu_inst": cpi, "cycles_per_valu_inst_source": cpi_src,
       "valu_issue_frac": (valu * cpi) / (gui * 1024.0) if gui else None,
       "valu_issue_frac_if_2_cycles": (valu * 2.07) / (gui * 1024.0) if gui else None,
       "valu_issue_frac_if_4_cycles": (valu * 4.13) / (gui * 1024.0) if gui else None,
       "note": "GRBM_GUI_ACTIVE = kernel cycles (summed over dispatches); 1024 SIMDs; SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU is a count, "
               "not a time (profiles/r3_issue_rate.json): the issue time is the count x the measured cost of the kernel's instruction mix"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

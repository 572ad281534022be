#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the PMC summaries of scripts/profile_r1.sh:
HBM bytes per candidate pair of project_kernel = FETCH_SIZE (KB) x 2 [gfx950: 128-byte
requests are tallied at 64 B; checked against TCC_EA0_RDREQ_128B] + WRITE_SIZE (KB).
usage: make_traffic_json.py <dir with *_pmc.csv and trace_bench.json> <out.json>"""
import csv, glob, json, os, re, sys

# the projection of a step: project_kernel (sparse levels), project_staged_kernel (dense listed levels), project_entries_kernel (dense final level)
PROJ = re.compile(r"project_(kernel|staged_kernel|entries_kernel)")

d, out = sys.argv[1], sys.argv[2]
vals = {}
disp = 0
for path in glob.glob(os.path.join(d, "*_pmc.csv")):
    for row in csv.DictReader(open(path)):
        if PROJ.search(row["Name"]):
            vals[row["Counter"]] = vals.get(row["Counter"], 0.0) + float(row["Sum"])
            disp = max(disp, int(row["Dispatches"]))
bench = json.loads(open(os.path.join(d, "fetch_bench.json")).read().strip().splitlines()[-1])
# the profiled command runs warmup + steps passes; every pass presents the same pairs
passes = bench["steps"] + bench["warmup"]
pairs = bench["pairs_per_step_rank0"] * passes
rd128 = vals.get("TCC_EA0_RDREQ_128B_sum", 0.0)
res = {
    "kernel": "project_kernel + project_staged_kernel + project_entries_kernel (the projections of a step's levels)",
    "command": "bench.py " + " ".join(bench.get("argv", ["--ranges 16384 --steps 2 --warmup 1 --cpu-sample 0"])) + " (scripts/profile_r*.sh, separate --pmc passes)",
    "dispatches": disp,
    "pairs": pairs,
    "FETCH_SIZE_KB_sum": vals.get("FETCH_SIZE"),
    "WRITE_SIZE_KB_sum": vals.get("WRITE_SIZE"),
    "TCC_EA0_RDREQ_sum": vals.get("TCC_EA0_RDREQ_sum"),
    "TCC_EA0_RDREQ_128B_sum": rd128,
    "TCC_EA0_WRREQ_64B_sum": vals.get("TCC_EA0_WRREQ_64B_sum"),
    "fetch_bytes_per_pair_reported": vals["FETCH_SIZE"] * 1024 / pairs,
    "fetch_bytes_per_pair_corrected_x2": vals["FETCH_SIZE"] * 1024 * 2 / pairs,
    "fetch_bytes_per_pair_from_128B_requests": rd128 * 128 / pairs if rd128 else None,
    "write_bytes_per_pair": vals["WRITE_SIZE"] * 1024 / pairs,
    "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, HBM section); x2 is exact when "
            "RDREQ == RDREQ_128B",
}
res["hbm_bytes_per_pair"] = res["fetch_bytes_per_pair_corrected_x2"] + res["write_bytes_per_pair"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

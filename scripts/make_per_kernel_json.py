#!/usr/bin/env python3
"""profiles/<tag>_per_kernel.json: every kernel that takes >= 1 % of a headline step, with its time from the kernel trace
and what the PMC passes of the same command saw of it -- HBM bytes (FETCH_SIZE x 2 [gfx950: 128-byte requests tallied at
64 B; exact when TCC_EA0_RDREQ == TCC_EA0_RDREQ_128B, which the ea pass shows] + WRITE_SIZE), VALU instructions priced with
the kernel's own static mix (scripts/valu_mix.py, profiles/r3_issue_rate.json), occupancy (SQ_WAVE_CYCLES / kernel cycles
/ 1024 SIMDs), s_waitcnt share and LDS bank conflicts where those passes were taken.  bench.py puts the list into
roofline.per_kernel.
usage: make_per_kernel_json.py <dir with trace_kernel_stats.csv, *_pmc.csv, trace_bench.json> <out.json> [--asm kernels.s]"""
import csv, glob, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12
SIMDS = 1024
CLK = 2.4e9  # engine clock: GRBM_GUI_ACTIVE of round 4's passes / kernel time = 2.41 GHz (rocprofv3's rows per dispatch differ from pass to pass, the time does not)

d, out = sys.argv[1], sys.argv[2]
asm = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None


def short(name):
    name = re.sub(r"\.kd$", "", name)
    m = re.match(r"_ZN4impg(?:12_GLOBAL__N_1)?\d+([A-Za-z_0-9]+?)(I[A-Z].*)?E", name)
    if m:
        base = m.group(1)
        t = re.match(r"I((?:L[bjim]\d+E?)+)", m.group(2) or "")
        if t:
            base += "<" + ",".join(re.findall(r"L[bjim](\d+)", t.group(1))) + ">"
        return base
    if "rocprim" in name:
        k = re.search(r"(onesweep_histograms|radix_sort_onesweep|radix_sort_block_sort|radix_sort_merge|lookback_scan|histogram|block_sort|merge|scan)", name)
        vt = "pairs" if re.search(r"iterationIS\d_Lb0EPK[jy]P[jy]S\d_S\w_", name) and "PKyPyS" in name else ""
        return "rocprim::" + (k.group(1) if k else "kernel") + (" (" + vt + ")" if vt else "")
    return name[:48]


bench = json.loads(open(os.path.join(d, "trace_bench.json")).read().strip().splitlines()[-1])
passes = bench["steps"] + bench["warmup"]
trace = list(csv.DictReader(open(os.path.join(d, "trace_kernel_stats.csv"))))
pmc = {}  # kernel -> counter -> (sum, rows)
for path in sorted(glob.glob(os.path.join(d, "*_pmc.csv"))):
    for row in csv.DictReader(open(path)):
        c = pmc.setdefault(row["Name"], {})
        if row["Counter"] not in c:  # (a counter taken in two passes -- SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE -- counts once: the first file's)
            c[row["Counter"]] = (float(row["Sum"]), int(row["Dispatches"]))
step_ns = sum(int(r["TotalDurationNs"]) for r in trace if "cigar_" not in r["Name"] and "tiles_kernel" not in r["Name"] and "entries_kernelEPK17" not in r["Name"]) / passes
res = []
for r in trace:
    ns = int(r["TotalDurationNs"]) / passes
    if any(x in r["Name"] for x in ("cigar_", "tiles_kernel", "entries_kernelEPK17")):  # the index build, once per process
        continue
    if ns < 0.01 * step_ns:
        continue
    c = pmc.get(r["Name"], {})
    calls = int(r["Calls"])
    e = {"name": short(r["Name"]), "mangled": r["Name"][:96], "calls_per_step": calls / passes, "ms_per_step": ns / 1e6, "share_of_step": ns / step_ns,
         "vgprs": int(r["VGPRs"]) if r["VGPRs"].isdigit() else None}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd = c["FETCH_SIZE"][0] * 1024 * 2 / passes
        wr = c["WRITE_SIZE"][0] * 1024 / passes
        e.update({"hbm_read_GB_per_step": rd / 1e9, "hbm_written_GB_per_step": wr / 1e9, "hbm_TBs": (rd + wr) / ns / 1e3,
                  "hbm_frac": (rd + wr) / (ns * 1e-9) / HBM_PEAK})
        if "TCC_EA0_RDREQ_sum" in c and c["TCC_EA0_RDREQ_sum"][0]:
            e["rdreq_128B_share"] = c.get("TCC_EA0_RDREQ_128B_sum", (0, 0))[0] / c["TCC_EA0_RDREQ_sum"][0]
    if "SQ_INSTS_VALU" in c:
        cyc = ns * 1e-9 * CLK * passes  # the kernel's cycles over all its dispatches, from the un-instrumented trace
        valu = c["SQ_INSTS_VALU"][0]
        cpi, src = 3.1, "default (no asm given)"
        if asm:
            key = re.sub(r"\.kd$", "", r["Name"])
            key = key[len("_Z"):min(len(key), 60)]
            try:
                j = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "valu_mix.py"), "--asm", asm, "--kernel", key],
                                                       stderr=subprocess.DEVNULL))
                if j.get("static_valu"):
                    cpi, src = j["cycles_per_valu_inst"], "static mix of the kernel (valu_mix.py)"
            except Exception:
                pass
        e.update({"valu_insts_per_step": valu / passes, "cycles_per_valu_inst": cpi, "cycles_per_valu_inst_source": src,
                  "valu_issue_frac": valu * cpi / (cyc * SIMDS)})
        if "SQ_WAVE_CYCLES" in c:  # (counted in units of four cycles)
            e["waves_per_simd_mean"] = c["SQ_WAVE_CYCLES"][0] * 4.0 / (cyc * SIMDS)
        if "SQ_WAIT_INST_ANY" in c and c.get("SQ_WAVE_CYCLES", (0, 0))[0]:
            e["wait_share_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"][0] / c["SQ_WAVE_CYCLES"][0]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", (0, 0))[0]:
        e["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"][0] / c["SQ_LDS_IDX_ACTIVE"][0]
    res.append(e)
doc = {"command": "bench.py " + " ".join(bench.get("argv", [])) + " (scripts/profile_r2.sh: kernel trace, then separate --pmc passes)",
       "step_ms_of_kernels": step_ns / 1e6, "passes": passes, "peak_hbm_TBs": HBM_PEAK / 1e12,
       "note": "hbm_frac = (FETCH_SIZE x 2 + WRITE_SIZE) / kernel time / 8 TB/s; valu_issue_frac = SQ_INSTS_VALU x the measured issue cost of the "
               "kernel's instruction mix / (kernel time x 2.4 GHz x 1024 SIMDs); waves_per_simd_mean = SQ_WAVE_CYCLES x 4 / the same cycles; "
               "times are the un-instrumented trace pass's",
       "kernels": res}
json.dump(doc, open(out, "w"), indent=1)
for e in res:
    print("%-44s %6.2f ms %5.1f%%  hbm %s  valu %s  waves/SIMD %s" % (e["name"], e["ms_per_step"], 100 * e["share_of_step"],
          "%.2f" % e["hbm_frac"] if e.get("hbm_frac") is not None else "-", "%.2f" % e["valu_issue_frac"] if e.get("valu_issue_frac") else "-",
          "%.1f" % e["waves_per_simd_mean"] if e.get("waves_per_simd_mean") else "-"))

#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output (ROCm 7.2 default) into small CSV
summaries: per-kernel time (what `--stats` prints) and per-kernel PMC sums.
usage: rocpd_summary.py <results.db> <out_prefix>"""
import sqlite3
import sys


def main(db_path, out):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.workgroup_size_x) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,WorkgroupSize\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8]))
    pmc = list(cur.execute("select name from rocpd_info_pmc"))
    if pmc:
        rows = list(cur.execute(
            "select s.kernel_name, i.name, count(*), sum(p.value), avg(p.value) from rocpd_pmc_event p "
            "join rocpd_kernel_dispatch d on p.event_id=d.event_id join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
            "join rocpd_info_pmc i on p.pmc_id=i.id group by s.kernel_name, i.name order by 4 desc"))
        with open(out + "_pmc.csv", "w") as f:
            f.write("Name,Counter,Dispatches,Sum,MeanPerDispatch\n")
            for r in rows:
                f.write('"%s",%s,%d,%.6g,%.6g\n' % r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

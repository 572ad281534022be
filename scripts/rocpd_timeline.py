#!/usr/bin/env python3
"""GPU timeline of a rocprofv3 --kernel-trace run (rocpd SQLite): for the last `frac` of the run, the time the GPU
ran at least one kernel, the time it ran two or more side by side, the idle time, the longest idle gaps with the
kernels on either side.  usage: rocpd_timeline.py <results.db> <out.json> [frac=0.5 | last milliseconds]"""
import json
import sqlite3
import sys


def main(db, out, frac=0.5):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select d.start, d.end, s.kernel_name, d.queue_id from rocpd_kernel_dispatch d "
                            "join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * frac if frac <= 1.0 else t1 - frac * 1e6  # (a value above 1: the last `frac` milliseconds)
    rows = [r for r in rows if r[0] >= cut]
    ev = []
    for a, b, _, _ in rows:
        ev.append((a, 1)); ev.append((b, -1))
    ev.sort()
    busy = over = 0
    depth, last = 0, ev[0][0]
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    span = max(r[1] for r in rows) - rows[0][0]
    gaps = []
    end, prev = rows[0][1], rows[0][2]
    for a, b, name, q in rows[1:]:
        if a > end: gaps.append((a - end, prev[:60], name[:60]))
        if b > end: end, prev = b, name
    gaps.sort(reverse=True)
    hist = {}
    for g, p, n in gaps:
        k = (p.split("(")[0][-40:], n.split("(")[0][-40:])
        c = hist.setdefault(k, [0, 0]); c[0] += 1; c[1] += g
    top = sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]
    per = {}
    for a, b, name, q in rows:
        c = per.setdefault(name.split("(")[0][-56:], [0, 0]); c[0] += 1; c[1] += b - a
    per = sorted(per.items(), key=lambda kv: -kv[1][1])[:30]
    res = {"span_ms": span / 1e6, "busy_ms": busy / 1e6, "overlap_ms": over / 1e6, "idle_ms": (span - busy) / 1e6,
           "kernels": [{"name": k, "calls": v[0], "ms": v[1] / 1e6} for k, v in per],
           "dispatches": len(rows), "queues": len(set(r[3] for r in rows)),
           "idle_by_neighbours": [{"after": k[0], "before": k[1], "count": v[0], "ms": v[1] / 1e6} for k, v in top]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("idle_by_neighbours", "kernels")}))
    for e in res["kernels"][:int(sys.argv[4]) if len(sys.argv) > 4 else 0]:
        print("%9.3f ms %5d  %s" % (e["ms"], e["calls"], e["name"]))
    for g, p_, n_ in gaps[:8]:
        print("   gap %7.3f ms  %s -> %s" % (g / 1e6, p_.split("(")[0][-36:], n_.split("(")[0][-36:]))
    for e in res["idle_by_neighbours"][:14]:
        print("%8.3f ms %5d  %s -> %s" % (e["ms"], e["count"], e["after"], e["before"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.5)

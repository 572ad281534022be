#!/usr/bin/env python3
"""End-to-end wall time of `impg-gpu query` on BASELINE config 3 (1 M-record PAF, 10 k ranges, -x -m 3 -d 1000, BED):
index from the PAF vs from a saved index, then lookup + projection on the GPU, result assembly, merge and text on
the host.  (GPU box; bounded: 10 k ranges = 2.1e8 result rows before the merge.)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import impg_amd
tmp = tempfile.gettempdir()
paf = os.path.join(tmp, "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
bed = impg_amd.synth_bed(7, n)
bedf = os.path.join(tmp, "q%d.bed" % n)
with open(bedf, "w") as f:
    for t, s, e in zip(bed["target_id"], bed["start"], bed["end"]):
        f.write("%s\t%d\t%d\n" % (impg_amd.synth_seq_name(int(t)), s, e))
cli = os.path.join(ROOT, "impg_amd", "impg-gpu")
saved = os.path.join(tmp, "headline.impghbm")
t = time.perf_counter(); subprocess.run([cli, "index", "-a", paf, "-i", saved], check=True); t_index = time.perf_counter() - t
out = os.path.join(tmp, "out.bed")
for label, src in (("from the PAF", ["-a", paf]), ("from the saved index", ["-i", saved])):
    for flags in (["-d", "1000"], ["-d", "1000", "-x", "-m", "3"], ["-d", "1000", "-x", "-m", "3", "--host-merge"]):
        t = time.perf_counter()
        with open(out, "wb") as fo:
            r = subprocess.run([cli, "query", "-v", "1"] + src + ["-b", bedf, "-o", "bed"] + flags, stdout=fo, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t
        assert r.returncode == 0, r.stderr.decode()[-500:]
        print("%-22s %-18s %6.2f s wall, %7.1f MB of BED, %d rows" % (label, " ".join(flags), dt, os.path.getsize(out) / 1e6,
                                                                      0))
        print("   " + r.stderr.decode().strip().splitlines()[-1])
print("impg-gpu index: %.2f s" % t_index)
os.remove(saved); os.remove(out)

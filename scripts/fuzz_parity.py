#!/usr/bin/env python3
"""Differential soak test on the GPU box: random PAFs, random ranges, random parameters; the HIP engine
(through the C ABI) against the CPU oracle, row for row -- results, CIGARs, projection counts, BED / PAF /
BEDPE text.  usage: fuzz_parity.py <seconds> [first_seed] [big]     (big: thousands of records on few
sequences -- dense windows, the wave-per-range emit pass, frontiers past the lookup-order threshold)"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import impg_amd
from oracle import oracle as o
from tests.paf_gen import random_paf, random_ranges
from tests.test_gpu_fullsize import checksum

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
BIG = len(sys.argv) > 3
t_end = time.time() + budget
n_cases = n_rows = 0
tmp = tempfile.mkdtemp()
REPEAT = int(os.environ.get("FUZZ_REPEAT", "0"))  # > 0: the first seed over and over (hunting a failure that comes and goes)
while time.time() < t_end:
    if REPEAT and n_cases >= REPEAT:
        break
    rng = np.random.default_rng(seed)
    n_seq = int(rng.integers(2, 14))
    seq_len = int(rng.choice([2500, 8000, 30000, 120000]))
    max_ops = int(rng.choice([6, 30, 150, 700]))
    n_rec = int(rng.integers(20, 700))
    if BIG:
        n_seq, n_rec, seq_len = int(rng.integers(2, 6)), int(rng.integers(1500, 7000)), int(rng.choice([8000, 30000]))
        max_ops = int(rng.choice([6, 30, 150]))
    weird, incons, self_aln = bool(rng.random() < 0.4), bool(rng.random() < 0.3), bool(rng.random() < 0.6)
    n_files = int(rng.choice([1, 1, 2, 3]))
    paths = []
    for k in range(n_files):
        text, _ = random_paf(seed * 7 + k, max(5, n_rec // n_files), n_seq=n_seq, seq_len=seq_len, max_ops=max_ops, weird=weird,
                             inconsistent=incons, self_aln=self_aln)
        p = os.path.join(tmp, "f%d_%d.paf" % (seed, k))
        open(p, "w").write(text)
        paths.append(p)
    bidir = bool(rng.random() < 0.8)
    order = impg_amd.ORDER_COITREES if rng.random() < 0.8 else impg_amd.ORDER_SORTED
    # one GPU, or the index sharded over 2..4 ranks that share the GPU (the multi handle: same entry points)
    world = int(rng.choice([1, 1, 2, 3, 4]))
    devs = None if world == 1 else [0] * world
    # round 3: the index built on the device (default) or by the host builder, with or without its prefix lines
    for k2 in ("IMPG_BUILD_HOST", "IMPG_PREFIX_LINES"):
        os.environ.pop(k2, None)
    if rng.random() < 0.2:
        os.environ["IMPG_BUILD_HOST"] = "1"
    if rng.random() < 0.15:
        os.environ["IMPG_PREFIX_LINES"] = "0"
    g = impg_amd.GpuImpg.from_paf(paths, bidirectional=bidir, order=order, devices=devs, lanes=int(rng.integers(1, 3)))
    g.set_option("walk_kernel", int(rng.choice([0, 1, 1, 2])))       # the per-query walk: off / DFS / DFS + small BFS batches
    g.set_option("filter_covered", int(rng.choice([0, 0, 1, 2])))
    g.set_option("segment_parts", int(rng.choice([0, 0, 2, 3, 17])))    # the update's queries cut into slices (forced)
    o.set_sorted_visits(order == impg_amd.ORDER_SORTED)  # both order policies have an exact checker
    c = o.OracleIndex(paf_paths=paths, bidirectional=bidir, preparse=True)
    g.set_option("locality_min", int(rng.choice([0, 1, 4096])))
    for kv in filter(None, os.environ.get("FUZZ_OPTS", "").split(",")):  # e.g. FUZZ_OPTS=regroup_entries=0 to bisect a failing seed
        k, v = kv.split("=")
        g.set_option(k, int(v))
    if rng.random() < 0.3:
        g.set_option("chunk_ranges", int(rng.integers(1, 40)))
    if rng.random() < 0.2:
        g.set_option("pair_budget", int(rng.choice([1024, 5000, 100000])))  # levels that outgrow it split the chunk
    ranges = random_ranges(seed + 1, int(rng.integers(100, 600)) if BIG else int(rng.integers(5, 120)), g.num_seqs(), seq_len, max_len=int(min(rng.choice([300, 3000, seq_len // 2]), seq_len - 1)),
                           min_len=int(rng.choice([1, 50, 150])))
    ranges = [(t, s, e) for (t, s, e) in ranges if e > s] or [(0, 0, min(seq_len, 500))]
    kw = {}
    if rng.random() < 0.7:
        kw.update(transitive=True, max_depth=int(rng.choice([1, 2]) if BIG else rng.choice([0, 1, 2, 3, 5])), min_transitive_len=int(rng.choice([0, 10, 101, 500])),
                  min_distance_between_ranges=int(rng.choice([0, 10, 200])))
        if rng.random() < 0.3:
            kw["dfs"] = True
        if kw["max_depth"] == 0 and kw["min_transitive_len"] < 101:
            kw["max_depth"] = 3  # keep unlimited-depth cases from exploding
    if rng.random() < 0.3:
        kw["min_output_length"] = int(rng.choice([0, 100, 1000]))
    if rng.random() < 0.3:
        kw["min_identity"] = float(rng.choice([0.3, 0.7, 0.95]))
    if rng.random() < 0.25:
        kw["multi_impg"] = True
    cigar = bool(rng.random() < 0.5)  # (on a sharded index the slices' ops follow the hits home)
    if rng.random() < 0.3:
        kw["consider_strandness"] = True
    mask = None
    if kw.get("transitive") and rng.random() < 0.35:  # masked_regions: one map for the batch
        mask = {}
        for sid in range(g.num_seqs()):
            if rng.random() < 0.8:
                cuts = np.unique(rng.integers(0, seq_len, size=2 * int(rng.integers(0, 10))))
                rs = [(int(cuts[2 * i]), int(cuts[2 * i + 1])) for i in range(len(cuts) // 2)]
                mask[sid] = (int(seq_len if rng.random() < 0.8 else rng.integers(1, seq_len + 500)), rs)
        cigar = False  # (the oracle's masked entry point returns rows only)
    keep = None
    if rng.random() < 0.25:  # subset filter: the host's per-sequence verdict
        keep = (rng.random(g.num_seqs()) < rng.choice([0.2, 0.6, 0.9])).astype(np.uint8)
        cigar = False
    params = impg_amd.make_params(store_cigar=cigar, **kw)
    try:
        res = g.query_batch(ranges, params, masked_regions=mask, subset_keep=keep)
    except Exception:
        print("FAILED CALL seed", seed, "world", world, "files", n_files, "cigar", cigar, "kw", kw, "mask", mask is not None, "keep", keep is not None, flush=True)
        raise
    total = 0
    for i, (t, s, e) in enumerate(ranges):
        if cigar:
            want, wcg = c.query_cigar(t, s, e, **kw)
            got_cg = res.cigars(i)
            assert [x.tolist() for x in got_cg] == [x.tolist() for x in wcg], ("cigar", seed, i, kw)
        else:
            want = c.query(t, s, e, masked_regions=mask, subset_keep=keep, **kw)
        if res[i].tolist() != want.tolist():
            # Always leave enough behind to tell WHICH side moved: the rows, the set-up, and both sides asked again
            # (a mismatch that does not repeat on a second call is a race or an uninitialised read, and the side
            # whose answer changed is the one that has it).
            gl, wl = res[i].tolist(), want.tolist()
            print("MISMATCH seed", seed, "range", i, (t, s, e), "world", world, "files", n_files, "order", order, "bidir", bidir,
                  "cigar", cigar, "kw", kw, "got", len(gl), "want", len(wl), flush=True)
            for k in range(max(len(gl), len(wl))):
                a = gl[k] if k < len(gl) else None
                b = wl[k] if k < len(wl) else None
                if a != b:
                    print("  row", k, "got", a, "want", b, flush=True)
            for rep in range(3):
                g2 = g.query_batch(ranges, params, masked_regions=mask, subset_keep=keep)[i].tolist()
                w2 = (c.query_cigar(t, s, e, **kw)[0] if cigar else c.query(t, s, e, masked_regions=mask, subset_keep=keep, **kw)).tolist()
                print("  again", rep, "engine same as before:", g2 == gl, " oracle same as before:", w2 == wl, " agree now:", g2 == w2, flush=True)
        assert res[i].tolist() == want.tolist(), ("rows", seed, i, (t, s, e), kw, mask)
        total += c.last_projection_count()
        n_rows += len(want)
    assert res.projected == total, ("projected", seed, kw, world, res.projected, total)
    # the counting form of the same batch (slots in lookup order, pairs regrouped by entry -- or not): per-range counts
    # and checksums must be those of the rows above
    if mask is None and keep is None:
        g.set_option("free_slot_order", int(rng.integers(0, 2)))
        g.set_option("regroup_entries", int(rng.integers(0, 2)))
        st, cnt, ck = g.query_batch_stats(ranges, impg_amd.make_params(**kw))
        assert st.projected == total, ("stats projected", seed, kw, world)
        assert cnt.tolist() == [len(res[i]) - 1 for i in range(len(ranges))], ("stats counts", seed, kw, world)
        assert [int(x) for x in ck] == [checksum(res[i][1:]) for i in range(len(ranges))], ("stats checksums", seed, kw, world)
        g.set_option("free_slot_order", 1)
        g.set_option("regroup_entries", 1)
    # round 6: the same batch with its rows left in HBM (single GPU; plain and BFS without store_cigar / MultiImpg): the
    # attributed layout's rows as multisets per range + its recomputed counts and checksums, both ordered layouts row for row
    if world == 1 and mask is None and keep is None and not kw.get("dfs") and not kw.get("multi_impg") and not cigar:
        pdev = impg_amd.make_params(**kw)
        mol = kw.get("min_output_length") if kw.get("transitive") else None
        dr = g.query_batch_device(ranges, pdev)
        assert dr.projected == total, ("device rows projected", seed, kw)
        got = [[] for _ in ranges]
        for k in range(len(dr.parts())):
            first, level, qid, co, src, fr = dr.part_to_host(k)
            live = qid != np.uint32(0xFFFFFFFF)
            if mol is not None:
                live &= np.abs(co[:, 1].astype(np.int64) - co[:, 0]) >= mol
            f = fr[src[live]]
            for q, r, tg in zip((first + f["range_idx"]).tolist(), np.column_stack([qid[live], co[live]]).tolist(), f["target_id"].tolist()):
                got[q].append((r[0], r[1], r[2], tg, r[3], r[4]))
        for i in range(len(ranges)):
            assert sorted(got[i]) == sorted(tuple(int(x) for x in r) for r in res[i][1:].tolist()), ("device rows", seed, i, kw)
        cnt2, ck2 = dr.check()
        dr.free()
        assert cnt2.tolist() == [len(res[i]) - 1 for i in range(len(ranges))], ("device rows counts", seed, kw)
        assert [int(x) for x in ck2] == [checksum(res[i][1:]) for i in range(len(ranges))], ("device rows checksums", seed, kw)
        for layout in (impg_amd._lib.ROWS_ORDERED, impg_amd._lib.ROWS_ORDERED_SLOTS):
            do = g.query_batch_device(ranges, pdev, layout=layout)
            for k in range(len(do.parts())):
                first, rows, off = do.ordered_to_host(k)
                for j in range(len(off) - 1):
                    r = rows[off[j]:off[j + 1]]
                    assert r[r["query_id"] != 0xFFFFFFFF].tolist() == res[first + j].tolist(), ("ordered rows", layout, seed, first + j, kw)
            do.free()
    # text outputs on the ranges long enough for perform_query's validation
    mtl = kw.get("min_transitive_len", 101)
    ok = [i for i, (t, s, e) in enumerate(ranges) if e - s >= mtl]
    if ok and not kw.get("multi_impg") and mask is None and keep is None:
        sub = [ranges[i] for i in ok]
        d = int(rng.choice([-1, 0, 30, 1000]))
        names = ["n%d" % i for i in ok]
        if cigar:
            r2 = g.query_batch(sub, params)
            for fmt in ("paf", "bedpe"):
                try:
                    want = "".join(c.query_paf(g.seq_name(t), s, e, range_name=names[k], merge_distance=d, fmt=fmt, **kw)
                                   for k, (t, s, e) in enumerate(sub))
                except RuntimeError:
                    continue  # a range with no row to drop: the reference panics
                assert r2.paf(names, merge_distance=d, params=params, fmt=fmt) == want, (fmt, seed, d, kw)
        else:
            r2 = g.query_batch(sub, params)
            want = "".join(c.query_bed(g.seq_name(t), s, e, range_name=names[k], merge_distance=d, **kw) for k, (t, s, e) in enumerate(sub))
            assert r2.bed(names, merge_distance=d, params=params) == want, ("bed", seed, d, kw)
            if True:  # both merges + the text on the device (every rank its own ranges on a sharded index)
                got_dev = g.query_batch_bed(sub, params, merge_distance=d, range_names=names)
                if got_dev != want:
                    a, b = got_dev.splitlines(), want.splitlines()
                    k = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
                    print("device bed differs at line", k, "of", len(a), len(b), "\n got ", a[max(0, k - 2):k + 3], "\n want", b[max(0, k - 2):k + 3])
                assert got_dev == want, ("device bed", seed, d, kw)
    n_cases += 1
    for p in paths:
        os.remove(p)
    if not REPEAT:
        seed += 1
o.set_sorted_visits(False)
print("fuzz ok: %d cases, %d result rows compared, seeds up to %d" % (n_cases, n_rows, seed - 1))

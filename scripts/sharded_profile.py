#!/usr/bin/env python3
"""Where does a sharded step go?  (1 rank, RCCL; GPU box)"""
import os, sys, time, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import impg_amd
from impg_amd import sharded
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
paf = os.path.join(tempfile.gettempdir(), "impg_synth_1000000_seed42.paf")
if not os.path.exists(paf):
    impg_amd.synth_paf_text(paf, 42, 1_000_000)
eng = sharded.ShardedImpg.from_paf(paf, 0, 1, device=0)
eng.chunk_ranges = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
if os.environ.get("IMPG_SLICE"): eng.backend.slice_records = int(os.environ["IMPG_SLICE"])
if os.environ.get("IMPG_NO_LOC"): eng.local.set_option("locality_min", 0)
if os.environ.get("IMPG_REORDER"):  # take the home-side reorder even with one rank: "native" or "argsort"
    eng.always_reorder = True
    if os.environ["IMPG_REORDER"] == "argsort":
        def _argsort(hits, n_front):
            perm = torch.argsort(hits[:, 0].to(torch.int64), stable=True)
            return hits[perm].contiguous()
        eng.backend.reorder = _argsort
T = collections.defaultdict(float)
def timed(obj, name, label=None):
    f = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); T[label or name] += time.perf_counter() - t
        return r
    setattr(obj, name, w)
timed(eng, "_all_to_all_rows"); timed(eng, "_any"); timed(eng.backend, "expand"); timed(eng.backend, "update"); timed(eng.backend, "begin")
timed(eng, "_hop"); timed(eng.backend, "reorder"); timed(eng.backend, "route")
bed = impg_amd.synth_bed(7, 100000)
ids = np.array([eng.local.seq_id(impg_amd.synth_seq_name(i)) for i in range(200)], dtype=np.uint32)
r = np.zeros(len(bed), dtype=impg_amd.RANGE_DTYPE)
r["target_id"] = ids[bed["target_id"]]; r["start"], r["end"] = bed["start"], bed["end"]
d = torch.from_numpy(r.view(np.uint8)).cuda()
p = impg_amd.make_params(transitive=True, max_depth=3)
eng.query_batch_stats(d, len(r), p)
T.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
st = eng.query_batch_stats(d, len(r), p)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("step %.1f ms, projected %d; engine stage ms: lookup %.1f project %.1f update %.1f" % (dt * 1e3, st.projected, st.ms_lookup, st.ms_project, st.ms_update))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-18s %.1f ms" % (k, v * 1e3))
dist.destroy_process_group()

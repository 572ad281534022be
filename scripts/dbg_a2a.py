"""Evidence for ShardedImpg.A2A_ROUND_BYTES: all_to_all_single of one large message on one rank (RCCL 2.26,
ROCm 7.0, MI355X): past 1 GiB the rows that come back differ from the rows sent (observed with both
torch-written and engine-written buffers, depending on the allocation); at or below 0.64 GB they never did."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import impg_amd
from tests.paf_gen import random_paf
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
text, names = random_paf(1, 50, n_seq=5, seq_len=10000)
p = os.path.join(tempfile.gettempdir(), "r.paf"); open(p, "w").write(text)
g = impg_amd.GpuImpg.from_paf(p)
for n in (10_000_000, 40_000_000, 70_000_000, 90_000_000):
    fr = torch.randint(0, 1000, (n, 4), dtype=torch.int32, device="cuda")
    for mode in ("torch-written", "engine-written"):
        if mode == "torch-written":
            src = fr.clone(); src[:, 3] = torch.arange(n, dtype=torch.int32, device="cuda")
        else:
            src = torch.empty_like(fr); torch.cuda.synchronize()
            g.stage_route(fr.data_ptr(), n, 1, src.data_ptr())
        torch.cuda.synchronize()
        out = torch.empty_like(src)
        dist.all_to_all_single(out.view(-1), src.view(-1), output_split_sizes=[n * 4], input_split_sizes=[n * 4])
        torch.cuda.synchronize()
        bad = int((out != src).any(dim=1).sum())
        print("n %d (%.2f GB) %s: a2a mismatching rows %d" % (n, n * 16 / 1e9, mode, bad))
dist.destroy_process_group()

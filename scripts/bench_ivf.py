"""cfg 5 of BASELINE.json at one-GPU scale: IVF-4096 over clustered synthetic unit vectors
(Gaussian centres + noise), nprobe sweep, recall@10 vs the flat kernel on the
same shard, queries/s and the fraction of the shard each batch touches."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd.engine import Engine, HipTimer
from rassengine_amd.ivf import IvfIndex

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--centres", type=int, default=8192)
ap.add_argument("--sigma", type=float, default=1.0)
ap.add_argument("--train-rows", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--queries", type=int, default=1024)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--iid", action="store_true", help="worst case: iid Gaussian rows instead of clusters")
ap.add_argument("--slab", default="f32", help="the IVF's list-ordered copy of the rows: 'f32' | 'bf16' (rass_ivf_build_ex)")
ap.add_argument("--fine-factor", type=int, default=None, help="train_centroids(fine_factor=...): 1 = one level (rounds 1-3a), default: the library's (4)")
ap.add_argument("--seeding", default=None, help="train_centroids(seeding=...): 'random' | 'repair' (default: the library's)")
a = ap.parse_args()

dim = 1024
dev = torch.device("cuda", 0)
eng = Engine(0, dim)
flat = eng.open_index("ivf-bench", capacity_rows=a.rows)
g = torch.Generator(device=dev); g.manual_seed(7)
centres = torch.randn((a.centres, dim), generator=g, device=dev)
centres /= centres.norm(dim=1, keepdim=True)
t0 = time.perf_counter()
chunk = 262144
for lo in range(0, a.rows, chunk):
    n = min(chunk, a.rows - lo)
    if a.iid:
        x = torch.randn((n, dim), generator=g, device=dev)
    else:
        lab = torch.randint(0, a.centres, (n,), generator=g, device=dev)
        x = centres[lab] + a.sigma * torch.randn((n, dim), generator=g, device=dev) / dim ** 0.5
    torch.cuda.synchronize()
    flat.add_device(x.data_ptr(), n, normalize=True)
    eng.synchronize()
gen_s = time.perf_counter() - t0
t0 = time.perf_counter()
from rassengine_amd.ivf import train_centroids
kw = {"seeding": a.seeding} if a.seeding else {}
if a.fine_factor is not None:
    kw["fine_factor"] = a.fine_factor
cent = train_centroids(flat, a.nlist, train_rows=a.train_rows, iters=a.iters, seed=1, **kw)
torch.cuda.synchronize()
train_s = time.perf_counter() - t0
ivf = IvfIndex.build(flat, nlist=a.nlist, centroids=cent, dtype=a.slab)
build_s = time.perf_counter() - t0
sizes = ivf.list_sizes
if a.iid:
    q = torch.randn((a.queries, dim), generator=g, device=dev)
else:
    lab = torch.randint(0, a.centres, (a.queries,), generator=g, device=dev)
    q = centres[lab] + a.sigma * torch.randn((a.queries, dim), generator=g, device=dev) / dim ** 0.5
q = q.contiguous()
torch.cuda.synchronize()
B, k = a.batch, a.k
out_s = torch.empty((B, k), device=dev)
# ground truth: the flat fused scan (itself oracle-validated)
truth = torch.empty((a.queries, k), dtype=torch.int64, device=dev)
stream = eng.stream
tm = HipTimer()
tm.start(stream)
for b in range(0, a.queries, B):
    flat.search_device(q[b:b + B].data_ptr(), B, k, out_s.data_ptr(), truth[b:b + B].data_ptr())
tm.stop(stream)
flat_ms = tm.elapsed_ms()
truth_h = truth.cpu().numpy()
esize = 2 if a.slab == "bf16" else 4
res = {"workload": f"IVF-{a.nlist} over {a.rows} x {dim} {'iid' if a.iid else 'clustered'} rows, {a.slab} slab, top-{k}, batch {B}",
       "gen_s": round(gen_s, 1), "build_s": round(build_s, 1), "train_s": round(train_s, 1), "sigma": a.sigma,
       "list_len_mean": float(sizes.mean()),
       "list_len_max": int(sizes.max()), "empty_lists": int((sizes == 0).sum()),
       "flat_qps": round(a.queries / flat_ms * 1e3, 1), "sweep": []}
got = torch.empty((a.queries, k), dtype=torch.int64, device=dev)
for nprobe in (1, 2, 4, 8, 16, 32, 64, 128):
    for b in range(0, min(a.queries, 4 * B), B):  # warm-up
        ivf.search_device(q[b:b + B].data_ptr(), B, k, nprobe, out_s.data_ptr(), got[b:b + B].data_ptr())
    eng.synchronize()
    tm.start(stream)
    for b in range(0, a.queries, B):
        ivf.search_device(q[b:b + B].data_ptr(), B, k, nprobe, out_s.data_ptr(), got[b:b + B].data_ptr())
    tm.stop(stream)
    ms_groups = tm.elapsed_ms()
    # the same queries in calls of 1 024 (rass_ivf_search_device_batch, round 4): the rate reported
    all_s = torch.empty((a.queries, k), device=dev)
    step = min(1024, a.queries)
    for b in range(0, a.queries, step):
        ivf.search_device_batch(q[b:b + step].data_ptr(), min(step, a.queries - b), k, nprobe, all_s[b:].data_ptr(), got[b:].data_ptr())
    eng.synchronize()
    tm.start(stream)
    for b in range(0, a.queries, step):
        ivf.search_device_batch(q[b:b + step].data_ptr(), min(step, a.queries - b), k, nprobe, all_s[b:].data_ptr(), got[b:].data_ptr())
    tm.stop(stream)
    ms = tm.elapsed_ms()
    got_h = got.cpu().numpy()
    recall = float(np.mean([len(set(got_h[r]) & set(truth_h[r])) / k for r in range(a.queries)]))
    # rows the fine scans touch (the union of a batch's probed lists): the host API reports it; 8 batches sampled
    q_h = q[:8 * B].cpu().numpy()
    _, _, scanned = ivf.search(q_h, k, nprobe)
    scanned_per_batch = scanned / 8
    us_per_batch = ms / (a.queries / B) * 1e3
    probed_bytes = scanned_per_batch * dim * esize + a.nlist * dim * 4          # fine scans + the coarse scan over the centroids
    res["sweep"].append({"nprobe": nprobe, "recall_at_10": round(recall, 4), "qps": round(a.queries / ms * 1e3, 1), "qps_group_by_group": round(a.queries / ms_groups * 1e3, 1),
                         "us_per_batch": round(us_per_batch, 1), "scanned_rows_per_batch": round(scanned_per_batch),
                         "scanned_fraction": round(scanned_per_batch / a.rows, 5),
                         "probed_TBps": round(probed_bytes / (us_per_batch * 1e-6) / 1e12, 3)})
print(json.dumps(res))

"""Experiment: two-level training for corpora with MORE clusters than lists (sigma = 2, 8 192 clusters, IVF-4096): Lloyd from nlist
seeds stalls at recall ~0.8 however long it runs (profiles/r03_ivf4096_2M_sigma2_seeding_experiments.txt).  Over-cluster first
(K' = f x nlist fine lists: a cluster then has a list or two of its own), group the fine means into nlist groups, and give every
row the group of its FINE list."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd.engine import Engine
from rassengine_amd import ivf as I

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--centres", type=int, default=8192)
ap.add_argument("--sigma", type=float, default=2.0)
ap.add_argument("--factors", default="2,4,8")
ap.add_argument("--train-rows", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=8)
a = ap.parse_args()
dim, dev = 1024, torch.device("cuda", 0)
eng = Engine(0, dim)
flat = eng.open_index("probe2", capacity_rows=a.rows)
g = torch.Generator(device=dev); g.manual_seed(7)
centres = torch.randn((a.centres, dim), generator=g, device=dev)
centres /= centres.norm(dim=1, keepdim=True)
for lo in range(0, a.rows, 262144):
    n = min(262144, a.rows - lo)
    lab = torch.randint(0, a.centres, (n,), generator=g, device=dev)
    x = centres[lab] + a.sigma * torch.randn((n, dim), generator=g, device=dev) / dim ** 0.5
    torch.cuda.synchronize(); flat.add_device(x.data_ptr(), n, normalize=True); eng.synchronize()
qlab = torch.randint(0, a.centres, (256,), generator=g, device=dev)
q = (centres[qlab] + a.sigma * torch.randn((256, dim), generator=g, device=dev) / dim ** 0.5).cpu().numpy()
_, truth = flat.search(q, 10)

def recall(ivf):
    out = []
    for nprobe in (1, 4, 16, 64):
        _, ids, _ = ivf.search(q, 10, nprobe)
        out.append(round(float(np.mean([len(set(ids[r]) & set(truth[r])) / 10 for r in range(256)])), 4))
    return out

def group_fine(fine, counts, nlist, iters=15, seed=0):
    """spherical k-means over the fine means (weighted by their list sizes) -> group id per fine list, group means"""
    gg = torch.Generator(device="cpu"); gg.manual_seed(seed)
    K = fine.shape[0]
    w = counts.clamp(min=0).float()
    cent = fine[torch.randperm(K, generator=gg)[:nlist].to(fine.device)].clone()
    for _ in range(iters):
        sim = fine @ cent.T
        grp = sim.argmax(dim=1)
        sums = torch.zeros_like(cent).index_add_(0, grp, fine * w[:, None])
        empty = sums.norm(dim=1) == 0
        sums[empty] = fine[torch.randint(0, K, (int(empty.sum()),), generator=gg).to(fine.device)]
        cent = sums / sums.norm(dim=1, keepdim=True)
    grp = (fine @ cent.T).argmax(dim=1)
    sums = torch.zeros_like(cent).index_add_(0, grp, fine * w[:, None])
    ok = sums.norm(dim=1) > 0
    cent[ok] = sums[ok] / sums[ok].norm(dim=1, keepdim=True)
    return grp, cent

res = {"workload": f"{a.rows} rows, {a.centres} centres, sigma {a.sigma}, IVF-{a.nlist}", "variants": []}
t0 = time.perf_counter()
cent = I.train_centroids(flat, a.nlist, train_rows=a.train_rows, iters=10, seed=1)
ivf = I.IvfIndex.build(flat, nlist=a.nlist, centroids=cent)
res["variants"].append({"training": "one level (library default), 10 iterations", "build_s": round(time.perf_counter() - t0, 1), "recall@10 at nprobe 1/4/16/64": recall(ivf),
                        "list_len_max": int(ivf.list_sizes.max())})
ivf.close()
for f in [int(v) for v in a.factors.split(",")]:
    t0 = time.perf_counter()
    K = f * a.nlist
    fine = I.train_centroids(flat, K, train_rows=a.train_rows, iters=a.iters, seed=1, seeding="random")
    with I._engine_on_torch_stream(flat):
        fa, _ = I.kmeans_assign(flat, fine)                      # every row's fine list
        _, counts = I.kmeans_accumulate(flat, fa, K, 0, 1, -(-flat.rows // 32))
    grp, coarse = group_fine(fine, counts, a.nlist)
    t_train = time.perf_counter() - t0
    assign_fine = grp[fa[:flat.rows].long()].to(torch.int32).cpu().numpy()
    for name, assign in (("rows follow their FINE list's group", assign_fine), ("rows go to the best COARSE mean", None)):
        t1 = time.perf_counter()
        ivf = I.IvfIndex.build(flat, nlist=a.nlist, centroids=coarse, assign=assign)
        res["variants"].append({"training": f"two levels, K' = {f} x nlist, {a.iters} fine iterations; {name}", "train_s": round(t_train, 1),
                                "build_s": round(time.perf_counter() - t1, 1), "recall@10 at nprobe 1/4/16/64": recall(ivf),
                                "list_len_max": int(ivf.list_sizes.max()), "empty_lists": int((ivf.list_sizes == 0).sum())})
        ivf.close()
    print(json.dumps(res["variants"][-2:]), flush=True)
print(json.dumps(res))

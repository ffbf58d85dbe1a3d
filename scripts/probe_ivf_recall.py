"""Why is IVF recall what it is on the wide-cluster corpus?  The probe's SEMANTICS are pinned
elsewhere (tests/test_gpu_cfg5.py: result == brute force restricted to the nprobe best lists), so recall is a
property of the list structure.  This script measures that structure on the bench's data (8 192 Gaussian centres,
x = c + sigma * N(0,1)/sqrt(d), IVF-4096):

  cohesion   for every TRUE cluster, the share of its rows that sit in the cluster's majority list
             (1.0 = k-means kept the cluster together; a query can then find all its neighbours in one list)
  hit@p      how often the majority list of the query's own cluster is among the query's p best lists
  recall@10  vs the flat scan, per nprobe

for several training budgets.  torch only (no oracle): flat-scan truth comes from the engine itself."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd.engine import Engine
from rassengine_amd.ivf import IvfIndex, train_centroids

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2_000_000)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--centres", type=int, default=8192)
ap.add_argument("--sigma", type=float, default=1.0)
ap.add_argument("--queries", type=int, default=256)
a = ap.parse_args()
dim, dev = 1024, torch.device("cuda", 0)
eng = Engine(0, dim)
flat = eng.open_index("probe", capacity_rows=a.rows)
g = torch.Generator(device=dev); g.manual_seed(7)
centres = torch.randn((a.centres, dim), generator=g, device=dev)
centres /= centres.norm(dim=1, keepdim=True)
labs = []
for lo in range(0, a.rows, 262144):
    n = min(262144, a.rows - lo)
    lab = torch.randint(0, a.centres, (n,), generator=g, device=dev)
    x = centres[lab] + a.sigma * torch.randn((n, dim), generator=g, device=dev) / dim ** 0.5
    torch.cuda.synchronize()
    flat.add_device(x.data_ptr(), n, normalize=True)
    eng.synchronize()
    labs.append(lab.cpu())
lab = torch.cat(labs).numpy()
qlab = torch.randint(0, a.centres, (a.queries,), generator=g, device=dev)
q = (centres[qlab] + a.sigma * torch.randn((a.queries, dim), generator=g, device=dev) / dim ** 0.5).cpu().numpy()
qlab = qlab.cpu().numpy()
_, truth = flat.search(q, 10)
same_cluster = float(np.mean(lab[truth] == qlab[:, None]))
out = {"workload": f"{a.rows} rows, {a.centres} centres, sigma {a.sigma}, IVF-{a.nlist}",
       "true_top10_in_query_cluster": round(same_cluster, 4), "variants": []}
for name, train_rows, iters, seeding in (("random seeds, 1M-row sample, 10 iterations (rounds 1-2)", 1_000_000, 10, "random"),
                                         ("repair, 1M-row sample, 10 iterations", 1_000_000, 10, "repair"),
                                         ("random seeds, all rows, 30 iterations", 0, 30, "random"),
                                         ("repair, all rows, 30 iterations", 0, 30, "repair")):
    t0 = time.perf_counter()
    cent = train_centroids(flat, a.nlist, train_rows=train_rows, iters=iters, seed=1, seeding=seeding)
    ivf = IvfIndex.build(flat, nlist=a.nlist, centroids=cent)
    build_s = time.perf_counter() - t0
    assign = ivf.assign
    # cohesion: rows of cluster c in its majority list
    key = lab.astype(np.int64) * a.nlist + assign
    uniq, cnt = np.unique(key, return_counts=True)
    cl = uniq // a.nlist
    order = np.lexsort((cnt, cl))
    last = np.r_[cl[order][1:] != cl[order][:-1], True]
    maj_list = np.full(a.centres, -1, dtype=np.int64); maj_cnt = np.zeros(a.centres)
    maj_list[cl[order][last]] = (uniq[order] % a.nlist)[last]; maj_cnt[cl[order][last]] = cnt[order][last]
    size = np.bincount(lab, minlength=a.centres)
    cohesion = float((maj_cnt[size > 0] / size[size > 0]).mean())
    lists_per_cluster = float(np.bincount(cl, minlength=a.centres)[size > 0].mean())
    cn = (cent / cent.norm(dim=1, keepdim=True)).cpu().numpy()
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    coarse = np.argsort(-(qn @ cn.T), axis=1)
    row = {"training": name, "build_s": round(build_s, 1), "cluster_cohesion": round(cohesion, 4),
           "lists_per_cluster": round(lists_per_cluster, 2), "list_len_max": int(ivf.list_sizes.max()), "sweep": []}
    for nprobe in (1, 4, 16, 64, 128):
        _, ids, scanned = ivf.search(q, 10, nprobe)
        rec = float(np.mean([len(set(ids[r]) & set(truth[r])) / 10 for r in range(a.queries)]))
        hit = float(np.mean([maj_list[qlab[r]] in coarse[r, :nprobe] for r in range(a.queries)]))
        row["sweep"].append({"nprobe": nprobe, "recall_at_10": round(rec, 4), "own_cluster_list_probed": round(hit, 4),
                             "scanned_fraction_per_batch": round(scanned / (a.queries / 32) / a.rows, 5)})
    out["variants"].append(row)
    ivf.close()
print(json.dumps(out))

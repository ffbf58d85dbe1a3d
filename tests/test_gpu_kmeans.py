"""K9(i) on the GPU: the k-means kernels of the IVF build (csrc/kmeans.hip) through the C ABI against plain fp32
torch references of the same ops (a floating-point kernel: the one place a torch reference is the yardstick).

  * rass_kmeans_assign: row -> arg max cosine over the centroids.  The chosen list's score must be the row's
    maximum to within fp32 summation order (1e-6 on unit vectors), equal to torch's argmax wherever the runner-up
    is further away than that, and the reported best score within 2e-6 of fp64.
  * rass_kmeans_accumulate: per-list sums / counts equal torch's index_add (atomics reorder the fp32 adds: 1e-4
    relative).
  * train_centroids: one iteration reproduces a torch Lloyd step from the same seeds; empty lists are re-seeded;
    the sampled variant touches only the sampled blocks.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dim,nlist,step", [(1000, 1024, 64, 1), (4099, 1024, 500, 1), (70000, 1024, 4096, 1),
                                              (5000, 256, 33, 1), (9000, 640, 100, 3), (33, 128, 7, 1)])
def test_assign_and_accumulate_match_torch(gpu, n, dim, nlist, step):
    torch = gpu
    from rassengine_amd import ivf
    from rassengine_amd.engine import Engine
    g = torch.Generator(device="cpu")
    g.manual_seed(n + nlist)
    x = torch.randn((n, dim), generator=g)
    cent = torch.randn((nlist, dim), generator=g)
    cent[nlist // 2] = x[5]                      # a row that IS a centroid: cosine 1
    eng = Engine(0, dim)
    try:
        idx = eng.open_index("km", capacity_rows=n)       # capacity % 32 may be 16: the half-block guard
        idx.add(x.numpy(), normalize=True)
        cd = cent.cuda()
        total_blocks = -(-n // 32)
        n_blocks = -(-total_blocks // step)
        with ivf._engine_on_torch_stream(idx):
            assign, best, _slab = ivf.kmeans_assign(idx, cd, 0, step, n_blocks, with_best=True)
            sums, counts = ivf.kmeans_accumulate(idx, assign, nlist, 0, step, n_blocks)
            torch.cuda.synchronize()
        rows = (torch.arange(n_blocks * 32) // 32) * step * 32 + torch.arange(n_blocks * 32) % 32
        valid = rows < n
        xs = torch.nn.functional.normalize(x.double(), dim=1)[rows[valid]]
        cn = torch.nn.functional.normalize(cent.double(), dim=1)
        scores = xs @ cn.T                                        # fp64 truth
        top2 = scores.topk(min(2, nlist), dim=1)
        a = assign.cpu()[valid].long()
        chosen = scores.gather(1, a[:, None])[:, 0]
        assert bool((a >= 0).all()) and bool((a < nlist).all())
        assert bool((top2.values[:, 0] - chosen <= 1e-6).all()), float((top2.values[:, 0] - chosen).max())
        clear = (top2.values[:, 0] - top2.values[:, -1]) > 2e-6 if nlist > 1 else torch.ones_like(chosen, dtype=torch.bool)
        assert bool((a[clear] == top2.indices[:, 0][clear]).all())
        assert bool(((best.cpu()[valid].double() - chosen).abs() <= 2e-6).all())
        if step == 1 and n > 5:
            assert int(a[5]) == nlist // 2 and abs(float(best[5]) - 1.0) < 1e-6
        # accumulate vs index_add on the very assignment the kernel produced
        ref_s = torch.zeros((nlist, dim), dtype=torch.float64).index_add_(0, a, xs)
        ref_c = torch.bincount(a, minlength=nlist).double()
        assert torch.equal(counts.cpu().double(), ref_c)
        err = (sums.cpu().double() - ref_s).abs().max()
        assert float(err) <= 1e-4 * max(1.0, float(ref_s.abs().max())), float(err)
    finally:
        eng.close()


def test_train_centroids_is_a_lloyd_iteration_and_reseeds_empty_lists(gpu):
    torch = gpu
    from rassengine_amd import ivf
    from rassengine_amd.engine import Engine
    rng = np.random.default_rng(0)
    centres = rng.standard_normal((20, 256)).astype(np.float32)
    x = (centres[rng.integers(0, 20, size=6000)] + 0.3 * rng.standard_normal((6000, 256))).astype(np.float32)
    eng = Engine(0, 256)
    try:
        idx = eng.open_index("km-train")
        idx.add(x, normalize=True)
        c0 = ivf.train_centroids(idx, nlist=20, iters=0, seed=3, seeding="random", fine_factor=1)   # the seeds
        c1 = ivf.train_centroids(idx, nlist=20, iters=1, seed=3, seeding="random", fine_factor=1)
        xn = torch.nn.functional.normalize(torch.from_numpy(x), dim=1)
        lab = (xn @ c0.cpu().T).argmax(dim=1)
        sums = torch.zeros((20, 256)).index_add_(0, lab, xn)
        ref = sums / (sums.norm(dim=1, keepdim=True) + 1e-9)
        live = torch.bincount(lab, minlength=20) > 0
        assert torch.allclose(c1.cpu()[live], ref[live], atol=1e-5)
        assert torch.allclose(c1.norm(dim=1).cpu(), torch.ones(20), atol=1e-5)    # re-seeded rows are unit too
        c8 = ivf.train_centroids(idx, nlist=20, iters=8, seed=3, seeding="random", fine_factor=1)
        assign = ivf.assign_rows(idx, c8)
        assert assign.shape == (6000,) and assign.dtype == np.int32
        # k-means found the 20 planted clusters: every list is (almost) pure
        truth = (xn @ torch.nn.functional.normalize(torch.from_numpy(centres), dim=1).T).argmax(dim=1).numpy()
        purity = np.mean([np.bincount(truth[assign == l], minlength=20).max() / max(1, (assign == l).sum())
                          for l in range(20) if (assign == l).any()])
        assert purity > 0.9, purity
        # a strided sample (every 4th block) trains too and more lists than clusters leaves no NaN behind
        c_s = ivf.train_centroids(idx, nlist=64, train_rows=1500, iters=3, seed=1)
        assert bool(torch.isfinite(c_s).all()) and torch.allclose(c_s.norm(dim=1).cpu(), torch.ones(64), atol=1e-5)
    finally:
        eng.close()


def test_repair_merges_split_clusters_and_covers_missed_ones(gpu):
    """train_centroids(seeding="repair"): with as many lists as planted clusters, random seeds leave ~1/e of the clusters
    without a seed of their own and put two or more into others (which Lloyd then splits for good: it cannot move a
    centre out of a cluster it shares); merging near-duplicate means and re-seeding the freed centroids where no mean is
    close gives (almost) every cluster a list of its own."""
    torch = gpu
    from rassengine_amd import ivf
    from rassengine_amd.engine import Engine
    rng = np.random.default_rng(4)
    n_clusters, dim = 128, 256
    centres = rng.standard_normal((n_clusters, dim)).astype(np.float32)
    lab = rng.integers(0, n_clusters, size=40000)
    x = (centres[lab] + 0.5 * rng.standard_normal((40000, dim))).astype(np.float32)
    eng = Engine(0, dim)
    try:
        idx = eng.open_index("km-seed")
        idx.add(x, normalize=True)
        cn = torch.nn.functional.normalize(torch.from_numpy(centres), dim=1)

        def covered(cent):
            """planted clusters that are the nearest planted centre of at least one trained centroid"""
            owner = (torch.nn.functional.normalize(cent.cpu(), dim=1) @ cn.T).argmax(dim=1)
            return len(set(owner.tolist())) / n_clusters

        def cohesion(cent):
            a = ivf.assign_rows(idx, cent)
            return float(np.mean([np.bincount(a[lab == c]).max() / max(1, (lab == c).sum()) for c in range(n_clusters)]))

        c_rand = ivf.train_centroids(idx, nlist=n_clusters, iters=9, seed=2, seeding="random", fine_factor=1)
        c_rep = ivf.train_centroids(idx, nlist=n_clusters, iters=9, seed=2, seeding="repair", fine_factor=1)
        assert c_rep.shape == (n_clusters, dim) and torch.allclose(c_rep.norm(dim=1).cpu(), torch.ones(n_clusters), atol=1e-5)
        cov_r, cov_p = covered(c_rand), covered(c_rep)
        coh_r, coh_p = cohesion(c_rand), cohesion(c_rep)
        print(f"clusters with a centroid of their own: random seeds {cov_r:.3f}, with repair {cov_p:.3f}; "
              f"cluster cohesion {coh_r:.3f} -> {coh_p:.3f}")
        assert cov_p >= 0.97 and cov_p > cov_r and coh_p > coh_r
        assert len(eng._indices) == 1                  # the scratch index of the repair is gone
        # the default (two levels: 4 x nlist fine lists grouped into nlist) keeps these tight clusters whole as well
        c_two = ivf.train_centroids(idx, nlist=n_clusters, iters=9, seed=2)
        assert c_two.shape == (n_clusters, dim) and torch.allclose(c_two.norm(dim=1).cpu(), torch.ones(n_clusters), atol=1e-5)
        coh_t = cohesion(c_two)
        print(f"two-level default: clusters with a centroid of their own {covered(c_two):.3f}, cohesion {coh_t:.3f}")
        assert coh_t >= 0.97
        with pytest.raises(ValueError):
            ivf.train_centroids(idx, nlist=16, iters=1, seeding="kmeans++")
    finally:
        eng.close()


def test_two_level_training_keeps_clusters_whole_when_there_are_more_clusters_than_lists(gpu):
    """The corpus shape of SURVEY §8d in small: twice as many planted clusters as lists, noisy rows (a row's cosine to its
    cluster's centre 0.45, two rows of one cluster 0.2).  One-level Lloyd from nlist seeds scatters the clusters that got no
    seed over all lists and stalls; the two-level default (fine_factor = 4) deals whole fine lists to the coarse lists: a
    cluster's rows end up together, and the 10 nearest neighbours of a query are found in its best list."""
    torch = gpu
    from rassengine_amd import ivf
    from rassengine_amd.engine import Engine
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    n_clusters, nlist, dim, n = 4096, 2048, 1024, 1_000_000
    centres = torch.nn.functional.normalize(torch.randn((n_clusters, dim), generator=g, device="cuda"), dim=1)
    lab = torch.randint(0, n_clusters, (n,), generator=g, device="cuda")
    qlab = torch.randint(0, n_clusters, (64,), generator=g, device="cuda")
    q = (centres[qlab] + 2.0 * torch.randn((64, dim), generator=g, device="cuda") / dim ** 0.5).cpu().numpy()
    lab_h = lab.cpu().numpy()
    eng = Engine(0, dim)
    try:
        idx = eng.open_index("km-two-level", capacity_rows=n)
        for lo in range(0, n, 250_000):           # 1 M rows x 1024 = 4 GB in the index; generated a quarter at a time
            x = centres[lab[lo:lo + 250_000]] + 2.0 * torch.randn((250_000, dim), generator=g, device="cuda") / dim ** 0.5
            torch.cuda.synchronize()
            idx.add_device(x.data_ptr(), 250_000, normalize=True)
            eng.synchronize()
        del x
        _, truth = idx.search(q, 10)

        def measure(cent):
            iv = ivf.IvfIndex.build(idx, nlist=nlist, centroids=cent)
            try:
                a = iv.assign
                key = lab_h.astype(np.int64) * nlist + a          # (cluster, list) pairs: the majority list's share per cluster
                uniq, cnt = np.unique(key, return_counts=True)
                best = np.zeros(n_clusters)
                np.maximum.at(best, uniq // nlist, cnt)
                coh = float(np.mean(best / np.maximum(1, np.bincount(lab_h, minlength=n_clusters))))
                _, got, _ = iv.search(q, 10, nprobe=1)
                rec = float(np.mean([len(set(got[r]) & set(truth[r])) / 10 for r in range(64)]))
                return coh, rec, int(iv.list_sizes.max())
            finally:
                iv.close()

        one = measure(ivf.train_centroids(idx, nlist=nlist, train_rows=500_000, iters=8, seed=1, fine_factor=1))
        two = measure(ivf.train_centroids(idx, nlist=nlist, train_rows=500_000, iters=8, seed=1))
        print(f"cohesion / recall@10 at nprobe 1 / longest list: one level {one}, two levels {two}")
        assert two[0] >= 0.9 and two[1] >= 0.9
        assert two[0] > one[0] + 0.05 and two[1] > one[1] + 0.05
        assert two[2] <= 8 * n // nlist
    finally:
        eng.close()

"""Multi-rank sharded search on real HIP shards: 2 ranks share the one GPU of the test box
(gloo rendezvous; the collectives move device tensors), each owns half of the rows; every rank
must end with exactly the single-index result (ids AND scores bit-identical) — the invariant
the 8-GPU RCCL path relies on."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_local, out_dir, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    # gloo: both ranks share GPU 0 (1-GPU test box); nccl (= RCCL over xGMI): one GPU per rank
    device = rank if backend == "nccl" else 0
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd.dist import HipShard, ShardedSearch
        from rassengine_amd.engine import Engine
        eng = Engine(device, 1024)
        idx = eng.open_index("shard", capacity_rows=n_local)
        idx.fill_synthetic(n_local, seed=77, row_id_base=rank * n_local)
        eng.synchronize()
        search = ShardedSearch(HipShard(idx, id_base=rank * n_local))
        g = torch.Generator(device="cpu")
        g.manual_seed(5)
        q_all = torch.randn((20, 1024), generator=g).cuda()
        q = q_all.clone() if rank == 0 else torch.zeros_like(q_all)   # only rank 0 holds the batch
        s, i = search.search(q, 10)
        torch.cuda.synchronize()
        # a 96-query batch in 3 launch groups: 2 collectives in all, same answers as group-by-group
        qb_all = torch.randn((96, 1024), generator=g).cuda()
        qb = qb_all.clone() if rank == 0 else torch.zeros_like(qb_all)
        sb, ib = search.search_batch(qb, 10, 32)
        for j in range(3):
            sj, ij = search.search(qb[32 * j:32 * j + 32].clone(), 10, broadcast=False)
            assert torch.equal(ij, ib[32 * j:32 * j + 32]) and torch.equal(sj, sb[32 * j:32 * j + 32])
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.cpu().numpy(), i=i.cpu().numpy(), q=q.cpu().numpy())
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_sharded_search_equals_single_index(gpu, tmp_path, backend):
    """gloo: two ranks on the one GPU of the test box.  nccl: BASELINE cfg 4's RCCL leg (query broadcast +
    ONE all-gather of packed per-shard top-k over xGMI) on two real GPUs — runs wherever >= 2 are visible,
    skipped on a 1-GPU box.  Same assertion for both: every rank == the single-index result, bit for bit."""
    import torch.multiprocessing as mp
    from rassengine_amd.engine import Engine
    if backend == "nccl" and gpu.cuda.device_count() < 2:
        pytest.skip("RCCL leg needs >= 2 GPUs (the driver's multi-GPU node); 1 visible here")
    n_local, world = 30000, 2
    mp.spawn(_worker, args=(world, _free_port(), n_local, str(tmp_path), backend), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    assert np.array_equal(r0["q"], r1["q"])                       # broadcast reached rank 1
    assert np.array_equal(r0["i"], r1["i"]) and np.array_equal(r0["s"], r1["s"])
    eng = Engine(0, 1024)
    try:
        whole = eng.open_index("whole", capacity_rows=world * n_local)
        whole.fill_synthetic(world * n_local, seed=77, row_id_base=0)   # Philox rows keyed by global id
        s, i = whole.search(r0["q"], 10)
    finally:
        eng.close()
    assert np.array_equal(r0["i"], i)
    assert np.array_equal(r0["s"], s)
    assert (r0["i"] >= n_local).any() and (r0["i"] < n_local).any()   # hits come from both shards


def _ivf_worker(rank, world, port, n_local, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd.dist import ShardedSearch
        from rassengine_amd.engine import Engine
        from rassengine_amd.ivf import IvfIndex, IvfShard, train_centroids
        torch.cuda.set_device(0)
        eng = Engine(0, 1024)
        idx = eng.open_index("shard", capacity_rows=n_local)
        idx.fill_synthetic(n_local, seed=78, row_id_base=rank * n_local)
        idx.delete(3)                                       # one tombstone per shard
        eng.synchronize()
        # shared centroids: k-means sums all-reduced over the ranks (every rank ends with the same ones)
        cent = train_centroids(idx, nlist=64, iters=4, seed=2)
        ivf = IvfIndex.build(idx, nlist=64, centroids=cent)
        g = torch.Generator(device="cuda")
        g.manual_seed(6)
        q_all = torch.randn((12, 1024), generator=g, device="cuda")
        q = q_all.clone() if rank == 0 else torch.zeros_like(q_all)
        out = {"cent": cent.cpu().numpy()}
        for nprobe in (64, 4):
            s, i = ShardedSearch(IvfShard(ivf, id_base=rank * n_local, nprobe=nprobe)).search(q, 10)
            torch.cuda.synchronize()
            out[f"s{nprobe}"] = s.cpu().numpy()
            out[f"i{nprobe}"] = i.cpu().numpy()
        # the same lists over a bf16 slab (rass_ivf_build_ex): every list probed, merged over the ranks
        ivf_b = IvfIndex.build(idx, nlist=64, centroids=cent, dtype="bf16")
        s, i = ShardedSearch(IvfShard(ivf_b, id_base=rank * n_local, nprobe=64)).search(q, 10)
        torch.cuda.synchronize()
        out["sb"] = s.cpu().numpy()
        out["ib"] = i.cpu().numpy()
        ivf_b.close()
        out["q"] = q.cpu().numpy()
        np.savez(os.path.join(out_dir, f"ivf_rank{rank}.npz"), **out)
        ivf.close()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_rank_ivf_shards_share_centroids_and_match_flat(gpu, tmp_path):
    """cfg 5's structure on 2 ranks: centroids trained with an all-reduce, per-rank inverted lists
    over the rank's own rows, per-shard top-k merged after one all-gather.  Probing every list
    must reproduce the flat single-index answer (tombstones included); a partial probe must be
    identical on both ranks and contain only true scores."""
    import torch.multiprocessing as mp
    from rassengine_amd.engine import Engine
    n_local, world = 20000, 2
    mp.spawn(_ivf_worker, args=(world, _free_port(), n_local, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(os.path.join(str(tmp_path), "ivf_rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "ivf_rank1.npz"))
    assert np.array_equal(r0["cent"], r1["cent"])
    for key in ("i64", "s64", "i4", "s4", "q"):
        assert np.array_equal(r0[key], r1[key]), key
    eng = Engine(0, 1024)
    try:
        whole = eng.open_index("whole", capacity_rows=world * n_local)
        whole.fill_synthetic(world * n_local, seed=78, row_id_base=0)
        whole.delete(3)
        whole.delete(n_local + 3)
        s, i = whole.search(r0["q"], 10)
    finally:
        eng.close()
    assert np.array_equal(r0["i64"], i) and np.array_equal(r0["s64"], s)
    # bf16 slabs on both ranks == ONE flat bf16 index over all rows (same rows rounded to bf16, same kernel arithmetic)
    assert np.array_equal(r0["ib"], r1["ib"]) and np.array_equal(r0["sb"], r1["sb"])
    eng = Engine(0, 1024)
    try:
        whole_b = eng.open_index("whole-b", capacity_rows=world * n_local, dtype="bf16")
        whole_b.fill_synthetic(world * n_local, seed=78, row_id_base=0)
        whole_b.delete(3)
        whole_b.delete(n_local + 3)
        sb, ib = whole_b.search(r0["q"], 10)
    finally:
        eng.close()
    assert np.array_equal(r0["ib"], ib) and np.array_equal(r0["sb"], sb)
    truth = {(qq, int(a)): float(b) for qq in range(i.shape[0]) for a, b in zip(i[qq], s[qq])}
    hits = 0
    for qq in range(i.shape[0]):
        for a, b in zip(r0["i4"][qq], r0["s4"][qq]):
            if (qq, int(a)) in truth:
                assert truth[(qq, int(a))] == float(b)
                hits += 1
    assert hits > 0


def _serving_worker(rank, world, port, out_dir, backend, prefilter="off"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    device = rank if backend == "nccl" else 0
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import config, embedding, indexer, serving
        from rassengine_amd.docstore import REGISTRY
        from rassengine_amd.engine import Engine
        from tests.helpers import HashEmbedder
        from tests.test_serving_gloo import _scenario
        config.RASS_PREFILTER = prefilter       # the shards' candidate mode (hip_shard_factory / hip_shard_loader read it)
        front = serving.start(serving.hip_shard_factory(device, 1024), 1024, torch.device("cuda", device),
                              shard_loader=serving.hip_shard_loader(device, 1024))
        if rank != 0:
            assert front is None
            open(os.path.join(out_dir, f"worker{rank}.done"), "w").write("ok")
            return
        embedding.set_embedder(HashEmbedder(1024))
        sharded = _scenario(indexer, embedding, REGISTRY, config, "rass-idx-user1")
        idx = REGISTRY.get("rass-idx-user1").index
        assert isinstance(idx, serving.ShardedIndex) and sorted(set(idx._owner_rank)) == [0, 1]
        # persistence through the front on real shard files (rass_index_save keeps the global ids)
        import asyncio
        from rassengine_amd.docstore import IndexState
        st = REGISTRY.get("rass-idx-user1")
        prefix = os.path.join(out_dir, "saved-user1")
        st.save(prefix)
        st2 = IndexState.load("rass-idx-restored", prefix, front.load_index)
        qv = asyncio.run(embedding.embed_query("chunk number 12 about topic5 and drug0"))
        a, b = st.index.search(qv, 10), st2.index.search(qv, 10)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        assert st2.index.count == st.index.count and st2.index.rows == st.index.rows
        assert np.array_equal(st2.index.get_row(int(a[1][0, 0])), st.index.get_row(int(a[1][0, 0])))
        front.shutdown()
        # the same scenario on ONE HIP index in this process (always the exact scan)
        config.RASS_PREFILTER = "off"
        REGISTRY.clear()
        eng = Engine.get(device, 1024)
        REGISTRY.set_index_factory(lambda name: eng.open_index("single-" + name))
        single = _scenario(indexer, embedding, REGISTRY, config, "rass-idx-user1")
        np.savez(os.path.join(out_dir, "serving.npz"), **{"sharded_" + k: np.asarray(v) for k, v in sharded.items()},
                 **{"single_" + k: np.asarray(v) for k, v in single.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_shim_over_sharded_hip_index_equals_single_hip_index(gpu, tmp_path, backend):
    """rassengine_amd.serving on real HIP shards: rank 0 serves HipIndexer / store_fhir_docs_in_opensearch over
    a ShardedIndex (rows dealt round-robin by batch, global ids assigned through rass_index_add_ex), rank 1
    follows in worker_loop; everything the shim returns must equal the single-index run bit for bit."""
    import torch.multiprocessing as mp
    if backend == "nccl" and gpu.cuda.device_count() < 2:
        pytest.skip("RCCL leg needs >= 2 GPUs; 1 visible here")
    mp.spawn(_serving_worker, args=(2, _free_port(), str(tmp_path), backend), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "worker1.done"))       # clean collective shutdown
    z = np.load(os.path.join(str(tmp_path), "serving.npz"))
    keys = sorted(k[len("single_"):] for k in z.files if k.startswith("single_"))
    for k in keys:
        a, b = z["sharded_" + k], z["single_" + k]
        if a.dtype.kind == "f":
            assert a.shape == b.shape and np.array_equal(a, b), k
        else:
            assert a.tolist() == b.tolist(), (k, a, b)
    assert int(z["single_count"]) == 90 and int(z["single_rows"]) == 92 and len(z["single_sem_ids"]) == 10


@pytest.mark.parametrize("prefilter", ["int8", "bf16"])
def test_sharded_shards_in_a_prefilter_mode_equal_the_single_exact_index(gpu, tmp_path, prefilter):
    """RASS_PREFILTER on a multi-GPU front: every shard scans its int8 (bf16) copy for candidates — with caller-assigned
    global ids, masked doc_type / patient filters and tombstones in play — and rescores them exactly; the whole scenario
    (uploads, overwrites, every search shape of the shim, save / load) still equals ONE exact index bit for bit."""
    import torch.multiprocessing as mp
    mp.spawn(_serving_worker, args=(2, _free_port(), str(tmp_path), "gloo", prefilter), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "worker1.done"))
    z = np.load(os.path.join(str(tmp_path), "serving.npz"))
    for k in sorted(k[len("single_"):] for k in z.files if k.startswith("single_")):
        a, b = z["sharded_" + k], z["single_" + k]
        if a.dtype.kind == "f":
            assert a.shape == b.shape and np.array_equal(a, b), k
        else:
            assert a.tolist() == b.tolist(), (k, a, b)


def _ivf_docs(n, start=0):
    return [{"doc_id": f"n-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
             "unstructuredText": f"note {i} topic{i % 13} drug{i % 7} ward{i % 5}"} for i in range(start, start + n)]


def _ivf_serving_scenario(indexer, embedding, REGISTRY, name, build):
    """Uneven uploads, then `build(index)` (the sharded run builds its IVFs there), then MORE uploads (the flat delta),
    an overwrite of a covered doc and of a delta doc; returns comparable plain data."""
    import asyncio
    docs = _ivf_docs(3600)
    a = 0
    for n in (700, 300, 700, 300, 700, 300):                 # round-robin by batch: rank 0 holds 2 100 rows, rank 1 900
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs[a:a + n], None, name))
        a += n
    build(REGISTRY.get(name).index)
    for n in (250, 350):                                     # after the build: the delta, on both ranks
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs[a:a + n], None, name))
        a += n
    asyncio.run(indexer.store_fhir_docs_in_opensearch(
        [], [dict(docs[7], unstructuredText="entirely new words here"), dict(docs[3100], unstructuredText="other words")],
        None, name))
    ix = indexer.HipIndexer(None, name)
    out = {}
    for key, text, k, kw in (("a", "note 77 topic12 drug0 ward2", 10, {}), ("new", "entirely new words here", 5, {}),
                             ("delta", "note 3333 topic5 drug1 ward3", 8, {}), ("pat", "note 300 topic1 drug6 ward0", 10, {"patient_id": "p0"}),
                             ("deep", "note 12 topic12 drug5 ward2", 50, {}), ("other", "other words", 3, {"patient_id": "p1"})):
        h = ix.semantic_search(asyncio.run(embedding.embed_query(text)), k=k, **kw)
        out[key + "_ids"] = [d["doc_id"] for d, _ in h]
        out[key + "_scores"] = [float(x) for _, x in h]
    st = REGISTRY.get(name)
    out["count"], out["rows"] = int(st.index.count), int(st.index.rows)
    return out


def _serving_ivf_worker(rank, world, port, out_dir, ivf_dtype="f32"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import config, embedding, indexer, serving
        from rassengine_amd.docstore import REGISTRY, IndexState
        from rassengine_amd.engine import Engine
        from tests.helpers import HashEmbedder
        config.RASS_KNN_PREFETCH = 0
        config.RASS_IVF_DTYPE = ivf_dtype                  # the slab dtype every rank builds (ShardedIndex.build_ivf's default)
        front = serving.start(serving.hip_shard_factory(0, 1024), 1024, torch.device("cuda", 0),
                              shard_loader=serving.hip_shard_loader(0, 1024))
        if rank != 0:
            assert front is None
            open(os.path.join(out_dir, f"ivfworker{rank}.done"), "w").write("ok")
            return
        import asyncio
        embedding.set_embedder(HashEmbedder(1024))
        name = "rass-idx-ivf"
        # nlist 32 on shards of 2 100 and 900 rows: sized from the rank-local rows the two ranks would disagree on one vs
        # two training levels (2 100 // 16 = 131 -> 128 fine lists; 900 // 16 = 56 < 2 x 32 -> one level) and hang in
        # mismatched collectives; every list probed => the answers must equal ONE flat index bit for bit
        sharded = _ivf_serving_scenario(indexer, embedding, REGISTRY, name, lambda index: index.build_ivf(nlist=32, nprobe=32))
        idx = REGISTRY.get(name).index
        assert isinstance(idx, serving.ShardedIndex) and idx._ivf_covered == 3000 and idx.rows == 3602
        # a partial probe: every hit carries its true score, the delta is always scanned
        idx.build_ivf(nlist=32, nprobe=2)
        assert idx._ivf_builds == 2 and idx._ivf_covered == 3602
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], _ivf_docs(40, 5000), None, name))
        q = asyncio.run(embedding.embed_query("note 5017 topic12 drug5 ward2"))
        part = idx.search(q, 10)
        full = idx.search(q, 40)                             # k > 32: the exact flag -> flat shards
        truth = {int(i): float(sv) for i, sv in zip(full[1][0], full[0][0])}
        assert int(part[1][0, 0]) == int(full[1][0, 0]) == 3602 + 17           # the delta row finds itself
        common = [int(i) for i in part[1][0] if int(i) in truth]
        assert common and all(truth[i] == float(sv) for i, sv in zip(part[1][0], part[0][0]) if int(i) in truth)
        # persistence: every rank saves its rows AND its IVF; the restored index answers like the live one
        st = REGISTRY.get(name)
        prefix = os.path.join(out_dir, "saved-ivf")
        st.save(prefix)
        assert len([f for f in os.listdir(out_dir) if f.endswith(".ivf")]) == world
        st2 = IndexState.load("rass-idx-ivf-restored", prefix, front.load_index)
        b = st2.index.search(q, 10)
        assert np.array_equal(part[1], b[1]) and np.array_equal(part[0], b[0])
        front.shutdown()
        REGISTRY.clear()
        eng = Engine.get(0, 1024)
        REGISTRY.set_index_factory(lambda nm: eng.open_index("single-" + nm))
        single = _ivf_serving_scenario(indexer, embedding, REGISTRY, name, lambda index: None)
        np.savez(os.path.join(out_dir, "serving_ivf.npz"), **{"sharded_" + k: np.asarray(v) for k, v in sharded.items()},
                 **{"single_" + k: np.asarray(v) for k, v in single.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ivf_dtype", ["f32", "int8"])
def test_sharded_ivf_with_delta_behind_the_shim_equals_single_flat_index(gpu, tmp_path, ivf_dtype):
    """VERDICT r3 #2b on 2 ranks: OP_IVF_BUILD (shared centroids over UNEQUAL shards), appends / overwrites after the
    build land in the shards' flat deltas, every list probed == one flat HIP index bit for bit through HipIndexer — with
    fp32 slabs and with int8 slabs (int8 candidates + exact re-rank per shard; the scenario's k are <= 10 or > 32)."""
    import torch.multiprocessing as mp
    mp.spawn(_serving_ivf_worker, args=(2, _free_port(), str(tmp_path), ivf_dtype), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ivfworker1.done"))
    z = np.load(os.path.join(str(tmp_path), "serving_ivf.npz"))
    keys = sorted(k[len("single_"):] for k in z.files if k.startswith("single_"))
    assert "deep_ids" in keys and len(z["single_deep_ids"]) == 50
    for k in keys:
        a, b = z["sharded_" + k], z["single_" + k]
        if a.dtype.kind == "f":
            assert a.shape == b.shape and np.array_equal(a, b), k
        else:
            assert a.tolist() == b.tolist(), (k, a, b)
    assert z["single_new_ids"][0] == "n-7" and z["single_delta_ids"][0] == "n-3333" and int(z["single_rows"]) == 3602


def _dp_ingest_worker(rank, world, port, out_dir, model_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import embedding, indexer, serving
        from rassengine_amd.docstore import REGISTRY
        from rassengine_amd.encoder import HipSentenceEncoder
        from rassengine_amd.engine import Engine
        front = serving.start(serving.hip_shard_factory(0, 128), 128, torch.device("cuda", 0),
                              encoder_factory=serving.hip_encoder_factory(model_dir, 0))
        if rank != 0:
            open(os.path.join(out_dir, f"worker{rank}.done"), "w").write("ok")
            return
        import asyncio
        enc = HipSentenceEncoder.from_dir(model_dir, device=0)
        embedding.set_embedder(enc)
        words = [f"w{i}" for i in range(40)]
        rng = np.random.default_rng(3)
        docs = [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}",
                 "unstructuredText": " ".join(rng.choice(words, size=int(rng.integers(3, 40))))} for i in range(600)]
        docs[9]["unstructuredText"] = ""

        def run(name):
            asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs, None, name))       # 3 encoder batches
            ix = indexer.HipIndexer(None, name)
            out = {}
            for key, d, kw in (("a", 17, {}), ("b", 411, {"patient_id": "p0"}), ("c", 599, {})):
                h = ix.semantic_search(asyncio.run(embedding.embed_query(docs[d]["unstructuredText"])), k=5, **kw)
                out[key + "_ids"] = [x["doc_id"] for x, _ in h]
                out[key + "_scores"] = [float(v) for _, v in h]
            st = REGISTRY.get(name)
            out["rows"], out["count"] = int(st.index.rows), int(st.index.count)
            return out
        sharded = run("rass-idx-dp")
        idx = REGISTRY.get("rass-idx-dp").index
        assert isinstance(idx, serving.ShardedIndex) and idx.can_encode and sorted(set(idx._owner_rank)) == [0, 1]
        front.shutdown()
        REGISTRY.clear()
        eng = Engine.get(0, 128)
        REGISTRY.set_index_factory(lambda name: eng.open_index("single-" + name))
        single = run("rass-idx-dp")
        np.savez(os.path.join(out_dir, "dp.npz"), **{"sharded_" + k: np.asarray(v) for k, v in sharded.items()},
                 **{"single_" + k: np.asarray(v) for k, v in single.items()})
    finally:
        dist.destroy_process_group()


def test_data_parallel_ingest_with_hip_encoders_on_two_ranks(gpu, tmp_path):
    """SURVEY 8e on real kernels: two ranks (sharing the test GPU), each with its own HIP encoder and shard; rank 0
    tokenises, each rank encodes the batches dealt to it straight into its shard.  Ids, order and scores equal the
    single-engine ingest of the same texts (the encoder's output for a sequence does not depend on its batch)."""
    import torch.multiprocessing as mp
    from rassengine_amd.encoder import EncoderConfig, synthetic_vocab, write_random_model_dir
    model_dir = str(tmp_path / "tiny")
    vocab = synthetic_vocab(300)
    vocab[200:240] = [f"w{i}" for i in range(40)]
    write_random_model_dir(model_dir, EncoderConfig(vocab_size=300, hidden=128, layers=2, heads=2, intermediate=512,
                                                    max_positions=512), seed=5, vocab=vocab)
    mp.spawn(_dp_ingest_worker, args=(2, _free_port(), str(tmp_path), model_dir), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "worker1.done"))
    z = np.load(os.path.join(str(tmp_path), "dp.npz"))
    for k in sorted(k[len("single_"):] for k in z.files if k.startswith("single_")):
        a, b = z["sharded_" + k], z["single_" + k]
        if a.dtype.kind == "f":
            assert a.shape == b.shape and np.allclose(a, b, rtol=0, atol=2e-6), (k, a, b)
        else:
            assert a.tolist() == b.tolist(), (k, a, b)
    assert int(z["single_rows"]) == 600 and z["single_a_ids"][0] == "d17" and z["single_c_ids"][0] == "d599"


def _peer_worker(rank, world, port, n_local, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd.dist import HipShard, PeerMergeSearch, ShardedSearch
        from rassengine_amd.engine import Engine
        eng = Engine(0, 1024)
        idx = eng.open_index("shard", capacity_rows=n_local)
        idx.fill_synthetic(n_local, seed=91, row_id_base=rank * n_local)
        eng.synchronize()
        shard = HipShard(idx, id_base=rank * n_local)
        peer = PeerMergeSearch(shard)
        gather = ShardedSearch(shard)
        g = torch.Generator(device="cpu")
        g.manual_seed(8)
        out = {}
        for step, (nq, k) in enumerate([(32, 10), (5, 32), (17, 3), (32, 10), (1, 1)] * 6):   # 30 steps: both parities
            q_all = torch.randn((nq, 1024), generator=g).cuda()
            q = q_all.clone() if rank == 0 else torch.zeros_like(q_all)
            res = peer.search(q, k)
            s2, i2 = gather.search(q.clone(), k, broadcast=False)        # q already holds the broadcast queries
            if rank == 0:
                s1, i1 = res
                peer.check()
                out[f"ok{step}"] = np.array(bool(torch.equal(i1, i2) and torch.equal(s1, s2)))
                out[f"both{step}"] = np.array(bool((i1 >= n_local).any() and (i1 < n_local).any()) or nq * k < 4)
            else:
                assert res is None
        torch.cuda.synchronize()
        peer.close()
        if rank == 0:
            np.savez(os.path.join(out_dir, "peer.npz"), **out)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_peer_store_merge_equals_all_gather_merge(gpu, tmp_path):
    """per-shard top-k records stored straight into rank 0's buffer (HIP IPC mapping) + system-scope
    flags + bounded wait, instead of the all-gather: 30 steps of varying (nq, k) on 2 ranks sharing the test GPU must
    give rank 0 exactly the all-gather path's result, hits from both shards included."""
    import torch.multiprocessing as mp
    mp.spawn(_peer_worker, args=(2, _free_port(), 20000, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(str(tmp_path), "peer.npz"))
    assert len(z.files) == 60
    assert all(bool(z[k]) for k in z.files), [k for k in z.files if not bool(z[k])]

"""The multi-GPU index behind the boundary on CPU: world-size 2 and 3 ``gloo`` runs of
rassengine_amd.serving — rank 0 serves ``HipIndexer`` / ``store_fhir_docs_in_opensearch`` over a
``ShardedIndex``, the other ranks sit in ``worker_loop`` — with oracle-backed shards (test doubles).
store -> semantic_search through the reference-shaped shim must equal the single-index result: same doc ids in
the same order, same scores; overwrite (tombstone on the owning rank), patient / doc_type filters, count
reduced over ranks, the embedding quirk (row fetched from its owner) and a clean collective shutdown."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleServingShard:
    """serving.HipServingShard's surface on CPU tensors; arithmetic = the CPU oracle."""

    def __init__(self, dim):
        self.dim = dim
        self.device = torch.device("cpu")
        self._x = np.zeros((0, dim), dtype=np.float32)
        self._tags = np.zeros((0,), dtype=np.int32)
        self._gid = np.zeros((0,), dtype=np.int64)

    rows = property(lambda self: self._x.shape[0])
    count = property(lambda self: int((self._tags != -1).sum()))

    def add(self, vecs, tags, normalize, first_global_id):
        from oracle import oracle as O
        v = vecs.numpy().astype(np.float32)
        if normalize:
            v = O.normalize_ref(v).astype(np.float32)
        first = self.rows
        self._x = np.concatenate([self._x, v])
        self._tags = np.concatenate([self._tags, tags.numpy().astype(np.int32)])
        self._gid = np.concatenate([self._gid, first_global_id + np.arange(v.shape[0], dtype=np.int64)])
        return first

    def delete(self, ordinal):
        self._tags[ordinal] = -1

    def save(self, path):
        with open(path, "wb") as f:
            np.savez(f, x=self._x, tags=self._tags, gid=self._gid)

    @classmethod
    def load(cls, dim, path):
        z = np.load(path)
        s = cls(dim)
        s._x, s._tags, s._gid = z["x"], z["tags"], z["gid"]
        return s

    def get_row(self, ordinal):
        return torch.from_numpy(self._x[ordinal].copy())

    def search_packed(self, queries, k, filt, mask, after=None):
        from oracle import oracle as O
        from rassengine_amd.serving import HipServingShard
        nq = queries.shape[0]
        qn = O.normalize_ref(queries.numpy().copy()).astype(np.float32)
        kk = k if after is None else max(self.rows, 1)
        s, i = O.search(self._x, qn, kk, tags=self._tags, qfilter=None if filt is None else filt.numpy().copy(),
                        qmask=None if mask is None else mask.numpy().copy())
        if after is not None:      # keep only what ranks strictly behind (after_score, after_row), then the best k
            a_s, a_r = after[0].numpy(), after[1].numpy()
            s2 = np.full((nq, k), -np.inf)
            i2 = np.full((nq, k), -1, dtype=np.int64)
            for q in range(nq):
                keep = [(sv, iv) for sv, iv in zip(s[q], i[q]) if iv >= 0 and
                        (np.float32(sv) < a_s[q] or (np.float32(sv) == a_s[q] and iv > a_r[q]))][:k]
                for j, (sv, iv) in enumerate(keep):
                    s2[q, j], i2[q, j] = sv, iv
            s, i = s2, i2
        gids = np.full(i.shape, -1, dtype=np.int64)
        if self.rows:
            gids = np.where(i >= 0, self._gid[np.clip(i, 0, None)], -1).astype(np.int64)
        ids_off, size = HipServingShard.record_bytes(nq, k)
        rec = np.zeros(size, dtype=np.uint8)
        rec[:nq * k * 4] = s.astype(np.float32).view(np.uint8).reshape(-1)
        rec[ids_off:] = gids.view(np.uint8).reshape(-1)
        return torch.from_numpy(rec)

    def merge_packed(self, gathered, world, nq, k):
        from oracle import oracle as O
        from rassengine_amd.serving import HipServingShard
        ids_off, size = HipServingShard.record_bytes(nq, k)
        g = gathered.numpy().reshape(world, size)
        ls = np.stack([g[r, :nq * k * 4].copy().view(np.float32).reshape(nq, k) for r in range(world)])
        li = np.stack([g[r, ids_off:].copy().view(np.int64).reshape(nq, k) for r in range(world)])
        s, i = O.merge(ls.astype(np.float64), li)
        return s.astype(np.float32), i


def _docs(n):
    return [{"doc_id": f"text-note-{i}", "doc_type": "unstructured" if i % 5 else "structured",
             "patientId": f"p{i % 3}", "unstructuredText": f"chunk number {i} about topic{i % 7} and drug{i % 4}"}
            for i in range(n)]


def _scenario(indexer, embedding, REGISTRY, config, name):
    """What a FastAPI process does through the reference-shaped shim; returns comparable plain data."""
    import asyncio
    docs = _docs(90)
    out = {}
    # three uploads (three batches: dealt to different ranks), then an overwrite of two docs
    for a in range(0, 90, 30):
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs[a:a + 30], None, name))
    asyncio.run(indexer.store_fhir_docs_in_opensearch(
        [], [dict(docs[7], unstructuredText="entirely new words here"), dict(docs[40], unstructuredText="other text")],
        None, name))
    ix = indexer.HipIndexer(None, name)
    q = asyncio.run(embedding.embed_query("chunk number 12 about topic5 and drug0"))
    out["has"] = ix.has_any_data()
    hits = {"sem": ix.semantic_search(q, k=10), "pat": ix.semantic_search(q, k=10, patient_id="p1"),
            "hyb": ix.hybrid_search("x", q, k=7), "hs": ix.hybrid_structured_search("x", q, k=5, patient_id="p0"),
            "new": ix.semantic_search(asyncio.run(embedding.embed_query("entirely new words here")), k=3),
            "none": ix.semantic_search(q, k=5, patient_id="nobody"),
            # k > 32: continuation passes across the shards (the reference passes top_k through, main.py:2882)
            "k50": ix.semantic_search(q, k=50), "k70p": ix.semantic_search(q, k=70, patient_id="p2")}
    for key, h in hits.items():
        out[key + "_ids"] = [d["doc_id"] for d, _ in h]
        out[key + "_scores"] = [float(s) for _, s in h]
    st = REGISTRY.get(name)
    out["count"] = int(st.index.count)
    out["rows"] = int(st.index.rows)
    config.RASS_RETURN_EMBEDDING = True
    out["emb"] = np.asarray(ix.semantic_search(q, k=2)[1][0]["embedding"], dtype=np.float32)
    config.RASS_RETURN_EMBEDDING = False
    return out


def _prefetch_scenario(indexer, embedding, config, name):
    """12 concurrent ask()-shaped coroutines (embed_query -> ensure_index_exists -> a synchronous search,
    app/main.py:2800-2885): their k-NN scans are shared at the second await (rassengine_amd/prefetch.py) — over a
    sharded index through its QueryBatcher, one OP_SEARCH of 12 queries — and must equal the serial answers."""
    import asyncio
    from rassengine_amd import prefetch
    reqs = [(f"chunk number {i} about topic{i % 7} and drug{i % 4}", 6, None if i % 3 else f"p{i % 3}") for i in range(12)]

    async def ask(q, k, pid):
        emb = await embedding.embed_query(q)
        await indexer.ensure_index_exists(None, name)
        return indexer.HipIndexer(None, name).semantic_search(query_emb=emb, k=k, patient_id=pid, query=q)

    async def burst():
        return await asyncio.gather(*[ask(*r) for r in reqs])

    mode0 = config.RASS_KNN_PREFETCH
    try:
        config.RASS_KNN_PREFETCH = 0
        serial = asyncio.run(burst())
        config.RASS_KNN_PREFETCH = 1
        prefetch.reset_stats()
        shared = asyncio.run(burst())
    finally:
        config.RASS_KNN_PREFETCH = mode0
    assert [[(d["doc_id"], s) for d, s in h] for h in shared] == [[(d["doc_id"], s) for d, s in h] for h in serial]
    assert prefetch.stats["answered"] >= 10, prefetch.stats
    return {"pf_ids": [d["doc_id"] for h in shared for d, _ in h], "pf_scores": [float(s) for h in shared for _, s in h]}


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import config, embedding, indexer, serving
        from rassengine_amd.docstore import REGISTRY
        from tests.helpers import HashEmbedder, OracleIndex
        front = serving.start(lambda name: OracleServingShard(1024), 1024, torch.device("cpu"),
                              shard_loader=lambda name, path: OracleServingShard.load(1024, path))
        if rank != 0:
            assert front is None                       # the worker left its loop through the collective shutdown
            open(os.path.join(out_dir, f"worker{rank}.done"), "w").write("ok")
            return
        embedding.set_embedder(HashEmbedder(1024))
        sharded = _scenario(indexer, embedding, REGISTRY, config, "rass-idx-user1")
        sharded.update(_prefetch_scenario(indexer, embedding, config, "rass-idx-user1"))
        # round-robin by batch really spread the rows: every rank holds some
        idx = REGISTRY.get("rass-idx-user1").index
        assert isinstance(idx, serving.ShardedIndex)
        assert sorted(set(idx._owner_rank)) == list(range(min(world, 4)))
        # persistence through the front: every rank saves its shard, rank 0 the manifest; two generations, then a
        # load under another name must answer exactly like the live index (and the first generation's files go)
        from rassengine_amd.docstore import IndexState
        import asyncio
        st = REGISTRY.get("rass-idx-user1")
        prefix = os.path.join(out_dir, "saved-user1")
        st.save(prefix)
        gen1 = sorted(f for f in os.listdir(out_dir) if ".g000001." in f)
        assert len(gen1) == 1 + 2 * world, gen1                 # manifest + per rank: the shard file and its extent table
        st.save(prefix)
        assert not [f for f in os.listdir(out_dir) if ".g000001." in f]
        st2 = IndexState.load("rass-idx-restored", prefix, front.load_index)
        assert isinstance(st2.index, serving.ShardedIndex) and st2.index.rows == st.index.rows
        assert st2.index.count == st.index.count
        qv = asyncio.run(embedding.embed_query("chunk number 12 about topic5 and drug0"))
        PM = 0x00FFFFFF
        a = st.index.search(qv, 10)
        b = st2.index.search(qv, 10)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        st2.index.delete(int(a[1][0, 0]))                       # the restored extent tables find the owner
        assert int(st2.index.search(qv, 10)[1][0, 0]) == int(a[1][0, 1])
        first_new = st2.index.add(np.ones((3, 1024), dtype=np.float32))
        assert first_new == st.index.rows and st2.index.count == st.index.count - 1 + 3
        # incremental persistence on the sharded index: a delta segment (rows read back shard by shard through OP_GETROW)
        # on top of the snapshot, replayed by load through the front's add / delete
        asyncio.run(indexer.store_fhir_docs_in_opensearch(
            [], [{"doc_id": f"late-{i}", "doc_type": "unstructured", "patientId": "p1", "unstructuredText": f"late chunk {i} about topic5"}
                 for i in range(9)] + [{"doc_id": "text-f-3", "doc_type": "unstructured", "patientId": "p0", "unstructuredText": "rewritten late"}],
            None, "rass-idx-user1"))
        assert st.save_delta(prefix) is True
        st3 = IndexState.load("rass-idx-restored-delta", prefix, front.load_index)
        assert st3.index.rows == st.index.rows and st3.index.count == st.index.count and st3.doc_row == st.doc_row
        qd = asyncio.run(embedding.embed_query("late chunk 4 about topic5"))
        a3, b3 = st.index.search(qd, 10), st3.index.search(qd, 10)
        assert np.array_equal(a3[1], b3[1]) and np.array_equal(a3[0], b3[0])
        front.shutdown()
        front.shutdown()                               # idempotent
        # the same scenario on ONE index in this very process (same HashEmbedder hash seed)
        REGISTRY.clear()
        REGISTRY.set_index_factory(lambda name: OracleIndex(1024))
        single = _scenario(indexer, embedding, REGISTRY, config, "rass-idx-user1")
        single.update(_prefetch_scenario(indexer, embedding, config, "rass-idx-user1"))
        np.savez(os.path.join(out_dir, "rank0.npz"), **{"sharded_" + k: np.asarray(v) for k, v in sharded.items()},
                 **{"single_" + k: np.asarray(v) for k, v in single.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_store_and_search_through_the_shim_equal_the_single_index(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(1, world):
        assert os.path.exists(os.path.join(str(tmp_path), f"worker{r}.done"))
    z = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    keys = sorted(k[len("single_"):] for k in z.files if k.startswith("single_"))
    assert "sem_ids" in keys and "emb" in keys
    for k in keys:
        a, b = z["sharded_" + k], z["single_" + k]
        if a.dtype.kind == "f":
            assert a.shape == b.shape and np.array_equal(a, b), k       # bit-identical scores / stored row
        else:
            assert a.tolist() == b.tolist(), (k, a, b)
    assert bool(z["single_has"]) and int(z["single_count"]) == 90 and int(z["single_rows"]) == 92
    assert len(z["single_sem_ids"]) == 10 and len(z["single_none_ids"]) == 0
    assert z["single_new_ids"][0] == "text-note-7"
    assert len(z["single_hs_ids"]) > 0                                  # structured rows of patient p0 exist


def _dp_scenario(indexer, embedding, REGISTRY, name):
    """Ingest through store_fhir_docs_in_opensearch with NO embed_fn: on a sharded index whose ranks have encoders
    the texts are embedded data-parallel (ShardedIndex.add_texts); on a single index by the process's embedder."""
    import asyncio
    docs = [{"doc_id": f"n-{i}", "doc_type": "unstructured", "patientId": f"p{i % 4}",
             "unstructuredText": f"note {i} mentions topic{i % 11} drug{i % 5} ward{i % 3}"} for i in range(700)]
    docs[13]["unstructuredText"] = "   "            # blank: a zero row, never a hit
    asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs[:600], None, name))      # 3 encoder batches
    asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs[600:], None, name))
    asyncio.run(indexer.store_fhir_docs_in_opensearch([], [dict(docs[5], unstructuredText="fresh words only")], None, name))
    ix = indexer.HipIndexer(None, name)
    out = {}
    for key, text, kw in (("a", "note 77 mentions topic0 drug2 ward2", {}), ("b", "fresh words only", {}),
                          ("c", "note 300 mentions topic3 drug0 ward0", {"patient_id": "p0"})):
        h = ix.semantic_search(asyncio.run(embedding.embed_query(text)), k=8, **kw)
        out[key + "_ids"] = [d["doc_id"] for d, _ in h]
        out[key + "_scores"] = [float(x) for _, x in h]
    st = REGISTRY.get(name)
    out["count"], out["rows"] = int(st.index.count), int(st.index.rows)
    out["row13"] = np.asarray(st.index.get_row(13), dtype=np.float32)
    return out


def _dp_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import embedding, indexer, serving
        from rassengine_amd.docstore import REGISTRY
        from tests.helpers import OracleIndex, TokenHashEncoder
        encoders = []

        def make_encoder():
            encoders.append(TokenHashEncoder(1024))
            return encoders[-1]
        front = serving.start(lambda name: OracleServingShard(1024), 1024, torch.device("cpu"), encoder_factory=make_encoder)
        seen = encoders[0].encoded_seqs if encoders else 0
        open(os.path.join(out_dir, f"encoded{rank}.txt"), "w").write(str(seen))
        if rank != 0:
            return
        embedding.set_embedder(TokenHashEncoder(1024))          # queries only: the ingest must not use it
        sharded = _dp_scenario(indexer, embedding, REGISTRY, "rass-idx-dp")
        idx = REGISTRY.get("rass-idx-dp").index
        assert isinstance(idx, serving.ShardedIndex) and idx.can_encode
        assert sorted(set(idx._owner_rank)) == list(range(world))
        open(os.path.join(out_dir, "encoded0.txt"), "w").write(str(encoders[0].encoded_seqs))
        front.shutdown()
        REGISTRY.clear()
        REGISTRY.set_index_factory(lambda name: OracleIndex(1024))
        single = _dp_scenario(indexer, embedding, REGISTRY, "rass-idx-dp")
        np.savez(os.path.join(out_dir, "rank0.npz"), **{"sharded_" + k: np.asarray(v) for k, v in sharded.items()},
                 **{"single_" + k: np.asarray(v) for k, v in single.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_ingest_equals_the_single_index(world, tmp_path):
    """ingest is data-parallel — every rank encodes the batches dealt to it into its own shard."""
    mp.spawn(_dp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    keys = sorted(k[len("single_"):] for k in z.files if k.startswith("single_"))
    for k in keys:
        a, b = z["sharded_" + k], z["single_" + k]
        if a.dtype.kind == "f":
            assert a.shape == b.shape and np.array_equal(a, b), k
        else:
            assert a.tolist() == b.tolist(), (k, a, b)
    assert int(z["single_rows"]) == 701 and int(z["single_count"]) == 700
    assert z["single_b_ids"][0] == "n-5" and "n-13" not in z["single_a_ids"].tolist()
    assert not z["single_row13"].any()                                       # the blank text is a zero row
    # every rank really encoded: 700 non-blank texts spread over the ranks' own encoders
    done = [int(open(os.path.join(str(tmp_path), f"encoded{r}.txt")).read()) for r in range(world)]
    assert all(d > 0 for d in done) and sum(done) == 700, done


def _failing_encoder_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import serving
        from tests.helpers import TokenHashEncoder

        class Flaky(TokenHashEncoder):
            def encode_flat(self, ids, cu):
                if dist.get_rank() == 1:
                    raise RuntimeError("device lost")
                return super().encode_flat(ids, cu)
        front = serving.start(lambda name: OracleServingShard(64), 64, torch.device("cpu"), encoder_factory=lambda: Flaky(64),
                              shard_loader=lambda name, path: OracleServingShard.load(64, path))
        if rank != 0:
            open(os.path.join(out_dir, f"worker{rank}.done"), "w").write("ok")
            return
        ix = front.open_index("flaky")
        texts = [f"alpha beta {i}" for i in range(600)]            # batches 0, 2 -> rank 0; batch 1 -> rank 1 (fails)
        assert ix.add_texts(texts) == 0 and ix.rows == 600 and ix.count == 600
        q = TokenHashEncoder(64).encode(["alpha beta 300", "alpha beta 10"])
        s_, i_ = ix.search(q, 3)
        assert i_[1, 0] == 10                                      # rank 0's batch: encoded
        assert 300 not in i_[0].tolist() and not ix.get_row(300).any()   # rank 1's batch: zero rows, never a hit
        with pytest.raises(serving.CollectiveFailure):             # a load nobody can satisfy: refused everywhere ...
            json_path = os.path.join(out_dir, "bogus.json")
            open(json_path, "w").write('{"format": "rass-sharded-1", "world": %d, "base": "nope", "rows": 0, '
                                       '"batches": 0, "runs": [], "deleted": []}' % world)
            front.load_index("ghost", json_path)
        assert ix.add_texts(["gamma delta"]) == 600                # ... and the service goes on
        front.shutdown()
    finally:
        dist.destroy_process_group()


def test_encoder_failure_and_refused_load_keep_the_ranks_in_step(tmp_path):
    mp.spawn(_failing_encoder_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "worker1.done"))


def _failing_shard_worker(rank, world, port, out_dir, fail_rank):
    """a shard whose add / search raises on ONE rank (rank 0 itself or a worker) must fail the operation on
    EVERY rank together, keep the workers in the loop, never reuse a global id, and leave row -> doc mapping intact."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import asyncio
        from rassengine_amd import embedding, indexer, serving
        from rassengine_amd.docstore import REGISTRY
        from tests.helpers import HashEmbedder, TokenHashEncoder
        D = 64

        class FlakyShard(OracleServingShard):
            def add(self, vecs, tags, normalize, first_global_id):
                t = tags.numpy()
                if dist.get_rank() == fail_rank and (t == 999).any():
                    raise MemoryError("slab growth failed")
                return super().add(vecs, tags, normalize, first_global_id)

            def search_packed(self, queries, k, filt, mask, after=None):
                if dist.get_rank() == fail_rank and float(queries[0, 0]) == 12345.0:
                    raise RuntimeError("scan launch failed")
                return super().search_packed(queries, k, filt, mask, after)

        front = serving.start(lambda name: FlakyShard(D), D, torch.device("cpu"), encoder_factory=lambda: TokenHashEncoder(D))
        if rank != 0:
            open(os.path.join(out_dir, f"worker{rank}.done"), "w").write("ok")   # left the loop through the shutdown
            return
        assert dist.get_backend(front.server.ctl) == "gloo"            # the idle wait is a host-side socket read
        ix = front.open_index("flaky")
        rng = np.random.default_rng(0)

        def batch(n, tag=1):
            return rng.standard_normal((n, D)).astype(np.float32), np.full(n, tag, dtype=np.int32)

        va, ta = batch(30)
        assert ix.add(va, ta) == 0                                     # batch 0 -> rank 0
        vb, tb = batch(30)
        assert ix.add(vb, tb) == 30                                    # batch 1 -> rank 1
        if fail_rank == 1:
            vc, tc = batch(5)
            assert ix.add(vc, tc) == 60                                # batch 2 -> rank 0, so that batch 3 -> rank 1
        rows0, live0 = ix.rows, ix.count
        bad_v, bad_t = batch(7, tag=999)
        with pytest.raises(serving.CollectiveFailure) as ei:
            ix.add(bad_v, bad_t)                                       # the owner (= fail_rank) raises
        assert ei.value.failed_ranks == (fail_rank,)
        assert ix.rows == rows0 + 7 and ix.count == live0              # the ids are burnt, nothing was stored
        with pytest.raises(IndexError):
            ix.get_row(rows0 + 3)                                      # a hole
        vd, td = batch(4)
        first = ix.add(vd, td)
        assert first == rows0 + 7                                      # never the failed batch's ids again
        s_, i_ = ix.search(vd, 1)
        assert i_[:, 0].tolist() == list(range(first, first + 4))      # the new rows answer under their own ids
        assert np.allclose(ix.get_row(first + 2), vd[2] / (np.linalg.norm(vd[2]) + 1e-9), atol=1e-6)

        # a search that fails on one rank fails everywhere, once; the next one is served
        qbad = vd[:1].copy()
        qbad[0, 0] = 12345.0
        with pytest.raises(serving.CollectiveFailure):
            ix.search(qbad, 3)
        assert ix.search(vd[:1], 1)[1][0, 0] == first

        # a multi-chunk add whose SECOND chunk fails: the first chunk is rolled back (tombstoned), ids burnt
        n_big = serving.ADD_CHUNK_ROWS + 10
        vbig = rng.standard_normal((n_big, D)).astype(np.float32)
        tbig = np.ones(n_big, dtype=np.int32)
        tbig[-1] = 999
        # make sure this batch's owner is the failing rank
        while ix._batches % world != fail_rank:
            ix.add(*batch(1))
        rows1, live1 = ix.rows, ix.count
        with pytest.raises(serving.CollectiveFailure):
            ix.add(vbig, tbig)
        assert ix.rows == rows1 + n_big and ix.count == live1
        assert ix.search(vbig[:1], 1)[1][0, 0] != rows1                # the rolled-back row is no hit

        # data-parallel ingest: the append of ONE rank's batch fails -> the round fails as a whole, the other rank's
        # batch is rolled back, the cursor moved past all of it
        texts = [f"alpha beta {i}" for i in range(600)]                # 3 batches over 2 ranks
        ttags = np.ones(600, dtype=np.int32)
        owners = [(ix._batches + j) % world for j in range(3)]
        ttags[256 * owners.index(fail_rank)] = 999
        rows2, live2 = ix.rows, ix.count
        with pytest.raises(serving.CollectiveFailure):
            ix.add_texts(texts, tags=ttags)
        # (the cursor moved past every round that was posted: 512 texts per round of 2 ranks, the failing round included)
        assert ix.rows in (rows2 + 512, rows2 + 600) and ix.count == live2
        rows3 = ix.rows
        assert ix.add_texts(["gamma delta"]) == rows3 and ix.count == live2 + 1

        # through the shim: a failed store registers nothing and shifts nothing (row -> doc stays right)
        REGISTRY.clear()
        REGISTRY.set_index_factory(front.open_index)
        embedding.set_embedder(HashEmbedder(D))

        def docs(a, b, patient="p1"):
            return [{"doc_id": f"d{i}", "doc_type": "unstructured", "patientId": patient,
                     "unstructuredText": f"chunk number {i} about topic{i % 7}"} for i in range(a, b)]
        from rassengine_amd import config
        config.EMBED_DIM = D
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs(0, 20), None, "flaky-shim", embed_fn=lambda t: _embed(t)))
        st = REGISTRY.get("flaky-shim")
        while st.index._batches % world != fail_rank:
            asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs(1000 + st.index._batches, 1001 + st.index._batches), None,
                                                              "flaky-shim", embed_fn=lambda t: _embed(t)))
        # the 999th patient code does not exist; make the tag 999 through the shard's eyes: patch tag_of for one store
        orig_tag_of = st.tag_of
        st.tag_of = lambda d: 999
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs(20, 30), None, "flaky-shim", embed_fn=lambda t: _embed(t)))
        st.tag_of = orig_tag_of
        assert "d25" not in st.doc_row                                  # logged, nothing registered (main.py:1279-1281)
        asyncio.run(indexer.store_fhir_docs_in_opensearch([], docs(30, 40), None, "flaky-shim", embed_fn=lambda t: _embed(t)))
        ixr = indexer.HipIndexer(None, "flaky-shim")
        for i in (3, 17, 33, 39):
            q = asyncio.run(_embed([f"chunk number {i} about topic{i % 7}"]))
            hits = ixr.semantic_search(q, k=1)
            assert hits and hits[0][0]["doc_id"] == f"d{i}", (i, hits)
        front.shutdown()
    finally:
        dist.destroy_process_group()


async def _embed(texts):
    from tests.helpers import HashEmbedder
    return HashEmbedder(64).encode(list(texts))


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_a_failing_shard_fails_every_rank_together_and_corrupts_nothing(fail_rank, tmp_path):
    mp.spawn(_failing_shard_worker, args=(2, _free_port(), str(tmp_path), fail_rank), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "worker1.done"))


def test_extent_table():
    from rassengine_amd.serving import Extents
    e = Extents()
    e.append(0, 0, 30)
    e.append(60, 30, 30)       # the batch in between went to another rank
    e.append(90, 60, 5)        # contiguous in both spaces: merged into the previous run
    assert len(e.gid) == 2
    assert e.ordinal_of(0) == 0 and e.ordinal_of(29) == 29 and e.ordinal_of(30) is None and e.ordinal_of(59) is None
    assert e.ordinal_of(60) == 30 and e.ordinal_of(94) == 64 and e.ordinal_of(95) is None and e.ordinal_of(-1) is None


# ------------------------------------------------------------------ OP_IVF_BUILD / the flat delta on CPU doubles (VERDICT r3 #2b)
class OracleIvfServingShard(OracleServingShard):
    """The IVF surface of ``serving.HipServingShard`` on the CPU oracle: ``build_ivf`` (a collective: every rank ends with
    rank 0's centroids), ``search_packed(..., exact=, nprobe=)`` = brute force restricted to the rows of each query's
    nprobe best lists U the rows appended since the build (the delta), tombstones honoured in both."""

    def __init__(self, dim):
        super().__init__(dim)
        self._cent = None
        self._assign = None
        self._covered = 0
        self._nprobe = 1
        self.searches = []          # (exact, nprobe, used_ivf) per search_packed call

    def build_ivf(self, nlist, nprobe, dtype="f32", group=None):
        cent = torch.from_numpy(self._x[:nlist].copy()) if dist.get_rank() == 0 else torch.zeros((nlist, self.dim))
        if dist.get_world_size() > 1:
            dist.broadcast(cent, src=0, group=group)          # shared centroids, as the k-means all-reduce gives the HIP shards
        self._cent = cent.numpy()
        self._assign = np.argmax(self._x @ self._cent.T, axis=1)
        self._covered = self.rows // 32 * 32
        self._nprobe = nprobe

    def search_packed(self, queries, k, filt, mask, after=None, exact=False, nprobe=0):
        use = self._cent is not None and not exact and after is None
        self.searches.append((bool(exact), int(nprobe), use))
        if not use:
            return super().search_packed(queries, k, filt, mask, after)
        from oracle import oracle as O
        from rassengine_amd.serving import HipServingShard
        nq = queries.shape[0]
        qn = O.normalize_ref(queries.numpy().copy()).astype(np.float32)
        npb = min(int(nprobe) or self._nprobe, self._cent.shape[0])
        s = np.full((nq, k), -np.inf)
        gids = np.full((nq, k), -1, dtype=np.int64)
        for q in range(nq):
            lists = np.argsort(-(qn[q] @ self._cent.T), kind="stable")[:npb]
            member = np.ones(self.rows, dtype=bool)
            member[:self._covered] = np.isin(self._assign[:self._covered], lists)
            rows = np.nonzero(member)[0]
            sq, iq = O.search(self._x[rows], qn[q:q + 1], k, tags=self._tags[rows],
                              qfilter=None if filt is None else filt.numpy()[q:q + 1].copy(),
                              qmask=None if mask is None else mask.numpy()[q:q + 1].copy())
            ok = iq[0] >= 0
            s[q, :ok.sum()] = sq[0][ok]
            gids[q, :ok.sum()] = self._gid[rows[iq[0][ok]]]
        ids_off, size = HipServingShard.record_bytes(nq, k)
        rec = np.zeros(size, dtype=np.uint8)
        rec[:nq * k * 4] = s.astype(np.float32).view(np.uint8).reshape(-1)
        rec[ids_off:] = gids.view(np.uint8).reshape(-1)
        return torch.from_numpy(rec)


def _ivf_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import config, serving
        D = 64
        shards = []

        def make(name):
            shards.append(OracleIvfServingShard(D))
            return shards[-1]
        front = serving.start(make, D, torch.device("cpu"), install_registry=False)
        if rank != 0:
            # the worker followed every command: its shard built an IVF twice and answered exact + probed searches
            sh = shards[0]
            assert sh._cent is not None and sh._covered > 0
            assert any(e for e, _, _ in sh.searches) and any(u for _, _, u in sh.searches)
            open(os.path.join(out_dir, f"ivf{rank}.done"), "w").write("ok")
            return
        from tests.helpers import OracleIndex
        rng = np.random.default_rng(3)
        ix = front.open_index("ivf-cpu")
        ref = OracleIndex(D)
        for n in (130, 70, 150, 50):                           # unequal shards: rank 0 holds 280 rows, rank 1 120
            v = rng.standard_normal((n, D)).astype(np.float32)
            t = rng.integers(1, 4, size=n).astype(np.int32)
            assert ix.add(v, t) == ref.add(v, t)
        with pytest.raises(ValueError):
            ix.build_ivf()                                      # RASS_IVF_NLIST is 0: an explicit nlist is needed
        ix.build_ivf(nlist=8, nprobe=8)
        assert ix._ivf_covered == 400 and ix.epoch == (400, 0, 1)
        for n in (33, 21):                                     # the delta, on both ranks
            v = rng.standard_normal((n, D)).astype(np.float32)
            t = rng.integers(1, 4, size=n).astype(np.int32)
            assert ix.add(v, t) == ref.add(v, t)
        for r in (5, 131, 399, 410, 440):                      # covered rows of both ranks and delta rows
            ix.delete(r)
            ref.delete(r)
        q = rng.standard_normal((7, D)).astype(np.float32)
        q[0] = ref._rows[420] * 3.0                             # a delta row
        f = np.array([1, 2, 3, 1, 2, 3, 1], dtype=np.int32)
        for args in ((10,), (10, f), (45,), (33, f)):          # k > 32: every pass carries the EXACT flag
            a, b = ix.search(q, *args), ref.search(q, *args)
            assert np.array_equal(a[1], b[1]) and np.allclose(a[0], b[0], atol=1e-6), args
        assert int(ix.search(q[:1], 1)[1][0, 0]) == 420
        sh = shards[0]
        assert [s for s in sh.searches if s[0]] and all(not u for e, _, u in sh.searches if e)
        assert all(n == 8 for e, n, _ in sh.searches if not e)       # nprobe travels with the command
        # a partial probe: true scores only, the delta row is always found; then a rebuild that absorbs the delta
        ix.build_ivf(nlist=8, nprobe=1)
        a, b = ix.search(q, 10), ref.search(q, 10)
        assert int(a[1][0, 0]) == 420
        truth = {(r, int(i)): float(sv) for r in range(7) for i, sv in zip(b[1][r], b[0][r])}
        got = [(r, int(i), float(sv)) for r in range(7) for i, sv in zip(a[1][r], a[0][r]) if i >= 0]
        assert got and all(abs(truth[(r, i)] - sv) <= 1e-6 for r, i, sv in got if (r, i) in truth)
        assert ix._ivf_covered == 454 and ix._ivf_builds == 2 and shards[0]._covered == (280 + 33) // 32 * 32
        # the automatic policy: RASS_IVF_NLIST > 0 -> rank 0 posts the rebuild once the delta passes the threshold
        config.RASS_IVF_NLIST, config.RASS_IVF_NPROBE, config.RASS_IVF_MIN_ROWS, config.RASS_IVF_REBUILD_FRACTION = 8, 8, 100, 0.1
        v = rng.standard_normal((40, D)).astype(np.float32)
        ix.add(v)
        ref.add(v)
        assert ix._ivf_builds == 2                              # 40 <= 0.1 x 454
        v = rng.standard_normal((30, D)).astype(np.float32)
        ix.add(v)
        ref.add(v)
        assert ix._ivf_builds == 3 and ix._ivf_covered == 524 and ix._nprobe == 8
        a, b = ix.search(q, 10), ref.search(q, 10)
        assert np.array_equal(a[1], b[1])
        # the manifest keeps the IVF bookkeeping
        ix.save(os.path.join(out_dir, "ivf-cpu.manifest"))
        import json
        man = json.load(open(os.path.join(out_dir, "ivf-cpu.manifest")))
        assert man["ivf"] == {"covered": 524, "nprobe": 8}
        front.shutdown()
        open(os.path.join(out_dir, "ivf0.done"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_ivf_build_op_delta_and_exact_flag_on_two_ranks(tmp_path):
    mp.spawn(_ivf_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ivf0.done")) and os.path.exists(os.path.join(str(tmp_path), "ivf1.done"))


# ------------------------------------------------------------------ a PARTIAL append, then save / load
def _partial_append_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rassengine_amd import serving
        D = 32

        class PartialShard(OracleServingShard):
            """An append that stores PART of its batch and then fails (a slab growth that ran out of memory half-way)."""

            def add(self, vecs, tags, normalize, first_global_id):
                if (tags.numpy() == 999).any():
                    super().add(vecs[:3], torch.ones(3, dtype=torch.int32), normalize, first_global_id)
                    raise MemoryError("slab growth failed after 3 rows")
                return super().add(vecs, tags, normalize, first_global_id)

        front = serving.start(lambda name: PartialShard(D), D, torch.device("cpu"), install_registry=False,
                              shard_loader=lambda name, path: PartialShard.load(D, path))
        if rank != 0:
            open(os.path.join(out_dir, f"partial{rank}.done"), "w").write("ok")
            return
        rng = np.random.default_rng(5)
        ix = front.open_index("partial")

        def batch(n, tag=1):
            return rng.standard_normal((n, D)).astype(np.float32), np.full(n, tag, dtype=np.int32)
        va, ta = batch(20)
        ix.add(va, ta)                                             # batch 0 -> rank 0
        vb, tb = batch(20)
        ix.add(vb, tb)                                             # batch 1 -> rank 1
        vc, tc = batch(10)
        ix.add(vc, tc)                                             # batch 2 -> rank 0
        bad_v, bad_t = batch(7, tag=999)
        with pytest.raises(serving.CollectiveFailure):
            ix.add(bad_v, bad_t)                                   # batch 3 -> rank 1: stores 3 rows, then raises
        assert ix.rows == 57 and ix.count == 50                    # the 7 ids are burnt, the 3 stored rows are tombstones
        vd, td = batch(15)
        ix.add(vd, td)                                             # batch 4 -> rank 0
        ve, te = batch(12)
        assert ix.add(ve, te) == 72                                # batch 5 -> rank 1: its ordinals start BEHIND the 3 dead rows
        q = np.concatenate([ve[4:5] * 2.0, vb[7:8], bad_v[1:2], rng.standard_normal((3, D)).astype(np.float32)])
        before = ix.search(q, 6)
        assert int(before[1][0, 0]) == 72 + 4 and int(before[1][1, 0]) == 20 + 7
        assert int(before[1][2, 0]) not in range(50, 57)           # a row of the failed batch is never a hit
        row_e = ix.get_row(72 + 4)
        man = os.path.join(out_dir, "partial.manifest")
        ix.save(man)
        assert os.path.exists(os.path.join(out_dir, "partial.manifest.shard1of2.ext"))
        back = front.load_index("partial-restored", man)           # refused before round 4: 35 rows in the file, 32 in the runs
        after = back.search(q, 6)
        assert np.array_equal(before[1], after[1]) and np.array_equal(before[0], after[0])
        assert np.array_equal(back.get_row(72 + 4), row_e) and back.rows == ix.rows and back.count == ix.count
        back.delete(72 + 4)                                        # the restored extent table finds the right ordinal
        assert int(back.search(q[:1], 1)[1][0, 0]) != 72 + 4
        assert back.add(*batch(5)) == 84
        front.shutdown()
        open(os.path.join(out_dir, "partial0.done"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_partial_append_then_save_and_load(tmp_path):
    mp.spawn(_partial_append_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "partial0.done")) and os.path.exists(os.path.join(str(tmp_path), "partial1.done"))

"""N > 1 path on CPU: world_size-2 (and 3) ``gloo`` runs of rassengine_amd.dist.ShardedSearch
with an oracle-backed shard (tests double).  Checks the partition, the id offsets, the
broadcast + all-gather order and that every rank ends with the single-shard result."""
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


class OracleShard:
    """LocalShard protocol on CPU tensors; arithmetic = the CPU oracle."""

    def __init__(self, xn_local, id_base):
        self.xn = xn_local
        self.id_base = id_base
        self.device = torch.device("cpu")

    def search_local(self, queries, k):
        from oracle import oracle as O
        qn = O.normalize_ref(queries.numpy()).astype(np.float32)
        s, i = O.search(self.xn, qn, k, id_base=self.id_base)
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(i)

    def merge(self, list_scores, list_ids):
        from oracle import oracle as O
        s, i = O.merge(list_scores.numpy().astype(np.float64), list_ids.numpy())
        return torch.from_numpy(s.astype(np.float32)), torch.from_numpy(i)


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from rassengine_amd.dist import ShardedSearch, shard_bounds
        xn = O.synthetic_unit_rows(n_total, 256, 77)            # every rank regenerates, keeps its block
        lo, hi = shard_bounds(n_total, world)[rank]
        search = ShardedSearch(OracleShard(xn[lo:hi], lo))
        rng = np.random.default_rng(5)
        q_all = rng.standard_normal((6, 256)).astype(np.float32)
        q = torch.from_numpy(q_all.copy()) if rank == 0 else torch.zeros((6, 256))  # only src holds the batch
        s, i = search.search(q, 7)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.numpy(), i=i.numpy(), q=q.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1001), (3, 500)])
def test_sharded_search_equals_single_shard(world, n_total, tmp_path, oracle):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    xn = oracle.synthetic_unit_rows(n_total, 256, 77)
    q_all = np.random.default_rng(5).standard_normal((6, 256)).astype(np.float32)
    rs, ri = oracle.search(xn, oracle.normalize_ref(q_all).astype(np.float32), 7)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.array_equal(got["q"], q_all)                      # broadcast reached every rank
        assert np.array_equal(got["i"], ri), f"rank {r}"
        assert np.array_equal(got["s"], rs.astype(np.float32))


def test_shard_bounds_cover_rows_exactly():
    from rassengine_amd.dist import shard_bounds
    for n, w in [(10, 3), (0, 2), (1_000_000, 8), (7, 8), (10_000_000, 8)]:
        b = shard_bounds(n, w)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1

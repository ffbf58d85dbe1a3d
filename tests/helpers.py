"""Test doubles: a CPU stand-in for FlatIndex (oracle-backed) so the host-side shim logic
can be exercised without a GPU.  Lives under tests/ — never imported by the product."""
import zlib

import numpy as np

from oracle import oracle as O


class OracleIndex:
    """Same surface as rassengine_amd.engine.FlatIndex; arithmetic = the CPU oracle."""

    def __init__(self, dim=1024):
        self.dim = dim
        self._rows = np.zeros((0, dim), dtype=np.float32)
        self._tags = np.zeros((0,), dtype=np.int32)

    @property
    def rows(self):
        return self._rows.shape[0]

    @property
    def count(self):
        return int((self._tags != -1).sum())

    def add(self, vecs, tags=None, normalize=True):
        v = np.ascontiguousarray(vecs, dtype=np.float32)
        first = self.rows
        if normalize:
            v = O.normalize_ref(v).astype(np.float32)
        t = np.zeros(v.shape[0], dtype=np.int32) if tags is None else np.asarray(tags, dtype=np.int32)
        self._rows = np.concatenate([self._rows, v])
        self._tags = np.concatenate([self._tags, t])
        return first

    def delete(self, row):
        if not 0 <= row < self.rows:
            raise IndexError(row)
        self._tags[row] = -1

    def get_row(self, row):
        return self._rows[row].copy()

    def get_rows(self, first_row, n):
        return self._rows[first_row:first_row + n].copy()

    def search(self, queries, k, q_filter=None, q_filter_mask=None):
        qn = O.normalize_ref(np.ascontiguousarray(queries, dtype=np.float32)).astype(np.float32)
        s, i = O.search(self._rows, qn, k, tags=self._tags, qfilter=q_filter, qmask=q_filter_mask)
        return s.astype(np.float32), i


class HashEmbedder:
    """Deterministic text -> vector stand-in for the encoder (host-logic tests only)."""

    def __init__(self, dim=1024):
        self.dim = dim
        self.calls = []

    def encode(self, texts):
        self.calls.append(list(texts))
        out = np.zeros((len(texts), self.dim), dtype=np.float32)
        for r, t in enumerate(texts):
            for w in t.lower().split():
                rng = np.random.default_rng(zlib.crc32(w.encode("utf-8")))   # not hash(): that is salted per process
                out[r] += rng.standard_normal(self.dim).astype(np.float32)
        return out * 3.0  # deliberately un-normalised


class TokenHashEncoder:
    """Process-independent stand-in for the sentence encoder with the surface serving.py's data-parallel ingest
    uses: ``tokenize(texts) -> (ids, cu)`` on rank 0, ``encode_flat(ids, cu) -> [n, dim]`` on every rank, and
    ``encode(texts)`` for the single-index path.  A token id is crc32(word) (no PYTHONHASHSEED dependence: the ranks
    are different processes), a sequence's vector the sum of its tokens' seeded Gaussian vectors."""

    def __init__(self, dim=1024):
        self.dim = dim
        self.encoded_seqs = 0

    def tokenize(self, texts):
        import zlib
        seqs = [[zlib.crc32(w.encode("utf-8")) % 30000 + 1 for w in t.lower().split()] for t in texts]
        cu = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum([len(q) for q in seqs], out=cu[1:])
        ids = np.concatenate([np.asarray(q, dtype=np.int32) for q in seqs]) if cu[-1] else np.zeros(0, np.int32)
        return ids, cu

    def encode_flat(self, ids, cu):
        n = len(cu) - 1
        out = np.zeros((n, self.dim), dtype=np.float32)
        for r in range(n):
            for tok in np.asarray(ids[int(cu[r]):int(cu[r + 1])]):
                out[r] += np.random.default_rng(int(tok)).standard_normal(self.dim).astype(np.float32)
        self.encoded_seqs += n
        return out * 3.0  # deliberately un-normalised

    def encode(self, texts):
        ids, cu = self.tokenize(texts)
        return self.encode_flat(ids, cu)

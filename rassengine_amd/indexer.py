"""``HipIndexer`` — drop-in for the k-NN half of the reference's ``OpenSearchIndexer``
(app/main.py:1395-2150) — and ``store_fhir_docs`` / ``ensure_index_exists`` for the write
side (app/main.py:350-579, 1211-1282).

Same names, argument meaning and error behaviour as the reference:

* ``OpenSearchIndexer(client, index_name)`` is built per request (2802) -> construction is
  a dictionary lookup;
* ``semantic_search(query_emb, k, filter_clause, patient_id)`` returns
  ``[(doc_dict, float(score))]`` best first, ``[]`` on an empty query embedding (1534-1535)
  and on ANY exception (1558-1560: log + ``[]``);
* ``ask()`` calls it with ``query=`` as well (2879-2885), which the reference's own
  signature does not accept (TypeError -> HTTP 500): accepted and ignored here;
* the query is re-normalised with ``e / (||e|| + 1e-9)`` (1536-1537) — on the GPU;
* ``patient_id`` -> the ``term: patientId`` filter (1549) as an exact PRE-filter (the int32
  row tag), which is what the bool/filter query asks for;
* ``filter_clause``: ``ask()`` passes the NER entity list there (2770), which makes the
  reference's query malformed and returns ``[]``.  Tolerated, not
  replicated: only a ``{"term": {"patientId": ...}}`` dict is honoured, anything else ignored;
* ``_score``: OpenSearch k-NN cosinesimil reports ``1 / (2 - cos)`` (unverified
  offline, so configurable: ``RASS_SCORE_MODE=opensearch|cosine``).  ``ask()`` never reads it.

The 8 text / aggregate query builders of the reference (exact_match_search, structured_search,
...) are BM25 / Lucene text search: out of scope; the knn sub-score of the
hybrid builders is exposed as ``knn_scores`` (§8f-1).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from . import config
from .docstore import REGISTRY, IndexState

logger = logging.getLogger("rassengine_amd")

TOP_K = config.TOP_K


def _score_out(cos: float, mode: Optional[str] = None) -> float:
    mode = mode or config.RASS_SCORE_MODE
    if mode == "cosine":
        return float(cos)
    return float(1.0 / (2.0 - float(cos)))  # OpenSearch k-NN cosinesimil: 1 / (1 + (1 - cos))


def _patient_from_filter(filter_clause: Any) -> Optional[Any]:
    if isinstance(filter_clause, dict):
        term = filter_clause.get("term")
        if isinstance(term, dict) and "patientId" in term:
            return term["patientId"]
    return None


class HipIndexer:
    """Exact cosine k-NN over the HBM-resident index named ``index_name``."""

    def __init__(self, client: Any = None, index_name: str = ""):
        self.client = client          # kept for signature parity; unused (no HTTP hop any more)
        self.index_name = index_name

    # ------------------------------------------------------------------ app/main.py:1470-1478
    def has_any_data(self) -> bool:
        try:
            st = REGISTRY.get(self.index_name, create=False)
            return st is not None and st.live_count() > 0
        except Exception:
            return False

    # ------------------------------------------------------------------ app/main.py:1527-1560
    def semantic_search(self, query_emb: np.ndarray, k: int = TOP_K, filter_clause: Optional[Dict] = None,
                        patient_id: Optional[str] = None, query: Optional[str] = None, **_ignored
                        ) -> List[Tuple[Dict, float]]:
        if query_emb is None or np.size(query_emb) == 0:
            return []
        try:
            return self._knn(query_emb, k, filter_clause, patient_id, boost=1.0, score_mode=None)
        except Exception as e:  # reference: log and return [] (1558-1560)
            logger.error(f"Semantic search error: {e}")
            return []

    # knn sub-clause of hybrid_search (1595, boost 2.0), hybrid_structured_search (1754, 2.0),
    # multi_intent_search (2003, 1.5): score = boost * knn_score, to be summed with the BM25
    # `should` clauses by whoever keeps a text engine.
    def knn_scores(self, query_emb: np.ndarray, k: int = TOP_K, boost: float = 1.0,
                   filter_clause: Optional[Dict] = None, patient_id: Optional[str] = None
                   ) -> List[Tuple[Dict, float]]:
        if query_emb is None or np.size(query_emb) == 0:
            return []
        try:
            return self._knn(query_emb, k, filter_clause, patient_id, boost=boost, score_mode=None)
        except Exception as e:
            logger.error(f"kNN score error: {e}")
            return []

    def hybrid_search(self, query: str, query_emb: np.ndarray, k: int = TOP_K, filter_clause: Optional[Dict] = None,
                      patient_id: Optional[str] = None) -> List[Tuple[Dict, float]]:
        """The knn `should` clause of app/main.py:1562-1615 (boost 2.0); the two multi_match
        clauses need a text engine and contribute 0 here."""
        if not query or not query.strip() or query_emb is None or np.size(query_emb) == 0:
            return []  # 1570-1571
        return self.knn_scores(query_emb, k, boost=2.0, filter_clause=filter_clause, patient_id=patient_id)

    async def asemantic_search(self, query_emb: np.ndarray, k: int = TOP_K, filter_clause: Optional[Dict] = None,
                               patient_id: Optional[str] = None, query: Optional[str] = None, **_ignored
                               ) -> List[Tuple[Dict, float]]:
        """Awaitable ``semantic_search``: concurrent callers on one event loop share scans
        through the index's ``QueryBatcher`` (up to 32 queries per HBM pass) instead of
        blocking the loop with one scan each (the reference calls the sync client inline,
        app/main.py:1552)."""
        if query_emb is None or np.size(query_emb) == 0:
            return []
        try:
            st: Optional[IndexState] = REGISTRY.get(self.index_name, create=False)
            if st is None:
                return []
            prep = self._prepare(st, query_emb, k, filter_clause, patient_id)
            if prep is None:
                return []
            q, k_eff, q_filter = prep
            if st.batcher is None:
                from .batcher import QueryBatcher
                st.batcher = QueryBatcher(st.index)
            scores, ids = await st.batcher.search(q[0], k_eff, int(q_filter[0]) if q_filter is not None else -1)
            return self._hits(st, scores, ids, 1.0, None)
        except Exception as e:
            logger.error(f"Semantic search error: {e}")
            return []

    # ---------------------------------------------------------------------------- internals
    @staticmethod
    def _prepare(st: IndexState, query_emb, k, filter_clause, patient_id):
        q = np.asarray(query_emb, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        q = np.ascontiguousarray(q[:1])  # the reference searches row 0 only: (…)[0].tolist(), 1537
        pid = patient_id if patient_id else _patient_from_filter(filter_clause)
        q_filter = None
        if pid:
            code = st.patients.lookup(pid)
            if code is None:
                return None  # term filter on a patient that was never indexed
            q_filter = np.array([code], dtype=np.int32)
        return q, max(1, min(int(k), 32)), q_filter

    def _knn(self, query_emb, k, filter_clause, patient_id, boost, score_mode) -> List[Tuple[Dict, float]]:
        st: Optional[IndexState] = REGISTRY.get(self.index_name, create=False)
        if st is None:
            return []
        prep = self._prepare(st, query_emb, k, filter_clause, patient_id)
        if prep is None:
            return []
        q, k_eff, q_filter = prep
        scores, ids = st.index.search(q, k_eff, q_filter=q_filter)
        return self._hits(st, scores[0], ids[0], boost, score_mode)

    @staticmethod
    def _hits(st: IndexState, scores, ids, boost, score_mode) -> List[Tuple[Dict, float]]:
        out: List[Tuple[Dict, float]] = []
        with st.lock:
            for cos, row in zip(scores, ids):
                if row < 0:
                    break
                doc = st.row_doc[int(row)] if int(row) < len(st.row_doc) else None
                if doc is None:
                    continue
                hit = dict(doc)
                if config.RASS_RETURN_EMBEDDING:
                    hit["embedding"] = st.index.get_row(int(row)).tolist()  # reference quirk 5
                out.append((hit, boost * _score_out(cos, score_mode)))
        return out


# --------------------------------------------------------------------------------- write side
async def ensure_index_exists(client: Any, index_name: str) -> None:
    """app/main.py:350-579: create the per-user cosine index when absent; errors are printed
    and swallowed (578-579)."""
    try:
        REGISTRY.get(index_name, create=True)
    except Exception as e:
        print(f"[Error] OpenSearch Index could not be created: {e}")


def add_documents(index_name: str, docs: List[Dict], embeddings: np.ndarray) -> List[int]:
    """Append ``docs`` with their (un-normalised) ``embeddings`` [n, dim]; the GPU normalises
    (app/main.py:1249-1251).  ``_id = doc_id`` overwrite semantics (1260): an existing doc_id is
    tombstoned first.  Returns the row ids."""
    st = REGISTRY.get(index_name, create=True)
    emb = np.ascontiguousarray(embeddings, dtype=np.float32)
    if emb.ndim != 2 or emb.shape[0] != len(docs):
        raise ValueError(f"embeddings {emb.shape} do not match {len(docs)} docs")
    with st.lock:
        tags = np.array([st.patients.encode(d.get("patientId")) for d in docs], dtype=np.int32)
        # duplicates inside one batch: the last one wins, as with sequential bulk index ops
        last = {}
        for i, d in enumerate(docs):
            last[d.get("doc_id")] = i
        for d in docs:
            old = st.doc_row.pop(d.get("doc_id"), None)
            if old is not None:
                st.index.delete(old)
                st.row_doc[old] = None
        first = st.index.add(emb, tags=tags, normalize=True)
        rows = list(range(first, first + len(docs)))
        for i, (d, r) in enumerate(zip(docs, rows)):
            if len(st.row_doc) <= r:
                st.row_doc.extend([None] * (r + 1 - len(st.row_doc)))
            if last[d.get("doc_id")] == i:
                st.row_doc[r] = d
                st.doc_row[d.get("doc_id")] = r
            else:
                st.index.delete(r)  # superseded inside the same batch
        return rows


async def store_fhir_docs_in_opensearch(structured_docs: List[Dict], unstructured_docs: List[Dict], client: Any,
                                        index_name: str, embed_fn=None) -> None:
    """app/main.py:1211-1282 with the HTTP hops removed: structured docs are kept by doc_id
    (they carry no embedding, so k-NN never returns them — same as the reference); the
    unstructured docs are embedded (``embed_texts_in_batches``), normalised and indexed."""
    await ensure_index_exists(client, index_name)
    st = REGISTRY.get(index_name, create=True)
    if structured_docs:
        try:
            with st.lock:
                for doc in structured_docs:
                    st.structured[doc["doc_id"]] = doc
            logger.info(f"Indexed {len(structured_docs)} structured docs, errors: []")
        except Exception as e:
            logger.error(f"Structured docs indexing error: {e}")
    if not unstructured_docs:
        return
    if embed_fn is None:
        from .embedding import embed_texts_in_batches as embed_fn
    un_texts = [d["unstructuredText"] for d in unstructured_docs]
    embeddings = await embed_fn(un_texts, batch_size=config.BATCH_SIZE)
    try:
        add_documents(index_name, unstructured_docs, embeddings)
        logger.info(f"Indexed {len(unstructured_docs)} unstructured docs, errors: []")
    except Exception as e:
        logger.error(f"Unstructured docs indexing error: {e}")


def install(module) -> None:
    """Rebind the reference's hot-path names on an imported ``main`` / ``embedding_gen``
    module: routes, chunk_text, Prisma and LLM code stay untouched."""
    from . import embedding
    for name, obj in (("OpenSearchIndexer", HipIndexer), ("ensure_index_exists", ensure_index_exists),
                      ("store_fhir_docs_in_opensearch", store_fhir_docs_in_opensearch),
                      ("ollama_embed_text", embedding.ollama_embed_text),
                      ("embed_texts_in_batches", embedding.embed_texts_in_batches),
                      ("embed_query", embedding.embed_query)):
        if hasattr(module, name):
            setattr(module, name, obj)

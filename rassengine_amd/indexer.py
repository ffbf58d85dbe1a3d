"""``HipIndexer`` — drop-in for the reference's ``OpenSearchIndexer`` (app/main.py:1395-2150) —
and ``store_fhir_docs_in_opensearch`` / ``ensure_index_exists`` for the write side
(app/main.py:350-579, 1211-1282).

What moves to the GPU and what stays:

* the four knn-bearing builders are answered here: ``semantic_search`` (1527-1560), ``hybrid_search``
  (knn clause 1595, boost 2.0), ``hybrid_structured_search`` (1754, boost 2.0, ``term: doc_type =
  structured`` 1765) and ``multi_intent_search`` (2003, boost 1.5) — exact cosine over the HBM index;
* ``has_any_data`` (1470-1478) is answered from the host-side maps;
* the eight BM25 / aggregate builders (``exact_match_search`` 1480, ``structured_search`` 1617,
  ``aggregate_search`` 1777, ``comparison_search`` 1810, ``temporal_search`` 1866, ``explanatory_search``
  1920, ``entity_specific_search`` 2029, ``document_fetch_search`` 2120) are Lucene text search: every
  attribute this class does not define is DELEGATED to a lazily built instance of the module's own
  ``OpenSearchIndexer`` (``install()`` keeps it), so ``ask()``'s 11-entry method table (2855-2867) and the
  DOCUMENT_FETCH branch (2805) keep working.  When a text engine is kept, the BM25 ``should`` clauses of
  the three hybrid builders are fetched from it and summed with the knn sub-score by ``doc_id``.

Same names, argument meaning and error behaviour as the reference:

* ``OpenSearchIndexer(client, index_name)`` is built per request (2802) -> construction is O(1);
* every knn-bearing method returns ``[(doc_dict, float(score))]`` best first, ``[]`` on an empty query
  embedding / blank query text (1534, 1570, 1712, 1970) and on ANY exception (log + ``[]``, 1558-1560);
* ``ask()`` passes ``query=`` to ``semantic_search`` too (2879-2885), which the reference's own signature
  does not accept (TypeError -> HTTP 500): accepted and ignored here;
* the query is re-normalised with ``e / (||e|| + 1e-9)`` (1536-1537) — on the GPU; only row 0 is searched
  (``[0].tolist()``);
* ``patient_id`` -> ``term: patientId`` (1549) as an exact PRE-filter on the row tag; ``k`` is passed
  through as is (k > 32 is served in passes; nothing is clamped);
* ``filter_clause``: ``ask()`` passes the NER entity list there (2770), which makes the reference's query
  malformed.  Tolerated, not replicated: only ``{"term": {"patientId": ...}}`` /
  ``{"term": {"doc_type": ...}}`` dicts are honoured, anything else is ignored;
* quirk 3 (``hybrid_structured_search`` raises ``KeyError`` without filter / patient, 1764): not
  replicated — the ``doc_type`` filter is applied on its own;
* ``_score``: OpenSearch k-NN cosinesimil reports ``1 / (2 - cos)`` (unverified offline, so
  configurable: ``RASS_SCORE_MODE=opensearch|cosine``); a knn ``should`` clause contributes
  ``boost * _score``.  ``ask()`` never reads it.
"""
from __future__ import annotations

import asyncio
import inspect
import logging
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from . import config, prefetch
from .docstore import REGISTRY, IndexState

logger = logging.getLogger("rassengine_amd")

TOP_K = config.TOP_K

# the reference's knn boosts
BOOST_HYBRID = 2.0             # app/main.py:1595
BOOST_HYBRID_STRUCTURED = 2.0  # app/main.py:1754
BOOST_MULTI_INTENT = 1.5       # app/main.py:2003
# how much deeper than k the two ranked lists are read before they are fused by doc_id
FUSION_OVERFETCH = 4


def _score_out(cos: float, mode: Optional[str] = None) -> float:
    mode = mode or config.RASS_SCORE_MODE
    if mode == "cosine":
        return float(cos)
    return float(1.0 / (2.0 - float(cos)))  # OpenSearch k-NN cosinesimil: 1 / (1 + (1 - cos))


def _term_from_filter(filter_clause: Any, field: str) -> Optional[Any]:
    if isinstance(filter_clause, dict):
        term = filter_clause.get("term")
        if isinstance(term, dict) and field in term:
            return term[field]
    return None


def _empty(query_emb) -> bool:
    return query_emb is None or np.size(query_emb) == 0


class HipIndexer:
    """Exact cosine k-NN over the HBM-resident index named ``index_name``; everything that is not
    k-NN is delegated to the original ``OpenSearchIndexer`` (``_original_cls``, set by ``install``)."""

    _original_cls: Optional[type] = None

    def __init__(self, client: Any = None, index_name: str = ""):
        self.client = client          # handed to the delegate; the k-NN path has no HTTP hop any more
        self.index_name = index_name
        self._delegate = None

    # ------------------------------------------------------------------ delegation (BM25 builders)
    def _original(self):
        if self._delegate is None:
            cls = type(self)._original_cls
            if cls is None:
                return None
            self._delegate = cls(self.client, self.index_name)
        return self._delegate

    def __getattr__(self, name: str):
        # only reached for attributes this class does not define: the 8 text / aggregate builders,
        # text_fields / keyword_fields / date_fields, ...
        if name.startswith("_"):
            raise AttributeError(name)
        orig = self._original()
        if orig is None:
            raise AttributeError(
                f"{type(self).__name__!s} has no attribute {name!r}: it is one of the reference's text-search "
                "builders and no original OpenSearchIndexer was kept (use rassengine_amd.indexer.install(module))")
        return getattr(orig, name)

    def _text_engine(self):
        """The delegate, when it can actually reach a text engine (a client was given)."""
        return self._original() if self.client else None

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
        if _empty(query_emb):
            return []
        try:
            return self._knn(query_emb, k, filter_clause, patient_id, boost=1.0)
        except Exception as e:  # reference: log and return [] (1558-1560)
            logger.error(f"Semantic search error: {e}")
            return []

    def knn_scores(self, query_emb: np.ndarray, k: int = TOP_K, boost: float = 1.0,
                   filter_clause: Optional[Dict] = None, patient_id: Optional[str] = None,
                   doc_type: Optional[str] = None) -> List[Tuple[Dict, float]]:
        """The knn ``should`` clause on its own: ``boost * _score`` per hit."""
        if _empty(query_emb):
            return []
        try:
            return self._knn(query_emb, k, filter_clause, patient_id, boost=boost, doc_type=doc_type)
        except Exception as e:
            logger.error(f"kNN score error: {e}")
            return []

    # ------------------------------------------------------------------ app/main.py:1562-1615
    def hybrid_search(self, query: str, query_emb: np.ndarray, k: int = TOP_K, filter_clause: Optional[Dict] = None,
                      patient_id: Optional[str] = None) -> List[Tuple[Dict, float]]:
        if not query or not query.strip() or _empty(query_emb):
            return []  # 1570-1571
        return self._hybrid("hybrid_search", "Hybrid search error", query, query_emb, k, filter_clause, patient_id,
                            BOOST_HYBRID, None)

    # ------------------------------------------------------------------ app/main.py:1710-1775
    def hybrid_structured_search(self, query: str, query_emb: np.ndarray, k: int = TOP_K,
                                 filter_clause: Optional[Dict] = None, patient_id: Optional[str] = None
                                 ) -> List[Tuple[Dict, float]]:
        if not query or not query.strip() or _empty(query_emb):
            return []  # 1712-1713
        # term: doc_type = structured (1765).  The reference stores no embedding on structured docs
        # (1222-1240), so its knn clause matches none of them; here the filter selects the rows whose
        # doc_type byte says "structured" — none with the reference's ingest, all of them for a
        # deployment that embeds its structured docs.
        return self._hybrid("hybrid_structured_search", "Hybrid structured search error", query, query_emb, k,
                            filter_clause, patient_id, BOOST_HYBRID_STRUCTURED, "structured")

    # ------------------------------------------------------------------ app/main.py:1962-2027
    def multi_intent_search(self, query: str, query_emb: np.ndarray, k: int = TOP_K,
                            filter_clause: Optional[Dict] = None, patient_id: Optional[str] = None
                            ) -> List[Tuple[Dict, float]]:
        if not query or not query.strip() or _empty(query_emb):
            return []  # 1970-1971
        return self._hybrid("multi_intent_search", "Multi-intent search error", query, query_emb, k, filter_clause,
                            patient_id, BOOST_MULTI_INTENT, None)

    async def asemantic_search(self, query_emb: np.ndarray, k: int = TOP_K, filter_clause: Optional[Dict] = None,
                               patient_id: Optional[str] = None, query: Optional[str] = None, **_ignored
                               ) -> List[Tuple[Dict, float]]:
        """Awaitable ``semantic_search``: concurrent callers on one event loop share scans
        through the index's ``QueryBatcher`` (up to 32 queries per HBM pass) instead of
        blocking the loop with one scan each (the reference calls the sync client inline,
        app/main.py:1552)."""
        if _empty(query_emb):
            return []
        try:
            st: Optional[IndexState] = REGISTRY.get(self.index_name, create=False)
            if st is None:
                return []
            prep = self._prepare(st, query_emb, k, filter_clause, patient_id, None)
            if prep is None:
                return []
            q, k_eff, (fval, fmask) = prep
            eng = getattr(st.index, "engine", None)
            if eng is not None and hasattr(eng, "search_multi") and k_eff <= 32:
                # one batcher per ENGINE: concurrent users' per-user indices share scan launches
                if getattr(eng, "_cross_batcher", None) is None:
                    from .batcher import CrossIndexBatcher
                    eng._cross_batcher = CrossIndexBatcher(eng)
                scores, ids = await eng._cross_batcher.search(st.index, q[0], k_eff, fval, fmask)
                return self._hits(st, scores, ids, 1.0, None)
            if st.batcher is None:
                from .batcher import QueryBatcher
                st.batcher = QueryBatcher(st.index)
            scores, ids = await st.batcher.search(q[0], k_eff, fval, fmask)
            return self._hits(st, scores, ids, 1.0, None)
        except Exception as e:
            logger.error(f"Semantic search error: {e}")
            return []

    # ---------------------------------------------------------------------------- internals
    @staticmethod
    def _prepare(st: IndexState, query_emb, k, filter_clause, patient_id, doc_type):
        q = np.asarray(query_emb, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        q = np.ascontiguousarray(q[:1])  # the reference searches row 0 only: (…)[0].tolist(), 1537
        # the reference ANDs its filter list (filter_clause, then term: patientId, then — hybrid_structured_search — term:
        # doc_type, 1543-1550 / 1760-1765): two different values for one field match nothing
        f_pid, f_dtype = _term_from_filter(filter_clause, "patientId"), _term_from_filter(filter_clause, "doc_type")
        if (patient_id and f_pid is not None and str(f_pid) != str(patient_id)) or \
                (doc_type and f_dtype is not None and str(f_dtype) != str(doc_type)):
            return None
        pid = patient_id if patient_id else f_pid
        dtype = doc_type if doc_type else f_dtype
        flt = st.filter_for(pid, dtype)
        if flt is None:
            return None  # term filter on a value that was never indexed
        return q, max(1, int(k)), flt

    def _knn(self, query_emb, k, filter_clause, patient_id, boost, doc_type=None, score_mode=None
             ) -> List[Tuple[Dict, float]]:
        st: Optional[IndexState] = REGISTRY.get(self.index_name, create=False)
        if st is None:
            return []
        prep = self._prepare(st, query_emb, k, filter_clause, patient_id, doc_type)
        if prep is None:
            return []
        q, k_eff, (fval, fmask) = prep
        parked = prefetch.take(st, q, k_eff, fval, fmask)
        if parked is not None:      # this request's scan was shared with the other requests in flight
            return self._hits(st, parked[0], parked[1], boost, score_mode)
        if fmask:
            scores, ids = st.index.search(q, k_eff, q_filter=np.array([fval], dtype=np.int32),
                                          q_filter_mask=np.array([fmask], dtype=np.int32))
        else:
            scores, ids = st.index.search(q, k_eff)
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

    def _hybrid(self, method: str, err_label: str, query, query_emb, k, filter_clause, patient_id, boost, doc_type
                ) -> List[Tuple[Dict, float]]:
        """``bool.should`` of the reference's hybrid builders: score = sum of the matching clauses.
        The knn clause comes from the HBM index; the text clauses from the kept text engine (when there
        is one), whose own knn clause matches nothing because no vector is stored there any more.  The
        two ranked lists are read FUSION_OVERFETCH x k deep and summed by ``doc_id``."""
        try:
            k = max(1, int(k))
            text = self._text_engine()
            depth = k * FUSION_OVERFETCH if text is not None else k
            knn = self._knn(query_emb, depth, filter_clause, patient_id, boost=boost, doc_type=doc_type)
            if text is None:
                return knn[:k]
            try:
                bm25 = getattr(text, method)(query, query_emb, k=depth, filter_clause=filter_clause,
                                             patient_id=patient_id)
            except Exception as e:  # e.g. the reference's own KeyError in hybrid_structured_search (quirk 3)
                logger.error(f"{err_label} (text clauses): {e}")
                bm25 = []
            fused: Dict[Any, List] = {}
            for doc, s in knn:
                fused[doc.get("doc_id", id(doc))] = [doc, float(s)]
            for doc, s in bm25 or []:
                key = doc.get("doc_id", id(doc))
                if key in fused:
                    fused[key][1] += float(s)
                else:
                    fused[key] = [doc, float(s)]
            ranked = sorted(fused.values(), key=lambda e: -e[1])
            return [(d, s) for d, s in ranked[:k]]
        except Exception as e:
            logger.error(f"{err_label}: {e}")
            return []


# --------------------------------------------------------------------------------- write side
_ORIGINALS: Dict[int, Dict[str, Any]] = {}   # id(module) -> the names install() replaced


def _originals_for(fn_globals_name: Optional[str] = None) -> Dict[str, Any]:
    for rec in _ORIGINALS.values():
        if fn_globals_name is None or rec.get("__name__") == fn_globals_name:
            return rec
    return {}


async def ensure_index_exists(client: Any, index_name: str) -> None:
    """app/main.py:350-579: create the per-user cosine index when absent; errors are printed
    and swallowed (578-579).  When a text engine is kept (``client`` given and the module's original
    function was recorded by ``install``) its index — the ~90 text fields — is ensured as well."""
    st = None
    try:
        st = REGISTRY.get(index_name, create=True)
    except Exception as e:
        print(f"[Error] OpenSearch Index could not be created: {e}")
    orig = _originals_for().get("ensure_index_exists")
    if client and orig is not None:
        try:
            await orig(client, index_name)
        except Exception as e:
            print(f"[Error] OpenSearch Index could not be created: {e}")
    # ask() awaits this right after embed_query and right before its synchronous search (app/main.py:2800-2802):
    # the one place where the k-NN scans of concurrent requests can share a launch (prefetch.py)
    await prefetch.run(st)


def add_documents(index_name: str, docs: List[Dict], embeddings: Optional[np.ndarray],
                  texts: Optional[List[str]] = None) -> List[int]:
    """Append ``docs`` with their (un-normalised) ``embeddings`` [n, dim]; the GPU normalises
    (app/main.py:1249-1251).  ``_id = doc_id`` overwrite semantics (1260): the new rows are appended
    FIRST and the rows they supersede are tombstoned only after the append succeeded, so a failed add
    (OOM on slab growth, HIP error) loses nothing.  Returns the row ids.  With ``texts`` instead of
    ``embeddings`` (a multi-GPU index whose ranks have encoders, ``index.can_encode``) the texts are embedded
    where their rows will live: data-parallel over the ranks, no vector leaves its GPU."""
    st = REGISTRY.get(index_name, create=True)
    if texts is not None:
        if len(texts) != len(docs):
            raise ValueError(f"{len(texts)} texts do not match {len(docs)} docs")
        emb = None
    else:
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        if emb.ndim != 2 or emb.shape[0] != len(docs):
            raise ValueError(f"embeddings {emb.shape} do not match {len(docs)} docs")
    with st.lock:
        tags = np.array([st.tag_of(d) for d in docs], dtype=np.int32)
        # duplicates inside one batch: the last one wins, as with sequential bulk index ops
        last = {}
        for i, d in enumerate(docs):
            last[d.get("doc_id")] = i
        if emb is None:
            first = st.index.add_texts(list(texts), tags=tags, normalize=True)
        else:
            first = st.index.add(emb, tags=tags, normalize=True)      # raises -> nothing was changed
        rows = list(range(first, first + len(docs)))
        for d in docs:
            old = st.doc_row.pop(d.get("doc_id"), None)
            if old is not None:
                st.index.delete(old)
                st.row_doc[old] = None
                st.note_deleted(old)
        for i, (d, r) in enumerate(zip(docs, rows)):
            if len(st.row_doc) <= r:
                st.row_doc.extend([None] * (r + 1 - len(st.row_doc)))
            if last[d.get("doc_id")] == i:
                st.row_doc[r] = d
                st.doc_row[d.get("doc_id")] = r
            else:
                st.index.delete(r)  # superseded inside the same batch
        return rows


async def _call_embed(embed_fn, texts: List[str]) -> np.ndarray:
    """main.py's embed_texts_in_batches takes batch_size (240-242), embedding_gen.py's does not (173)."""
    try:
        params = inspect.signature(embed_fn).parameters
    except (TypeError, ValueError):
        params = {}
    if "batch_size" in params:
        return await embed_fn(texts, batch_size=config.BATCH_SIZE)
    return await embed_fn(texts)


async def store_fhir_docs_in_opensearch(structured_docs: List[Dict], unstructured_docs: List[Dict], client: Any,
                                        index_name: str, embed_fn=None) -> None:
    """app/main.py:1211-1282 (embedding_gen.py:1061-1132) with the vector hops removed: the
    unstructured docs are embedded (``embed_texts_in_batches``), normalised and indexed in HBM.
    Structured docs carry no embedding (k-NN never returns them — same as the reference); they are
    kept by doc_id and, when a text engine is kept (``client`` given, ``install`` recorded the module's
    ``bulk``), bulk-indexed there like the reference does — as are the unstructured docs' TEXT (without
    the 1024-float ``embedding`` field), so the BM25 builders keep seeing every document."""
    await ensure_index_exists(client, index_name)
    st = REGISTRY.get(index_name, create=True)
    bulk = _originals_for().get("bulk") if client else None

    def _bulk(docs: List[Dict], label: str) -> None:
        if bulk is None or not docs:
            return
        actions = [{"_op_type": "index", "_index": index_name, "_id": d["doc_id"], "_source": d,
                    "_routing": d.get("patientId")} for d in docs]
        step = max(1, config.BATCH_SIZE)
        for a in range(0, len(actions), step):
            try:
                ok, errors = bulk(client, actions[a:a + step])
                logger.info(f"Indexed {ok} {label} docs, errors: {errors}")
            except Exception as e:
                logger.error(f"{label.capitalize()} docs indexing error: {e}")

    if structured_docs:
        try:
            with st.lock:
                for doc in structured_docs:
                    st.structured[doc["doc_id"]] = doc
                    st.note_structured(doc["doc_id"])
            if bulk is None:
                logger.info(f"Indexed {len(structured_docs)} structured docs, errors: []")
        except Exception as e:
            logger.error(f"Structured docs indexing error: {e}")
        _bulk(structured_docs, "structured")
    if not unstructured_docs:
        return
    un_texts = [d["unstructuredText"] for d in unstructured_docs]
    # a multi-GPU index whose ranks hold encoders embeds the texts where their rows will live
    data_parallel = embed_fn is None and getattr(st.index, "can_encode", False)
    if embed_fn is None:
        from .embedding import embed_texts_in_batches as embed_fn
    if not data_parallel:
        embeddings = await _call_embed(embed_fn, un_texts)     # an embedding error propagates, as in the reference
    try:
        if data_parallel:
            await asyncio.to_thread(add_documents, index_name, unstructured_docs, None, un_texts)
        else:
            add_documents(index_name, unstructured_docs, embeddings)
        if bulk is None:
            logger.info(f"Indexed {len(unstructured_docs)} unstructured docs, errors: []")
    except Exception as e:
        logger.error(f"Unstructured docs indexing error: {e}")
        return
    _bulk(unstructured_docs, "unstructured")   # text only: the vectors live in HBM


# --------------------------------------------------------------------------------- install
def make_indexer_class(original_cls: Optional[type]) -> type:
    """A ``HipIndexer`` subclass that delegates the BM25 builders to ``original_cls``."""
    if original_cls is not None and isinstance(original_cls, type) and issubclass(original_cls, HipIndexer):
        return original_cls  # already installed
    if original_cls is None or original_cls is object or not callable(original_cls):
        return HipIndexer
    return type("HipIndexer", (HipIndexer,), {"_original_cls": original_cls, "__doc__": HipIndexer.__doc__})


def install(module) -> None:
    """Rebind the reference's hot-path names on an imported ``main`` / ``embedding_gen`` module
    (SURVEY §8b): routes, chunk_text, Prisma and LLM code stay untouched.

    * ``OpenSearchIndexer`` -> a ``HipIndexer`` subclass that KEEPS the module's own class for the eight
      BM25 / aggregate builders (``ask()`` builds a table of all eleven search methods before it
      dispatches, app/main.py:2855-2867, and calls ``document_fetch_search`` at 2805);
    * ``embed_*``: the flavour of the module being patched — ``app/main.py:225-274`` raises on errors and
      returns ``np.array([])`` for an empty list; ``app/embedding_gen.py:152-192`` takes no ``batch_size``,
      returns ``zeros((0, EMBED_DIM))`` and turns errors into zero vectors;
    * ``ensure_index_exists`` / ``store_fhir_docs_in_opensearch``: vectors to HBM; the module's originals and
      its ``bulk`` are remembered so a kept text engine still receives the text.
    Idempotent."""
    from . import embedding
    rec = _ORIGINALS.get(id(module))
    if rec is None:
        rec = {"__name__": getattr(module, "__name__", None)}
        for name in ("OpenSearchIndexer", "ensure_index_exists", "store_fhir_docs_in_opensearch", "bulk",
                     "ollama_embed_text", "embed_texts_in_batches", "embed_query"):
            if hasattr(module, name):
                rec[name] = getattr(module, name)
        _ORIGINALS[id(module)] = rec
    orig_embed = rec.get("embed_texts_in_batches")
    gen_flavour = False
    if callable(orig_embed):
        try:
            gen_flavour = "batch_size" not in inspect.signature(orig_embed).parameters
        except (TypeError, ValueError):
            gen_flavour = False
    elif str(rec.get("__name__") or "").endswith("embedding_gen"):
        gen_flavour = True
    emb = embedding.GEN_FLAVOUR if gen_flavour else embedding.MAIN_FLAVOUR
    bindings = (("OpenSearchIndexer", make_indexer_class(rec.get("OpenSearchIndexer"))),
                ("ensure_index_exists", ensure_index_exists),
                ("store_fhir_docs_in_opensearch", store_fhir_docs_in_opensearch),
                ("ollama_embed_text", emb["ollama_embed_text"]),
                ("embed_texts_in_batches", emb["embed_texts_in_batches"]),
                ("embed_query", emb["embed_query"]))
    for name, obj in bindings:
        if hasattr(module, name):
            setattr(module, name, obj)


def uninstall(module) -> None:
    """Put the module's own names back (tests)."""
    rec = _ORIGINALS.pop(id(module), None)
    if rec:
        for name, obj in rec.items():
            if name != "__name__" and name != "bulk":
                setattr(module, name, obj)

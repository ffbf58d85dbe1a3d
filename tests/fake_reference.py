"""A stand-in for the reference's `app/main.py` / `app/embedding_gen.py` MODULE SURFACE, for the drop-in
tests: the names `install()` rebinds, a 12-method stub `OpenSearchIndexer` (the text engine the BM25 builders
talk to, recording every call) and `ask_shaped()`, which makes the calls `ask()` makes around the hot path
(app/main.py:2800-2892) in the same order and with the same argument shapes:

    query_emb = await embed_query(query)                       2800
    await ensure_index_exists(os_client, index_name)           2801
    os_indexer = OpenSearchIndexer(os_client, index_name)      2802
    DOCUMENT_FETCH -> os_indexer.document_fetch_search(query, k=, filter_clause=, patient_id=)   2804-2807
    a dict of ELEVEN bound methods of os_indexer               2855-2867
    AGGREGATE -> search_method(query, filter_clause=, patient_id=)                                2869-2873
    SEMANTIC / HYBRID / HYBRID_STRUCTURED / MULTI_INTENT -> (query=, query_emb=, k=, filter_clause=, patient_id=)
    everything else -> (query=, k=, filter_clause=, patient_id=)                                  2875-2892

Nothing here is the reference's code; module-level names are resolved at call time exactly as there, which is
what makes rebinding them a drop-in.  Test infrastructure only."""
import types

import numpy as np

CALLS = []          # (method name, kwargs) seen by the stub text engine
BULKED = []         # actions handed to bulk()
ENSURED = []        # index names the ORIGINAL ensure_index_exists was asked for

TEXT_METHODS = ["exact_match_search", "structured_search", "aggregate_search", "comparison_search",
                "temporal_search", "explanatory_search", "entity_specific_search", "document_fetch_search"]
KNN_METHODS = ["semantic_search", "hybrid_search", "hybrid_structured_search", "multi_intent_search"]


class StubTextEngineIndexer:
    """The module's own OpenSearchIndexer: all 12 builders + has_any_data, BM25-ish canned answers."""

    def __init__(self, client, index_name):
        self.client = client
        self.index_name = index_name
        self.text_fields = ["unstructuredText^3"]
        self.keyword_fields = ["patientGender^3"]
        self.date_fields = ["patientDOB"]

    def has_any_data(self):
        return bool(self.client)

    def _text_hits(self, name, **kw):
        CALLS.append((name, kw))
        docs = getattr(self.client, "text_docs", [])
        return [(dict(d), float(s)) for d, s in docs[: kw.get("k", 3)]]

    def aggregate_search(self, query, filter_clause=None, patient_id=None):
        CALLS.append(("aggregate_search", {"query": query, "filter_clause": filter_clause, "patient_id": patient_id}))
        return {"aggregations": {"n": 1}}

    def semantic_search(self, query_emb, k=3, filter_clause=None, patient_id=None):   # no `query` parameter (1527)
        return self._text_hits("semantic_search", k=k)


def _make_text_method(name):
    def method(self, query, k=3, filter_clause=None, patient_id=None):
        return self._text_hits(name, query=query, k=k, filter_clause=filter_clause, patient_id=patient_id)
    method.__name__ = name
    return method


def _make_hybrid_method(name):
    def method(self, query, query_emb, k=3, filter_clause=None, patient_id=None):
        if name == "hybrid_structured_search" and not (filter_clause or patient_id):
            raise KeyError("filter")                                                  # quirk 3 (main.py:1764)
        return self._text_hits(name, query=query, k=k, filter_clause=filter_clause, patient_id=patient_id)
    method.__name__ = name
    return method


for _n in TEXT_METHODS:
    if _n != "aggregate_search":
        setattr(StubTextEngineIndexer, _n, _make_text_method(_n))
for _n in KNN_METHODS[1:]:
    setattr(StubTextEngineIndexer, _n, _make_hybrid_method(_n))


def make_module(name="main", gen_flavour=False):
    """A fresh module object with the reference's hot-path names bound to 'original' stand-ins."""
    m = types.ModuleType(name)
    m.OpenSearchIndexer = StubTextEngineIndexer

    async def ensure_index_exists(client, index_name):
        ENSURED.append(index_name)

    async def store_fhir_docs_in_opensearch(structured_docs, unstructured_docs, client, index_name):
        raise AssertionError("the original write path must not run once the engine is installed")

    async def ollama_embed_text(text):
        raise AssertionError("the original Ollama client must not run once the engine is installed")

    async def embed_query(query):
        raise AssertionError("the original Ollama client must not run once the engine is installed")

    if gen_flavour:
        async def embed_texts_in_batches(texts):
            raise AssertionError("original")
    else:
        async def embed_texts_in_batches(texts, batch_size=64):
            raise AssertionError("original")

    def bulk(client, actions):
        BULKED.extend(actions)
        return len(actions), []

    def chunk_text(text, chunk_size=512):
        words = text.split()
        return [" ".join(words[i:i + chunk_size]) for i in range(0, len(words), chunk_size)]

    m.ensure_index_exists = ensure_index_exists
    m.store_fhir_docs_in_opensearch = store_fhir_docs_in_opensearch
    m.ollama_embed_text = ollama_embed_text
    m.embed_texts_in_batches = embed_texts_in_batches
    if not gen_flavour:
        m.embed_query = embed_query
    m.bulk = bulk
    m.chunk_text = chunk_text

    async def ask_shaped(query, intent, top_k, os_client, index_name, filter_clause=None, primary_patient_id=None):
        query_emb = await m.embed_query(query)
        await m.ensure_index_exists(os_client, index_name)
        os_indexer = m.OpenSearchIndexer(os_client, index_name)
        if intent == "DOCUMENT_FETCH":
            return os_indexer.document_fetch_search(query, k=top_k, filter_clause=filter_clause,
                                                    patient_id=primary_patient_id)
        search_methods = {
            "SEMANTIC": os_indexer.semantic_search,
            "KEYWORD": os_indexer.exact_match_search,
            "HYBRID": os_indexer.hybrid_search,
            "STRUCTURED": os_indexer.structured_search,
            "HYBRID_STRUCTURED": os_indexer.hybrid_structured_search,
            "AGGREGATE": os_indexer.aggregate_search,
            "COMPARISON": os_indexer.comparison_search,
            "TEMPORAL": os_indexer.temporal_search,
            "EXPLANATORY": os_indexer.explanatory_search,
            "MULTI_INTENT": os_indexer.multi_intent_search,
            "ENTITY_SPECIFIC": os_indexer.entity_specific_search,
        }
        search_method = search_methods.get(intent, os_indexer.hybrid_search)
        if intent == "AGGREGATE":
            return search_method(query, filter_clause=filter_clause, patient_id=primary_patient_id)
        if intent in ["SEMANTIC", "HYBRID", "HYBRID_STRUCTURED", "MULTI_INTENT"]:
            return search_method(query=query, query_emb=query_emb, k=top_k, filter_clause=filter_clause,
                                 patient_id=primary_patient_id)
        return search_method(query=query, k=top_k, filter_clause=filter_clause, patient_id=primary_patient_id)

    m.ask_shaped = ask_shaped
    return m


class FakeClient:
    """Truthy `os_client` whose text engine 'holds' some BM25-scored docs."""

    def __init__(self, text_docs=()):
        self.text_docs = list(text_docs)


def reset():
    CALLS.clear()
    BULKED.clear()
    ENSURED.clear()

"""Generates tests/golden/reference_boundary.json by EXECUTING the reference's own boundary code — ``class
OpenSearchIndexer`` (app/main.py:1395-2150) and ``store_fhir_docs_in_opensearch`` (app/main.py:1211-1282) — lifted from
/root/reference/app/main.py by ast WITHOUT running module scope (which needs dotenv / prisma / opensearchpy and fetches
HF models), against a RECORDING fake OpenSearch client / fake ``bulk`` / fake ``embed_texts_in_batches``.

What is stored is data only: the INPUTS (seeded query vectors, k, filter clauses, patient ids, docs, raw embeddings) and
what the reference EMITTED for them — the k-NN request bodies (normalised ``vector``, ``size``, ``k``, ``terminate_after``,
filter terms, ``routing``, clause boosts), its return values for canned hits / empty embeddings / a raising client, and the
bulk actions of the write side (``_op_type``, ``_index``, ``_id``, ``_routing``, the normalised ``embedding`` rows, the
slicing into bulks of BATCH_SIZE).  No reference source text is written anywhere.

Run from the repo root (build container only: /root/reference does not exist on the GPU box):
    python tests/golden/make_reference_boundary_fixtures.py
"""
import ast
import asyncio
import copy
import json
import logging
import os
import re  # noqa: F401
from datetime import datetime, timedelta, timezone  # noqa: F401
from typing import Any, Dict, List, Optional, Tuple, Union  # noqa: F401

import numpy as np

REF = "/root/reference/app/main.py"
HERE = os.path.dirname(os.path.abspath(__file__))
DIM = 128          # the reference's code does not depend on EMBED_DIM; 128 keeps the fixture small (one 1024-d case too)
TOP_K = 3          # app/main.py:88 default
BATCH_SIZE = 64    # app/main.py:78 default


class RecordingClient:
    """Truthy ``os_client`` stand-in: records every ``search`` request, answers with canned hits."""

    def __init__(self, hits=None, fail=False):
        self.requests = []
        self.hits = hits or []
        self.fail = fail

    def search(self, index=None, body=None, routing=None, **kw):
        self.requests.append({"index": index, "body": copy.deepcopy(body), "routing": routing, "extra": sorted(kw)})
        if self.fail:
            raise RuntimeError("connection refused")
        return {"hits": {"hits": [{"_source": dict(s), "_score": sc} for s, sc in self.hits]}}

    def count(self, index=None, **kw):
        if self.fail:
            raise RuntimeError("connection refused")
        return {"count": len(self.hits)}


def lift():
    """exec the two definitions in a namespace that provides the module-level names they use."""
    bulked = []

    def bulk(client, actions):
        bulked.append(copy.deepcopy(actions))
        return len(actions), []

    embed_calls = []

    async def ensure_index_exists(client, index_name):
        return None

    ns = {"List": List, "Dict": Dict, "Optional": Optional, "Tuple": Tuple, "Any": Any, "Union": Union, "np": np, "re": re,
          "json": json, "datetime": datetime, "timedelta": timedelta, "timezone": timezone, "OpenSearch": object,
          "logger": logging.getLogger("reference"), "TOP_K": TOP_K, "BATCH_SIZE": BATCH_SIZE, "EMBED_DIM": DIM,
          "bulk": bulk, "ensure_index_exists": ensure_index_exists, "print": lambda *a, **k: None}

    async def embed_texts_in_batches(texts, batch_size=BATCH_SIZE):
        embed_calls.append((list(texts), batch_size))
        return ns["_fake_embeddings"](texts)

    ns["embed_texts_in_batches"] = embed_texts_in_batches
    tree = ast.parse(open(REF, encoding="utf-8").read())
    for node in tree.body:
        if (isinstance(node, ast.ClassDef) and node.name == "OpenSearchIndexer") or \
                (isinstance(node, ast.AsyncFunctionDef) and node.name == "store_fhir_docs_in_opensearch"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
    return ns, bulked, embed_calls


def f32(a):
    return np.asarray(a, dtype=np.float32)


def summarise(req):
    """What the hot path is pinned on, out of a recorded request: size / terminate_after / routing, the knn clause (where it
    sits, its normalised vector, k, boost), the filter list, and of the TEXT clauses only their kind and boost (their field
    lists are Lucene's business and the reference's literals: not stored)."""
    body = req["body"]
    q = body["query"]
    out = {"index": req["index"], "routing": req["routing"], "size": body.get("size"), "terminate_after": body.get("terminate_after"),
           "filter": None, "should": None, "minimum_should_match": None}
    if "knn" in q:
        out["knn_at"], knn = "query", q["knn"]["embedding"]
    else:
        b = q["bool"]
        out["filter"] = b.get("filter")
        out["minimum_should_match"] = b.get("minimum_should_match")
        if "must" in b:
            out["knn_at"], knn = "bool.must", b["must"][0]["knn"]["embedding"]
        else:
            knn = None
            out["should"] = []
            for c in b["should"]:
                (kind, spec), = c.items()
                if kind == "knn":
                    out["knn_at"], knn = "bool.should", spec["embedding"]
                    out["should"].append({"kind": "knn", "boost": spec["embedding"].get("boost")})
                else:
                    out["should"].append({"kind": kind, "boost": spec.get("boost") if isinstance(spec, dict) else None})
    out["knn"] = {"vector": knn["vector"], "k": knn["k"], "boost": knn.get("boost")}
    return out


def main():
    ns, bulked, embed_calls = lift()
    Indexer = ns["OpenSearchIndexer"]
    rng = np.random.default_rng(20261004)
    canned = [({"doc_id": "d-1", "patientId": "p1", "doc_type": "unstructured", "unstructuredText": "alpha"}, 0.91),
              ({"doc_id": "d-2", "patientId": "p2", "doc_type": "unstructured", "unstructuredText": "beta"}, 0.77)]
    out = {"dim": DIM, "top_k_default": TOP_K, "batch_size": BATCH_SIZE, "search": [], "store": None, "has_any_data": []}

    # ---- read side: the four knn-bearing builders
    queries = {"q_scaled": f32(rng.standard_normal((1, DIM)) * 7.5), "q_tiny": f32(rng.standard_normal((1, DIM)) * 1e-4),
               "q_two_rows": f32(rng.standard_normal((2, DIM))), "q_zero": np.zeros((1, DIM), dtype=np.float32),
               "q_1024": f32(rng.standard_normal((1, 1024)) * 3.0)}
    knn_methods = (("semantic_search", False), ("hybrid_search", True), ("hybrid_structured_search", True),
                   ("multi_intent_search", True))
    variants = [{"k": None, "filter_clause": None, "patient_id": None},
                {"k": 5, "filter_clause": None, "patient_id": "p1"},
                {"k": 10, "filter_clause": {"term": {"doc_type": "unstructured"}}, "patient_id": None},
                {"k": 7, "filter_clause": {"term": {"patientId": "p2"}}, "patient_id": "p2"},
                {"k": 4, "filter_clause": ["Condition", "diabetes"], "patient_id": None}]       # ask() passes the NER list (2770)
    for name, takes_text in knn_methods:
        for qn in ("q_scaled", "q_tiny", "q_two_rows", "q_1024"):
            for var in (variants if qn == "q_scaled" else variants[:2]):
                client = RecordingClient(canned)
                ix = Indexer(client, "rass-idx-user1")
                kw = {key: val for key, val in var.items() if val is not None}
                args = (("what about diabetes",) if takes_text else ()) + (queries[qn].copy(),)
                try:
                    ret = getattr(ix, name)(*args, **kw)
                    raised = None
                except Exception as e:          # the reference's own failure mode is data too (quirk 3: KeyError)
                    ret, raised = None, type(e).__name__
                out["search"].append({"method": name, "query": qn, "text": "what about diabetes" if takes_text else None,
                                      "kwargs": kw, "requests": [summarise(r) for r in client.requests], "returned": ret,
                                      "raised": raised})
        # empty embedding / blank text / a failing client
        for label, text, q, client in (("empty_embedding", "x", np.array([]), RecordingClient(canned)),
                                       ("blank_text", "   ", queries["q_scaled"], RecordingClient(canned)),
                                       ("client_raises", "x", queries["q_scaled"], RecordingClient(canned, fail=True))):
            if label == "blank_text" and not takes_text:
                continue
            ix = Indexer(client, "rass-idx-user1")
            args = ((text,) if takes_text else ()) + (q,)
            try:
                ret, raised = getattr(ix, name)(*args), None
            except Exception as e:
                ret, raised = None, type(e).__name__
            out["search"].append({"method": name, "case": label, "text": text if takes_text else None,
                                  "n_requests": len(client.requests), "returned": ret, "raised": raised})
    for client, label in ((RecordingClient(canned), "two_docs"), (RecordingClient([]), "no_docs"),
                          (RecordingClient(canned, fail=True), "client_raises"), (None, "no_client")):
        out["has_any_data"].append({"case": label, "out": Indexer(client, "i").has_any_data()})

    # ---- write side
    structured = [{"doc_id": "Condition-1", "doc_type": "structured", "patientId": "p1", "conditionCodeText": "diabetes"},
                  {"doc_id": "Observation-9", "doc_type": "structured", "patientId": None, "observationValue": "7.1"}]
    unstructured = [{"doc_id": f"text-note-{i}", "doc_type": "unstructured", "patientId": f"p{i % 3}" if i % 11 else None,
                     "unstructuredText": f"chunk number {i} about topic{i % 7}"} for i in range(70)]
    unstructured[40]["doc_id"] = "text-note-5"                 # the same _id twice in one upload: the later action wins
    unstructured[69]["unstructuredText"] = "   "                # a blank chunk: the fake embedder returns a ZERO row for it
    raw = f32(rng.standard_normal((70, DIM)) * 4.0)
    raw[69] = 0.0
    ns["_fake_embeddings"] = lambda texts: raw[:len(texts)].copy()
    docs_in = copy.deepcopy(unstructured)
    asyncio.run(ns["store_fhir_docs_in_opensearch"](copy.deepcopy(structured), unstructured, RecordingClient(), "rass-idx-user1"))
    actions = [a for chunk in bulked for a in chunk]
    out["store"] = {
        "structured_docs": structured, "unstructured_docs": docs_in, "raw_embeddings": raw.tolist(),
        "embed_calls": [{"n_texts": len(t), "batch_size": b} for t, b in embed_calls],
        "bulk_sizes": [len(c) for c in bulked],
        "actions": [{"_op_type": a["_op_type"], "_index": a["_index"], "_id": a["_id"], "_routing": a["_routing"],
                     "doc_type": a["_source"].get("doc_type"),
                     "embedding": a["_source"].get("embedding")} for a in actions],
    }
    bulked.clear()
    asyncio.run(ns["store_fhir_docs_in_opensearch"](structured, unstructured, None, "rass-idx-user1"))
    out["store"]["no_client_bulks"] = len(bulked)
    with open(os.path.join(HERE, "reference_boundary.json"), "w", encoding="utf-8") as f:
        out["queries"] = {k: v.tolist() for k, v in queries.items()}
        json.dump(out, f, ensure_ascii=True, separators=(",", ":"))
    print({"search_cases": len(out["search"]), "actions": len(out["store"]["actions"]), "bulk_sizes": out["store"]["bulk_sizes"],
           "bytes": os.path.getsize(os.path.join(HERE, "reference_boundary.json"))})


if __name__ == "__main__":
    main()

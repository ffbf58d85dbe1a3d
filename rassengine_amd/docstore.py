"""Host-side metadata that OpenSearch kept next to the vectors: ``row -> doc dict``,
``doc_id -> row`` (overwrite semantics of ``_id=doc_id``, app/main.py:1260) and the
patientId dictionary behind the integer row tags (the reference's ``_routing`` /
``term: patientId`` filter, app/main.py:1263, 1549).  Pure Python; the vectors live in HBM.

One ``IndexState`` per index name; a process-global registry makes
``HipIndexer(client, index_name)`` O(1) (the reference builds it per request, 2802).
"""
from __future__ import annotations

import threading
from typing import Any, Dict, List, Optional

PATIENT_NONE = 0  # tag of rows without a patientId


class PatientDictionary:
    """Dictionary-encodes patientId strings to int32 codes >= 1 (0 = no patient)."""

    def __init__(self):
        self._code: Dict[str, int] = {}
        self._name: List[Optional[str]] = [None]

    def encode(self, patient_id: Optional[Any]) -> int:
        if patient_id is None or patient_id == "":
            return PATIENT_NONE
        key = str(patient_id)
        code = self._code.get(key)
        if code is None:
            code = len(self._name)
            self._code[key] = code
            self._name.append(key)
        return code

    def lookup(self, patient_id: Optional[Any]) -> Optional[int]:
        """Code of a known patient, None when it was never indexed (filter matches nothing)."""
        if patient_id is None or patient_id == "":
            return None
        return self._code.get(str(patient_id))

    def __len__(self) -> int:
        return len(self._name) - 1


class IndexState:
    """Everything the shim keeps per index name besides the HBM slab."""

    def __init__(self, name: str, index):
        self.name = name
        self.index = index                      # FlatIndex-like: add / delete / search / count
        self.lock = threading.RLock()
        self.row_doc: List[Optional[dict]] = []  # row id -> stored doc (None = tombstoned)
        self.doc_row: Dict[str, int] = {}        # doc_id -> live row
        self.structured: Dict[str, dict] = {}    # structured docs carry no embedding (app/main.py:1222-1240)
        self.patients = PatientDictionary()
        self.batcher = None                      # QueryBatcher, created on first async search

    def live_count(self) -> int:
        return len(self.doc_row) + len(self.structured)

    # ---- persistence: the vectors go to `<prefix>.rass` (rass_index_save), the
    # host-side metadata OpenSearch used to hold (docs, doc_id map, patient dictionary) to
    # `<prefix>.meta.json`.  Replaces OpenSearch's durability for this path.
    def save(self, prefix: str) -> None:
        import json
        with self.lock:
            self.index.save(prefix + ".rass")
            meta = {"version": 1, "name": self.name, "row_doc": self.row_doc, "structured": self.structured,
                    "patients": self.patients._name[1:]}
            with open(prefix + ".meta.json", "w", encoding="utf-8") as f:
                json.dump(meta, f)

    @classmethod
    def load(cls, name: str, prefix: str, index_loader) -> "IndexState":
        """``index_loader(name, path) -> FlatIndex-like`` (e.g. ``Engine.load_index``)."""
        import json
        with open(prefix + ".meta.json", encoding="utf-8") as f:
            meta = json.load(f)
        st = cls(name, index_loader(name, prefix + ".rass"))
        st.row_doc = meta["row_doc"]
        st.structured = meta["structured"]
        for p in meta["patients"]:
            st.patients.encode(p)
        st.doc_row = {d["doc_id"]: r for r, d in enumerate(st.row_doc) if d is not None}
        return st


class Registry:
    """index name -> IndexState; the engine behind it is created lazily on first use."""

    def __init__(self):
        self._lock = threading.Lock()
        self._states: Dict[str, IndexState] = {}
        self._factory = None

    def set_index_factory(self, factory) -> None:
        """``factory(index_name) -> FlatIndex-like``.  Default: the process-global HIP engine."""
        with self._lock:
            self._factory = factory

    def _default_factory(self, name: str):
        from . import config
        from .engine import Engine
        return Engine.get(config.RASS_DEVICE, config.EMBED_DIM).open_index(name)

    def get(self, name: str, create: bool = True) -> Optional[IndexState]:
        with self._lock:
            st = self._states.get(name)
            if st is None and create:
                factory = self._factory or self._default_factory
                st = self._states[name] = IndexState(name, factory(name))
            return st

    def put(self, st: IndexState) -> None:
        with self._lock:
            self._states[st.name] = st

    def exists(self, name: str) -> bool:
        with self._lock:
            return name in self._states

    def drop(self, name: str) -> None:
        with self._lock:
            self._states.pop(name, None)

    def clear(self) -> None:
        with self._lock:
            self._states.clear()


REGISTRY = Registry()

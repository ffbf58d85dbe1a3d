"""Host-side metadata that OpenSearch kept next to the vectors: ``row -> doc dict``,
``doc_id -> row`` (overwrite semantics of ``_id=doc_id``, app/main.py:1260), the patientId
dictionary behind the integer row tags (the reference's ``_routing`` / ``term: patientId``
filter, app/main.py:1263, 1549) and the ``doc_type`` dictionary (``term: doc_type`` of
hybrid_structured_search, app/main.py:1765).  Pure Python; the vectors live in HBM.

Row tag (int32, on the device): ``patient code | doc_type code << 24`` — bits 0..23 the patientId
code (0 = none), bits 24..30 the doc_type code (0 = none); -1 = tombstone.  A masked compare in
the scan kernel serves either filter or both (include/rass_engine.h, RASS_TAG_*).

One ``IndexState`` per index name; a process-global registry makes
``HipIndexer(client, index_name)`` O(1) (the reference builds it per request, 2802).
"""
from __future__ import annotations

import json
import os
import threading
from typing import Any, Dict, List, Optional, Tuple

PATIENT_NONE = 0  # tag of rows without a patientId
TAG_PATIENT_MASK = 0x00FFFFFF
TAG_DOCTYPE_SHIFT = 24
TAG_DOCTYPE_MASK = 0x7F000000


class PatientDictionary:
    """Dictionary-encodes strings to int32 codes >= 1 (0 = none).  Used for patientId
    (24 bits) and doc_type (7 bits)."""

    def __init__(self, max_code: int = TAG_PATIENT_MASK):
        self._code: Dict[str, int] = {}
        self._name: List[Optional[str]] = [None]
        self._max = max_code

    def encode(self, patient_id: Optional[Any]) -> int:
        if patient_id is None or patient_id == "":
            return PATIENT_NONE
        key = str(patient_id)
        code = self._code.get(key)
        if code is None:
            code = len(self._name)
            if code > self._max:
                raise OverflowError(f"dictionary full ({self._max} distinct values)")
            self._code[key] = code
            self._name.append(key)
        return code

    def lookup(self, patient_id: Optional[Any]) -> Optional[int]:
        """Code of a known value, None when it was never indexed (filter matches nothing)."""
        if patient_id is None or patient_id == "":
            return None
        return self._code.get(str(patient_id))

    def names(self) -> List[str]:
        return list(self._name[1:])

    def __len__(self) -> int:
        return len(self._name) - 1


def compose_tag(patient_code: int, doctype_code: int) -> int:
    return (int(patient_code) & TAG_PATIENT_MASK) | (int(doctype_code) << TAG_DOCTYPE_SHIFT)


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
        self.doc_types = PatientDictionary(max_code=0x7F)
        self.batcher = None                      # QueryBatcher, created on first async search
        self.generation = 0                      # bumped by every save()

    def live_count(self) -> int:
        return len(self.doc_row) + len(self.structured)

    def tag_of(self, doc: dict) -> int:
        return compose_tag(self.patients.encode(doc.get("patientId")), self.doc_types.encode(doc.get("doc_type")))

    def filter_for(self, patient_id: Optional[Any], doc_type: Optional[str]) -> Optional[Tuple[int, int]]:
        """(value, mask) of the masked tag compare for the given term filters; (-1, 0) = no filter;
        None = a filter on a value that was never indexed (matches nothing)."""
        value = mask = 0
        if patient_id not in (None, ""):
            code = self.patients.lookup(patient_id)
            if code is None:
                return None
            value |= code
            mask |= TAG_PATIENT_MASK
        if doc_type not in (None, ""):
            code = self.doc_types.lookup(doc_type)
            if code is None:
                return None
            value |= code << TAG_DOCTYPE_SHIFT
            mask |= TAG_DOCTYPE_MASK
        return (value, mask) if mask else (-1, 0)

    # ---- persistence: the vectors go to `<prefix>.g<generation>.rass` (rass_index_save),
    # the host-side metadata OpenSearch used to hold (docs, doc_id map, dictionaries) to `<prefix>.meta.json`,
    # which NAMES the vector file of its generation.  Replaces OpenSearch's durability for this path.
    #
    # Crash safety: both files are written under temporary names and fsynced; the vector file is renamed to
    # its (new, unique) generation name first, the manifest is renamed LAST (os.replace is atomic), and only
    # then is the previous generation's vector file removed.  At every instant `<prefix>.meta.json` is either
    # the old manifest (whose vector file still exists) or the new one (whose vector file is complete).
    def save(self, prefix: str) -> None:
        with self.lock:
            # never reuse a file name the live manifest references: a state built WITHOUT load() starts at generation 0
            # and would otherwise overwrite `<prefix>.g000001.rass` in place before the new manifest is committed
            gen = max(self.generation, self._manifest_generation(prefix)) + 1
            prev = self._manifest_vectors(prefix)
            vec_name = f"{os.path.basename(prefix)}.g{gen:06d}.rass"
            vec_path = os.path.join(os.path.dirname(prefix) or ".", vec_name)
            tmp_vec = vec_path + ".tmp"
            self.index.save(tmp_vec)            # rass_index_save fsyncs before closing
            os.replace(tmp_vec, vec_path)
            meta = {"version": 2, "name": self.name, "generation": gen, "vectors": vec_name,
                    "rows": int(self.index.rows), "live": int(self.index.count),
                    "row_doc": self.row_doc, "structured": self.structured,
                    "patients": self.patients.names(), "doc_types": self.doc_types.names()}
            tmp_meta = prefix + ".meta.json.tmp"
            with open(tmp_meta, "w", encoding="utf-8") as f:
                json.dump(meta, f)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp_meta, prefix + ".meta.json")
            _fsync_dir(os.path.dirname(prefix) or ".")
            self.generation = gen
            if prev and prev != vec_path and os.path.exists(prev):
                # a sharded index's vector "file" is a manifest naming one shard file per rank
                for f in (self.index.saved_files(prev) if hasattr(self.index, "saved_files") else []):
                    if os.path.exists(f):
                        os.remove(f)
                os.remove(prev)

    @staticmethod
    def _manifest_generation(prefix: str) -> int:
        """Generation of the manifest on disk (0 when there is none / it is unreadable); a vector file of a later
        generation left behind by a crash between the two renames counts too."""
        gen = 0
        try:
            with open(prefix + ".meta.json", encoding="utf-8") as f:
                gen = int(json.load(f).get("generation", 0))
        except (OSError, ValueError, TypeError):
            pass
        d, base = os.path.dirname(prefix) or ".", os.path.basename(prefix)
        try:
            for fn in os.listdir(d):
                if fn.startswith(base + ".g") and fn.endswith(".rass"):
                    digits = fn[len(base) + 2:-5]
                    if digits.isdigit():
                        gen = max(gen, int(digits))
        except OSError:
            pass
        return gen

    @staticmethod
    def _manifest_vectors(prefix: str) -> Optional[str]:
        try:
            with open(prefix + ".meta.json", encoding="utf-8") as f:
                meta = json.load(f)
            return os.path.join(os.path.dirname(prefix) or ".", meta["vectors"]) if "vectors" in meta else None
        except (OSError, ValueError, KeyError):
            return None

    @classmethod
    def load(cls, name: str, prefix: str, index_loader) -> "IndexState":
        """``index_loader(name, path) -> FlatIndex-like`` (e.g. ``Engine.load_index``).  The pair is
        rejected when the manifest and the vector file disagree (rows, tombstones)."""
        with open(prefix + ".meta.json", encoding="utf-8") as f:
            meta = json.load(f)
        vec = os.path.join(os.path.dirname(prefix) or ".", meta["vectors"]) if "vectors" in meta else prefix + ".rass"
        index = index_loader(name, vec)
        row_doc = meta["row_doc"]
        rows, live = int(index.rows), int(index.count)
        dead = sum(1 for d in row_doc if d is None)
        if len(row_doc) != rows or rows - live != dead or meta.get("rows", rows) != rows or meta.get("live", live) != live:
            raise ValueError(f"{prefix}: manifest and vector file disagree (manifest rows {len(row_doc)} / tombstones "
                             f"{dead}, vector file rows {rows} / tombstones {rows - live}): refusing to load")
        st = cls(name, index)
        st.row_doc = row_doc
        st.structured = meta["structured"]
        st.generation = int(meta.get("generation", 0))
        for p in meta["patients"]:
            st.patients.encode(p)
        for t in meta.get("doc_types", []):
            st.doc_types.encode(t)
        st.doc_row = {d["doc_id"]: r for r, d in enumerate(st.row_doc) if d is not None}
        return st


def _fsync_dir(path: str) -> None:
    try:
        fd = os.open(path, os.O_RDONLY)
        try:
            os.fsync(fd)
        finally:
            os.close(fd)
    except OSError:
        pass


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
        eng = Engine.get(config.RASS_DEVICE, config.EMBED_DIM)
        if config.RASS_IVF_NLIST > 0:       # approximate index + flat delta for later inserts (ivf.IvfBackedIndex)
            from .ivf import open_backed_index
            idx = open_backed_index(eng, name)
        else:
            idx = eng.open_index(name)
        if config.RASS_PREFILTER != "off":  # searches of k <= 16: bf16 / int8 candidate scan + exact fp32 re-rank (engine.set_prefilter)
            idx.set_prefilter(config.RASS_PREFILTER)
        return idx

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

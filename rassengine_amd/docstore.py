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

import numpy as np

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
        # incremental persistence (save_delta): what changed since the last save() / save_delta()
        self._ckpt_rows = 0                      # rows of the index that are on disk
        self._dead_since: List[int] = []         # rows < _ckpt_rows tombstoned since
        self._structured_dirty: set = set()      # structured doc_ids (re)written since
        self._delta_seq = 0                      # delta segments written on top of the current snapshot

    def live_count(self) -> int:
        return len(self.doc_row) + len(self.structured)

    # ---- the write path reports what it changes (indexer.add_documents / store_fhir_docs_in_opensearch)
    def note_deleted(self, row: int) -> None:
        if row < self._ckpt_rows:
            self._dead_since.append(int(row))

    def note_structured(self, doc_id: str) -> None:
        self._structured_dirty.add(doc_id)

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
            # the snapshot covers everything: the delta log of the previous generation goes with it
            self._ckpt_rows, self._dead_since, self._structured_dirty, self._delta_seq = int(self.index.rows), [], set(), 0
            self._remove_deltas(prefix, keep_generation=None)
            if prev and prev != vec_path and os.path.exists(prev):
                # a sharded index's vector "file" is a manifest naming one shard file per rank
                for f in (self.index.saved_files(prev) if hasattr(self.index, "saved_files") else []):
                    if os.path.exists(f):
                        os.remove(f)
                os.remove(prev)

    # ---- incremental persistence (the reference's durability is incremental: one bulk per 64 docs,
    # app/main.py:1253-1282).  save() writes a SNAPSHOT (every vector, every doc: O(corpus)); save_delta() appends ONE segment
    # holding only what changed since the last save() / save_delta(): the rows appended since (read back from the index:
    # the stored, normalised bits), their docs, the rows tombstoned since, the structured docs rewritten since — O(delta).
    # `<prefix>.deltas.json` (tiny, replaced atomically, LAST) lists the segments of the manifest's generation; load() replays
    # them in order on top of the snapshot.  A segment or list left behind by a crash before that replace is ignored.
    def save_delta(self, prefix: str) -> bool:
        """Returns False (nothing written) when there is no snapshot of THIS state under ``prefix`` to append to, or the index
        cannot hand its rows out (``get_rows``): the caller then takes a snapshot (``save``)."""
        with self.lock:
            get_rows = getattr(self.index, "get_rows", None)
            if get_rows is None or self.generation == 0 or self._manifest_generation_only(prefix) != self.generation:
                return False
            rows_now = int(self.index.rows)
            n_new = rows_now - self._ckpt_rows
            if n_new == 0 and not self._dead_since and not self._structured_dirty:
                return True
            vecs = np.ascontiguousarray(get_rows(self._ckpt_rows, n_new), dtype=np.float32) if n_new else np.zeros((0, 0), np.float32)
            head = {"version": 1, "generation": self.generation, "seq": self._delta_seq + 1, "first_row": self._ckpt_rows,
                    "n": n_new, "dim": int(vecs.shape[1]) if n_new else 0,
                    "row_doc": self.row_doc[self._ckpt_rows:rows_now] + [None] * max(0, rows_now - len(self.row_doc)),
                    "dead": sorted(set(self._dead_since)),
                    "structured": {k: self.structured[k] for k in sorted(self._structured_dirty) if k in self.structured},
                    "patients": self.patients.names(), "doc_types": self.doc_types.names(),
                    "rows_after": rows_now, "live_after": int(self.index.count)}
            d, base = os.path.dirname(prefix) or ".", os.path.basename(prefix)
            seg = f"{base}.g{self.generation:06d}.d{self._delta_seq + 1:06d}.delta"
            tmp = os.path.join(d, seg + ".tmp")
            with open(tmp, "wb") as f:
                f.write(json.dumps(head).encode("utf-8") + b"\n")
                f.write(vecs.tobytes())
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp, os.path.join(d, seg))
            listing = self._read_delta_list(prefix)
            segs = listing["segments"] if listing and listing.get("generation") == self.generation else []
            tmp_l = prefix + ".deltas.json.tmp"
            with open(tmp_l, "w", encoding="utf-8") as f:
                json.dump({"generation": self.generation, "segments": segs + [seg]}, f)
                f.flush()
                os.fsync(f.fileno())
            os.replace(tmp_l, prefix + ".deltas.json")
            _fsync_dir(d)
            self._ckpt_rows, self._dead_since, self._structured_dirty = rows_now, [], set()
            self._delta_seq += 1
            return True

    @staticmethod
    def _read_delta_list(prefix: str) -> Optional[dict]:
        try:
            with open(prefix + ".deltas.json", encoding="utf-8") as f:
                return json.load(f)
        except (OSError, ValueError):
            return None

    @classmethod
    def _remove_deltas(cls, prefix: str, keep_generation: Optional[int]) -> None:
        d, base = os.path.dirname(prefix) or ".", os.path.basename(prefix)
        try:
            for fn in os.listdir(d):
                if fn.startswith(base + ".g") and (fn.endswith(".delta") or fn.endswith(".delta.tmp")):
                    if keep_generation is None or not fn.startswith(f"{base}.g{keep_generation:06d}."):
                        os.remove(os.path.join(d, fn))
            if keep_generation is None and os.path.exists(prefix + ".deltas.json"):
                os.remove(prefix + ".deltas.json")
        except OSError:
            pass

    @staticmethod
    def _manifest_generation_only(prefix: str) -> int:
        try:
            with open(prefix + ".meta.json", encoding="utf-8") as f:
                return int(json.load(f).get("generation", 0))
        except (OSError, ValueError, TypeError):
            return 0

    def _replay_deltas(self, prefix: str) -> None:
        """load(): the segments `<prefix>.deltas.json` lists for this generation, in order, on top of the snapshot."""
        listing = self._read_delta_list(prefix)
        if not listing or int(listing.get("generation", -1)) != self.generation:
            return
        d = os.path.dirname(prefix) or "."
        for k, seg in enumerate(listing.get("segments", [])):
            with open(os.path.join(d, seg), "rb") as f:
                head = json.loads(f.readline().decode("utf-8"))
                n, dim = int(head["n"]), int(head["dim"])
                raw = f.read(n * dim * 4)
            if int(head.get("generation", -1)) != self.generation or int(head.get("seq", -1)) != k + 1 or \
                    int(head["first_row"]) != int(self.index.rows) or len(raw) != n * dim * 4 or len(head["row_doc"]) != n:
                raise ValueError(f"{seg}: delta segment does not continue the index (generation / order / rows / length)")
            for pname in head["patients"]:
                self.patients.encode(pname)
            for t in head.get("doc_types", []):
                self.doc_types.encode(t)
            if n:
                vecs = np.frombuffer(raw, dtype=np.float32).reshape(n, dim).copy()
                tags = np.array([self.tag_of(doc) if doc is not None else 0 for doc in head["row_doc"]], dtype=np.int32)
                first = self.index.add(vecs, tags=tags, normalize=False)     # the stored bits, not re-normalised
                if int(first) != int(head["first_row"]):
                    raise ValueError(f"{seg}: rows landed at {first}, expected {head['first_row']}")
            for r in head["dead"]:
                self.index.delete(int(r))
                doc = self.row_doc[int(r)]
                if doc is not None and self.doc_row.get(doc.get("doc_id")) == int(r):
                    del self.doc_row[doc.get("doc_id")]
                self.row_doc[int(r)] = None
            for i, doc in enumerate(head["row_doc"]):
                r = int(head["first_row"]) + i
                self.row_doc.append(doc)
                if doc is None:
                    self.index.delete(r)
                else:
                    self.doc_row[doc["doc_id"]] = r
            self.structured.update(head["structured"])
            if int(self.index.rows) != int(head["rows_after"]) or int(self.index.count) != int(head["live_after"]):
                raise ValueError(f"{seg}: after the replay the index holds {self.index.rows} rows / {self.index.count} live, "
                                 f"the segment recorded {head['rows_after']} / {head['live_after']}")
            self._delta_seq = k + 1
        self._ckpt_rows = int(self.index.rows)

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
        try:    # a restored index takes the configured candidate mode like a new one (the default factory below)
            from . import config
            if config.RASS_PREFILTER != "off" and hasattr(index, "set_prefilter") and not getattr(index, "prefilter", False):
                index.set_prefilter(config.RASS_PREFILTER)
        except Exception:   # e.g. a bf16 corpus: no prefilter mode — the exact scan serves
            pass
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
        st._ckpt_rows = rows
        st._replay_deltas(prefix)
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

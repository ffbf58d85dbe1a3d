"""Drop-in replacements for the reference's Ollama embedding client
(app/main.py:225-274; variant in app/embedding_gen.py:152-192): same names, signatures,
dtypes, shapes and blank-text behaviour, with the per-text HTTP POST replaced by batched
calls into the in-process encoder.

Two flavours, because the reference's two modules differ and ``install()`` binds the
one that belongs to the module it patches:

* ``MAIN_FLAVOUR`` (app/main.py): ``embed_texts_in_batches(texts, batch_size=BATCH_SIZE)``; empty input ->
  ``np.array([])`` (shape (0,), 246-247); embed errors RAISE (``raise_for_status``, 235);
* ``GEN_FLAVOUR`` (app/embedding_gen.py): ``embed_texts_in_batches(texts)`` — no ``batch_size`` (173); empty
  input -> ``np.zeros((0, EMBED_DIM), float32)`` (174-175); an embed error is printed and that text gets a ZERO
  vector (168-170); no ``embed_query`` in that module.

Contract reproduced:

* ``ollama_embed_text(text)``: blank text -> ``[0.0] * EMBED_DIM`` (227-228), else a list of
  EMBED_DIM floats; errors raise (``raise_for_status``, 235);
* ``embed_texts_in_batches(texts, batch_size)``: order-preserving, C-contiguous fp32
  ``[n, EMBED_DIM]``; empty input -> ``np.array([])`` (shape (0,), 246-247);
* ``embed_query(query)``: blank -> ``np.array([])`` whose ``.size == 0`` is what
  ``semantic_search`` tests (1534); else ``[1, EMBED_DIM]`` fp32; no query prompt prefix.

The reference bounds concurrency with ``Semaphore(MAX_EMBED_CONCURRENCY)`` around one
request per text (app/main.py:250-260); here every embed request of the process — the query of each
``/ask`` (app/main.py:2800), single texts, upload slices — goes through ONE ``batcher.EmbedBatcher``:
requests that are in flight together are answered by one varlen encoder forward on the batcher's
worker thread (the event loop is never blocked), a lone request by a forward of its own.
``RASS_EMBED_BATCH_MAX=0`` turns the coalescing off (one encoder call per request, on a worker thread).
"""
from __future__ import annotations

import asyncio
import threading
from typing import List, Optional, Protocol

import numpy as np

from . import config, prefetch

BATCH_SIZE = config.BATCH_SIZE
EMBED_DIM = config.EMBED_DIM


class Embedder(Protocol):
    dim: int

    def encode(self, texts: List[str]) -> np.ndarray:
        """Pooled sentence embeddings, fp32 [len(texts), dim] (not necessarily normalised)."""


_embedder: Optional[Embedder] = None
_lock = threading.Lock()
_batcher = None          # batcher.EmbedBatcher, made on first use
UPLOAD_SLICE = 1024      # texts per upload entry: the encoder pipelines tokeniser and GPU inside one call, and a
                         # query arriving during an upload waits for at most one slice (small entries go first)


def set_embedder(embedder: Optional[Embedder]) -> None:
    global _embedder
    with _lock:
        _embedder = embedder


def get_batcher():
    """The process-wide embed micro-batcher (None when ``RASS_EMBED_BATCH_MAX`` is 0)."""
    global _batcher
    if config.RASS_EMBED_BATCH_MAX <= 0:
        return None
    with _lock:
        if _batcher is None:
            from .batcher import EmbedBatcher
            # the embedder is looked up per call, so set_embedder() takes effect without a new batcher
            _batcher = EmbedBatcher(lambda texts: get_embedder().encode(texts), max_seqs=config.RASS_EMBED_BATCH_MAX,
                                    max_delay_ms=config.RASS_EMBED_BATCH_DELAY_MS,
                                    quiet_us=config.RASS_EMBED_BATCH_QUIET_US)
        return _batcher


def reset_batcher() -> None:
    """Stops the worker thread (tests; a new batcher is made on the next request)."""
    global _batcher
    with _lock:
        b, _batcher = _batcher, None
    if b is not None:
        b.close()


def get_embedder() -> Embedder:
    """The process-global encoder; loaded from ``RASS_MODEL_DIR`` on first use."""
    global _embedder
    with _lock:
        if _embedder is None:
            if not config.RASS_MODEL_DIR:
                raise RuntimeError(
                    "no encoder loaded: set RASS_MODEL_DIR to a local directory with the "
                    f"{config.EMBED_MODEL_NAME} weights and vocab.txt, or call set_embedder()")
            from .encoder import HipSentenceEncoder
            _embedder = HipSentenceEncoder.from_dir(config.RASS_MODEL_DIR, device=config.RASS_DEVICE)
        return _embedder


async def _encode_nonblank(texts: List[str]) -> np.ndarray:
    """Blank texts -> zero rows (app/main.py:227-228); the rest go through the encoder — coalesced with whatever
    other requests are in flight (``EmbedBatcher``)."""
    keep = [i for i, t in enumerate(texts) if t.strip()]
    if not keep:
        return np.zeros((len(texts), get_embedder().dim), dtype=np.float32)
    sub = [texts[i] for i in keep]
    b = get_batcher()
    if b is not None:
        vecs = await b.embed(sub)
    else:
        vecs = np.asarray(await asyncio.to_thread(lambda: get_embedder().encode(sub)), dtype=np.float32)
    if len(keep) == len(texts):
        return vecs
    out = np.zeros((len(texts), vecs.shape[1]), dtype=np.float32)
    out[keep] = vecs
    return out


async def ollama_embed_text(text: str) -> List[float]:
    """Get an embedding for a single text (app/main.py:225-237)."""
    if not text.strip():
        return [0.0] * EMBED_DIM
    vec = await _encode_nonblank([text])
    return vec[0].tolist()


async def embed_texts_in_batches(texts: List[str], batch_size: int = BATCH_SIZE) -> np.ndarray:
    """Embeds a list of texts in batches (app/main.py:240-263)."""
    if not texts:
        return np.array([])
    # The reference's batch_size bounds how many HTTP requests are gathered at once (one per text).
    # Here a slice is ONE encoder call that batches on its own (<= 256 sequences / 131 072 tokens per
    # forward, tokenisation overlapped), so slices are made large enough to keep the GPU busy; the
    # result does not depend on the slicing (order-preserving, each text embedded independently).
    # A short list (<= RASS_EMBED_BATCH_MAX texts) shares its forward with concurrent requests.
    step = max(int(batch_size), UPLOAD_SLICE)
    all_embeddings = []
    for i in range(0, len(texts), step):
        all_embeddings.append(await _encode_nonblank(texts[i:i + step]))
    return np.ascontiguousarray(np.concatenate(all_embeddings, axis=0), dtype=np.float32)


async def embed_query(query: str) -> np.ndarray:
    """Embedding of a single query, shape [1, EMBED_DIM] (app/main.py:266-274)."""
    if not query.strip():
        return np.array([])
    # the reference builds np.array([emb_list], dtype=np.float32) from ollama_embed_text's list of Python floats; the
    # encoder's fp32 row IS that array (float32 -> float -> float32 is exact), without 1 024 boxed floats per query
    prefetch.embed_enter()
    try:
        vec = await _encode_nonblank([query])
    finally:
        prefetch.embed_exit()
    out = np.ascontiguousarray(vec[:1], dtype=np.float32)
    # ask() awaits ensure_index_exists next (app/main.py:2801): the k-NN scans of concurrent requests are shared there
    prefetch.remember(out)
    return out


# ------------------------------------------------------------------ app/embedding_gen.py:152-192
async def gen_ollama_embed_text(text: str) -> List[float]:
    """embedding_gen.py:152-170: as above, but an error is printed and a zero vector returned."""
    if not text.strip():
        return [0.0] * EMBED_DIM
    try:
        vec = await _encode_nonblank([text])
        return vec[0].tolist()
    except Exception as ex:
        print("[ERROR] Ollama embed request:", ex)
        return [0.0] * EMBED_DIM


async def _encode_or_zero(texts: List[str]) -> np.ndarray:
    """One encoder request for the slice; if it fails, every text is retried on its own so that only the
    texts that really fail become zero rows (the reference's errors are per text, 168-170)."""
    try:
        return await _encode_nonblank(texts)
    except Exception as ex:
        if len(texts) == 1:
            print("[ERROR] Ollama embed request:", ex)
            return np.zeros((1, EMBED_DIM), dtype=np.float32)
    rows = []
    for t in texts:
        try:
            rows.append(await _encode_nonblank([t]))
        except Exception as ex:
            print("[ERROR] Ollama embed request:", ex)
            rows.append(np.zeros((1, EMBED_DIM), dtype=np.float32))
    return np.concatenate(rows, axis=0)


async def gen_embed_texts_in_batches(texts: List[str]) -> np.ndarray:
    """embedding_gen.py:173-192."""
    if not texts:
        return np.zeros((0, EMBED_DIM), dtype=np.float32)
    step = max(int(BATCH_SIZE), UPLOAD_SLICE)   # see embed_texts_in_batches: a slice is one self-batching encoder call
    out = []
    for i in range(0, len(texts), step):
        out.append(await _encode_or_zero(texts[i:i + step]))
    return np.ascontiguousarray(np.concatenate(out, axis=0), dtype=np.float32)


MAIN_FLAVOUR = {"ollama_embed_text": ollama_embed_text, "embed_texts_in_batches": embed_texts_in_batches,
                "embed_query": embed_query}
GEN_FLAVOUR = {"ollama_embed_text": gen_ollama_embed_text, "embed_texts_in_batches": gen_embed_texts_in_batches,
               "embed_query": embed_query}

"""Configuration: the same environment names the reference reads at import
(app/main.py:60-108, app/embedding_gen.py:39-70), plus ``RASS_*`` for the engine."""
import os


def _int(name: str, default: int) -> int:
    try:
        return int(os.getenv(name, default))
    except (TypeError, ValueError):
        return default


EMBED_MODEL_NAME = os.getenv("OLLAMA_EMBED_MODEL", "mxbai-embed-large:latest")  # app/main.py:67
BATCH_SIZE = _int("BATCH_SIZE", 64)            # app/main.py:78
CHUNK_SIZE = _int("CHUNK_SIZE", 512)           # app/main.py:79
EMBED_DIM = _int("EMBED_DIM", 1024)            # app/main.py:80
OPENSEARCH_INDEX_NAME = os.getenv("OPENSEARCH_INDEX_NAME", "")  # app/main.py:87
TOP_K = _int("TOP_K", 3)                       # app/main.py:88
SHARD_COUNT = _int("SHARD_COUNT", 1)           # app/main.py:89 (here: informational; shards = GPUs)

RASS_DEVICE = _int("RASS_DEVICE", _int("LOCAL_RANK", 0))   # the GPU this process drives
RASS_MODEL_DIR = os.getenv("RASS_MODEL_DIR", "")           # local dir with encoder weights + vocab.txt
RASS_SCORE_MODE = os.getenv("RASS_SCORE_MODE", "opensearch")  # "opensearch": 1/(2-cos); "cosine": raw
RASS_RETURN_EMBEDDING = os.getenv("RASS_RETURN_EMBEDDING", "0") == "1"
RASS_POOLING = os.getenv("RASS_POOLING", "")               # "cls" | "mean" | "" (from the model dir)
# embed micro-batcher (batcher.EmbedBatcher): sequences per coalesced forward (0 = off: one encoder call per request),
# the longest a request waits for company from idle, and the quiet gap that ends the wait early
RASS_EMBED_BATCH_MAX = _int("RASS_EMBED_BATCH_MAX", 64)
try:
    RASS_EMBED_BATCH_DELAY_MS = float(os.getenv("RASS_EMBED_BATCH_DELAY_MS", "0.2"))
    RASS_EMBED_BATCH_QUIET_US = float(os.getenv("RASS_EMBED_BATCH_QUIET_US", "50"))
except ValueError:
    RASS_EMBED_BATCH_DELAY_MS, RASS_EMBED_BATCH_QUIET_US = 0.2, 50.0
# k-NN prefetch at ask()'s `await ensure_index_exists` (prefetch.py): 0 = off, 1 = when other requests are in flight, 2 = always
RASS_KNN_PREFETCH = _int("RASS_KNN_PREFETCH", 1)
# IVF behind the boundary (ivf.IvfPolicy): 0 = every index stays flat (exact; the default).  > 0: an index of at least
# RASS_IVF_MIN_ROWS rows gets an IVF-<nlist> (probed with RASS_IVF_NPROBE lists per query) + a flat delta for the rows
# appended since; the IVF is rebuilt once the delta exceeds RASS_IVF_REBUILD_FRACTION of the rows it covers
RASS_IVF_NLIST = _int("RASS_IVF_NLIST", 0)
RASS_IVF_NPROBE = _int("RASS_IVF_NPROBE", 8)
RASS_IVF_MIN_ROWS = _int("RASS_IVF_MIN_ROWS", 262144)
RASS_IVF_DTYPE = os.getenv("RASS_IVF_DTYPE", "f32")            # the IVF's own copy of the rows: "f32" | "bf16" | "int8" (+ exact re-rank, k <= 16)
# Candidate scan of a flat index's searches with k <= 16: "off" (exact fp32 scan, the default), "bf16" (half the bytes per pass)
# or "int8" (a quarter: per-row-scaled int8 copy); the 32 candidates per query are re-scored exactly in fp32 either way
# (include/rass_engine.h: rass_index_set_prefilter).  The reference's own index is approximate (HNSW, app/main.py:563-572).
RASS_PREFILTER = os.getenv("RASS_PREFILTER", "off").strip().lower() or "off"
if RASS_PREFILTER not in ("off", "bf16", "int8"):
    raise ValueError(f"RASS_PREFILTER must be off, bf16 or int8, not {RASS_PREFILTER!r}")
try:
    RASS_IVF_REBUILD_FRACTION = float(os.getenv("RASS_IVF_REBUILD_FRACTION", "0.25"))
except ValueError:
    RASS_IVF_REBUILD_FRACTION = 0.25


def get_index_name(user_id: str) -> str:
    """app/main.py:346-347."""
    return f"{OPENSEARCH_INDEX_NAME}-{user_id}"

"""The in-process sentence encoder that replaces the Ollama ``/embeddings`` hop
(reference app/main.py:225-237, model ``mxbai-embed-large``, app/main.py:67).

* ``HipSentenceEncoder`` — loads a LOCAL model directory (``config.json`` +
  ``model.safetensors`` in Hugging Face BERT naming + ``vocab.txt``; nothing is fetched) into
  the HIP encoder behind the C ABI (``rass_encoder_*``) and turns texts into pooled fp32
  vectors, batched and varlen-packed (one GPU forward per batch instead of one HTTP request
  per text).
* ``WordPieceTokenizer`` — BERT basic + WordPiece tokenisation (lower-casing, accent
  stripping, punctuation split, greedy longest-match-first), ``[CLS] ... [SEP]``, truncated to
  the model's window: what the model server does to the reference's ``chunk_text`` chunks
  (<= CHUNK_SIZE words, app/main.py:2160-2170, which overflow 512 tokens and get truncated).
* ``write_random_model_dir`` — seeded random weights of a given architecture for parity tests
  and the benchmark (no weights or vocab exist offline).

Pooling: ``cls`` or ``mean``; taken from ``RASS_POOLING``, else from the
sentence-transformers ``1_Pooling/config.json`` of the model dir, else ``cls``.
"""
from __future__ import annotations

import ctypes
import json
import os
import unicodedata
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

from . import _native as N
from . import config as rcfg


# ----------------------------------------------------------------------------------- config
@dataclass
class EncoderConfig:
    vocab_size: int = 30522
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    intermediate: int = 4096
    max_positions: int = 512
    layer_norm_eps: float = 1e-12
    pooling: str = "cls"       # "cls" | "mean"
    normalize: bool = False    # the reference normalises downstream (app/main.py:1249-1251)

    @classmethod
    def from_dir(cls, path: str) -> "EncoderConfig":
        with open(os.path.join(path, "config.json"), encoding="utf-8") as f:
            c = json.load(f)
        if c.get("hidden_act", "gelu") not in ("gelu",):
            raise ValueError(f"unsupported hidden_act {c.get('hidden_act')!r} (erf GELU only)")
        if c.get("position_embedding_type", "absolute") != "absolute":
            raise ValueError("only absolute position embeddings are supported")
        pooling = rcfg.RASS_POOLING or "cls"
        pool_cfg = os.path.join(path, "1_Pooling", "config.json")
        if not rcfg.RASS_POOLING and os.path.exists(pool_cfg):
            with open(pool_cfg, encoding="utf-8") as f:
                p = json.load(f)
            if p.get("pooling_mode_mean_tokens"):
                pooling = "mean"
            elif p.get("pooling_mode_cls_token"):
                pooling = "cls"
        return cls(vocab_size=int(c["vocab_size"]), hidden=int(c["hidden_size"]), layers=int(c["num_hidden_layers"]),
                   heads=int(c["num_attention_heads"]), intermediate=int(c["intermediate_size"]),
                   max_positions=min(int(c.get("max_position_embeddings", 512)), 512),
                   layer_norm_eps=float(c.get("layer_norm_eps", 1e-12)), pooling=pooling)

    def to_hf_dict(self) -> Dict:
        return {"architectures": ["BertModel"], "model_type": "bert", "vocab_size": self.vocab_size,
                "hidden_size": self.hidden, "num_hidden_layers": self.layers, "num_attention_heads": self.heads,
                "intermediate_size": self.intermediate, "max_position_embeddings": self.max_positions,
                "layer_norm_eps": self.layer_norm_eps, "hidden_act": "gelu", "type_vocab_size": 2,
                "position_embedding_type": "absolute", "hidden_dropout_prob": 0.0,
                "attention_probs_dropout_prob": 0.0}


class _CConfig(ctypes.Structure):
    _fields_ = [("vocab_size", ctypes.c_int32), ("hidden", ctypes.c_int32), ("layers", ctypes.c_int32),
                ("heads", ctypes.c_int32), ("intermediate", ctypes.c_int32), ("max_positions", ctypes.c_int32),
                ("pooling", ctypes.c_int32), ("normalize", ctypes.c_int32), ("layer_norm_eps", ctypes.c_float)]


def weight_names(layers: int) -> List[str]:
    names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
             "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
    per = ["attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
           "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"]
    for l in range(layers):
        for p in per:
            names += [f"encoder.layer.{l}.{p}.weight", f"encoder.layer.{l}.{p}.bias"]
    return names


def load_weights(path: str) -> Dict[str, np.ndarray]:
    """fp32 arrays by bare BERT name (a leading ``bert.`` / ``model.`` prefix is stripped)."""
    from safetensors import safe_open
    out: Dict[str, np.ndarray] = {}
    with safe_open(os.path.join(path, "model.safetensors"), framework="pt") as f:
        for k in f.keys():
            name = k
            for prefix in ("bert.", "model."):
                if name.startswith(prefix):
                    name = name[len(prefix):]
            out[name] = f.get_tensor(k).float().contiguous().numpy()
    return out


def random_weights(cfg: EncoderConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded random weights by bare BERT name (bit-identical on every host: numpy PCG64), scaled so activations
    stay O(1) through ``cfg.layers`` post-LN layers.  No weights exist offline: parity tests and the
    benchmark run on these."""
    rng = np.random.default_rng(seed)
    H, I = cfg.hidden, cfg.intermediate
    shapes = {"embeddings.word_embeddings.weight": (cfg.vocab_size, H),
              "embeddings.position_embeddings.weight": (cfg.max_positions, H),
              "embeddings.token_type_embeddings.weight": (2, H)}
    tensors: Dict[str, np.ndarray] = {}
    for name in weight_names(cfg.layers):
        if name in shapes:
            t = rng.standard_normal(shapes[name], dtype=np.float32) * 0.05
        elif "LayerNorm.weight" in name:
            t = 1.0 + 0.1 * rng.standard_normal(H, dtype=np.float32)
        elif "LayerNorm.bias" in name:
            t = 0.05 * rng.standard_normal(H, dtype=np.float32)
        elif name.endswith(".bias"):
            n = I if "intermediate.dense" in name else H
            t = 0.02 * rng.standard_normal(n, dtype=np.float32)
        else:
            shape = (I, H) if "intermediate.dense" in name else (H, I) if ".output.dense" in name and \
                "attention" not in name else (H, H)
            t = rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(shape[1]))
        tensors[name] = np.ascontiguousarray(t, dtype=np.float32)
    return tensors


def write_random_model_dir(path: str, cfg: EncoderConfig, seed: int = 0, vocab: Optional[Sequence[str]] = None) -> None:
    """``random_weights`` in the Hugging Face BERT layout, plus config.json / 1_Pooling / vocab.txt."""
    from safetensors.numpy import save_file
    os.makedirs(path, exist_ok=True)
    H = cfg.hidden
    tensors = random_weights(cfg, seed)
    save_file(tensors, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w", encoding="utf-8") as f:
        json.dump(cfg.to_hf_dict(), f)
    os.makedirs(os.path.join(path, "1_Pooling"), exist_ok=True)
    with open(os.path.join(path, "1_Pooling", "config.json"), "w", encoding="utf-8") as f:
        json.dump({"word_embedding_dimension": H, "pooling_mode_cls_token": cfg.pooling == "cls",
                   "pooling_mode_mean_tokens": cfg.pooling == "mean"}, f)
    if vocab is None:
        vocab = synthetic_vocab(cfg.vocab_size)
    with open(os.path.join(path, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")


def synthetic_vocab(size: int) -> List[str]:
    """A deterministic stand-in vocab.txt: specials, single characters (so nothing is [UNK]),
    common suffix pieces, then filler."""
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789.,;:!?()[]{}-_'\"/\\%&+*=<>@#$")
    toks += chars + ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"]
    words = ["the", "of", "and", "patient", "history", "diabetes", "blood", "pressure", "note", "chunk", "number",
             "about", "topic", "condition", "drug", "pain", "heart", "type", "is", "what", "with", "for", "in", "on"]
    toks += words + ["##s", "##ing", "##ed", "##tion", "##es", "##ly"]
    seen = set()
    out = []
    for t in toks:
        if t not in seen:
            seen.add(t)
            out.append(t)
    i = 0
    while len(out) < size:
        t = f"tok{i}"
        i += 1
        if t not in seen:
            seen.add(t)
            out.append(t)
    return out[:size]


# -------------------------------------------------------------------------------- tokenizer
def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF) or
            (0x2A700 <= cp <= 0x2B73F) or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF) or
            (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


class WordPieceTokenizer:
    """BERT (uncased) BasicTokenizer + WordpieceTokenizer."""

    def __init__(self, vocab: Iterable[str], lower_case: bool = True, max_chars_per_word: int = 100):
        self.vocab = {t: i for i, t in enumerate(vocab)}
        self.lower = lower_case
        self.max_chars = max_chars_per_word
        for special in ("[UNK]", "[CLS]", "[SEP]"):
            if special not in self.vocab:
                raise ValueError(f"vocab lacks {special}")
        self.unk, self.cls, self.sep = self.vocab["[UNK]"], self.vocab["[CLS]"], self.vocab["[SEP]"]

    @classmethod
    def from_file(cls, path: str, lower_case: bool = True) -> "WordPieceTokenizer":
        with open(path, encoding="utf-8") as f:
            return cls([line.rstrip("\n") for line in f if line.rstrip("\n") != ""], lower_case)

    def _basic(self, text: str) -> List[str]:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or (unicodedata.category(ch) in ("Cc", "Cf") and ch not in "\t\n\r"):
                continue
            if _is_cjk(cp):
                out.append(f" {ch} ")
            elif ch in "\t\n\r" or unicodedata.category(ch) == "Zs":
                out.append(" ")
            else:
                out.append(ch)
        words = []
        for tok in "".join(out).split():
            if self.lower:
                tok = tok.lower()
                tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
            cur = ""
            for ch in tok:
                if _is_punct(ch):
                    if cur:
                        words.append(cur)
                        cur = ""
                    words.append(ch)
                else:
                    cur += ch
            if cur:
                words.append(cur)
        return words

    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk]
        ids, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                piece = word[start:end]
                if start > 0:
                    piece = "##" + piece
                if piece in self.vocab:
                    cur = self.vocab[piece]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            ids.append(cur)
            start = end
        return ids

    def encode(self, text: str, max_len: int = 512) -> List[int]:
        """``[CLS] pieces... [SEP]``, truncated to ``max_len`` tokens in total."""
        ids: List[int] = []
        for w in self._basic(text):
            ids.extend(self._wordpiece(w))
            if len(ids) >= max_len - 2:
                break
        return [self.cls] + ids[: max_len - 2] + [self.sep]


class CppWordPieceTokenizer:
    """The same tokenizer in host C++ behind the C ABI (``rass_tokenizer_*``), multi-threaded
    over a batch; used by ``HipSentenceEncoder`` so ingest is not bottlenecked on Python."""

    def __init__(self, vocab_path: str, lower_case: bool = True):
        self._L = N.lib()
        h = ctypes.c_void_p()
        N.check("rass_tokenizer_create",
                self._L.rass_tokenizer_create(vocab_path.encode(), 1 if lower_case else 0, ctypes.byref(h)))
        self._h = h

    @property
    def vocab_size(self) -> int:
        return int(self._L.rass_tokenizer_vocab_size(self._h))

    def encode(self, text: str, max_len: int = 512) -> List[int]:
        raw = text.encode("utf-8", "replace")
        out = np.empty(max_len, dtype=np.int32)
        n = N.check("rass_tokenizer_encode",
                    self._L.rass_tokenizer_encode(self._h, raw, len(raw), int(max_len),
                                                  out.ctypes.data_as(ctypes.c_void_p)))
        return out[:n].tolist()

    def encode_batch(self, texts: Sequence[str], max_len: int = 512, n_threads: int = 0):
        """-> (ids int32 [total], cu_seqlens int32 [n+1]) ready for ``rass_encode``."""
        n = len(texts)
        raws = [t.encode("utf-8", "replace") for t in texts]
        arr = (ctypes.c_char_p * n)(*raws)
        lens = np.array([len(r) for r in raws], dtype=np.int64)
        ids = np.empty(max(n, 1) * max_len, dtype=np.int32)
        cu = np.zeros(n + 1, dtype=np.int32)
        total = self._L.rass_tokenizer_encode_batch(self._h, arr, lens.ctypes.data_as(ctypes.c_void_p), n,
                                                    int(max_len), ids.ctypes.data_as(ctypes.c_void_p),
                                                    cu.ctypes.data_as(ctypes.c_void_p), int(n_threads))
        N.check("rass_tokenizer_encode_batch", int(total) if total < 0 else 0)
        return ids[:int(total)], cu

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.rass_tokenizer_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------- encoder
class HipSentenceEncoder:
    """texts -> pooled fp32 vectors [n, hidden] on one GPU (an ``Embedder`` for embedding.py)."""

    def __init__(self, cfg: EncoderConfig, weights: Dict[str, np.ndarray], tokenizer: Optional[WordPieceTokenizer],
                 device: int = 0, max_batch_tokens: int = 131072, max_batch_seqs: int = 256):
        self.cfg = cfg
        self.dim = cfg.hidden
        self.tokenizer = tokenizer
        self.max_batch_tokens = max_batch_tokens
        self.max_batch_seqs = max_batch_seqs
        self._L = N.lib()
        c = _CConfig(cfg.vocab_size, cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate, cfg.max_positions,
                     1 if cfg.pooling == "mean" else 0, 1 if cfg.normalize else 0, cfg.layer_norm_eps)
        h = ctypes.c_void_p()
        N.check("rass_encoder_create", self._L.rass_encoder_create(int(device), ctypes.byref(c), ctypes.byref(h)))
        self._h = h
        for name in weight_names(cfg.layers):
            if name not in weights:
                raise KeyError(f"model is missing tensor {name}")
            w = np.ascontiguousarray(weights[name], dtype=np.float32)
            N.check("rass_encoder_set_weight",
                    self._L.rass_encoder_set_weight(self._h, name.encode(), w.ctypes.data_as(ctypes.c_void_p), w.size))
        N.check("rass_encoder_finalize", self._L.rass_encoder_finalize(self._h))

    @classmethod
    def from_dir(cls, path: str, device: int = 0, **kw) -> "HipSentenceEncoder":
        cfg = EncoderConfig.from_dir(path)
        vocab_path = os.path.join(path, "vocab.txt")
        tok = CppWordPieceTokenizer(vocab_path) if os.path.exists(vocab_path) else None
        return cls(cfg, load_weights(path), tok, device=device, **kw)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.rass_encoder_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        """The encoder's own hipStream_t (as int): what ``rass_encode`` and ``rass_encode_device(stream=NULL)``
        run on.  Hand it to ``Engine.set_stream`` for the device-resident ingest hand-off."""
        return int(self._L.rass_encoder_get_stream(self._h) or 0)

    @property
    def handle(self) -> int:
        """The ``rass_encoder_t*`` as an int (what ``torch.ops.rass.encode`` takes)."""
        return int(self._h.value or 0)

    def stats(self) -> Dict[str, int]:
        """Forwards / sequences / tokens since the encoder was created (``rass_encoder_stats``)."""
        out = (ctypes.c_int64 * 3)()
        N.check("rass_encoder_stats", self._L.rass_encoder_stats(self._h, out))
        return {"forwards": int(out[0]), "sequences": int(out[1]), "tokens": int(out[2])}

    def encode_device(self, d_token_ids_ptr: int, d_cu_seqlens_ptr: int, nseq: int, total_tokens: int,
                      max_seqlen: int, d_out_ptr: int, stream_ptr: int = 0) -> None:
        """Device-resident, asynchronous forward (``rass_encode_device``); ``stream_ptr`` 0 = ``self.stream``."""
        N.check("rass_encode_device",
                self._L.rass_encode_device(self._h, ctypes.c_void_p(d_token_ids_ptr), ctypes.c_void_p(d_cu_seqlens_ptr),
                                           int(nseq), int(total_tokens), int(max_seqlen), ctypes.c_void_p(d_out_ptr),
                                           ctypes.c_void_p(stream_ptr or 0)))

    def encode_ids(self, seqs: Sequence[Sequence[int]]) -> np.ndarray:
        """Pooled embeddings of already-tokenised sequences (each ``[CLS] ... [SEP]``)."""
        n = len(seqs)
        if n == 0:
            return np.empty((0, self.dim), dtype=np.float32)
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
        cu = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=cu[1:])
        ids = np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs]) if cu[-1] else np.empty(0, np.int32)
        return self._encode_flat(ids, cu)

    def _encode_flat(self, ids: np.ndarray, cu: np.ndarray) -> np.ndarray:
        """Packed token ids + cumulative sequence offsets -> pooled vectors; greedy batches of at most
        ``max_batch_seqs`` sequences / ``max_batch_tokens`` tokens, no per-token Python work."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        cu = np.asarray(cu, dtype=np.int64)
        n = cu.shape[0] - 1
        out = np.empty((n, self.dim), dtype=np.float32)
        lens = np.diff(cu)
        bad = np.nonzero((lens < 1) | (lens > self.cfg.max_positions))[0]
        if bad.size:
            j = int(bad[0])
            raise ValueError(f"sequence {j} has {int(lens[j])} tokens (1..{self.cfg.max_positions} allowed)")
        i = 0
        while i < n:
            j_tok = int(np.searchsorted(cu, cu[i] + self.max_batch_tokens, side="right")) - 1
            j = min(n, i + self.max_batch_seqs, j_tok)
            if j <= i:
                raise ValueError("a single sequence exceeds max_batch_tokens")
            sub_ids = ids[cu[i]:cu[j]]
            sub_cu = np.ascontiguousarray(cu[i:j + 1] - cu[i], dtype=np.int32)
            N.check("rass_encode",
                    self._L.rass_encode(self._h, sub_ids.ctypes.data_as(ctypes.c_void_p),
                                        sub_cu.ctypes.data_as(ctypes.c_void_p), j - i,
                                        out[i:j].ctypes.data_as(ctypes.c_void_p)))
            i = j
        return out

    def tokenize(self, texts: Sequence[str]):
        """(packed token ids int32, cumulative offsets int64) of ``texts``, each ``[CLS] ... [SEP]`` (serving.py's
        data-parallel ingest tokenises on rank 0 and encodes on every rank)."""
        if self.tokenizer is None:
            raise RuntimeError("this model directory has no vocab.txt")
        if hasattr(self.tokenizer, "encode_batch"):
            ids, cu = self.tokenizer.encode_batch(list(texts), self.cfg.max_positions)
            return np.asarray(ids, dtype=np.int32), np.asarray(cu, dtype=np.int64)
        seqs = [self.tokenizer.encode(t, self.cfg.max_positions) for t in texts]
        cu = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum([len(q) for q in seqs], out=cu[1:])
        ids = np.concatenate([np.asarray(q, dtype=np.int32) for q in seqs]) if cu[-1] else np.empty(0, np.int32)
        return ids, cu

    def encode_flat(self, ids: np.ndarray, cu: np.ndarray) -> np.ndarray:
        """Pooled vectors of packed, already-tokenised sequences (``tokenize``'s output, or a slice of it)."""
        return self._encode_flat(ids, cu)

    def encode(self, texts: List[str]) -> np.ndarray:
        if self.tokenizer is None:
            raise RuntimeError("this model directory has no vocab.txt: use encode_ids()")
        if not hasattr(self.tokenizer, "encode_batch"):
            return self.encode_ids([self.tokenizer.encode(t, self.cfg.max_positions) for t in texts])
        # C++ tokenizer: one multi-threaded call per slice of texts; the next slice is tokenised on a
        # worker thread while the GPU encodes the current one (both calls release the GIL).
        step = self.max_batch_seqs
        slices = [texts[a:a + step] for a in range(0, len(texts), step)]
        if not slices:
            return np.empty((0, self.dim), dtype=np.float32)
        if len(slices) == 1:
            ids, cu = self.tokenizer.encode_batch(slices[0], self.cfg.max_positions)
            return self._encode_flat(ids, cu)
        from concurrent.futures import ThreadPoolExecutor
        outs = []
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(self.tokenizer.encode_batch, slices[0], self.cfg.max_positions)
            for nxt in slices[1:] + [None]:
                ids, cu = fut.result()
                if nxt is not None:
                    fut = pool.submit(self.tokenizer.encode_batch, nxt, self.cfg.max_positions)
                outs.append(self._encode_flat(ids, cu))
        return np.concatenate(outs, axis=0)

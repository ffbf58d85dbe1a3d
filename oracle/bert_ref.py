"""Encoder oracle O3 — TEST INFRASTRUCTURE ONLY, never imported by the product.

A plain PyTorch fp32 CPU forward of the same BERT-class architecture with the same weights
(read from the same local model directory): Hugging Face ``BertModel`` when ``transformers``
is importable, cross-checked by the hand-written ``forward_plain`` below (so the oracle
does not silently depend on a library's attention implementation).

Parity status: "unpinned" — the reference's embedding arithmetic is llama.cpp's BERT
forward inside Ollama running the ``mxbai-embed-large`` GGUF (reference app/main.py:67,
225-237); neither the server nor the weights exist offline and the reference holds no
golden embedding.  The oracle restates the published BERT forward (post-LN encoder, erf
GELU, eps 1e-12) that model implements.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Sequence

import numpy as np
import torch


def _load(path: str):
    from safetensors import safe_open
    w: Dict[str, torch.Tensor] = {}
    with safe_open(os.path.join(path, "model.safetensors"), framework="pt") as f:
        for k in f.keys():
            name = k[5:] if k.startswith("bert.") else k
            w[name] = f.get_tensor(k).float()
    with open(os.path.join(path, "config.json"), encoding="utf-8") as f:
        cfg = json.load(f)
    return cfg, w


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def forward_plain(path: str, seqs: Sequence[Sequence[int]], bf16_weights: bool = False) -> List[np.ndarray]:
    """Last hidden states [len(seq), H] per sequence; fp32 (fp64 accumulate-free) torch ops only.
    ``bf16_weights`` rounds the matrices to bf16 first (what the HIP encoder stores)."""
    cfg, w = _load(path)
    H, L, A = cfg["hidden_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"]
    eps = cfg.get("layer_norm_eps", 1e-12)
    d = H // A

    def mat(name):
        t = w[name]
        return t.bfloat16().float() if bf16_weights else t

    outs = []
    with torch.no_grad():
        for ids in seqs:
            ids_t = torch.tensor(list(ids), dtype=torch.long)
            S = ids_t.numel()
            x = mat("embeddings.word_embeddings.weight")[ids_t] + mat("embeddings.token_type_embeddings.weight")[0] + \
                mat("embeddings.position_embeddings.weight")[:S]
            x = _ln(x, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], eps)
            for l in range(L):
                p = f"encoder.layer.{l}."
                q = x @ mat(p + "attention.self.query.weight").T + w[p + "attention.self.query.bias"]
                k = x @ mat(p + "attention.self.key.weight").T + w[p + "attention.self.key.bias"]
                v = x @ mat(p + "attention.self.value.weight").T + w[p + "attention.self.value.bias"]
                q, k, v = (t.view(S, A, d).transpose(0, 1) for t in (q, k, v))
                att = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(d), dim=-1) @ v
                ctx = att.transpose(0, 1).reshape(S, H)
                y = ctx @ mat(p + "attention.output.dense.weight").T + w[p + "attention.output.dense.bias"] + x
                x = _ln(y, w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"], eps)
                h = torch.nn.functional.gelu(x @ mat(p + "intermediate.dense.weight").T + w[p + "intermediate.dense.bias"])
                z = h @ mat(p + "output.dense.weight").T + w[p + "output.dense.bias"] + x
                x = _ln(z, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], eps)
            outs.append(x.numpy())
    return outs


def forward_hf(path: str, seqs: Sequence[Sequence[int]]) -> List[np.ndarray]:
    """Same through Hugging Face ``BertModel`` (fp32 CPU, padded batch + attention mask)."""
    from transformers import BertConfig, BertModel
    cfg, w = _load(path)
    hf = BertConfig(**{k: v for k, v in cfg.items() if k not in ("architectures", "model_type")})
    model = BertModel(hf, add_pooling_layer=False).eval()
    missing, unexpected = model.load_state_dict(w, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    S = max(len(s) for s in seqs)
    ids = torch.zeros((len(seqs), S), dtype=torch.long)
    mask = torch.zeros((len(seqs), S), dtype=torch.long)
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = torch.tensor(list(s))
        mask[i, : len(s)] = 1
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask).last_hidden_state
    return [out[i, : len(s)].numpy() for i, s in enumerate(seqs)]


def pool(hidden: Sequence[np.ndarray], mode: str = "cls", normalize: bool = False) -> np.ndarray:
    """cls = first token; mean = mean over the sequence's (real) tokens; optional reference
    normalise e / (||e|| + 1e-9) (app/main.py:1249-1251)."""
    rows = [h[0] if mode == "cls" else h.mean(axis=0) for h in hidden]
    e = np.stack(rows).astype(np.float32)
    if normalize:
        e = e / (np.linalg.norm(e, axis=1, keepdims=True) + 1e-9)
    return e

"""cfg 3 of BASELINE.json: end-to-end ingest (encode -> normalise + pack into the HBM index)
of synthetic chunks with the BERT-large-class HIP encoder (seeded random weights), then
top-10 search over what was ingested.  Device-resident hand-off: the encoder's pooled
output goes straight to rass_index_add_device (no host round trip)."""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd import _native as N
from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, weight_names
from rassengine_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=16384)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--profile", choices=["fixed512", "varlen"], default="fixed512")
ap.add_argument("--layers", type=int, default=24)
ap.add_argument("--from-text", action="store_true",
                help="start from raw text: C++ WordPiece tokenisation (overlapped with the GPU) -> encode -> host "
                     "embeddings -> index add, i.e. what store_fhir_docs_in_opensearch does per upload")
ap.add_argument("--cpu-baseline-chunks", type=int, default=0,
                help="also time the fp32 PyTorch-CPU forward of the same architecture (stand-in for Ollama's "
                     "CPU embed path, BASELINE.md s3) on this many 512-token chunks, batch 1 and batch 8")
a = ap.parse_args()

cfg = EncoderConfig(layers=a.layers, pooling="mean")
rng = np.random.default_rng(0)
H, I = cfg.hidden, cfg.intermediate
w = {}
for name in weight_names(cfg.layers):
    if name.endswith("word_embeddings.weight"): shape = (cfg.vocab_size, H)
    elif name.endswith("position_embeddings.weight"): shape = (cfg.max_positions, H)
    elif name.endswith("token_type_embeddings.weight"): shape = (2, H)
    elif "LayerNorm" in name: shape = (H,)
    elif name.endswith(".bias"): shape = (I,) if "intermediate" in name else (H,)
    elif "intermediate.dense" in name: shape = (I, H)
    elif ".output.dense" in name and "attention" not in name: shape = (H, I)
    else: shape = (H, H)
    t = rng.standard_normal(shape, dtype=np.float32) * (0.03 if len(shape) == 2 else 0.05)
    if "LayerNorm.weight" in name: t = 1 + t
    w[name] = t
tok = None
if a.from_text:
    import tempfile
    from rassengine_amd.encoder import CppWordPieceTokenizer, synthetic_vocab
    vocab = synthetic_vocab(cfg.vocab_size)
    vdir = tempfile.mkdtemp()
    with open(os.path.join(vdir, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    tok = CppWordPieceTokenizer(os.path.join(vdir, "vocab.txt"))
enc = HipSentenceEncoder(cfg, w, tok, device=0)
eng = Engine(0, H)
idx = eng.open_index("ingest", capacity_rows=a.chunks)
# encoder and index on ONE stream (the encoder's own, non-blocking): the add is ordered behind the forward.
# torch's default stream is pointer 0, which rass_encode_device reads as "my own stream": passing it to both
# (as round 1 did) left the forward and the add on two unordered streams.
stream = enc.stream
eng.set_stream(stream)
L = N.lib()
rng = np.random.default_rng(99)
n_batches = (a.chunks + a.batch - 1) // a.batch
out = torch.empty((a.batch, H), device="cuda")

def make_batch(nb):
    lens = np.full(nb, 512) if a.profile == "fixed512" else rng.integers(64, 513, size=nb)
    ids = rng.integers(0, cfg.vocab_size, size=int(lens.sum())).astype(np.int32)
    cu = np.zeros(nb + 1, dtype=np.int32); np.cumsum(lens, out=cu[1:])
    return torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), int(lens.sum()), int(lens.max())

batches = [make_batch(min(a.batch, a.chunks - b * a.batch)) for b in range(min(n_batches, 8))]  # token data resident in HBM
def ingest_one(b):
    d_ids, d_cu, total, mx = batches[b % len(batches)]
    nb = d_cu.numel() - 1
    N.check("enc", L.rass_encode_device(enc._h, ctypes.c_void_p(d_ids.data_ptr()), ctypes.c_void_p(d_cu.data_ptr()),
                                        nb, total, mx, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
    idx.add_device(out.data_ptr(), nb, normalize=True)
    return nb, total
if a.from_text:
    words = [v for v in vocab if v.isalpha() and len(v) > 2][:5000]
    trng = np.random.default_rng(5)
    n_words = 380 if a.profile == "fixed512" else None
    texts = [" ".join(trng.choice(words, size=n_words or int(trng.integers(40, 400)))) + ". BP 120/80 mmHg, HbA1c 7.2%."
             for _ in range(a.chunks)]
    enc.encode(texts[:a.batch])
    t0 = time.perf_counter()
    emb = enc.encode(texts)                       # tokenise (worker thread) || GPU encode, host embeddings out
    t_embed = time.perf_counter() - t0
    idx.add(emb, normalize=True)
    eng.synchronize()
    dt = time.perf_counter() - t0
    n_tok = None
    print(json.dumps({"workload": f"text ingest {a.chunks} chunks (~{n_words or 'varlen'} words), tokenise + encode + add",
                      "chunks_per_s": round(a.chunks / dt, 1), "embed_only_chunks_per_s": round(a.chunks / t_embed, 1),
                      "seconds": round(dt, 3), "index_rows": idx.rows}))
    sys.exit(0)
ingest_one(0); torch.cuda.synchronize()
eng.drop_index("ingest"); idx = eng.open_index("ingest", capacity_rows=a.chunks)
t0 = time.perf_counter(); chunks = tokens = 0
for b in range(n_batches):
    nb, total = ingest_one(b); chunks += nb; tokens += total
torch.cuda.synchronize()
dt = time.perf_counter() - t0
q = torch.randn((32, H), device="cuda")
os_ = torch.empty((32, 10), device="cuda"); oi = torch.empty((32, 10), dtype=torch.int64, device="cuda")
idx.search_device(q.data_ptr(), 32, 10, os_.data_ptr(), oi.data_ptr()); torch.cuda.synchronize()
flops_tok = cfg.layers * 2 * (4 * H * H + 2 * H * I)
cpu = None
if a.cpu_baseline_chunks > 0:
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    hf = BertModel(BertConfig(vocab_size=cfg.vocab_size, hidden_size=H, num_hidden_layers=cfg.layers,
                              num_attention_heads=cfg.heads, intermediate_size=I, max_position_embeddings=512),
                   add_pooling_layer=False).eval()
    n = a.cpu_baseline_chunks
    ids_cpu = torch.randint(0, cfg.vocab_size, (n, 512))
    with torch.no_grad():
        hf(input_ids=ids_cpu[:1])  # warm
        t0 = time.perf_counter()
        for i in range(n):          # one request per text, like the reference (app/main.py:252-260)
            hf(input_ids=ids_cpu[i:i + 1])
        t_b1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(0, n, 8):
            hf(input_ids=ids_cpu[i:i + 8])
        t_b8 = time.perf_counter() - t0
    cpu = {"kind": "port (PyTorch-CPU fp32 BertModel, same architecture, random weights)", "threads": torch.get_num_threads(),
           "chunks": n, "chunks_per_s_batch1": round(n / t_b1, 3), "chunks_per_s_batch8": round(n / t_b8, 3)}
print(json.dumps({"cpu_baseline": cpu, "workload": f"ingest {chunks} chunks, batch {a.batch}, {a.profile}, BERT-large-class bf16 (random weights)",
                  "chunks_per_s": round(chunks / dt, 1), "tokens_per_s": round(tokens / dt), "seconds": round(dt, 3),
                  "linear_TFLOPs": round(tokens * flops_tok / dt / 1e12, 1), "index_rows": idx.rows,
                  "extrapolated_1M_chunks_s": round(1e6 / (chunks / dt), 1)}))

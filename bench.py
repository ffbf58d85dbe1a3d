#!/usr/bin/env python3
"""Headline benchmark: exact cosine top-10 queries/sec over an N x 1024-d corpus resident
in HBM (BASELINE.json metric; N=1 workload = configs[1], "1M x 1024-d flat cosine top-10,
1 x MI355X, precomputed embeddings").

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N = 1 runs BASELINE configs[1] (1 M rows on the GPU); N > 1 runs BASELINE configs[3] by default: the FIXED
10 M x 1024 corpus row-sharded over the N ranks ("strong" scaling; --weak keeps 1 M rows per GPU instead), with
proof in the line itself that RCCL saw N ranks and that the merged result equals a CPU merge of the shards' lists.
At N = 1 the line also carries `ingest`: BASELINE configs[2]'s encoder forward (batch 256 x 512 tokens ->
rass_encode_device -> rass_index_add_device), timed OUTSIDE the search region, against the bf16 MFMA peak.

A step = one pass of the hot path over one batch of 1 024 synthetic queries.  The scan kernel
answers <= 32 queries per launch (two 16-wide MFMA N tiles), so a step is 32 LAUNCH GROUPS of
32 queries, each: query normalise, fused scan + top-k over the rank's shard, merge; for N > 1
also the query broadcast, the RCCL all-gather of per-shard top-k and the cross-shard merge
(each query is answered over all N shards).  (Round 1 called ONE launch group a step: 0.7 ms, so
a driver run of --warmup 5 --steps 20 timed 14 ms of GPU work that began 3.5 ms after the GPU
left idle — inside the ~20 ms the clocks take to settle — and hid 40 extra scans in front to
compensate.  A 1 024-query step is 22 ms: --warmup is exactly what runs, and it is enough.)
Inputs are in HBM when the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; not the 2:1-sparsity headline)
CONFIGS3_ROWS = 10_000_000      # BASELINE configs[3]: "10M x 1024-d sharded flat cosine top-10, 8 x MI355X"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--rows-global", type=int, default=-1,
                    help="a fixed global corpus split into contiguous row ranges over the ranks (STRONG scaling). "
                         "Default: 10000000 (BASELINE configs[3]) when --gpus > 1, off at --gpus 1 (configs[1])")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: weak scaling instead (every GPU holds --rows-per-gpu rows, corpus = N x that)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the encoder (configs[2]) leg of the N = 1 line")
    ap.add_argument("--no-ivf", action="store_true", help="skip the IVF (configs[4] per-GPU share) leg of the N = 1 line")
    ap.add_argument("--no-scale-ref", action="store_true",
                    help="skip the N = 1 line's strong_scaling_reference leg (BASELINE configs[3]'s 10 M-row corpus on ONE GPU: the "
                         "same-workload reference for the --gpus N > 1 lines, which shard that corpus)")
    ap.add_argument("--mode", choices=["flat", "ivf"], default="flat",
                    help="ivf (N >= 1): BASELINE configs[4] — IVF-4096 over --ivf-rows clustered rows PER GPU (12.5 M = the "
                         "100 M-row corpus / 8), shared centroids, per-shard probes, one all-gather of per-shard top-k")
    ap.add_argument("--ivf-rows", type=int, default=12_500_000, help="rows per GPU of the IVF leg / mode")
    ap.add_argument("--ivf-nlist", type=int, default=4096)
    ap.add_argument("--ivf-nprobe", type=int, default=8, help="lists probed per query and shard in --mode ivf's timed region")
    ap.add_argument("--ivf-queries", type=int, default=1024, help="queries of the IVF leg's recall / rate sweep")
    ap.add_argument("--ivf-dtype", choices=["f32", "bf16", "int8"], default="f32",
                    help="--mode ivf: the IVF's list-ordered copy of the rows: f32 (the parity path), bf16 (the bf16 scan's scores) or "
                         "int8 (int8 candidates + exact fp32 re-rank: the f32 IVF's scores); the N = 1 line's ivf leg sweeps f32 "
                         "and, flagged, int8")
    ap.add_argument("--ingest-batches", type=int, default=4, help="timed 256 x 512-token forwards of the ingest leg")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32, help="queries per scan launch (<= 32)")
    ap.add_argument("--launches-per-step", type=int, default=32,
                    help="scan launch groups per step: a step answers batch x this many queries (default 1024)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--query-pool", type=int, default=4096)
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--cpu-hnsw-rows", type=int, default=10_000,
                    help="rows of the HNSW restatement's sample (cpu_baseline.hnsw); 0 disables it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-group", action="store_true",
                    help="one engine call per launch group of 32 queries (rounds 1-3's N = 1 default) instead of one per step")
    ap.add_argument("--batched", action="store_true",
                    help="(the default since round 4) one engine call per 1 024-query step (rass_index_search_device_batch)")
    ap.add_argument("--merge", choices=["allgather", "peer"], default="allgather",
                    help="cross-shard exchange for N > 1: one RCCL all-gather (default) or peer stores into rank 0's "
                         "buffer + flags (SURVEY 8f-4; validated on 2 ranks sharing a GPU only)")
    ap.add_argument("--corpus-dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16: FLAGGED mode, never the headline: a bf16-only corpus (RASS_BF16), half the bytes per "
                         "scan, scores within ~1e-3 of the fp32 cosine; recall@k vs the fp32 index is reported")
    ap.add_argument("--prefilter", nargs="?", const="bf16", default=None, choices=["bf16", "int8"],
                    help="flagged mode (not the parity default): bf16 (half the bytes per pass) or int8 (a quarter) candidate "
                         "scan + exact fp32 re-rank; recall vs the flat scan is reported")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one rank per GPU; RASS_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box, gloo) lets ranks share a device
    share = os.environ.get("RASS_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll = "gloo (ranks sharing one GPU: a rehearsal, not a measurement)" if share else "RCCL"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from rassengine_amd.dist import HipShard, PeerMergeSearch, ShardedSearch, shard_bounds
    from rassengine_amd.engine import Engine, scan_kernel_name

    if args.mode == "ivf":
        result = ivf_mode(np, torch, dist, args, world, rank, local_rank, dev, coll)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    dim, B, k = args.dim, args.batch, args.k
    if args.rows_global < 0:
        args.rows_global = CONFIGS3_ROWS if (world > 1 and not args.weak) else 0
    if args.weak:
        args.rows_global = 0
    strong = args.rows_global > 0
    rows_global = args.rows_global if strong else args.rows_per_gpu * world
    row_lo, row_hi = shard_bounds(rows_global, world)[rank]
    n_local = row_hi - row_lo
    eng = Engine(device=local_rank, dim=dim)
    bf16 = args.corpus_dtype == "bf16"
    if bf16 and args.prefilter:
        raise SystemExit("--prefilter is a mode of the fp32 corpus")
    idx = eng.open_index("bench", capacity_rows=n_local, dtype=args.corpus_dtype)
    # Philox rows keyed by the GLOBAL row id: shard r regenerates exactly rows [row_lo, row_hi)
    idx.fill_synthetic(n_local, seed=1234, row_id_base=row_lo)
    if args.prefilter:
        idx.set_prefilter(args.prefilter)
    eng.synchronize()

    shard = HipShard(idx, id_base=row_lo)  # switches the engine to torch's current stream
    search = PeerMergeSearch(shard) if (args.merge == "peer" and world > 1) else ShardedSearch(shard)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)
    pool = torch.randn((args.query_pool, dim), generator=gen, device=dev)  # same on every rank (same seed)
    n_batches = args.query_pool // B
    q_buf = torch.empty((B, dim), device=dev)

    LPS = args.launches_per_step

    step_q = torch.empty((LPS * B, dim), device=dev)
    # N > 1: one engine call per step (rass_index_search_device_batch: one normalise + one merge launch for the
    # step's 32 launch groups), ONE broadcast and ONE all-gather.  N = 1 keeps one engine call per launch group —
    # the unit a caller of the reference's semantic_search issues — unless --batched: measured on MI355X the two
    # give the same queries/s (21.09 vs 21.10 ms per step): the chip is power-limited in this kernel, and what the
    # batch saves in launches and idle gaps (-0.5 ms) comes back as a lower clock in the back-to-back scans
    # (628 vs 611 us per launch).
    # (r4: with ONE sample pass for the step's 32 launch groups — kFlatSampleGroups — the batched call is 1.4-2 % ahead at N = 1
    # and is the default there too; --per-group keeps one engine call per launch group.)
    batched = isinstance(search, ShardedSearch) and B == 32 and not args.per_group

    def step(i: int):
        if batched:
            # the step's 1 024 queries travel in ONE broadcast, the per-shard top-k of its 32 launch groups in ONE
            # all-gather (2 collectives per step instead of 64), then one grouped merge launch
            if rank == 0:
                g0 = (i * LPS) % n_batches
                if g0 + LPS <= n_batches:   # the step's launch groups are consecutive pool batches: one device copy
                    step_q.copy_(pool[g0 * B:(g0 + LPS) * B])
                else:                       # (a pool that is not a multiple of the step: group by group, as before)
                    for j in range(LPS):
                        g = (i * LPS + j) % n_batches
                        step_q[j * B:(j + 1) * B].copy_(pool[g * B:(g + 1) * B])
            return search.search_batch(step_q, k, B)
        out = None
        for j in range(LPS):   # one step = LPS launch groups of B queries each
            g = (i * LPS + j) % n_batches
            if rank == 0:
                q_buf.copy_(pool[g * B:(g + 1) * B])
            out = search.search(q_buf, k)  # scan + merge (N = 1), or the peer-store exchange per group
        return out

    # EXACTLY --warmup untimed steps, nothing else.
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.kernel_timing_begin(args.steps * LPS)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    scan_ms, scan_launches = eng.kernel_timing_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    qps = B * LPS * args.steps / elapsed
    # algorithmic bytes of the dominant kernel: N_loc * D * 4 (fp32 scan); the bf16
    # candidate scan of the prefilter mode reads N_loc * D * 2
    i8_stride = (idx.row_stride + 511) // 512 * 512
    bytes_per_launch = n_local * i8_stride if args.prefilter == "int8" else \
        n_local * idx.row_stride * (2 if (args.prefilter or bf16) else 4)
    achieved = bytes_per_launch * scan_launches / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0

    result = {
        "metric": (f"queries/sec, cosine top-10 over N x 1024-d fp32 corpus in HBM through {args.prefilter} candidates + exact "
                   "fp32 re-rank (FLAGGED mode, not the headline)") if args.prefilter else
                  "queries/sec, exact cosine top-10 over N x 1024-d fp32 corpus in HBM" if not bf16 else
                  "queries/sec, cosine top-10 over N x 1024-d bf16 corpus in HBM (FLAGGED mode, not the headline)",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        # weak scaling in CORPUS ROWS (every GPU keeps rows_per_gpu rows, every query is answered over all
        # N shards): the ideal is CONSTANT queries/s while rows_global grows N x; the quantity that grows
        # N x is row_queries_per_s = value * rows_global (and config.aggregate_scan_GBps)
        "row_queries_per_s": round(qps * rows_global, 1),
        "vs_baseline": None,
        "dtype": "bf16" if bf16 else "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{rows_global} x {dim}-d flat cosine top-{k}, {world} x MI355X, precomputed embeddings "
                        + (("(BASELINE configs[3]: fixed corpus, row-sharded, all-gather merge)" if rows_global == CONFIGS3_ROWS
                            else "(fixed corpus, row-sharded)") if strong else
                           "(BASELINE configs[1])" if world == 1 else "(BASELINE configs[1] shard per GPU, weak scaling)"),
            "rows_per_gpu": n_local, "rows_global": rows_global, "dim": dim, "k": k, "query_batch": B,
            "queries_per_step": B * LPS, "launch_groups_per_step": LPS,
            "engine_calls_per_step": 1 if batched else LPS,
            "corpus_dtype": "bf16 only (fp32-accumulated bf16 MFMA)" if bf16 else
                            "f32" if not args.prefilter else f"f32 + {args.prefilter} candidate copy (exact fp32 re-rank)",
            "layout": "tile16b" if bf16 else "tile16", "mode": "prefilter" if args.prefilter else "flat",
            "cross_shard_exchange": ("peer stores + flags" if args.merge == "peer" else f"{coll} all-gather") if world > 1 else None,
            "sharding": f"row-sharded x{world}, {coll} all-gather merge" if world > 1 else "single shard",
            "aggregate_scan_GBps": round(bytes_per_launch * world * args.steps * LPS / elapsed / 1e9, 1),
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "kernel": scan_kernel_name(dim, B) if not (args.prefilter or bf16) else
            f"scan_i8_topk_kernel<{i8_stride // 512}, {1 if B <= 16 else 2}>" if args.prefilter == "int8" else
            f"scan_bf16_topk_kernel<{idx.row_stride // 256}, {1 if B <= 16 else 2}>", "bytes_per_launch": bytes_per_launch,
            "avg_launch_us": round(scan_ms / max(scan_launches, 1) * 1e3, 2), "launches": scan_launches,
        },
    }

    if world > 1:
        result.update(multi_gpu_proof(np, torch, dist, search, shard, pool, dev, rank, world, local_rank, row_lo, n_local,
                                      k, B, scan_ms, scan_launches, bytes_per_launch, isinstance(search, ShardedSearch)))
        result["scaling_note"] = (
            "strong scaling over a FIXED corpus: the N = 1 line of this bench is BASELINE configs[1] (1 M rows), not "
            "this corpus on one GPU (that number is the N = 1 line's strong_scaling_reference.value: ideal value_N = N x it); "
            "the workload-independent figure to compare across N is row_queries_per_s (= value x rows_global)") if strong else "weak scaling: rows_global grows N x; compare row_queries_per_s"

    # `traffic` = HBM bytes per launch from PMC counters.  They cannot be read from inside this process, so the
    # field stays null in an ordinary run; a run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` is summarised
    # by scripts/pmc_traffic.py into profiles/*pmc_traffic*.json, and the newest committed figure for THIS kernel
    # at THIS byte count is quoted under a key of its own (it was not measured by this run).
    result["roofline"]["traffic_committed_pmc"] = pmc_traffic(result["roofline"]["kernel"], bytes_per_launch)

    # Outside the timed region, single GPU only: the same scan at the other batch sizes SURVEY §8d
    # asks for (B <= 16 runs the NT=1 kernel variant, purely HBM-bound; B = 32 sits at the HBM / fp32-
    # MFMA corner).  Reported next to the headline, never as `value`.
    if world == 1 and not args.prefilter and not bf16 and rank == 0:
        others = []
        for Bo in (16, 1):
            if Bo == B:
                continue
            qo = pool[:Bo].contiguous()
            for _ in range(10):
                search.search(qo, k)
            torch.cuda.synchronize()
            n_o = 100
            eng.kernel_timing_begin(n_o)
            t0 = time.perf_counter()
            for _ in range(n_o):
                search.search(qo, k)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ms_o, launches_o = eng.kernel_timing_end()
            ach = bytes_per_launch * launches_o / (ms_o * 1e-3) / 1e9 if ms_o > 0 else 0.0
            others.append({"query_batch": Bo, "queries_per_s": round(Bo * n_o / el, 1), "kernel": scan_kernel_name(dim, Bo),
                           "avg_launch_us": round(ms_o / max(launches_o, 1) * 1e3, 2), "achieved": round(ach, 1),
                           "frac": round(ach / HBM_PEAK_GBPS, 4)})
        result["roofline_other_batches"] = others

    if rank == 0 and world == 1 and args.prefilter:
        # the flagged mode is approximate in principle: measure it against the exact flat scan
        qh = pool[:B].cpu().numpy()
        s_p, i_p = idx.search(qh, k)
        idx.set_prefilter(False)
        s_f, i_f = idx.search(qh, k)
        idx.set_prefilter(args.prefilter)
        result["prefilter_recall_vs_flat"] = float(np.mean([len(set(i_p[r]) & set(i_f[r])) / k for r in range(B)]))
        result["prefilter_scores_bit_identical"] = bool(np.array_equal(s_p[i_p == i_f], s_f[i_p == i_f]))

    if rank == 0 and world == 1 and bf16:
        # the flagged mode against the parity path: the same first rows as an fp32 index and as a bf16 index
        sample = min(args.cpu_sample_rows, n_local)
        ref = eng.open_index("bench-f32-sample", capacity_rows=sample)
        ref.fill_synthetic(sample, seed=1234, row_id_base=row_lo)
        sb = eng.open_index("bench-bf16-sample", capacity_rows=sample, dtype="bf16")
        sb.fill_synthetic(sample, seed=1234, row_id_base=row_lo)
        qh = pool[:4 * B].cpu().numpy()
        s_f, i_f = ref.search(qh, k)
        s_b, i_b = sb.search(qh, k)
        same = i_f == i_b
        result["bf16_vs_f32"] = {
            "sample_rows": sample, "queries": int(qh.shape[0]),
            "recall_at_k": float(np.mean([len(set(i_b[r]) & set(i_f[r])) / k for r in range(qh.shape[0])])),
            "max_abs_score_diff_on_common_ranks": float(np.abs(s_f[same] - s_b[same]).max()) if same.any() else None}
        if not args.no_cpu_baseline:
            result.update(cpu_baseline_and_recall(np, torch, eng, ref, pool, args, n_local, dim, B, k))
            result["recall_at_k_note"] = "recall_at_k / max_abs_cosine_err above are the fp32 parity path's on the sample"
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        result.update(cpu_baseline_and_recall(np, torch, eng, idx, pool, args, n_local, dim, B, k))
        if not args.prefilter:
            # the TIMED path's own output: the last launch group of the timed loop (32 queries over ALL rows of the shard)
            # against the oracle's fp64 ranking of the same rows — not a separate launch over a prefix
            g_last = ((args.warmup + args.steps - 1) * LPS + LPS - 1) % n_batches
            last = (out[0][-B:], out[1][-B:]) if batched else out
            result.update(timed_path_check(np, idx, pool[g_last * B:(g_last + 1) * B], last, n_local, k))

    if rank == 0 and world == 1 and not args.no_scale_ref and not bf16 and not args.prefilter and dim == 1024 and B == 32 \
            and rows_global == 1_000_000:
        # Outside the timed region: BASELINE configs[3]'s corpus (10 M x 1024 fp32 = 41 GB) on THIS one GPU, the same step
        # (1 024 queries, one engine call).  `bench.py --gpus N` (N > 1) shards exactly this corpus over N ranks ("strong"
        # scaling): its `value` is to be compared with N x THIS number, not with the 1 M-row headline above.
        try:
            big = eng.open_index("bench-cfg3", capacity_rows=CONFIGS3_ROWS)
            big.fill_synthetic(CONFIGS3_ROWS, seed=1234, row_id_base=0)
            eng.synchronize()
            bs = torch.empty((LPS * B, k), dtype=torch.float32, device=dev)
            bi = torch.empty((LPS * B, k), dtype=torch.int64, device=dev)
            big.search_device_batch(step_q.data_ptr(), LPS * B, k, bs.data_ptr(), bi.data_ptr())
            torch.cuda.synchronize()
            eng.kernel_timing_begin(2 * LPS)
            t1 = time.perf_counter()
            for _ in range(2):
                big.search_device_batch(step_q.data_ptr(), LPS * B, k, bs.data_ptr(), bi.data_ptr())
            torch.cuda.synchronize()
            el = time.perf_counter() - t1
            ms_b, launches_b = eng.kernel_timing_end()
            bytes_b = CONFIGS3_ROWS * idx.row_stride * 4
            result["strong_scaling_reference"] = {
                "workload": f"{CONFIGS3_ROWS} x {dim}-d flat cosine top-{k}, 1 x MI355X (BASELINE configs[3]'s corpus on one GPU)",
                "value": round(2 * LPS * B / el, 1), "unit": "queries/s", "steps": 2, "ms_per_step": round(el / 2 * 1e3, 2),
                "roofline": {"bound": "hbm", "achieved": round(bytes_b * launches_b / (ms_b * 1e-3) / 1e9, 1) if ms_b > 0 else None,
                             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "bytes_per_launch": bytes_b,
                             "avg_launch_us": round(ms_b / max(launches_b, 1) * 1e3, 1)},
                "note": "the --gpus N > 1 lines shard THIS corpus: ideal value_N = N x this value (not N x the 1 M-row headline)"}
            eng.drop_index("bench-cfg3")
        except Exception as e:  # noqa: BLE001 — the reference leg must not take the headline down
            result["strong_scaling_reference"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and not args.no_ingest and not bf16 and not args.prefilter and dim == 1024:
        # BASELINE configs[2]'s larger half (the "embedding" of "embedding + ANN"): after the timed search region and
        # outside it.  The search index is dropped first so the encoder does not share HBM bandwidth accounting with it.
        eng.synchronize()
        result["ingest"] = ingest_leg(np, torch, local_rank, args.ingest_batches)

    if rank == 0 and world == 1 and not args.no_ivf and not bf16 and not args.prefilter and dim == 1024:
        # BASELINE configs[4] at one GPU's share (12.5 M of the 100 M rows), outside the timed region: the search index is
        # dropped first (the IVF leg holds 51 GB of rows + the IVF's own 51 GB copy)
        eng.synchronize()
        eng.drop_index("bench")
        try:
            result["ivf"] = ivf_leg(np, torch, local_rank, args)
        except Exception as e:      # the headline must not die with an auxiliary leg
            result["ivf"] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if isinstance(search, PeerMergeSearch):
        search.check()
        search.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def multi_gpu_proof(np, torch, dist, search, shard, pool, dev, rank, world, local_rank, row_lo, n_local, k, B, scan_ms,
                    scan_launches, bytes_per_launch, can_check):
    """Outside the timed region: what makes an N > 1 line self-proving.  (1) who took part: world size and backend as
    torch.distributed reports them, and per rank the device index, PCI bus id, rows held and its own average scan launch
    (hipEvents) — N distinct bus ids = N GPUs.  (2) `sharded_equals_merge_of_shards`: for one batch of queries every rank's
    LOCAL top-k (its shard only, global ids) is gathered as Python objects, merged on the CPU with numpy lexsort
    (score desc, id asc) and compared BIT FOR BIT with what the engine's exchange + merge kernels returned."""
    props = torch.cuda.get_device_properties(dev)
    bus = None
    if hasattr(props, "pci_bus_id"):
        bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
    avg_us = scan_ms / max(scan_launches, 1) * 1e3
    mine = {"rank": rank, "device_index": local_rank, "pci_bus_id": bus, "device_name": props.name,
            "uuid": str(getattr(props, "uuid", "")) or None, "rows_held": n_local, "row_lo": row_lo,
            "avg_launch_us": round(avg_us, 2), "launches": scan_launches,
            "frac": round(bytes_per_launch / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if avg_us > 0 else None}
    ranks = [None] * world
    dist.all_gather_object(ranks, mine)
    out = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks,
           "distinct_pci_bus_ids": len({r["pci_bus_id"] for r in ranks if r["pci_bus_id"]}),
           "roofline_per_rank": {"avg_launch_us_min": min(r["avg_launch_us"] for r in ranks),
                                 "avg_launch_us_max": max(r["avg_launch_us"] for r in ranks),
                                 "frac_min": min(r["frac"] for r in ranks), "frac_max": max(r["frac"] for r in ranks)}}
    q = pool[:B].contiguous().clone()
    if can_check:
        merged_s, merged_i = search.search(q, k)          # broadcast + local scan + all-gather + merge kernel
        loc_s, loc_i = shard.search_local(q, k)           # this rank's shard alone, global ids
        torch.cuda.synchronize()
        lists = [None] * world
        dist.all_gather_object(lists, (loc_s.cpu().numpy(), loc_i.cpu().numpy()))
        if rank == 0:
            cs = np.concatenate([l[0] for l in lists], axis=1)        # [B, world * k]
            ci = np.concatenate([l[1] for l in lists], axis=1)
            ok = True
            ms, mi = merged_s.cpu().numpy(), merged_i.cpu().numpy()
            for r in range(B):
                live = ci[r] >= 0
                order = np.lexsort((ci[r][live], -cs[r][live].astype(np.float64)))[:k]
                ok = ok and np.array_equal(ci[r][live][order], mi[r][:order.size]) and \
                    np.array_equal(cs[r][live][order].view(np.uint32), ms[r][:order.size].view(np.uint32))
            out["sharded_equals_merge_of_shards"] = bool(ok)
            out["sharded_check"] = f"{B} queries, top-{k}: numpy lexsort merge of the {world} ranks' local lists vs the engine"
    else:
        out["sharded_equals_merge_of_shards"] = None
    return out


def ingest_leg(np, torch, device, timed_batches):
    """BASELINE configs[2] / SURVEY §8d cfg 3: the BERT-large-class encoder (24 x 1024 x 16 heads x 4096, vocab 30 522,
    bf16 weights — seeded random, no real weights offline) on batches of 256 chunks x 512 random tokens (seed 99),
    device-resident: rass_encode_device -> rass_index_add_device on the encoder's stream, 2 warm + `timed_batches`
    timed, bracketed by hipEvents on that stream.  FLOPs per SURVEY §8d: 604.0 MFLOP per token in the linears +
    98 304 x S per token in attention."""
    from rassengine_amd.encoder import EncoderConfig, HipSentenceEncoder, random_weights
    from rassengine_amd.engine import Engine, HipTimer
    cfg = EncoderConfig(pooling="mean")
    t0 = time.perf_counter()
    enc = HipSentenceEncoder(cfg, random_weights(cfg, seed=1), None, device=device)
    setup_s = time.perf_counter() - t0
    eng2 = Engine(device=device, dim=cfg.hidden)
    try:
        nseq, S = 256, 512
        warm = 2
        idx2 = eng2.open_index("bench-ingest", capacity_rows=(warm + timed_batches) * nseq)
        rng = np.random.default_rng(99)
        batches = []
        for _ in range(2):   # two different token batches, alternated (random data, not zeros: the clocks depend on it)
            ids = torch.from_numpy(rng.integers(0, cfg.vocab_size, size=nseq * S).astype(np.int32)).to(f"cuda:{device}")
            batches.append(ids)
        cu = torch.arange(0, (nseq + 1) * S, S, dtype=torch.int32, device=f"cuda:{device}")
        out = torch.empty((nseq, cfg.hidden), dtype=torch.float32, device=f"cuda:{device}")
        torch.cuda.synchronize()
        stream = enc.stream
        eng2.set_stream(stream)
        timer = HipTimer()

        def one(b):
            enc.encode_device(batches[b % 2].data_ptr(), cu.data_ptr(), nseq, nseq * S, S, out.data_ptr(), stream)
            idx2.add_device(out.data_ptr(), nseq, normalize=True)

        for b in range(warm):
            one(b)
        eng2.synchronize()
        timer.start(stream)
        for b in range(timed_batches):
            one(warm + b)
        timer.stop(stream)
        eng2.synchronize()
        ms = timer.elapsed_ms() / timed_batches
        rows = idx2.get_rows((warm + timed_batches - 1) * nseq, nseq)
        norms = np.linalg.norm(rows.astype(np.float64), axis=1)
        tokens = nseq * S
        flops = tokens * (604.0e6 + 98304.0 * S)
        tflops = flops / (ms * 1e-3) / 1e12
        st = enc.stats()
        return {
            "workload": "BASELINE configs[2] encoder leg: BERT-large-class (mxbai-embed-large shape) bf16, batch 256 x 512 "
                        "tokens, rass_encode_device -> rass_index_add_device (normalise + pack into the HBM index)",
            "chunks_per_s": round(nseq / (ms * 1e-3), 1), "tokens_per_s": round(tokens / (ms * 1e-3)),
            "ms_per_batch": round(ms, 3), "timed_batches": timed_batches, "warmup_batches": warm,
            "tflops": round(tflops, 1), "flop_per_batch": flops, "dtype": "bf16 (fp32 accumulate)",
            "roofline": {"bound": "mfma_bf16", "achieved": round(tflops, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4)},
            "rows_in_index": int(idx2.rows), "forwards": st["forwards"],
            "last_batch_rows_finite_and_unit_norm": bool(np.all(np.isfinite(rows)) and np.abs(norms - 1.0).max() < 1e-5),
            "extrapolated_1M_chunks_s": round(1e6 / (nseq / (ms * 1e-3)), 1),
            "weights": "seeded random (numpy PCG64 seed 1), data: random token ids seed 99", "setup_s": round(setup_s, 1),
        }
    finally:
        eng2.close()
        enc.close()


IVF_CENTRES, IVF_SIGMA, IVF_SEED = 8192, 1.0, 7       # SURVEY 8d cfg 5: 8 192 Gaussian centres (seed 7) + sigma-noise, normalised


def _ivf_fill(torch, flat, eng, rows, dev, iid, seed, centres):
    """`rows` synthetic rows straight into the flat index's slab (device-resident, 262 144 per add)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    dim = centres.shape[1]
    for lo in range(0, rows, 262144):
        n = min(262144, rows - lo)
        if iid:
            x = torch.randn((n, dim), generator=g, device=dev)
        else:
            lab = torch.randint(0, centres.shape[0], (n,), generator=g, device=dev)
            x = centres[lab] + IVF_SIGMA * torch.randn((n, dim), generator=g, device=dev) / dim ** 0.5
        torch.cuda.synchronize()
        flat.add_device(x.data_ptr(), n, normalize=True)
        eng.synchronize()


def _ivf_queries(torch, n, dev, iid, centres, seed=4321):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    dim = centres.shape[1]
    if iid:
        return torch.randn((n, dim), generator=g, device=dev).contiguous()
    lab = torch.randint(0, centres.shape[0], (n,), generator=g, device=dev)
    return (centres[lab] + IVF_SIGMA * torch.randn((n, dim), generator=g, device=dev) / dim ** 0.5).contiguous()


def _ivf_centres(torch, dev, dim):
    g = torch.Generator(device=dev)
    g.manual_seed(IVF_SEED)
    c = torch.randn((IVF_CENTRES, dim), generator=g, device=dev)
    return c / c.norm(dim=1, keepdim=True)


def _ivf_one_corpus(np, torch, eng, dev, rows, nlist, nq, k, iid, nprobes, int8_too=False):
    """Generate, build (two-level k-means + assignment + list-ordered copy), then per nprobe: queries/s over `nq` queries in
    32-query launch groups (hipEvents on the engine stream), recall@k against the flat scan of the SAME shard, the rows the
    fine scans touch (host API, 4 groups sampled) and probed bytes / time against the 8 TB/s HBM peak."""
    from rassengine_amd.engine import HipTimer
    from rassengine_amd.ivf import IvfIndex, train_centroids
    dim, B = 1024, 32
    centres = _ivf_centres(torch, dev, dim)
    name = "bench-ivf-iid" if iid else "bench-ivf"
    flat = eng.open_index(name, capacity_rows=rows)
    t0 = time.perf_counter()
    _ivf_fill(torch, flat, eng, rows, dev, iid, IVF_SEED + 1, centres)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    cent = train_centroids(flat, nlist, train_rows=0, iters=10, seed=1)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    ivf = IvfIndex.build(flat, nlist=nlist, centroids=cent)
    build_s = time.perf_counter() - t0
    try:
        q = _ivf_queries(torch, nq, dev, iid, centres)
        out_s = torch.empty((B, k), device=dev)
        truth = torch.empty((nq, k), dtype=torch.int64, device=dev)
        stream = eng.stream
        tm = HipTimer()
        tm.start(stream)
        for b in range(0, nq, B):
            flat.search_device(q[b:b + B].data_ptr(), B, k, out_s.data_ptr(), truth[b:b + B].data_ptr())
        tm.stop(stream)
        flat_ms = tm.elapsed_ms()
        truth_h = truth.cpu().numpy()
        got = torch.empty((nq, k), dtype=torch.int64, device=dev)
        stride = flat.row_stride

        def sweep_of(ivf, row_bytes):
            sweep = []
            for nprobe in nprobes:
                # group by group (one engine call per 32 queries: coarse scan, merge, plan, fine scan, merge = 5 launches each)
                for b in range(0, min(nq, 4 * B), B):
                    ivf.search_device(q[b:b + B].data_ptr(), B, k, nprobe, out_s.data_ptr(), got[b:b + B].data_ptr())
                eng.synchronize()
                tm.start(stream)
                for b in range(0, nq, B):
                    ivf.search_device(q[b:b + B].data_ptr(), B, k, nprobe, out_s.data_ptr(), got[b:b + B].data_ptr())
                tm.stop(stream)
                ms_groups = tm.elapsed_ms()
                by_group = got.cpu().numpy().copy()
                # the whole 1 024-query step in ONE call (rass_ivf_search_device_batch: one grouped coarse scan, one plan launch,
                # G fine scans, one grouped merge) — the rate reported; results must equal the group-by-group ones
                step = min(nq, 1024)
                all_s = torch.empty((nq, k), device=dev)
                for b in range(0, nq, step):
                    ivf.search_device_batch(q[b:b + step].data_ptr(), min(step, nq - b), k, nprobe, all_s[b:].data_ptr(), got[b:].data_ptr())
                eng.synchronize()
                tm.start(stream)
                for b in range(0, nq, step):
                    ivf.search_device_batch(q[b:b + step].data_ptr(), min(step, nq - b), k, nprobe, all_s[b:].data_ptr(), got[b:].data_ptr())
                tm.stop(stream)
                ms = tm.elapsed_ms()
                got_h = got.cpu().numpy()
                same = bool(np.array_equal(got_h, by_group))
                recall = float(np.mean([len(set(got_h[r]) & set(truth_h[r])) / k for r in range(nq)]))
                _, _, scanned = ivf.search(q[:4 * B].cpu().numpy(), k, nprobe)
                per_batch = scanned / 4
                us = ms / (nq / B) * 1e3
                probed = per_batch * row_bytes + nlist * stride * 4        # fine scans + the coarse scan per batch
                gbps = probed / (us * 1e-6) / 1e9
                sweep.append({"nprobe": nprobe, "queries_per_s": round(nq / ms * 1e3, 1), "recall_at_10": round(recall, 4),
                              "queries_per_s_group_by_group": round(nq / ms_groups * 1e3, 1), "batch_equals_group_by_group": same,
                              "us_per_batch": round(us, 1), "scanned_rows_per_batch": round(per_batch),
                              "scanned_fraction": round(per_batch / rows, 5),
                              "roofline": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                           "frac": round(gbps / HBM_PEAK_GBPS, 4), "bytes_per_batch": int(probed)}})
            return sweep

        sweep = sweep_of(ivf, stride * 4)
        flagged = None
        if int8_too:
            # FLAGGED (not the parity path): the same lists with an int8 copy of the slab — the fine scan reads a quarter of
            # the bytes for 32 candidates per query, rescored exactly from the fp32 copy (rass_ivf_build_ex, RASS_I8)
            ivf.close()
            t1 = time.perf_counter()
            ivf = IvfIndex.build(flat, nlist=nlist, centroids=cent, dtype="int8")
            flagged = {"slab": "fp32 + per-row-scaled int8 copy; 32 int8 candidates per query, exact fp32 re-rank (k <= 16)",
                       "build_s": round(time.perf_counter() - t1, 1),
                       "sweep": sweep_of(ivf, (stride + 511) // 512 * 512)}
        sizes = ivf.list_sizes
        return {"rows": rows, "data": "iid N(0,1) rows (the flagged WORST case: no cluster structure to find)" if iid else
                f"{IVF_CENTRES} Gaussian centres (seed {IVF_SEED}) + sigma {IVF_SIGMA} noise, normalised",
                "gen_s": round(gen_s, 1), "train_s": round(train_s, 1), "build_s": round(build_s, 1),
                "list_len_mean": round(float(sizes.mean()), 1), "list_len_max": int(sizes.max()),
                "empty_lists": int((sizes == 0).sum()), "flat_queries_per_s_same_shard": round(nq / flat_ms * 1e3, 1),
                "sweep": sweep, **({"int8_slab_flagged": flagged} if flagged else {})}
    finally:
        ivf.close()
        eng.drop_index(name)


def ivf_leg(np, torch, device, args):
    """BASELINE configs[4] / SURVEY 8d cfg 5 at ONE GPU's share: IVF-4096 (two-level k-means) over 12.5 M x 1024 fp32 rows,
    nprobe sweep with recall@10 against the flat scan of the same shard; iid rows of the same size as the worst case."""
    from rassengine_amd.engine import Engine
    dev = torch.device("cuda", device)
    eng = Engine(device=device, dim=1024)
    nprobes = (1, 2, 8, 32, 128)
    t0 = time.perf_counter()
    try:
        out = {"workload": f"BASELINE configs[4] per-GPU share: IVF-{args.ivf_nlist} over {args.ivf_rows} x 1024-d fp32 rows "
                           f"(100 M / 8), top-{args.k}, 32 queries per launch group, {args.ivf_queries} queries per point; "
                           "recall_at_10 = overlap with the exact flat scan of the same shard",
               "nlist": args.ivf_nlist, "training": "two-level spherical k-means (4 x nlist fine lists), 10 iterations, 1 M-row sample",
               "clustered": _ivf_one_corpus(np, torch, eng, dev, args.ivf_rows, args.ivf_nlist, args.ivf_queries, args.k,
                                            False, nprobes, int8_too=args.k <= 16)}
        out["iid_worst_case"] = _ivf_one_corpus(np, torch, eng, dev, args.ivf_rows, args.ivf_nlist, args.ivf_queries,
                                                args.k, True, nprobes)
        out["leg_s"] = round(time.perf_counter() - t0, 1)
        return out
    finally:
        eng.close()


def ivf_mode(np, torch, dist, args, world, rank, local_rank, dev, coll):
    """`--mode ivf`: BASELINE configs[4] — every GPU holds --ivf-rows clustered rows (12.5 M = 100 M / 8: at N = 8 this IS the
    100 M-row corpus; at smaller N the corpus is N x 12.5 M, because a 50 M-row shard plus its IVF copy does not fit one
    GPU), centroids trained by all ranks together (k-means sums all-reduced), every rank probes its own lists, ONE
    all-gather of per-shard top-k + merge.  A step = 1 024 queries in 32 launch groups.  Outside the timed region: recall@10
    against the FLAT sharded search of the same shards, the N > 1 self-proof, the per-batch probed bytes."""
    from rassengine_amd.dist import HipShard, ShardedSearch
    from rassengine_amd.engine import Engine
    from rassengine_amd.ivf import IvfIndex, IvfShard, train_centroids
    dim, B, k, nlist, nprobe = 1024, 32, args.k, args.ivf_nlist, args.ivf_nprobe
    rows = args.ivf_rows
    row_lo = rank * rows
    eng = Engine(device=local_rank, dim=dim)
    centres = _ivf_centres(torch, dev, dim)
    flat = eng.open_index("bench-ivf-shard", capacity_rows=rows)
    _ivf_fill(torch, flat, eng, rows, dev, False, IVF_SEED + 1 + rank, centres)
    t0 = time.perf_counter()
    cent = train_centroids(flat, nlist, train_rows=0, iters=10, seed=1)        # collective: shared centroids
    ivf = IvfIndex.build(flat, nlist=nlist, centroids=cent, dtype=args.ivf_dtype)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    shard = IvfShard(ivf, id_base=row_lo, nprobe=nprobe)        # switches the engine to torch's current stream
    search = ShardedSearch(shard)
    LPS = args.launches_per_step
    pool = _ivf_queries(torch, args.query_pool, dev, False, centres)           # same on every rank (same seed)
    n_batches = args.query_pool // B
    q_buf = torch.empty((B, dim), device=dev)

    from rassengine_amd import ops
    if LPS * B > 1024:
        raise SystemExit("--mode ivf answers a step of <= 1 024 queries per engine call")
    step_q = torch.empty((LPS * B, dim), device=dev)
    loc_s = torch.empty((LPS * B, k), dtype=torch.float32, device=dev)
    loc_i = torch.empty((LPS * B, k), dtype=torch.int64, device=dev)
    scanned_dev = torch.zeros((LPS,), dtype=torch.int64, device=dev)

    def step(i):
        # the step's 1 024 queries in ONE engine call per rank (rass_ivf_search_device_batch: one grouped coarse scan, one
        # plan launch, one fine-scan launch over every group's probed lists, one grouped merge), one broadcast, one
        # all-gather of (scores, ids) and one cross-shard merge
        if rank == 0:
            for j in range(LPS):
                g = (i * LPS + j) % n_batches
                step_q[j * B:(j + 1) * B].copy_(pool[g * B:(g + 1) * B])
        if world > 1:
            dist.broadcast(step_q, src=0)
        ivf.search_device_batch(step_q.data_ptr(), LPS * B, k, nprobe, loc_s.data_ptr(), loc_i.data_ptr(), 0, scanned_dev.data_ptr())
        ids = torch.where(loc_i >= 0, loc_i + row_lo, loc_i) if row_lo else loc_i
        if world == 1:
            return loc_s, ids
        gs = torch.empty((world, LPS * B, k), dtype=torch.float32, device=dev)
        gi = torch.empty((world, LPS * B, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gs.view(world * LPS * B, k), loc_s)
        dist.all_gather_into_tensor(gi.view(world * LPS * B, k), ids.contiguous())
        return ops.topk_merge(gs, gi)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.kernel_timing_begin(args.steps * LPS)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    scan_ms, scan_launches = eng.kernel_timing_end()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    qps = B * LPS * args.steps / elapsed
    # probed rows on THIS shard: the last step's per-group counts (the batch call reports them); one fine-scan launch per step
    per_step = float(scanned_dev.sum().item())
    per_batch = per_step / LPS
    stride = flat.row_stride
    # (nprobe <= 32: ONE fine-scan launch per step; deeper probes run group by group: LPS launches per step)
    groups_per_launch = max(1, round(LPS * args.steps / max(scan_launches, 1)))
    row_bytes = {"f32": stride * 4, "bf16": stride * 2, "int8": (stride + 511) // 512 * 512}[args.ivf_dtype]
    fine_bytes = per_step * row_bytes * groups_per_launch / LPS
    achieved = fine_bytes * scan_launches / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    rows_global = rows * world
    result = {
        "metric": "queries/sec, IVF cosine top-10 over N x 1024-d fp32 corpus in HBM (BASELINE configs[4])" +
                  ("" if args.ivf_dtype == "f32" else f" — FLAGGED: {args.ivf_dtype} slab" +
                   (" (int8 candidates + exact fp32 re-rank)" if args.ivf_dtype == "int8" else " (the bf16 scan's scores)")),
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "row_queries_per_s": round(qps * rows_global, 1), "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{rows_global} x {dim}-d IVF-{nlist} cosine top-{k}, {world} x MI355X, nprobe {nprobe} "
                               f"(BASELINE configs[4]: {rows} rows per GPU = 100 M / 8; the full 100 M at 8 GPUs)",
                   "rows_per_gpu": rows, "rows_global": rows_global, "dim": dim, "k": k, "query_batch": B, "nlist": nlist,
                   "nprobe": nprobe, "queries_per_step": B * LPS, "launch_groups_per_step": LPS, "engine_calls_per_step": 1,
                   "data": f"{IVF_CENTRES} Gaussian centres (seed {IVF_SEED}) + sigma {IVF_SIGMA} noise, normalised",
                   "cross_shard_exchange": f"{coll} all-gather" if world > 1 else None,
                   "sharding": f"row-sharded x{world}, shared centroids, {coll} all-gather merge" if world > 1 else "single shard",
                   "train_and_build_s": round(build_s, 1), "slab_dtype": args.ivf_dtype},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel": "scan_topk_f32_kernel<8, 2, 5, false> (kIvfGroups: the fine scans of a step's launch groups, one launch)"
                     if args.ivf_dtype == "f32" else "scan_i8_topk_kernel<2, 2, true> (one launch per group)" if args.ivf_dtype == "int8"
                     else "scan_bf16_topk_kernel<4, 2, false, true> (one launch per group)",
                     "bytes_per_launch": int(fine_bytes), "scanned_rows_per_batch": round(per_batch),
                     "launch_groups_per_launch": groups_per_launch,
                     "avg_launch_us": round(scan_ms / max(scan_launches, 1) * 1e3, 2), "launches": scan_launches},
    }
    # recall@10 against the exact FLAT sharded search of the same shards (all ranks take part), 256 queries
    flat_search = ShardedSearch(HipShard(flat, id_base=row_lo))
    hits = total = 0
    for b in range(0, 256, B):
        qb = pool[b:b + B].contiguous().clone()
        _, ti = flat_search.search(qb, k)
        _, gi = search.search(qb, k)
        torch.cuda.synchronize()
        th, gh = ti.cpu().numpy(), gi.cpu().numpy()
        hits += sum(len(set(gh[r]) & set(th[r])) for r in range(B))
        total += B * k
    result["recall_at_10_vs_flat_shards"] = round(hits / total, 4)
    if world > 1:
        shard.ivf.engine.set_stream(int(torch.cuda.current_stream(dev).cuda_stream))
        result.update(multi_gpu_proof(np, torch, dist, search, shard, pool, dev, rank, world, local_rank, row_lo, rows, k, B,
                                      scan_ms, scan_launches, int(fine_bytes), True))
    ivf.close()
    eng.close()
    return result


def pmc_traffic(kernel: str, bytes_per_launch: int):
    """HBM bytes per launch of the dominant kernel from the newest COMMITTED rocprofv3 PMC pass
    (`--pmc FETCH_SIZE` / `WRITE_SIZE` in runs of their own, gfx950 x2 correction; scripts/pmc_traffic.py)
    that holds THIS kernel at THIS byte count (+-10%); else null.  {"bytes": ..., "source": file}."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))):
        try:
            data = json.load(open(path))
        except Exception:
            continue
        for name, e in data.items():
            if not isinstance(e, dict):
                continue
            b = e.get("hbm_read_bytes_per_launch")
            if kernel in name and b and abs(b - bytes_per_launch) <= 0.1 * bytes_per_launch:
                best = {"bytes": round(b + e.get("hbm_write_bytes_per_launch", 0.0)),
                        "source": os.path.relpath(path, ROOT)}
    return best


def timed_path_check(np, idx, q_dev, out, n_local, k):
    """`out` = (scores, ids) the LAST search of the timed region returned for `q_dev`; checked against the oracle's fp64
    exact cosine over every row the GPU scanned (the rows are read back from the slab, 4 GB at 1 M rows)."""
    from oracle import oracle as O
    t0 = time.perf_counter()
    x = idx.get_rows(0, n_local)
    qn = O.normalize_ref(q_dev.cpu().numpy()).astype(np.float32)
    s64, i64 = O.search(x, qn, k, kind=O.KIND_F64, threads=O.usable_cpus())
    s_gpu, i_gpu = out[0].cpu().numpy(), out[1].cpu().numpy()
    nq = qn.shape[0]
    recall = float(np.mean([len(set(i_gpu[q]) & set(i64[q])) / k for q in range(nq)]))
    return {"recall_at_k_timed_path": recall,
            "timed_path_check": {"queries": nq, "rows": n_local, "ids_identical": bool(np.array_equal(i_gpu, i64)),
                                 "max_abs_cosine_err": float(np.abs(s_gpu.astype(np.float64) - s64).max()),
                                 "what": "the last launch group of the timed loop vs oracle fp64 exact cosine over all rows",
                                 "seconds": round(time.perf_counter() - t0, 1)}}


def cpu_baseline_and_recall(np, torch, eng, idx, pool, args, n_local, dim, B, k):
    """Rank 0, N=1 only.  (1) recall@k of the GPU path vs the oracle's fp64 ranking on a row
    prefix of the SAME corpus; (2) the CPU stand-in for the reference's OpenSearch k-NN lookup
    (oracle = exact cosine scan, fp32, OpenMP on all host cores) timed on that bounded sample
    and scaled linearly in rows to the full workload."""
    from oracle import oracle as O
    from rassengine_amd import _native as N

    sample = min(args.cpu_sample_rows, n_local)
    x = idx.get_rows(0, sample)                      # the very rows the GPU scans
    q_raw = pool[:B].cpu().numpy()
    qn = O.normalize_ref(q_raw).astype(np.float32)

    # GPU over the same prefix of the slab (zero copy), through the stateless C-ABI launcher
    L = N.lib()
    dev = pool.device
    ws = torch.empty(int(L.rass_scan_workspace_bytes(B, k)), dtype=torch.uint8, device=dev)
    out_s = torch.empty((B, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((B, k), dtype=torch.int64, device=dev)
    q_dev = pool[:B].contiguous()
    stream = int(torch.cuda.current_stream().cuda_stream)
    N.check("rass_scan_topk_f32", L.rass_scan_topk_f32(
        ctypes.c_void_p(idx.device_rows_ptr), sample, dim, idx.row_stride, None, ctypes.c_void_p(q_dev.data_ptr()), B,
        None, k, 0, ctypes.c_void_p(out_s.data_ptr()), ctypes.c_void_p(out_i.data_ptr()),
        ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(stream)))
    torch.cuda.synchronize()
    ids_gpu = out_i.cpu().numpy()
    s64, i64 = O.search(x, qn, k, kind=O.KIND_F64)
    recall = float(np.mean([len(set(ids_gpu[q]) & set(i64[q])) / k for q in range(B)]))
    max_dcos = float(np.abs(out_s.cpu().numpy().astype(np.float64) - s64).max())

    cores = O.usable_cpus()   # affinity mask capped by the cgroup quota, not the host's logical CPU count
    O.search(x[:20000], qn, k, kind=O.KIND_F32_BLOCKED, threads=cores)  # warm the thread pool
    t0 = time.perf_counter()
    reps = 0
    while True:
        O.search(x, qn, k, kind=O.KIND_F32_BLOCKED, threads=cores)
        reps += 1
        el = time.perf_counter() - t0
        if el >= 10.0 or reps >= 5000:
            break
    qps_sample = B * reps / el
    qps_full = qps_sample * sample / n_local  # brute force is linear in rows
    gflops = 2.0 * sample * dim * B * reps / el / 1e9

    # The reference's ALGORITHM on the same cores: HNSW with its parameters (m 48, ef_construction 400,
    # app/main.py:563-572; ef_search = the k-NN plugin's default 512), restated in oracle/hnsw.c.  Built on a
    # bounded prefix of the same corpus (the build is single-threaded, ~1 ms per row); its queries/s are those
    # of THAT sample — graph search is ~log N per query, so nothing is extrapolated — and its recall@k is
    # measured against the exact top-k of the same sample.
    hs = min(args.cpu_hnsw_rows, sample)
    hnsw = None
    if hs >= 1000:
        t0 = time.perf_counter()
        h = O.Hnsw(x[:hs])
        build_s = time.perf_counter() - t0
        nqh = 8 * B
        qh = O.normalize_ref(pool[:nqh].cpu().numpy()).astype(np.float32)
        h.search(qh[:B], k, O.Hnsw.EF_SEARCH, threads=cores)
        t0 = time.perf_counter()
        _, ids_h, evals = h.search(qh, k, O.Hnsw.EF_SEARCH, threads=cores)
        t_h = time.perf_counter() - t0
        _, truth = O.search(x[:hs], qh, k, kind=O.KIND_F64, threads=cores)
        hnsw = {"value": round(nqh / t_h, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                "recall_at_k": round(float(np.mean([len(set(ids_h[r]) & set(truth[r])) / k for r in range(nqh)])), 4),
                "m": O.Hnsw.M, "ef_construction": O.Hnsw.EF_CONSTRUCTION, "ef_search": O.Hnsw.EF_SEARCH,
                "sample_rows": hs, "build_s": round(build_s, 1),
                "distance_evals_per_query": round(evals / nqh, 1),
                "sample": f"oracle/hnsw.c (HNSW restatement with the reference's parameters), built on the first {hs} "
                          f"rows of the same corpus in {build_s:.1f} s (1 thread), {nqh} queries on {cores} threads; "
                          "queries/s and recall are of that sample, not extrapolated"}
        h.close()
    return {
        "recall_at_k": recall,
        "max_abs_cosine_err": max_dcos,
        "cpu_baseline": {
            "value": round(qps_full, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"oracle exact fp32 cosine top-{k}, register-blocked over 8 queries (AVX2, OpenMP, {cores} threads "
                      f"= the CPUs this job may use of {os.cpu_count()} logical), batch {B}, timed on the first "
                      f"{sample} rows of the same corpus for {el:.1f} s ({reps} passes, {gflops:.0f} GFLOP/s), scaled "
                      f"x{sample}/{n_local} to {n_local} rows; stand-in for OpenSearch k-NN (reference stack absent, "
                      "BASELINE.md s2)",
            "measured_qps_on_sample": round(qps_sample, 2), "sample_rows": sample, "gflops": round(gflops, 1),
            "hnsw": hnsw,
        },
    }


if __name__ == "__main__":
    main()

"""bench.py's JSON line, end to end on the GPU box (VERDICT r2 #2a, #4, #5): the N = 1 line carries `roofline`,
`cpu_baseline` and the `ingest` leg (BASELINE configs[2]'s encoder forward at batch 256 x 512); an N > 1 launch — two
ranks sharing the test GPU over gloo, the rehearsal mode — defaults to BASELINE configs[3] (the fixed 10 M-row corpus,
strong scaling) and proves in the line itself who took part and that the merged result is the merge of the shards."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_cpu_baseline_and_ingest(gpu):
    j = _line([sys.executable, "bench.py", "--steps", "3", "--warmup", "2", "--cpu-sample-rows", "50000",
               "--cpu-hnsw-rows", "2000", "--ingest-batches", "2", "--ivf-rows", "400000", "--ivf-nlist", "256",
               "--ivf-queries", "256"])
    assert j["n_gpus"] == 1 and j["scaling"] == "weak" and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert "configs[1]" in j["config"]["workload"] and j["config"]["rows_global"] == 1_000_000
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.5 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["bytes_per_launch"] == 1_000_000 * 1024 * 4
    assert j["recall_at_k"] == 1.0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    # configs[3]'s 10 M-row corpus on this one GPU: the same-workload reference of the --gpus N > 1 (strong scaling) lines
    ref = j["strong_scaling_reference"]
    assert "error" not in ref and "10000000 x 1024" in ref["workload"] and ref["roofline"]["bytes_per_launch"] == 10_000_000 * 4096
    assert 0.05 * j["value"] < ref["value"] < 0.2 * j["value"]           # ten times the rows per query
    ing = j["ingest"]
    assert ing["roofline"]["bound"] == "mfma_bf16" and ing["roofline"]["peak"] == 2500.0
    assert ing["last_batch_rows_finite_and_unit_norm"] is True and ing["rows_in_index"] == 4 * 256
    assert ing["chunks_per_s"] > 1500 and 0.2 < ing["roofline"]["frac"] < 1.0
    assert abs(ing["tflops"] - ing["flop_per_batch"] / (ing["ms_per_batch"] * 1e-3) / 1e12) < 1.0
    # the IVF leg (BASELINE configs[4] at one GPU's share; a small share here): nprobe sweep with recall against the flat
    # scan of the same shard and probed bytes / time against the HBM peak, clustered rows + iid rows as the worst case
    ivf = j["ivf"]
    assert "error" not in ivf and ivf["nlist"] == 256
    i8 = ivf["clustered"]["int8_slab_flagged"]["sweep"]                   # flagged: int8 candidates + exact re-rank on the same lists
    assert [p["nprobe"] for p in i8] == [1, 2, 8, 32, 128] and all(p["batch_equals_group_by_group"] for p in i8)
    assert all(abs(a["recall_at_10"] - b["recall_at_10"]) <= 0.02 for a, b in zip(i8, ivf["clustered"]["sweep"]))
    for key in ("clustered", "iid_worst_case"):
        c = ivf[key]
        assert c["rows"] == 400000 and [p["nprobe"] for p in c["sweep"]] == [1, 2, 8, 32, 128]
        for p in c["sweep"]:
            rf = p["roofline"]
            assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1.0 and p["queries_per_s"] > 0
            assert abs(rf["bytes_per_batch"] - (p["scanned_rows_per_batch"] * 4096 + 256 * 4096)) <= 4096     # (rounded mean)
            assert p["batch_equals_group_by_group"] is True and p["queries_per_s_group_by_group"] > 0
        rec = [p["recall_at_10"] for p in c["sweep"]]
        assert all(b >= a - 0.02 for a, b in zip(rec, rec[1:])), rec
        assert rec[-1] >= (0.95 if key == "clustered" else 0.3), rec        # iid rows have no lists worth probing: the worst case
    # cluster structure is what an IVF finds: at every nprobe the clustered corpus recalls more than the iid one
    assert all(c["recall_at_10"] > w["recall_at_10"] for c, w in zip(ivf["clustered"]["sweep"], ivf["iid_worst_case"]["sweep"]))
    assert ivf["clustered"]["sweep"][0]["queries_per_s"] > ivf["clustered"]["flat_queries_per_s_same_shard"]


def test_two_rank_line_defaults_to_configs3_and_proves_itself(gpu):
    port = 29500 + os.getpid() % 400
    j = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"],
              env={"RASS_BENCH_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "configs[3]" in j["config"]["workload"]
    assert j["config"]["rows_global"] == 10_000_000 and j["config"]["rows_per_gpu"] == 5_000_000
    assert j["ranks_seen"] == 2 and j["backend"] == "gloo"          # the rehearsal backend; a real run says nccl
    assert [r["rank"] for r in j["ranks"]] == [0, 1] and sum(r["rows_held"] for r in j["ranks"]) == 10_000_000
    assert j["ranks"][1]["row_lo"] == 5_000_000 and all(r["launches"] == 2 * 32 for r in j["ranks"])
    assert j["sharded_equals_merge_of_shards"] is True
    assert j["roofline_per_rank"]["avg_launch_us_min"] > 0
    assert "ingest" not in j and "cpu_baseline" not in j                # rank 0 at N = 1 only


def test_two_rank_ivf_mode_line(gpu):
    """`--mode ivf` (BASELINE configs[4]): shared centroids trained by both ranks, per-shard probes, one all-gather; the line
    proves who took part, that the merged result is the merge of the shards' lists, and reports recall against the FLAT
    sharded search of the same shards."""
    port = 29900 + os.getpid() % 90
    j = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--mode", "ivf", "--steps", "2",
               "--warmup", "1", "--ivf-rows", "300000", "--ivf-nlist", "128", "--ivf-nprobe", "128", "--launches-per-step", "8"],
              env={"RASS_BENCH_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "configs[4]" in j["config"]["workload"]
    assert j["config"]["rows_global"] == 600000 and j["config"]["nlist"] == 128 and j["config"]["nprobe"] == 128
    assert j["ranks_seen"] == 2 and [r["rank"] for r in j["ranks"]] == [0, 1] and j["ranks"][1]["row_lo"] == 300000
    assert j["sharded_equals_merge_of_shards"] is True
    assert j["recall_at_10_vs_flat_shards"] == 1.0       # every list probed on every shard == the flat sharded search
    rf = j["roofline"]
    # nprobe 128 > 32 runs group by group (the threshold path): 8 fine-scan launches per step; nprobe <= 32: one per step
    assert rf["bound"] == "hbm" and rf["launches"] == 2 * 8 and rf["launch_groups_per_launch"] == 1 and 0 < rf["frac"] < 1.0
    assert abs(rf["bytes_per_launch"] - rf["scanned_rows_per_batch"] * 4096) <= 4096
    j8 = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(port + 1), "bench.py", "--gpus", "2", "--mode", "ivf", "--steps", "2",
                "--warmup", "1", "--ivf-rows", "300000", "--ivf-nlist", "128", "--ivf-nprobe", "8", "--launches-per-step", "8"],
               env={"RASS_BENCH_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    rf8 = j8["roofline"]
    assert rf8["launches"] == 2 and rf8["launch_groups_per_launch"] == 8 and j8["sharded_equals_merge_of_shards"] is True
    assert abs(rf8["bytes_per_launch"] - 8 * rf8["scanned_rows_per_batch"] * 4096) <= 8 * 4096
    assert 0.3 < j8["recall_at_10_vs_flat_shards"] <= 1.0


def test_two_rank_ivf_mode_line_with_an_int8_slab(gpu):
    """`--mode ivf --ivf-dtype int8` (flagged): every rank's IVF keeps the int8 copy of its slab; int8 candidates + exact re-rank
    per shard, the same all-gather + merge; every list probed on every shard ≡ the flat sharded search."""
    port = 29700 + os.getpid() % 90
    j = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--mode", "ivf", "--ivf-dtype", "int8", "--steps", "2",
               "--warmup", "1", "--ivf-rows", "300000", "--ivf-nlist", "128", "--ivf-nprobe", "128", "--launches-per-step", "8"],
              env={"RASS_BENCH_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert j["n_gpus"] == 2 and "FLAGGED" in j["metric"] and j["config"]["slab_dtype"] == "int8"
    assert j["ranks_seen"] == 2 and j["sharded_equals_merge_of_shards"] is True
    assert j["recall_at_10_vs_flat_shards"] == 1.0
    rf = j["roofline"]
    assert "scan_i8_topk_kernel" in rf["kernel"] and abs(rf["bytes_per_launch"] - rf["scanned_rows_per_batch"] * 1024) <= 1024

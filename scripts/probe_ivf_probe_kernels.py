"""Kernel-level view of one IVF probe (for rocprofv3 --kernel-trace): IVF-4096 over clustered rows, 200 batches of 32 queries at
one nprobe; prints us per batch.  Usage: rocprofv3 --kernel-trace --stats ... -- python3 scripts/probe_ivf_probe_kernels.py --nprobe 2"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rassengine_amd.engine import Engine, HipTimer
from rassengine_amd.ivf import IvfIndex, train_centroids

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--nprobe", type=int, default=2)
ap.add_argument("--slab", default="f32")
ap.add_argument("--batches", type=int, default=200)
a = ap.parse_args()
dim, dev = 1024, torch.device("cuda", 0)
eng = Engine(0, dim)
flat = eng.open_index("p", capacity_rows=a.rows)
g = torch.Generator(device=dev); g.manual_seed(7)
centres = torch.randn((8192, dim), generator=g, device=dev); centres /= centres.norm(dim=1, keepdim=True)
for lo in range(0, a.rows, 262144):
    n = min(262144, a.rows - lo)
    lab = torch.randint(0, 8192, (n,), generator=g, device=dev)
    x = centres[lab] + torch.randn((n, dim), generator=g, device=dev) / dim ** 0.5
    torch.cuda.synchronize(); flat.add_device(x.data_ptr(), n, normalize=True); eng.synchronize()
cent = train_centroids(flat, 4096, train_rows=1_000_000, iters=8, seed=1)
ivf = IvfIndex.build(flat, nlist=4096, centroids=cent, dtype=a.slab)
lab = torch.randint(0, 8192, (32 * 8,), generator=g, device=dev)
q = (centres[lab] + torch.randn((32 * 8, dim), generator=g, device=dev) / dim ** 0.5).contiguous()
out_s = torch.empty((32, 10), device=dev); out_i = torch.empty((32, 10), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for b in range(8):
    ivf.search_device(q[32 * b:32 * b + 32].data_ptr(), 32, 10, a.nprobe, out_s.data_ptr(), out_i.data_ptr())
eng.synchronize()
tm = HipTimer(); tm.start(eng.stream)
for b in range(a.batches):
    ivf.search_device(q[32 * (b % 8):32 * (b % 8) + 32].data_ptr(), 32, 10, a.nprobe, out_s.data_ptr(), out_i.data_ptr())
tm.stop(eng.stream)
print(f"IVF-4096, {a.rows} rows, {a.slab} slab, nprobe {a.nprobe}: {tm.elapsed_ms() / a.batches * 1e3:.1f} us per 32-query batch", flush=True)

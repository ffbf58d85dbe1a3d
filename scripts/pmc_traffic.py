"""Summarise a `rocprofv3 --pmc FETCH_SIZE [WRITE_SIZE]` counter CSV into
profiles/<name>.json: per kernel, mean counter value per launch and the HBM bytes with the
gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE is in KiB and reports exactly 1/2 of
a wide coalesced streaming read; WRITE_SIZE is exact for 16-B-per-lane stores).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv \
        gpurun_out/pmc_write/*/*_counter_collection.csv profiles/r02_pmc_traffic.json

(one counter per run, as the guide prescribes; any number of counter CSVs, the last argument is the output)
"""
import collections
import csv
import json
import sys


def main(srcs, dst, how=None):
    agg = collections.defaultdict(list)
    for src in srcs:
        for r in csv.DictReader(open(src)):
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = {}
    for (kernel, counter), vals in agg.items():
        mean = sum(vals) / len(vals)
        e = out.setdefault(kernel, {})
        e["launches_" + counter] = len(vals)
        e[counter + "_KiB_mean"] = mean
        if counter == "FETCH_SIZE":
            e["hbm_read_bytes_per_launch"] = mean * 1024 * 2  # x2: gfx950 wide-coalesced-read correction
        if counter == "WRITE_SIZE":
            e["hbm_write_bytes_per_launch"] = mean * 1024
    if how:
        out["_how"] = how
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        if "scan_topk" in k or "scan_bf16" in k:
            print(k, v)


if __name__ == "__main__":
    main(sys.argv[1:-1], sys.argv[-1],
         how="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in runs of their own (python3 bench.py --steps 4 --warmup 1 "
             "--no-cpu-baseline), summarised by scripts/pmc_traffic.py; FETCH_SIZE is KiB x2 (gfx950 wide coalesced "
             "read correction), WRITE_SIZE KiB x1")

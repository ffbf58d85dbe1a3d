"""Summarise a `rocprofv3 --pmc FETCH_SIZE [WRITE_SIZE]` counter CSV into
profiles/<name>.json: per kernel, mean counter value per launch and the HBM bytes with the
gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE is in KiB and reports exactly 1/2 of
a wide coalesced streaming read; WRITE_SIZE is exact for 16-B-per-lane stores).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/*/*_counter_collection.csv profiles/r01_pmc_traffic.json
"""
import collections
import csv
import json
import sys


def main(src, dst):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(src)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = {}
    for (kernel, counter), vals in agg.items():
        mean = sum(vals) / len(vals)
        e = out.setdefault(kernel, {"launches": len(vals)})
        e[counter + "_KiB_mean"] = mean
        if counter == "FETCH_SIZE":
            e["hbm_read_bytes_per_launch"] = mean * 1024 * 2  # x2: gfx950 wide-coalesced-read correction
        if counter == "WRITE_SIZE":
            e["hbm_write_bytes_per_launch"] = mean * 1024
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        if "scan_topk" in k:
            print(k, v)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

"""Summarise rocprofv3 --pmc SQ counter CSVs (one or more passes) for one kernel into profiles/<name>.json: the mean
counter value per launch plus two derived fractions (MFMA pipe busy, waves stalled on an instruction).

    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out1 -- python3 bench.py ...
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d out2 -- python3 bench.py ...
    python scripts/pmc_sq_summary.py "scan_topk_f32_kernel<8, 2, 0, false>" out1/*/*counter_collection.csv out2/... profiles/x.json
"""
import collections
import csv
import json
import sys


def main(kernel, srcs, dst):
    agg = collections.defaultdict(list)
    for src in srcs:
        for r in csv.DictReader(open(src)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in agg.items()}
    out["_launches_per_counter"] = {k: len(v) for k, v in agg.items()}
    d = {}
    if "GRBM_GUI_ACTIVE" in out and "SQ_VALU_MFMA_BUSY_CYCLES" in out:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs / 8 XCDs = 128 SIMDs per XCD
        d["simd_cycles_per_launch"] = out["GRBM_GUI_ACTIVE"] * 128
        d["mfma_busy_frac"] = out["SQ_VALU_MFMA_BUSY_CYCLES"] / d["simd_cycles_per_launch"]
    if "SQ_WAVE_CYCLES" in out and "SQ_WAIT_INST_ANY" in out:
        d["wait_inst_any_frac_of_wave_cycles"] = out["SQ_WAIT_INST_ANY"] / out["SQ_WAVE_CYCLES"]
    if "SQ_WAVE_CYCLES" in out and "SQ_WAIT_ANY" in out:
        d["wait_any_frac_of_wave_cycles"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    d["note"] = f"per launch of {kernel}; GRBM_GUI_ACTIVE is summed over 8 XCDs"
    out["_derived"] = d
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:-1], sys.argv[-1])

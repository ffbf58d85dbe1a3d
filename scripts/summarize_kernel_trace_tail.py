"""Per-kernel mean duration over the LAST n dispatches of a rocprofv3 --kernel-trace CSV (the timed loop of a probe script,
after its set-up): python scripts/summarize_kernel_trace_tail.py <kernel_trace.csv> <n_dispatches>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tail = rows[-int(sys.argv[2]):]
agg = collections.defaultdict(list)
for r in tail:
    agg[r["Kernel_Name"][:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("  %5d x %8.1f us  %s" % (len(v), sum(v) / len(v), k))
    tot += sum(v)
t0, t1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
print("  kernel time %.0f us over a span of %.0f us (%d dispatches)" % (tot, (t1 - t0) / 1e3, len(tail)))

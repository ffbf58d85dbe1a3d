# One-off tuning aid: sweep RASS_SCAN_XCD_SKEW with the real bench data.
set -e
for B in 16 1; do
for s in 0 2 3 4 5; do
  echo "== B=$B skew $s"; RASS_SCAN_XCD_SKEW=$s timeout -k 10 120 python bench.py --no-cpu-baseline --batch $B --steps 300 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
done

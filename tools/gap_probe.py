"""Kernel durations and inter-kernel gaps of ONE small MSM from a rocprofv3 --kernel-trace csv (argv[1] = directory): what a
hipGraph could and could not remove."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f))]
rows.sort()
# the last MSM: kernels after the last gap > 200 us
start = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 200000:
        start = i
run = rows[start:]
busy = sum(e - s for s, e, _ in run)
gaps = [run[i][0] - run[i - 1][1] for i in range(1, len(run))]
print("kernels %d  span %.1f us  busy %.1f us  gaps total %.1f us  mean gap %.2f us  max gap %.1f us" % (
    len(run), (run[-1][1] - run[0][0]) / 1e3, busy / 1e3, sum(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3))
by = {}
for s, e, n in run:
    d = by.setdefault(n, [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-42s x%-3d %8.1f us  (%.1f us each)" % (n, c, t, t / c))
big = sorted(((run[i][0] - run[i - 1][1]) / 1e3, run[i - 1][2], run[i][2]) for i in range(1, len(run)))[-4:]
for g, a, b in reversed(big):
    print("  gap %.1f us between %s -> %s" % (g, a, b))

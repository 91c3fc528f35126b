#!/usr/bin/env python3
"""Timeline of one default step from a rocprofv3 --kernel-trace CSV: every kernel's start relative to the step's
first kernel, its duration and the gap in front of it, averaged over the steps found in the trace."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
steps, cur = [], []
for s, e, name in rows:
    if name.startswith("papr_estimate_kernel") and cur:
        steps.append(cur)
        cur = []
    if name.startswith("papr_estimate_kernel") or cur:
        cur.append((s, e, name))
if cur:
    steps.append(cur)
steps = [st for st in steps if len(st) == len(steps[len(steps) // 2])]
if len(steps) < 8:
    sys.exit("no steps found")
# bench.py runs the default table first and the 0.1 dB table second: report the two ends of the trace separately
k = min(15, len(steps) // 2 - 2)
for label, part in (("first %d steps (after 2)" % k, steps[2:2 + k]), ("last %d steps" % k, steps[-k:])):
  steps_all, steps = steps, part
  n = len(steps)
  print("%s, %d kernels each" % (label, len(steps[0])))
  acc = collections.OrderedDict()
  for st in steps:
      t0 = st[0][0]
      for k, (s, e, name) in enumerate(st):
          a = acc.setdefault((k, name[:48]), [0.0, 0.0, 0.0])
          a[0] += (s - t0) / 1e3
          a[1] += (e - s) / 1e3
          a[2] += (s - st[k - 1][1]) / 1e3 if k else 0.0
  for (k, name), a in acc.items():
      print("  %2d %-48s start %9.1f us  dur %8.1f us  gap before %6.1f us" % (k, name, a[0] / n, a[1] / n, a[2] / n))
  span = sum(st[-1][1] - st[0][0] for st in steps) / n / 1e3
  period = (steps[-1][0][0] - steps[0][0][0]) / (n - 1) / 1e3 if n > 1 else 0.0
  print("  first kernel start -> last kernel end: %.1f us; step period %.1f us" % (span, period))
  steps = steps_all

#!/usr/bin/env python3
"""Link utilisation of a traced StreamEstimator run (tools/r03_trace_stream.sh): large host-to-device copies of the second
(warm) run, the gaps between them and what ran inside the gaps."""
import csv
import glob
import re
import sys

root, label = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
cop, ker = [], []
for f in glob.glob(root + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]))
for f in glob.glob(root + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
big = sorted(c for c in cop if "HOST_TO_DEVICE" in c[2] and c[1] - c[0] > 300000)
big = big[len(big) // 2:]
t0, t1 = big[0][0], big[-1][1]
busy = sum(e - s for s, e, _ in big)
end = max(k[1] for k in ker if k[0] >= t0)
print("%s: %d copies, first copy -> last copy %.2f ms, link busy %.2f ms (%.0f%%), mean copy %.3f ms; last kernel ends %.2f ms after the last copy" % (
    label, len(big), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), busy / len(big) / 1e6, (end - t1) / 1e6))
gaps = []
for (s0, e0, _), (s1, e1, _) in zip(big, big[1:]):
    names = {}
    for k in ker:
        if k[0] < s1 and k[1] > e0:
            n = re.search(r"(k_\w+|__amd\w+)", k[2])
            n = n.group(1) if n else k[2][:20]
            names[n] = names.get(n, 0) + (min(k[1], s1) - max(k[0], e0)) / 1e3
    gaps.append(((s1 - e0) / 1e3, names))
print("   gaps (us):", " ".join("%.0f" % g for g, _ in gaps))
for g, names in gaps:
    if g > 150:
        print("   gap %.0f us holds" % g, {k: round(v) for k, v in sorted(names.items(), key=lambda x: -x[1])[:5]})

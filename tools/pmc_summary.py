#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs of one gpurun_out/<tag>/ directory into a small text file
(per kernel: mean counter value per dispatch).  Usage: tools/pmc_summary.py gpurun_out/<tag> [kernel-substring]"""
import collections
import csv
import re
import glob
import sys

def kname(full):
    m = re.search(r"(k_\w+(<[^>]*>)?)", full)
    return m.group(1) if m else full[:50]


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else ""
print("# source: %s (rocprofv3 --pmc <counters> --kernel-trace, separate passes)" % root)
try:
    import bench
    print("# kernel_source_sha: %s" % bench.kernel_source_sha())       # bench.py emits these counters only for the same kernels
    cfg = os.path.basename(os.path.normpath(root))                      # gpurun_out/<tag>/<config>
    if cfg in bench.CONFIG_SOURCES:
        print("# config_source_sha: %s" % bench.kernel_source_sha(cfg))  # ... or for the same files of this config's kernels
except Exception as e:      # noqa: BLE001
    print("# kernel_source_sha: unknown (%r)" % (e,))
# gpurun merges a call's files INTO the local gpurun_out/: a pass directory re-used by a later profile run holds the CSVs of
# the earlier runs (other builds) as well -- only the newest CSV of each pass directory is this build's
newest = {}
for path in glob.glob(root + "/pmc_*/*/*_counter_collection.csv"):
    d = os.path.dirname(path)
    if d not in newest or os.path.getmtime(path) > os.path.getmtime(newest[d]):
        newest[d] = path
for path in sorted(newest.values()):
    agg = collections.defaultdict(list)
    bygrid = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            agg[(kname(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
            bygrid[(kname(r["Kernel_Name"]), r["Counter_Name"])][r["Grid_Size"]].append(float(r["Counter_Value"]))
            meta[kname(r["Kernel_Name"])] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"])
    for (k, c), v in sorted(agg.items()):
        print("%-50s %-24s dispatches=%-3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
    # a kernel launched with several grid sizes in one step (the level-1 and level-2 searches of a GME step): one more line
    # per grid size, so that a launch can be rated on its own ("name[grid=N]")
    for (k, c), by in sorted(bygrid.items()):
        if len(by) > 1:
            for g, v in sorted(by.items(), key=lambda x: int(x[0])):
                print("%-50s %-24s dispatches=%-3d mean=%.6g" % ("%s[grid=%s]" % (k, g), c, len(v), sum(v) / len(v)))
    for k, m in meta.items():
        print("#   %s grid=%s wg=%s lds=%s vgpr=%s sgpr=%s" % ((k,) + m))

#!/usr/bin/env python3
"""Per-kernel resources as the compiler reports them (-Rpass-analysis=kernel-resource-usage, written to
global-motion-estimation_amd/csrc/build/*.remarks by the Makefile): VGPRs, VGPR / SGPR spills, scratch bytes per lane,
waves per SIMD, LDS.  `python tools/resource_table.py` prints the markdown table DESIGN.md carries between its
<!-- resources --> markers; `--check` exits non-zero when DESIGN.md's table differs from the build's.
usage: python tools/resource_table.py [--check] [--all]"""
import glob
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(REPO, "global-motion-estimation_amd", "csrc", "build")
FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "waves", "LDS Size [bytes/block]": "lds"}
# the instances bench.py's configs launch (launch plans of DESIGN.md section 5), in the table's order
BENCHED = ["k_exh_sea16p<3, 5, 36>", "k_exh_sea16p<5, 7, 38>", "k_exh_sea16p_mse<3, 5, 36>", "k_exh_sea16p_mse<5, 7, 38>",
           "k_exh_redo16<3, false>", "k_exh_redo16<3, true>", "k_exh_redo16<5, false>", "k_exh_redo16<5, true>",
           "k_exh_mfma16<3, 3, 4>", "k_exh_mfma16<5, 2, 4>", "k_exh_qsad16<3>", "k_exh_dot16<3>", "k_sqbox16",
           "k_walk16<0>", "k_walk16<1>", "k_walk16s<1, 1, true>", "k_walk16s<1, 2, true>", "k_dense2<1>",
           "k_fit_level", "k_compensate16", "k_pyrdown_lds"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    clean = []
    for n in out:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)
        clean.append(n)
    return clean


def kernels():
    rows = {}
    for path in sorted(glob.glob(os.path.join(BUILD, "*.remarks"))):
        cur = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"mangled": m.group(1), "file": os.path.basename(path).replace(".remarks", ".hip")}
                rows[m.group(1)] = cur
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?): (\d+)", line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    rows = list(rows.values())
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = n
    return rows


def table(rows, only_benched=True):
    by = {}
    for r in rows:
        by.setdefault(r["name"], r)
    names = [n for n in BENCHED if any(k == n or k.startswith(n + "<") or k.startswith(n) and n in ("k_fit_level", "k_compensate16", "k_pyrdown_lds", "k_sqbox16") for k in by)] if only_benched else sorted(by)
    lines = ["| kernel instance | file | VGPRs | AGPRs | VGPR spills | SGPR spills | scratch B/lane | waves/SIMD |", "|---|---|---|---|---|---|---|---|"]
    for n in names:
        cands = [k for k in by if k == n or (n in ("k_fit_level", "k_compensate16", "k_pyrdown_lds", "k_sqbox16") and k.startswith(n))]
        for k in sorted(cands):
            r = by[k]
            lines.append("| `%s` | %s | %d | %d | %d | %d | %d | %d |" % (k, r["file"], r.get("vgpr", -1), r.get("agpr", 0), r.get("vgpr_spill", -1),
                                                                       r.get("sgpr_spill", -1), r.get("scratch", -1), r.get("waves", -1)))
    return "\n".join(lines)


def main():
    rows = kernels()
    if not rows:
        sys.exit("no build/*.remarks: run make -C global-motion-estimation_amd/csrc first")
    t = table(rows, "--all" not in sys.argv)
    if "--check" in sys.argv:
        doc = open(os.path.join(REPO, "DESIGN.md")).read()
        m = re.search(r"<!-- resources -->\n(.*?)\n<!-- /resources -->", doc, re.S)
        if not m or m.group(1).strip() != t.strip():
            sys.exit("DESIGN.md's resource table differs from the build's remarks: python tools/resource_table.py and paste it between the markers")
        return
    print(t)


if __name__ == "__main__":
    main()

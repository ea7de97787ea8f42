#!/usr/bin/env python3
"""DESIGN.md section 5's table from the bench lines of a regeneration: python tools/design_table.py <tag> [previous tag]
reads gpurun_out/<tag>/<tag>_<config>_bench.json (falling back to profiles/) and profiles/<previous>_<config>_bench.json."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_final"
prev = sys.argv[2] if len(sys.argv) > 2 else "r03_final"
ROWS = [("exh720", "default", "720×480 bs16 sw16 exhaustive MAE (configs[1])"), ("exh720_brute", None, "same, brute force (`GME_EXH_BRUTE=1`)"),
        ("exh720_pan240x2", None, "same on pan240 ×2 (51 real frames, 640×480)"), ("exh720_pan240x2_noc2", None, "same, phase C2 off (`GME_SEA_QUOTA=0`)"),
        ("exh720_pan240seq", None, "same on pan240 (51 real frames, 320×240)"), ("exh720_race", None, "same on `race` (2 real frames)"),
        ("exh720_flat", None, "same on flat frames"), ("exh720_noise", None, "same on uniform noise (→ redo kernel)"),
        ("exh720mse", None, "720×480 exhaustive MSE (matrix cores)"), ("exh720mse_pan240x2", None, "same on pan240 ×2"), ("exh720mse_noise", None, "same on uniform noise"),
        ("exh720mse_vec", None, "720×480 exhaustive MSE on the vector unit (`GME_EXH_MFMA=0`)"), ("exh720mse_vec_pan240x2", None, "same on pan240 ×2"),
        ("exh720mse_vec_noise", None, "same on uniform noise"),
        ("tss720", None, "720×480 three-step MSE"), ("tdl720", None, "720×480 2-D log MSE"), ("dia720", None, "720×480 diamond MAE"),
        ("dia720mse", None, "720×480 diamond MSE"), ("tss_bs4sw2", None, "720×480 three-step bs 4 sw 2 MSE (the reference's default call)"),
        ("gme720", None, "720×480 full GME + compensate + PSNR (configs[2]), 4 ranges"), ("gme720_1stream", None, "same, one stream, blocking calls"),
        ("gme720dev", None, "same, opt-in device solve, 2 ranges"), ("gme720_1stream_devsolve", None, "same, opt-in device solve, one stream"),
        ("gme_pan240_bs12fd5", None, "320×240 real frames, GME at bs 12 / fd 5 (the slides' setting)"),
        ("exh1080", None, "1920×1080 exhaustive MAE sw 32"), ("exh1080mse", None, "1920×1080 exhaustive MSE sw 32 (configs[3] BBME)"),
        ("exh1080mse_mfma", None, "same on the matrix cores (opt-in, `GME_EXH_MFMA=1`)"),
        ("gme1080exh", None, "1920×1080 exhaustive-MSE GME + compensate (configs[3])"), ("gme1080", None, "1920×1080 diamond GME + compensate"),
        ("seq1080", None, "2000-frame 1080p sequence, diamond GME + gather (configs[4], 1 rank)")]


def load(t, name):
    for d in (os.path.join(REPO, "gpurun_out", t), os.path.join(REPO, "profiles")):
        p = os.path.join(d, "%s_%s_bench.json" % (t, name))
        if os.path.exists(p) and os.path.getsize(p):
            try:
                return json.loads(open(p).read().strip().splitlines()[-1])
            except Exception:
                pass
    return None


print("| config (1× MI355X, 2048 pairs per step unless the config says otherwise) | kernel of the launch plan | pairs/s round 3 → round 4 | ms/step | HBM fraction of the dominant kernel | CPU baseline (1 core, NumPy oracle) | parity (pairs vs C oracle) |")
print("|---|---|---|---|---|---|---|")
for name, alias, label in ROWS:
    d = load(tag, alias or name) or load(tag, name)
    if d is None:
        continue
    o = load(prev, alias or name) or load(prev, name)
    kern = d["roofline"]["kernel"].split(" tiles")[0].split(" grid")[0].split(" (")[0].split(" 1x")[0]
    cpu = d.get("cpu_baseline", {}).get("value")
    e = d.get("elimination")
    extra = " (scored %.2f %%, first UB left %.2f %%)" % (100 * e["surviving_fraction"], 100 * e.get("listed_fraction_before_ordered_rounds", e["surviving_fraction"])) if e else ""
    r = d["roofline"]
    frac = "%.3f" % r.get("hbm", r)["frac"] + (" (int8 MFMA %.3f algorithmic, %.3f issued)" % (r["frac"], r["issued_frac"]) if r["bound"] == "mfma" else "")
    print("| %s | `%s` | %s → **%s**%s | %.3f | %s | %s | %d / %d |" % (
        label, kern, ("%d" % round(o["value"])) if o else "—", "{:,}".format(round(d["value"])).replace(",", " "), extra, d["ms_per_step"],
        frac, ("%.4g" % cpu) if cpu else "", d["parity"]["pairs_checked"] - len(d["parity"].get("mismatching_pairs", [])), d["parity"]["pairs_checked"]))

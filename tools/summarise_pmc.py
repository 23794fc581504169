#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/steady_steps.py (tools/profile_pmc.sh) -> <out>/pmc_mfma.json and <out>/pmc_hbm.json:
per kernel family of the train step, MFMA busy share and HBM-side bytes / bandwidth per launch.

Units and corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 tallies the 128-byte
requests of wide streaming loads at 64 B); Infinity-Cache hits are counted (fabric traffic: an upper bound of HBM bytes);
SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES / SQ_WAVE_CYCLES quad-cycles.  Durations are the dispatches' own
(kernel_trace.csv of the same pass: a counter pass serialises the kernels, each duration is the kernel alone on the chip)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
FAMILIES = [
    ("conv_igemm (fwd / dgrad, LDS-DMA)", r"conv_igemm_glds_kernel|conv_igemm_halo_kernel", "mfma"),
    ("conv_igemm (register-staged, small Cin / Cout)", r"conv_igemm_kernel<", "mfma"),
    ("conv_wgrad", r"conv_wgrad_glds_kernel|conv_wgrad_glds_jobs_kernel|conv_wgrad_kernel", "mfma"),
    ("wgrad_reduce", r"wgrad_reduce_kernel|wgrad_jobs_reduce", "hbm"),
    ("bn_act", r"bn_act", "hbm"),
    ("bn_finalize", r"bn_finalize", "latency"),
    ("bn_silu_bwd", r"bn_silu_bwd", "hbm"),
    ("mca forward (stats / aggregate)", r"mca_stats|mca_aggregate|mca_prep|mca_finalize", "hbm / mfma"),
    ("mca backward (delta / dq / dkv)", r"mca_bwd", "hbm / mfma"),
    ("cos_attn (CAFM / TaskAligned)", r"cos_attn", "mfma (tiny)"),
    ("se_fuse / add_layernorm", r"se_fuse|add_layernorm", "hbm"),
    ("haar / pool / focus / copy_slice", r"haar|maxpool|focus|copy_slice|upsample", "hbm"),
    ("sgd_ema / ema / pack", r"sgd_ema|ema_update|pack_weights|grad_gather_check", "hbm"),
]


def newest(pat):
    f = glob.glob(pat, recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def family(name):
    for fam, pat, bound in FAMILIES:
        if re.search(pat, name):
            return fam, bound
    return None, None


def durations(d):
    tr = newest(os.path.join(d, "**", "*kernel_trace.csv"))
    dur = {}
    if tr:
        for r in csv.DictReader(open(tr)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return dur


def counters(d):
    cc = newest(os.path.join(d, "**", "*counter_collection.csv"))
    rows = collections.defaultdict(dict)
    names = {}
    if cc:
        for r in csv.DictReader(open(cc)):
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = rows[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
    return rows, names


def derive(e):
    """MFMA pipe occupancy from the per-launch sums.  GRBM_GUI_ACTIVE is reported once per XCD (8 instances, summed here): / 8 = the
    launch's active cycles; the chip has 256 CUs x 4 SIMDs = 1024 MFMA pipes.  SQ_BUSY_CU_CYCLES (cycles, summed over the CUs) gives the
    second figure: per SIMD-cycle of a CU that had work."""
    m = e.get("SQ_VALU_MFMA_BUSY_CYCLES_per_launch", 0.0)
    if e.get("GRBM_GUI_ACTIVE_per_launch"):
        e["mfma_busy_frac_of_all_simd_cycles"] = m / (e["GRBM_GUI_ACTIVE_per_launch"] / 8.0 * 1024.0)
    if e.get("SQ_BUSY_CU_CYCLES_per_launch"):
        e["mfma_busy_frac_of_busy_cu_simd_cycles"] = m / (e["SQ_BUSY_CU_CYCLES_per_launch"] * 4.0)
    return e


# ---- MFMA busy
sq, sq_names = counters(os.path.join(out, "pmc_sq"))
sq_dur = durations(os.path.join(out, "pmc_sq"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for did, c in sq.items():
    fam, bound = family(sq_names[did])
    if fam is None:
        continue
    a = agg[fam]
    a["launches"] += 1
    a["seconds"] += sq_dur.get(did, 0.0)
    for k, v in c.items():
        a[k] += v
    a["bound"] = bound
mf = {}
for fam, a in agg.items():
    e = {"bound": a["bound"], "launches_per_step": a["launches"] / nsteps, "avg_us": a["seconds"] / max(a["launches"], 1) * 1e6,
         "ms_per_step": a["seconds"] / nsteps * 1e3}
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
        if k in a:
            e[k + "_per_launch"] = a[k] / a["launches"]
    derive(e)
    mf[fam] = e
json.dump({"what": "MFMA pipe occupancy per kernel family of a steady-state TSCD-L train step (MI355X, rocprofv3 PMC)",
           "command": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 tools/steady_steps.py {nsteps}",
           "method": "counter values summed over their instances per dispatch, then averaged over the family's dispatches; "
                     "mfma_busy_frac_of_all_simd_cycles = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCD instances) x 1024 SIMDs) -- the gfx94x "
                     "MfmaUtil formula (ROCm 7.2 ships no gfx950 derived metrics; check: avg duration x ~2.4 GHz ~ GRBM_GUI_ACTIVE / 8); "
                     "mfma_busy_frac_of_busy_cu_simd_cycles = the same numerator over 4 x SQ_BUSY_CU_CYCLES",
           "families": mf}, open(os.path.join(out, "pmc_mfma.json"), "w"), indent=1)
# ---- HBM-side traffic
fe, fe_names = counters(os.path.join(out, "pmc_fetch"))
fe_dur = durations(os.path.join(out, "pmc_fetch"))
wr, wr_names = counters(os.path.join(out, "pmc_write"))
hb = collections.defaultdict(lambda: collections.defaultdict(float))
for did, c in fe.items():
    fam, bound = family(fe_names[did])
    if fam is None:
        continue
    hb[fam]["fetch_launches"] += 1
    hb[fam]["fetch_kib"] += c.get("FETCH_SIZE", 0.0)
    hb[fam]["seconds"] += fe_dur.get(did, 0.0)
    hb[fam]["bound"] = bound
for did, c in wr.items():
    fam, bound = family(wr_names[did])
    if fam is None:
        continue
    hb[fam]["write_launches"] += 1
    hb[fam]["write_kib"] += c.get("WRITE_SIZE", 0.0)
res = {}
for fam, a in hb.items():
    if not a["fetch_launches"] or not a["write_launches"]:
        continue
    f, w = a["fetch_kib"] / a["fetch_launches"], a["write_kib"] / a["write_launches"]
    byts = (2 * f + w) * 1024
    t = a["seconds"] / a["fetch_launches"]
    res[fam] = {"bound": a["bound"], "launches_per_step": a["fetch_launches"] / nsteps, "avg_us": t * 1e6, "fetch_size_kib_raw": f, "write_size_kib": w,
                "bytes_per_launch": byts, "gbps": byts / t / 1e9 if t > 0 else None, "frac_of_8_tbps": byts / t / 8e12 if t > 0 else None,
                "mb_per_step": byts * a["fetch_launches"] / nsteps / 1e6}
json.dump({"what": "HBM-side (fabric) traffic and bandwidth per kernel family of a steady-state TSCD-L train step (MI355X, rocprofv3 PMC)",
           "commands": [f"rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/steady_steps.py {nsteps}",
                        f"rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/steady_steps.py {nsteps}"],
           "method": "separate passes; KiB units; FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted: fabric "
                     "traffic, an upper bound of HBM bytes; bandwidth = bytes / the dispatch's own duration in the FETCH pass",
           "families": res}, open(os.path.join(out, "pmc_hbm.json"), "w"), indent=1)
for fam in sorted(set(mf) | set(res)):
    m, h = mf.get(fam, {}), res.get(fam, {})
    print(f"{fam:48s} {m.get('launches_per_step', 0):6.1f}/step {m.get('avg_us', 0):8.1f} us  mfma busy {100 * m.get('mfma_busy_frac_of_all_simd_cycles', 0):5.1f} %"
          f"  | {h.get('bytes_per_launch', 0) / 1e6:8.2f} MB/launch {h.get('gbps') or 0:8.0f} GB/s")

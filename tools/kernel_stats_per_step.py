#!/usr/bin/env python3
"""rocprofv3 *kernel_stats.csv of `tools/steady_steps.py N` -> per-step table grouped by kernel family.  Usage: kernel_stats_per_step.py <csv> <N>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
N = float(sys.argv[2])
fam = [("conv fwd/dgrad (igemm)", r"conv_igemm"), ("conv wgrad + reduce", r"conv_wgrad|wgrad_reduce"), ("bn finalize", r"bn_finalize"),
       ("bn apply (+act)", r"bn_act_kernel"), ("bn+silu bwd", r"bn_silu_bwd|colsum_partials"), ("mca fwd", r"mca_(prep|stats|aggregate|finalize)"),
       ("mca bwd", r"mca_bwd"), ("cos attention", r"cos_attn"), ("se / layernorm / rownorm", r"se_fuse|add_layernorm|row_normalize"),
       ("haar / pool / copy / focus", r"haar|maxpool|copy_slice|focus|zero_insert|upsample"), ("proposal / gather / nms / simota / loss", r"decode|select|gather|scatter|nms|simota|loss_|scale_grad"),
       ("optimizer / ema / pack", r"sgd_ema|ema_update|pack_"), ("bias / relu bwd", r"bias_relu"), ("torch elementwise / copies", r"at::native|rocclr|elementwise|Memcpy|fill")]
tot = {f: [0.0, 0.0] for f, _ in fam}
other = [0.0, 0.0]
for r in rows:
    nm, calls, ns = r["Name"], float(r["Calls"]), float(r["TotalDurationNs"])
    for f, pat in fam:
        if re.search(pat, nm):
            tot[f][0] += calls
            tot[f][1] += ns
            break
    else:
        other[0] += calls
        other[1] += ns
print(f"per step over {int(N)} steps   launches   ms")
s0 = s1 = 0.0
for f, _ in fam:
    print(f"  {f:42s} {tot[f][0] / N:8.1f} {tot[f][1] / N / 1e6:8.3f}")
    s0 += tot[f][0]
    s1 += tot[f][1]
print(f"  {'other':42s} {other[0] / N:8.1f} {other[1] / N / 1e6:8.3f}")
print(f"  {'TOTAL':42s} {(s0 + other[0]) / N:8.1f} {(s1 + other[1]) / N / 1e6:8.3f}")
print("top kernels (calls/step, avg us, ms/step):")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"  {float(r['Calls']) / N:7.1f} {float(r['AverageNs']) / 1e3:8.1f} {float(r['TotalDurationNs']) / N / 1e6:7.3f}  {r['Name'][:110]}")

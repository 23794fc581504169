#!/usr/bin/env python3
"""rocprofv3 *kernel_trace.csv of `tools/steady_steps.py N` -> which kernels make up the part of a train step that follows the host's
assignment chain (CAFM + TaskAligned forward, refined-head losses, backward, optimizer), and how much of that span is idle.

A step is cut at its optimizer launch (sgd_ema); inside a step the post phase starts at the first forward se_fuse launch (the first
kernel of CAFM, queued by matcher.finish once the chain has been joined) -- under the profiler the device is slower than the host's
chain, so the idle span in front of it is not a usable marker.  Medians over the last `steps` steps.
Usage: post_phase_kernels.py <kernel_trace.csv> [steps=40] [--seq]   (--seq: also the launch sequence of the last step, with queue ids)"""
import csv
import re
import statistics as st
import sys
from collections import defaultdict

rows = []
SEQ = "--seq" in sys.argv
argv = [a for a in sys.argv if a != "--seq"]
for r in csv.DictReader(open(argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
steps = int(argv[2]) if len(argv) > 2 else 40
cuts = [i for i, r in enumerate(rows) if re.search(r"sgd_ema", r[2])]
if len(cuts) < steps + 2:
    sys.exit(f"only {len(cuts)} optimizer launches in the trace")
cuts = cuts[-(steps + 1):]


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:86]


per_name = defaultdict(lambda: [0, 0.0])
span, busy, pre_span, gap_len, nlaunch, win_busy = [], [], [], [], [], []
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a + 1:b + 1]                       # kernels after the previous optimizer launch up to and including this one
    at = next((i for i, r in enumerate(seg) if re.search(r"se_fuse_kernel", r[2])), None)
    if at is None:
        continue
    best = seg[at][0] - max(r[1] for r in seg[:at])
    post = seg[at:]
    t0, t1 = post[0][0], max(r[1] for r in post)
    # busy time of the post phase = union of its launch intervals
    u, cur_s, cur_e = 0, post[0][0], post[0][1]
    for s, e, *_ in post[1:]:
        if s > cur_e:
            u += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    u += cur_e - cur_s
    span.append((t1 - t0) / 1e6)
    busy.append(u / 1e6)
    gap_len.append(best / 1e6)
    pre_span.append((seg[at - 1][1] - seg[0][0]) / 1e6 if at else 0.0)
    win_busy.append(sum(r[1] - r[0] for r in seg[:at]) / 1e6)
    nlaunch.append(len(post))
    last_post, last_pre = post, seg[:at]
    for s, e, n, _q in post:
        p = per_name[short(n)]
        p[0] += 1
        p[1] += (e - s) / 1e3

n = len(span)
print(f"{n} steps: launches before the wait {st.median(pre_span):.2f} ms span ({st.median(win_busy):.2f} ms of kernel time), device waits "
      f"{st.median(gap_len):.2f} ms for the host's chain,")
print(f"post phase {st.median(span):.2f} ms span, {st.median(busy):.2f} ms with a kernel running, {st.median(nlaunch):.0f} launches "
      f"({(st.median(span) - st.median(busy)) * 1e3 / st.median(nlaunch):.2f} us idle per launch)")
print("post-phase kernels (launches/step, avg us, ms/step):")
tot = 0.0
for name, (c, us) in sorted(per_name.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {c / n:6.1f} {us / c:8.1f} {us / n / 1e3:7.3f}  {name}")
    tot += us
print(f"  sum of all post-phase kernel durations {sum(v[1] for v in per_name.values()) / n / 1e3:.3f} ms/step")
if SEQ:
    qs = {}
    print("launch sequence of the last step BEFORE the post phase (us from the first launch, duration us, queue, kernel):")
    for s, e, n, q in last_pre:
        qs.setdefault(q, len(qs))
        print(f"  {(s - last_pre[0][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{qs[q]}  {short(n)[:70]}")
    print("launch sequence of the last step's post phase (us from its start, duration us, queue, kernel):")
    for s, e, n, q in last_post:
        qs.setdefault(q, len(qs))
        print(f"  {(s - last_post[0][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{qs[q]}  {short(n)[:70]}")

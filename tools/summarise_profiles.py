"""Turns gpurun_out/final/* into the committed profiles/ summaries (run in the build container after final_profiles.sh)."""
import csv, glob, json, collections, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in tools/)
fin = os.path.join(root, "gpurun_out", "final")
outdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles")   # on the GPU box: a directory under gpurun_out/ (merged back)
os.makedirs(outdir, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)      # gpurun merges runs into the same directory
stats = newest(os.path.join(fin, "stats", "*", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(outdir, f"{tag}_bench_kernel_stats.csv"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline   "
              "(3 warm-up + 50 early-window + 80 pre-roll + 10 timed train steps in the 500-proposal regime + 3 instrumented roofline steps + 35 roofline launches of the dominant shape + the 50-proposal regime's steps; "
              "per-step figures: profiles/*_steady_kernel_stats*.txt from tools/steady_steps.py)\n")
    out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows[:70]:
        out.write('"%s",%s,%s,%s,%s\n' % (r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
shutil.copy(os.path.join(fin, "bench_line.json"), os.path.join(outdir, f"{tag}_bench_line.json"))
def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0]); dom = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        nm = r["Kernel_Name"]
        if ("conv_igemm_glds_kernel<" in nm and ", 128, 2>" in nm) or "conv_igemm_v3_kernel" in nm or "conv_igemm_halo_kernel" in nm:
            v = float(r["Counter_Value"]); agg["all"][0] += 1; agg["all"][1] += v
            key = (nm.split("(")[0], int(r["Grid_Size"]))
            dom[key][0] += 1; dom[key][1] += v
    return agg, dom
fa, fd = load(newest(os.path.join(fin, "pmc_fetch", "*", "*counter_collection.csv")))
wa, wd = load(newest(os.path.join(fin, "pmc_write", "*", "*counter_collection.csv")))
n = fa["all"][0]; f = fa["all"][1] / n; w = wa["all"][1] / wa["all"][0]
# dominant shape: B16 72x72 256->256 3x3 -> 432 workgroups x 512 threads of the halo-staged variant (192-pixel x 256-channel tiles)
dk = [k for k in fd if "halo" in k[0] and k[1] == 432 * 512]
dd = None
if dk:
    k = dk[0]; fdv = fd[k][1] / fd[k][0]; wdv = wd[k][1] / wd[k][0]
    dd = {"shape": "B16 72x72 256->256 3x3 (conv_igemm_halo_kernel<6>: 192-row tiles, 432 workgroups)", "launches": fd[k][0], "fetch_size_kib_raw": fdv,
          "write_size_kib": wdv, "bytes_per_launch": (2 * fdv + wdv) * 1024, "algorithmic_bytes": 82944 * 256 * 2 * 2 + 256 * 2304 * 2}
out = {"what": "HBM-side traffic per launch of the LDS-DMA implicit-GEMM conv kernels (conv_igemm_glds_kernel<*,128,2> + conv_igemm_halo_kernel + conv_igemm_v3_kernel), rocprofv3 PMC counters (MI355X)",
       "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline",
                    "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline"],
       "method": "separate passes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2); counter unit KiB; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B-per-lane streaming loads at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted, so this is fabric traffic, an upper bound of HBM bytes",
       "launches": n, "all_launches": {"fetch_size_kib_raw": f, "write_size_kib": w, "bytes_per_launch": (2 * f + w) * 1024}, "dominant_shape": dd}
json.dump(out, open(os.path.join(outdir, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1200])
for r in rows[:12]:
    print(r["Name"][:70], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))

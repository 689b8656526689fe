"""Turn the raw rocprofv3 outputs merged back under gpurun_out/ into the committed summaries in profiles/.

    python tools/make_profiles.py <tag>      (e.g. r01_v4)
expects gpurun_out/prof (kernel stats), gpurun_out/pmc_fetch, gpurun_out/pmc_write, gpurun_out/bench.log
"""
import collections
import os
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1]


def one(pat):
    g = sorted(glob.glob(pat), key=os.path.getmtime)   # newest run (pid-named files do not sort by name)
    assert g, pat
    return g[-1]


shutil.copy(one("gpurun_out/prof/*/*kernel_stats.csv"), f"profiles/{tag}_bench_b32_384_kernel_stats.csv")
shutil.copy("gpurun_out/bench.log", f"profiles/{tag}_bench_b32_384.json")


def agg(path, cname):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            d[r["Kernel_Name"]][0] += 1
            d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


F = agg(one("gpurun_out/pmc_fetch/*/*counter_collection.csv"), "FETCH_SIZE")
W = agg(one("gpurun_out/pmc_write/*/*counter_collection.csv"), "WRITE_SIZE")
with open(f"profiles/{tag}_pmc_hbm_traffic_per_kernel.csv", "w") as fo:
    fo.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: bench.py --steps 1 --warmup 1 (B=32, 384x384)\n")
    fo.write("# counter values are KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of wide\n")
    fo.write("# coalesced reads (MI355X_MICROARCH.md, HBM); the x2 is calibrated for 16-byte loads only, so it is an upper bound for\n")
    fo.write("# the 4-byte gather loads of the implicit-GEMM kernels\n")
    fo.write("kernel,launches,fetch_KiB_per_launch_raw,write_KiB_per_launch,hbm_bytes_per_launch_corrected\n")
    for k in sorted(F, key=lambda k: -F[k][1]):
        n, fs = F[k]
        ws = W.get(k, [n, 0])[1]
        fo.write(f"\"{k}\",{n},{fs / n:.1f},{ws / n:.1f},{(2 * fs + ws) / n * 1024:.0f}\n")
fam = {"wfae_conv4x4s2_wgrad": "0, 4, 1, true", "wfae_conv4x4s2_up": "1, 3, 2, true", "wfae_conv4x4s2_down": "0, 2, 0, true",
       "wfae_wino_gemm_wgrad": "256, 2, 2, 0, 1, 1, true", "wfae_wino_gemm_down": "256, 2, 2, 0, 0, 0, true",
       "wfae_wino_gemm_up": "256, 2, 2, 1, 0, 0, true"}
out = {}
for ep, sig in fam.items():
    n = fs = ws = 0
    for k in F:
        if "gemm_kernel<" in k and sig in k:
            n += F[k][0]
            fs += F[k][1]
            ws += W.get(k, [0, 0])[1]
    if n:
        out[ep] = {"hbm_bytes_per_launch": (2 * fs + ws) / n * 1024, "fetch_raw_KiB": fs / n, "write_KiB": ws / n, "launches": n,
                   "source": f"profiles/{tag}_pmc_hbm_traffic_per_kernel.csv"}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
# dominant kernel = top row (by total time) of the kernel-trace stats; map it to the entry point / label that bench.py
# times with events, and record its PMC traffic under that label
rows = list(csv.DictReader(open(f"profiles/{tag}_bench_b32_384_kernel_stats.csv")))
top = rows[0]
ep = next((e for e, sig in fam.items() if "gemm_kernel<" in top["Name"] and sig in top["Name"]), None)
if ep is None:
    for key, e in (("bn_act_bwd_dx_kernel", "wfae_bn_act_bwd[dx]"), ("bn_act_bwd_reduce_kernel", "wfae_bn_act_bwd[reduce]"),
                   ("bn_act_fwd_kernel", "wfae_bn_act_fwd"), ("chan_reduce_kernel", "wfae_bn_stats_train"),
                   ("sgemm3_kernel<0,", "split_gemm[B=KxN]"), ("sgemm3_kernel<1,", "split_gemm[B=NxK]")):
        if key in top["Name"]:
            ep = e
if ep is not None and ep not in out and top["Name"] in F:
    n, fs = F[top["Name"]]
    ws_ = W.get(top["Name"], [n, 0])[1]
    out[ep] = {"hbm_bytes_per_launch": (2 * fs + ws_) / n * 1024, "fetch_raw_KiB": fs / n, "write_KiB": ws_ / n, "launches": n,
               "source": f"profiles/{tag}_pmc_hbm_traffic_per_kernel.csv"}
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
# every single-kernel label bench.py can name as dominant gets its PMC traffic, tagged with the kernel sources it was
# measured on (bench.py reports `traffic` only while that tag matches the sources it runs)
for key, e in (("bn_act_bwd_dx_kernel", "wfae_bn_act_bwd[dx]"), ("bn_act_bwd_reduce_kernel", "wfae_bn_act_bwd[reduce]"),
               ("bn_act_fwd_kernel", "wfae_bn_act_fwd"), ("chan_reduce_kernel<0>", "wfae_bn_stats_train"),
               ("sgemm3_kernel<0,", "split_gemm[B=KxN]"), ("sgemm3_kernel<1,", "split_gemm[B=NxK]")):
    n = fs = ws_ = 0
    for k in F:
        if key in k:
            n += F[k][0]
            fs += F[k][1]
            ws_ += W.get(k, [0, 0])[1]
    if n and e not in out:
        out[e] = {"hbm_bytes_per_launch": (2 * fs + ws_) / n * 1024, "fetch_raw_KiB": fs / n, "write_KiB": ws_ / n,
                  "launches": n, "source": f"profiles/{tag}_pmc_hbm_traffic_per_kernel.csv"}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_tag)
# whole-step HBM bytes (bench.py step_roofline.hbm_bytes_per_step_measured): all launches of the PMC run, which times
# `steps` steps after `warmup` untimed ones (2 in total); the 2 x FETCH bound and the raw counters side by side
nsteps = 2
out["_step_total"] = {"bytes_2xfetch_plus_write": (2 * sum(v[1] for v in F.values()) + sum(v[1] for v in W.values())) * 1024 / nsteps,
                      "bytes_raw_fetch_plus_write": (sum(v[1] for v in F.values()) + sum(v[1] for v in W.values())) * 1024 / nsteps,
                      "steps_in_run": nsteps}
# the same for --precision medium (bf16 activation storage) when tools/gpu_profile.sh collected it
if glob.glob("gpurun_out/pmc_fetch_medium/*/*counter_collection.csv") and glob.glob("gpurun_out/pmc_write_medium/*/*counter_collection.csv"):
    Fm = agg(one("gpurun_out/pmc_fetch_medium/*/*counter_collection.csv"), "FETCH_SIZE")
    Wm = agg(one("gpurun_out/pmc_write_medium/*/*counter_collection.csv"), "WRITE_SIZE")
    out["_step_total_bf16"] = {"bytes_2xfetch_plus_write": (2 * sum(v[1] for v in Fm.values()) + sum(v[1] for v in Wm.values())) * 1024 / nsteps,
                               "bytes_raw_fetch_plus_write": (sum(v[1] for v in Fm.values()) + sum(v[1] for v in Wm.values())) * 1024 / nsteps,
                               "steps_in_run": nsteps}
    with open(f"profiles/{tag}_medium_pmc_hbm_traffic_per_kernel.csv", "w") as fo:
        fo.write("# bench.py --precision medium (bf16 activation storage): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KiB\n")
        fo.write("kernel,launches,fetch_KiB_per_launch_raw,write_KiB_per_launch,hbm_bytes_per_launch_corrected\n")
        for k in sorted(Fm, key=lambda k: -Fm[k][1]):
            n, fs = Fm[k]
            ws = Wm.get(k, [n, 0])[1]
            fo.write(f"\"{k}\",{n},{fs / n:.1f},{ws / n:.1f},{(2 * fs + ws) / n * 1024:.0f}\n")
    shutil.copy(one("gpurun_out/prof_medium/*/*kernel_stats.csv"), f"profiles/{tag}_medium_bench_b32_384_kernel_stats.csv")
    shutil.copy("gpurun_out/bench_medium.log", f"profiles/{tag}_medium_bench_b32_384.json")
out["_kernel_source_tag"] = bench.kernel_source_tag()
out["_source"] = f"{tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1"
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump({"kernel": top["Name"], "entry_point": ep, "calls": int(top["Calls"]), "avg_ns": float(top["AverageNs"]),
           "source": f"profiles/{tag}_bench_b32_384_kernel_stats.csv"}, open("profiles/dominant_kernel.json", "w"), indent=1)
print(open("profiles/dominant_kernel.json").read())

"""Copies the summaries of a tools/round3_profiles.sh run (a directory under gpurun_out/) into
profiles/: the files the committed figures are quoted from.  Development aid, run here (no GPU).
    python tools/install_round3_profiles.py gpurun_out/r03_final2"""
import csv, glob, json, os, shutil, sys

src = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def put(a, b):
    os.makedirs(os.path.dirname(b), exist_ok=True)
    shutil.copyfile(a, b)


# the default bench run: kernel statistics, HBM counters, issue counters of the two sweeps
b = os.path.join(P, "r03_bench_soup1m_4k")
put(f"{src}/bench/kernel_stats.csv", f"{b}/kernel_stats.csv")
put(f"{src}/bench/pmc_summary.txt", f"{b}/pmc_summary.txt")
put(f"{src}/raster_counters.txt", f"{b}/raster_counters.txt")
put(f"{src}/cells_counters.txt", f"{b}/cells_counters.txt")
if os.path.exists(f"{src}/bench_plain.json"):
    line = [l for l in open(f"{src}/bench_plain.json") if l.startswith("{")][-1]
    open(f"{b}/bench_line.json", "w").write(line)
put(f"{src}/bench/bench_line.json", f"{b}/bench_line_under_rocprofv3.json")

# the general Moller-Trumbore sweep
g = os.path.join(P, "r03_general_rays")
os.makedirs(g, exist_ok=True)
out = []
for w in ("mesh100k_4k", "soup1m_4k"):
    line = json.loads([l for l in open(f"{src}/general_{w}.json") if l.startswith("{")][-1])
    f = glob.glob(f"{src}/general_{w}/*/*_counter_collection.csv")[0]
    per = {}
    for row in csv.DictReader(open(f)):
        if "k_sweep<" in row["Kernel_Name"]:
            per.setdefault(int(row["Dispatch_Id"]), {}).setdefault(row["Counter_Name"], 0.0)
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    d = per[max(per)]
    flop = 64 * (d["SQ_INSTS_VALU_ADD_F32"] + d["SQ_INSTS_VALU_MUL_F32"] + 2 * d["SQ_INSTS_VALU_FMA_F32"])
    out.append(f"{w}: rt_trace_rays on the 8 294 400 primary rays of the 3840x2160 frame, kernel k_sweep<1, false>, last launch of the program\n"
               f"   pairs {line['tests']:.6e}   HIP-event ms {line['ms']:.2f} (run under the counters)   pairs/s {line['tests_per_s']:.4e}\n"
               + "".join(f"   {k:26s} {v:.6e} wave instructions\n" for k, v in sorted(d.items()))
               + f"   executed fp32 flop = 64 x (ADD + MUL + 2 FMA) = {flop:.6e} = {flop / line['tests']:.2f} per pair "
               f"(the kernel's own count: {line['flop_per_test']}; required: within 15 %)\n"
               f"   fraction of the 157.3 TFLOP/s fp32 peak at {line['flop_per_test']} flop per pair: {line['frac']:.3f}\n")
    st = glob.glob(f"{src}/general_trace_{w}/*/*_kernel_stats.csv")
    if st:
        put(st[0], f"{g}/kernel_stats_{w}.csv")
        line2 = json.loads([l for l in open(f"{src}/general_trace_{w}.json") if l.startswith("{")][-1])
        out.append(f"   under --kernel-trace --stats only: {line2['ms']:.2f} ms, {line2['tests_per_s']:.4e} pairs/s, frac {line2['frac']:.3f}\n")
open(f"{g}/valu_f32_counters.txt", "w").write(
    "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES "
    "-- python3 tools/general_mt.py <workload>   (tools/round3_profiles.sh)\n\n" + "\n".join(out))

# strong-scaling prediction
bs = os.path.join(P, "r03_band_scaling")
put(f"{src}/band/final.txt", f"{bs}/end_of_round3.txt")
put(f"{src}/band/timeline_part3of8_final.txt", f"{bs}/timeline_part3of8_end_of_round3.txt")

put(f"{src}/other_workloads.jsonl", f"{P}/r03_other_workloads.jsonl")
put(f"{src}/valu_rate.txt", f"{P}/valu_rate_gfx950.txt")

# HBM traffic of the primary launch: (2 FETCH_SIZE + WRITE_SIZE) KiB
vals = {}
name = None
for l in open(f"{src}/bench/pmc_summary.txt"):
    if l.startswith("=="):
        name = l
    elif "k_sweep_raster_wave<" in (name or "") and "mean=" in l:
        vals[l.split()[0]] = float(l.split("mean=")[1])
line = json.loads(open(f"{b}/bench_line.json").read()) if os.path.exists(f"{b}/bench_line.json") else None
if vals and line:
    k = line["roofline"]["sweep_kernel"]
    key = f"soup1m_4k/raster-tile{k['tile_tris']}-ppl{k['pixels_per_lane']}-chunks{k['triangle_chunks']}"
    tj = os.path.join(P, "traffic.json")
    t = json.load(open(tj))
    t[key] = {"bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024, "fetch_size_kb": vals["FETCH_SIZE"],
              "write_size_kb": vals["WRITE_SIZE"],
              "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports KiB, and gfx950 tallies 128-byte read requests of wide coalesced loads at 64 bytes",
              "source": "profiles/r03_bench_soup1m_4k/pmc_summary.txt"}
    json.dump(t, open(tj, "w"), indent=1)
    print(key, t[key]["bytes"] / 1e9, "GB; writes", vals["WRITE_SIZE"] * 1024 / 1e9, "GB")
print("installed from", src)

"""Joins a rocprofv3 counter pass over tools/part_ranges.py with the pair tests that program
prints per any-hit range (development aid): apex_breakdown.py <rocprof dir> <log>."""
import collections, csv, glob, re, sys

root, log = sys.argv[1], sys.argv[2]
tests = [float(m.group(1)) for m in re.finditer(r"any-hit range .*? ([0-9.e+]+) tests,", open(log).read())]
trace = sorted(csv.DictReader(open(glob.glob(root + "/*/*_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
disp = [r for r in trace if "k_sweep_apex" in r["Kernel_Name"]][-len(tests):]
ctr = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(glob.glob(root + "/*/*_counter_collection.csv")[0])):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
print("any-hit ranges of the last frame: per pair figures are wave-level counts x 64 lanes / pair tests")
print(f"{'kernel':14} {'grid':>14} {'ms':>8} {'pairs':>10} {'T pairs/s':>9} {'VALU/pair':>9} {'SALU/pair':>9} {'LDS/pair':>8} "
      f"{'wait_inst/wave_cyc':>18} {'valu_busy/wave_cyc':>18} {'GHz':>5} {'issue frac':>10}")
tot = collections.defaultdict(float)
for r, t in zip(disp, tests):
    c = ctr[r["Dispatch_Id"]]
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    name = re.search(r"k_sweep_apex<\d+(, (false|true))?>", r["Kernel_Name"]).group(0)
    grid = f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    ghz = c["GRBM_GUI_ACTIVE"] / (ms * 1e-3) / 1e9 if "GRBM_GUI_ACTIVE" in c else float("nan")
    # issue slots: one VALU instruction per SIMD every two cycles = 1024 SIMDs x clock / 2 wave-instructions/s
    frac = c["SQ_INSTS_VALU"] / (ms * 1e-3) / (1024 * 2.4e9 / 2)
    print(f"{name:14} {grid:>14} {ms:8.3f} {t:10.3e} {t / ms / 1e9:9.2f} {c['SQ_INSTS_VALU'] * 64 / t:9.3f} "
          f"{c['SQ_INSTS_SALU'] * 64 / t:9.3f} {c['SQ_INSTS_LDS'] * 64 / t:8.3f} {c['SQ_WAIT_INST_ANY'] / max(c['SQ_WAVE_CYCLES'], 1):18.3f} "
          f"{c['SQ_ACTIVE_INST_VALU'] / max(c['SQ_WAVE_CYCLES'], 1):18.3f} {ghz:5.2f} {frac:10.3f}")
    for k, v in c.items():
        tot[k] += v
    tot["ms"] += ms
    tot["tests"] += t
print(f"all ranges: {tot['ms']:.2f} ms, {tot['tests']:.3e} pairs, {tot['tests'] / tot['ms'] / 1e9:.2f} T pairs/s, "
      f"VALU/pair {tot['SQ_INSTS_VALU'] * 64 / tot['tests']:.3f}, SALU/pair {tot['SQ_INSTS_SALU'] * 64 / tot['tests']:.3f}, "
      f"mean clock {tot['GRBM_GUI_ACTIVE'] / (tot['ms'] * 1e-3) / 1e9:.2f} GHz, "
      f"issue slots used (nominal 2.4 GHz) {tot['SQ_INSTS_VALU'] / (tot['ms'] * 1e-3) / (1024 * 2.4e9 / 2):.3f}")

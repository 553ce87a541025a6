"""Prints the kernel timeline of the last frame in a rocprofv3 kernel trace of tools/part_trace.py
(development aid): part_timeline.py <kernel_trace.csv> [all]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: from the last primary sweep that follows a shading kernel (a frame may have two sweeps)
sweeps = [i for i, r in enumerate(rows) if "k_sweep_raster" in r["Kernel_Name"]]
shades = [i for i, r in enumerate(rows) if "k_shade_final" in r["Kernel_Name"]]
ends = [i for i in shades if any(j > i for j in sweeps)]
idx = min(j for j in sweeps if j > ends[-1]) if ends else sweeps[0]
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    prev_end = e
    busy += e - s
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
    if len(sys.argv) > 2 or (e - s) > 20000 or gap > 10:
        print(f"{(s-t0)/1e6:8.3f} ms  gap {gap:7.1f} us  dur {(e-s)/1e6:7.3f} ms  grid "
              f"{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d} x {r['Grid_Size_Y']:>3} x {r['Grid_Size_Z']:>3}  {n}")
print(f"frame {(prev_end-t0)/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms")

"""Kernel timeline of the last frame in a rocprofv3 kernel trace of tools/frame_trace.py
(development aid): frame_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_transform" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"]); prev = None; busy = 0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0
    prev = e; busy += e - s
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:30]
    print(f"{(s-t0)/1e6:8.3f} ms  gap {gap:7.1f} us  dur {(e-s)/1e3:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):6d} x {r['Grid_Size_Y']:>3} x {r['Grid_Size_Z']:>3}  {n}")
print(f"frame {(prev-t0)/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms")

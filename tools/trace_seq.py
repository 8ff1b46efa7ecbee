"""Prints one occurrence of a kernel sequence from a rocprofv3 kernel trace csv:
   python tools/trace_seq.py <kernel_trace.csv> <first-kernel-substring> [count] [occurrence]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key, cnt = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 30
occ = int(sys.argv[4]) if len(sys.argv) > 4 else 5
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]][occ]
t0 = int(rows[idx]["Start_Timestamp"]); prev = None
for r in rows[idx:idx + cnt]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s start=%8.1f dur=%7.1f gap=%6.1f grid=%s,%s" % (r["Kernel_Name"][:44], (st - t0) / 1e3, (en - st) / 1e3,
          (st - prev) / 1e3 if prev else 0, r["Grid_Size_X"], r["Grid_Size_Y"]))
    prev = en

"""timeline of kernels in a rocprofv3 kernel trace between two occurrences of a marker kernel:
python scripts/timeline.py trace.csv <marker substr> <first occurrence index> <how many occurrences>"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
key, first, cnt = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0, i1 = starts[first], starts[min(first + cnt, len(starts) - 1)]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    m = re.search(r"(k_\w+|rocprim\w*|\w+Buffer\w*|elementwise\w*|\w+_kernel)", r["Kernel_Name"])
    print("%9.1f -> %9.1f us (%7.1f)  q%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), m.group(1)[:40] if m else r["Kernel_Name"][:40]))

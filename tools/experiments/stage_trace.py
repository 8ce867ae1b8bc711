"""Group a rocprofv3 kernel trace by (kernel, grid): calls, mean duration, ms per volume (8 volumes per batch)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nvol = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
d = collections.defaultdict(list)
for x in rows:
    n = x["Kernel_Name"].replace("void ", "").replace("fr3d::", "")
    key = (n[:44], x["Grid_Size_X"], x["Grid_Size_Y"], x["Grid_Size_Z"]) if "sor_step" not in n else ("k_sor_step", "", "", "")
    d[key].append((int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3)
byk = collections.defaultdict(float)
for k, v in d.items():
    byk[k[0]] += sum(v)
print("== per kernel, ms per volume")
for k, v in sorted(byk.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{k:46s} {v / nvol / 1e3:8.3f}")
print("== per kernel and grid: calls, mean us, ms per volume")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(f"{k[0]:46s} {'x'.join(k[1:]):22s} {len(v):5d} {sum(v) / len(v):9.1f} {sum(v) / nvol / 1e3:8.3f}")

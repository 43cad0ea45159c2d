"""Summarise a rocprofv3 --pmc counter_collection CSV for dsg kernels: per dispatch, counters side by side."""
import csv, sys, collections
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "dsg::"
rows = collections.OrderedDict()
with open(path) as f:
    for r in csv.DictReader(f):
        if pat not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        d = rows.setdefault(k, {"name": r["Kernel_Name"].split("(")[0][-60:], "grid": r["Grid_Size"],
                                "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({c for d in rows.values() for c in d if c not in ("name", "grid", "dur_us")})
print("dispatch  " + " ".join(f"{n[:22]:>22s}" for n in ["dur_us"] + names) + "  kernel/grid")
for k, d in rows.items():
    print(f"{k:8d}  " + " ".join(f"{d.get(n, 0):22.0f}" for n in ["dur_us"] + names) + f"  {d['name']} {d['grid']}")

"""Print the top kernels of a rocprofv3 --kernel-trace --stats CSV per iteration.  argv: csv iters [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
it = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / it / 1e6:.3f} ms per iteration, {sum(int(r['Calls']) for r in rows) / it:.0f} launches")
for r in rows[:top]:
    print(f"{float(r['TotalDurationNs']) / it / 1e6:8.3f} ms/it {int(r['Calls']) / it:7.1f} calls/it  {r['Name'][:140]}")

import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 22]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r.get("Percentage","")}')

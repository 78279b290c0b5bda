"""Print the top rows of a rocprofv3 kernel_stats.csv (name, calls, average us, percentage)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(4), str(round(float(r['AverageNs']) / 1e3, 1)).rjust(8), r['Percentage'])

#!/usr/bin/env python
"""Top rows of a rocprofv3 kernel_stats.csv as `name calls avg-us share` (tools only)."""
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:70].ljust(70), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])

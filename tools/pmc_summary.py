"""Summarise the rocprofv3 counter_collection csv files written by tools/gpu_round.sh: per
kernel name, the mean of each counter over its dispatches (summed over the dimension
instances rocprofv3 emits as separate rows)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, 'prof_*', '**', '*counter_collection.csv'), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))      # (kernel, counter) -> dispatch -> value
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].split('(')[0][:60]
            per[(k, row['Counter_Name'])][row['Dispatch_Id']] += float(row['Counter_Value'])
    print('--', os.path.relpath(f, out))
    for (k, c), d in sorted(per.items()):
        vals = list(d.values())
        print('%-62s %-22s n=%-4d mean=%.6g' % (k, c, len(vals), sum(vals) / len(vals)))

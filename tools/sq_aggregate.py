"""Collapse the SQ counter passes of tools/pmc_sq.sh (gpurun_out/pmc_sq/**/p*_counter_collection.csv) into one small file:
per kernel the MEAN value of every counter per launch (the launch sequence of a transition repeats, so a kernel's launches are
alike; the first squaring step's prescale variants are separate kernels).  -> gpurun_out/<tag>_sq_counters.json

    python tools/sq_aggregate.py r05
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_sq', '**', '*_counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
path = os.path.join(ROOT, 'gpurun_out', f'{tag}_sq_counters.json')
json.dump(out, open(path, 'w'), indent=1, sort_keys=True)
for k, v in out.items():
    if 'exp_fwd_march_kernel<false, 1' in k or 'exp_bwd_march_kernel<false, 1' in k:
        print(k, {c: round(x) for c, x in v.items()})
print('wrote', path)

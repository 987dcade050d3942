"""Collapse the rocprofv3 outputs of tools/profile_round.sh into the two small files that get committed under profiles/:
<tag>_kernel_stats.csv (per-kernel calls / total / average, from --stats) and <tag>_pmc_traffic.json (per-kernel mean
FETCH_SIZE / WRITE_SIZE per launch, raw KiB as the counters report them; bench.py applies the gfx950 corrections)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(os.path.join(out, 'trace', '**', 'k_kernel_stats.csv'), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(out, f'{tag}_kernel_stats.csv'))
traffic = collections.defaultdict(dict)
for sub, key in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, '**', '*_counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == key:
                acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    for k, v in acc.items():
        traffic[k][key + '_KiB_raw'] = sum(v) / len(v)
        traffic[k]['launches_' + key] = len(v)
json.dump(traffic, open(os.path.join(out, f'{tag}_pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
for k in sorted(traffic):
    if 'exp_' in k or 'perturb' in k:
        print(k, traffic[k])

"""Sum the counters of a rocprofv3 --pmc run per kernel: python tools/pmc_summary.py gpurun_out/pmc2 [kernel substring ...]"""
import collections, csv, glob, sys
root = sys.argv[1]; want = sys.argv[2:] or ['k_pulse', 'k_photons']
for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')[:36]
        agg[k][row['Counter_Name']] += float(row['Counter_Value'])
        key = (k, row['Dispatch_Id'])
        if key not in seen: seen.add(key); launches[k] += 1
    for k in agg:
        if any(w in k for w in want):
            print(f.split('/')[-3] if '/' in f else f, k, 'launches', launches[k], {c: f'{v / launches[k]:.4g}' for c, v in agg[k].items()})

"""Turn the output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under profiles/:
<tag>_bench.json (the bench line), <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_traffic.json
(FETCH_SIZE / WRITE_SIZE per launch, keyed by the engine's timer names).   python tools/make_profiles.py r1"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag)
dst = os.path.join(root, 'profiles')


def timer_name(kernel):
    k = kernel.replace('void ', '').split('(')[0]
    if k.startswith('k_photons<true'):
        return 'k_photons_count'
    if k.startswith('k_photons<false'):
        return 'k_photons_fill'
    if k.startswith('k_pulse_sparse'):
        return 'k_pulse_sparse'
    if k.startswith('k_pulse<'):
        return 'k_pulse_dense'
    return k.split('<')[0]


def per_launch(pattern, counter):
    agg, launches, seen = collections.defaultdict(float), collections.Counter(), set()
    files = glob.glob(os.path.join(src, pattern, '**', '*counter_collection.csv'), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:          # gpurun merges into gpurun_out/: older runs may still lie there
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != counter:
                continue
            k = timer_name(row['Kernel_Name'])
            agg[k] += float(row['Counter_Value'])
            if (k, row['Dispatch_Id']) not in seen:
                seen.add((k, row['Dispatch_Id'])); launches[k] += 1
    return {k: agg[k] / launches[k] for k in agg}


line = [l for l in open(os.path.join(src, 'bench.json')) if l.startswith('{')][-1]
json.dump(json.loads(line), open(os.path.join(dst, f'{tag}_bench.json'), 'w'), indent=1)
stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, f'{tag}_kernel_stats.csv'))
fetch, write = per_launch('fetch', 'FETCH_SIZE'), per_launch('write', 'WRITE_SIZE')
out = dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 1 --warmup 0, per launch '
                'averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half the bytes of coalesced '
                'streaming reads, MI355X_MICROARCH.md HBM section; narrower accesses are uncalibrated)', kernels={})
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out['kernels'][k] = dict(fetch_size_kb=f, write_size_kb=w, hbm_bytes=(2 * f + w) * 1024)
json.dump(out, open(os.path.join(dst, f'{tag}_traffic.json'), 'w'), indent=1)
print('wrote', [f for f in os.listdir(dst) if f.startswith(tag + '_')])

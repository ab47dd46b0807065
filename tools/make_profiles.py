"""Turn the output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under profiles/:
<tag>_bench.json        the bench line
<tag>_kernel_stats.csv  rocprofv3 --kernel-trace --stats (every dispatch of the profiled process, the cold first batch included)
<tag>_kernel_stats_steady.csv  the same trace without the first batch of the process: per kernel calls / average / min / max duration of the
                        dispatches that start behind the end of the first batch (its last k_pack) -- steady-state averages
<tag>_traffic.json      FETCH_SIZE / WRITE_SIZE per launch -> HBM bytes, keyed by the engine's timer names
<tag>_counters.json     SQ counters per launch and what they say limits each kernel
python tools/make_profiles.py r2"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r2'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', tag)
dst = os.path.join(root, 'profiles')
N_SIMD, N_CU, N_XCD = 1024, 256, 8


def timer_name(kernel):
    k = kernel.replace('void ', '').split('(')[0]
    if k.startswith('k_pulse_sparse'):
        return 'k_pulse_sparse'
    if k.startswith('k_pulse<'):
        return 'k_pulse_dense'
    return k.split('<')[0]


def per_launch(pattern):
    """{kernel: {counter: mean per launch}} of one --pmc pass"""
    agg, launches, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter(), set()
    files = glob.glob(os.path.join(src, pattern, '**', '*counter_collection.csv'), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:          # gpurun merges into gpurun_out/: older runs may still lie there
        for row in csv.DictReader(open(f)):
            k = timer_name(row['Kernel_Name'])
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            if (k, row['Dispatch_Id']) not in seen:
                seen.add((k, row['Dispatch_Id'])); launches[k] += 1
    return {k: {c: v / launches[k] for c, v in agg[k].items()} for k in agg}, dict(launches)


line = [l for l in open(os.path.join(src, 'bench.json')) if l.startswith('{')][-1]
bench = json.loads(line)
json.dump(bench, open(os.path.join(dst, f'{tag}_bench.json'), 'w'), indent=1)
stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if stats:
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, f'{tag}_kernel_stats.csv'))
# steady-state averages: the kernel trace itself, first batch of the process dropped (cold caches, first-touch allocations, code loading)
traces = glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True)
if traces:
    rows = list(csv.DictReader(open(max(traces, key=os.path.getmtime))))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    last = [r for r in rows if timer_name(r['Kernel_Name']) == 'k_pack']
    t_cut = int(last[0]['End_Timestamp']) if last else 0
    dur = collections.defaultdict(list)
    for r in rows:
        if int(r['Start_Timestamp']) > t_cut:
            dur[timer_name(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    total = sum(sum(v) for v in dur.values()) or 1
    with open(os.path.join(dst, f'{tag}_kernel_stats_steady.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 3), min(v), max(v)])
fetch, n_fetch = per_launch('fetch')
write, n_write = per_launch('write')
out = dict(note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 1 --warmup 0 (the full headline batch), '
                'per launch averages; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts half the bytes of coalesced '
                'streaming reads, MI355X_MICROARCH.md HBM section; narrower accesses are uncalibrated)',
           workload=bench.get('config'), kernels={})
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, {}).get('FETCH_SIZE', 0.0), write.get(k, {}).get('WRITE_SIZE', 0.0)
    out['kernels'][k] = dict(fetch_size_kb=f, write_size_kb=w, hbm_bytes=(2 * f + w) * 1024)
kms = bench.get('roofline', {}).get('kernels_ms', {})
tot = sum(v['hbm_bytes'] for k, v in out['kernels'].items() if k in kms)
out['pipeline'] = dict(hbm_bytes_all_timed_kernels=tot, algorithmic_bytes=bench.get('roofline', {}).get('algorithmic_bytes_per_launch'),
                       ratio=(tot / bench['roofline']['algorithmic_bytes_per_launch']) if bench.get('roofline', {}).get('algorithmic_bytes_per_launch') else None)
# every dispatch of one batch, runtime fills and copies (memsets of the accumulators!) included: per-launch mean x launches per batch, the
# number of batches in the profiled process taken from a kernel that runs once per batch
runs = max(1, n_fetch.get('k_counts', n_fetch.get('k_row_len', 1)))
every = sum((2 * fetch.get(k, {}).get('FETCH_SIZE', 0.0) * n_fetch.get(k, 0) + write.get(k, {}).get('WRITE_SIZE', 0.0) * n_write.get(k, 0)) * 1024
            for k in set(fetch) | set(write)) / runs
out['pipeline']['hbm_bytes_every_dispatch_per_batch'] = every
out['pipeline']['batches_in_profiled_process'] = runs
if out['pipeline']['algorithmic_bytes']:
    out['pipeline']['ratio_every_dispatch'] = every / out['pipeline']['algorithmic_bytes']
json.dump(out, open(os.path.join(dst, f'{tag}_traffic.json'), 'w'), indent=1)

# ---- SQ counters
passes = {}
for p in ('p1', 'p2', 'p3'):
    c, n = per_launch(p)
    for k, v in c.items():
        passes.setdefault(k, {}).update(v)
        passes[k]['launches_' + p] = n[k]
cnt = dict(note='rocprofv3 --pmc, three passes (p1-p3 of tools/profile_round.sh), python bench.py --steps 1 --warmup 0: the full headline batch; '
                'per launch averages summed over the chip.  SQ_* cycle counters are in quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over '
                'the 8 XCDs.  valu_busy = 4*SQ_ACTIVE_INST_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE/8) (rocprof\'s VALUBusy); lds_busy likewise with '
                'SQ_ACTIVE_INST_LDS per CU; wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave time spent waiting on any counter); '
                'waves_per_simd = SQ_LEVEL_WAVES-free estimate SQ_WAVE_CYCLES*4 / 1024 / (GRBM_GUI_ACTIVE/8).  limited_by: "hbm" if the measured HBM '
                'bytes / time exceed 60 % of 8 TB/s, else "valu" if valu_busy > 0.6, else "lds" if lds_busy > 0.6, else "latency".',
           workload=bench.get('config'), kernels={})
for k, v in sorted(passes.items()):
    if not k.startswith('k_'):
        continue
    gui = v.get('GRBM_GUI_ACTIVE', 0.0) / N_XCD
    d = dict(counters={c: x for c, x in v.items()})
    if gui > 0:
        d['valu_busy'] = round(4 * v.get('SQ_ACTIVE_INST_VALU', 0.0) / N_SIMD / gui, 4)
        d['lds_busy'] = round(4 * v.get('SQ_ACTIVE_INST_LDS', 0.0) / N_CU / gui, 4)
        d['vmem_busy'] = round(4 * v.get('SQ_ACTIVE_INST_VMEM', 0.0) / N_CU / gui, 4)
        d['waves_per_simd'] = round(4 * v.get('SQ_WAVE_CYCLES', 0.0) / N_SIMD / gui, 2)
    if v.get('SQ_WAVE_CYCLES'):
        d['wait_any'] = round(v.get('SQ_WAIT_ANY', 0.0) / v['SQ_WAVE_CYCLES'], 4)
    if v.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict'] = round(v.get('SQ_LDS_BANK_CONFLICT', 0.0) / v['SQ_LDS_IDX_ACTIVE'], 4)
    ms = kms.get(k)
    hb = out['kernels'].get(k, {}).get('hbm_bytes')
    if ms and hb is not None:
        d['ms'] = ms
        d['hbm_bytes'] = hb
        d['hbm_GBps'] = round(hb / (ms * 1e-3) / 1e9, 1)
        d['hbm_frac_of_peak'] = round(hb / (ms * 1e-3) / 8e12, 4)
        d['limited_by'] = ('hbm' if d['hbm_frac_of_peak'] > 0.6 else 'valu' if d.get('valu_busy', 0) > 0.6 else
                           'lds' if d.get('lds_busy', 0) > 0.6 else 'latency')
    cnt['kernels'][k] = d
json.dump(cnt, open(os.path.join(dst, f'{tag}_counters.json'), 'w'), indent=1)
print('wrote', sorted(f for f in os.listdir(dst) if f.startswith(tag + '_')))

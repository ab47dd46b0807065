#!/usr/bin/env python3
"""Headline benchmark: photoelectrons/s (+ raw_records MB/s) of the photon -> raw_records hot path on a batch of
10^6-PE S2 instructions (BASELINE.json configs[2]), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--instructions M] [--workload s2|s2map|mixed|nveto|s1]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the whole hot path (S2 electron/photon Monte Carlo, SPE gains, template scatter-add, digitisation, ZLE,
record packing) over one batch of synthetic instructions that is already resident in HBM.  For N > 1 (weak scaling: every rank
simulates its own M instructions; event clusters are independent) the step ends with the RCCL gather of the packed
raw_records on rank 0 -- the exchange step north_star names --, posted as one grouped send/recv and overlapped with the next
batch's kernels; `value` is that gathered rate.  The same run then times K steps WITHOUT the gather (`value_no_gather`: every rank
keeps its records in its own HBM, the sharded delivery of DESIGN.md 6) and the gather alone (`config.gather_ms_per_step`).
--workload s2 (default) is the headline; s2map = the same batch under a position dependent S2 pattern map (bright tiles), mixed =
configs[3] (S1 + S2 pairs, PMT afterpulses and noise on, synthetic tables), nveto = configs[4] (optical instructions at 1 MHz on 120
channels), s1 = configs[1]; all run through the same step / gather code.  Prints ONE JSON line (rank 0).
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from wfsim_amd import workloads as W               # noqa: E402
from wfsim_amd.physics import instruction_params   # noqa: E402
from wfsim_amd.resource import Resource            # noqa: E402
from wfsim_amd.scheduler import schedule           # noqa: E402
from wfsim_amd.distributed import gather_records, wait_gather   # noqa: E402
from wfsim_amd.workloads import s2_batch, bench_config          # noqa: E402,F401  (tests import them from here)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
DEFAULT_INSTRUCTIONS = dict(s2=1000, s2map=1000, mixed=10000, nveto=200000, s1=10000)


def algorithmic_bytes(counts):
    """B_alg = 21 P + 4 S_raw + 244 R + 70 N_inst"""
    return 21 * counts['n_photons'] + 4 * counts['n_raw_samples'] + 244 * counts['n_records'] + 70 * counts['n_instructions']


def _cpu_worker(job):
    """one process of the CPU baseline: `n` instructions of the bench batch through the C oracle (test infrastructure: the checker
    timed as the reported CPU baseline, never part of the measured GPU path)"""
    first, n, seed = job
    from oracle.oracle import make_oracle
    cfg = bench_config(seed)
    res = Resource(cfg)
    ins = s2_batch(n, first)
    order, key, cluster = schedule(ins, cfg)
    ip = instruction_params(ins[order], cfg, res)
    orc = make_oracle(cfg)
    t0 = time.perf_counter()
    orc.simulate(ins[order], (first + order).astype(np.uint32), ip)
    rec = orc.pack_records()
    return orc.n_pe, len(rec) // 244, time.perf_counter() - t0


def cpu_baseline(seed, n_sample, all_cores=True):
    """The C oracle (CPU restatement) on a bounded sample of the same workload: one thread (the `value`), and one
    process per host core of this GPU's share (`all_cores`).  Runs before anything touches the GPU (worker processes)."""
    n_pe, n_rec, dt = _cpu_worker((0, n_sample, seed))
    out = dict(value=n_pe / dt, unit='photoelectrons/s', cores=1, kind='port',
               sample=f'{n_sample} S2 instructions of the bench batch (10^4 e-, ~10^6 PE each), {dt:.1f} s, '
                      f'{n_rec} records, C oracle single thread')
    if all_cores:
        import multiprocessing as mp
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
        per = max(2, n_sample // 4)
        t0 = time.perf_counter()
        with mp.get_context('fork').Pool(cores) as pool:
            parts = pool.map(_cpu_worker, [(1000 + k * per, per, seed) for k in range(cores)])
        wall = time.perf_counter() - t0
        out['all_cores'] = dict(value=sum(p[0] for p in parts) / wall, unit='photoelectrons/s', cores=cores,
                                sample=f'{cores} processes x {per} instructions, {wall:.1f} s wall incl. process start')
    # for context only (not measured on this box: the reference cannot travel): the reference's own Python path in no-JIT mode
    # ran the same S2 shape at 5.7e5 PE/s on one core of the build container; its numba mode cannot be
    # timed offline (numba does not initialise with the installed numpy)
    out['reference_python_nojit_note'] = dict(value=5.7e5, unit='photoelectrons/s', cores=1, where='build container, SURVEY.md 6, not this box')
    return out


def self_launch(n, timeout_s=1500):
    """`python bench.py --gpus N` without a launcher: N child processes with the environment torch.distributed.run would give
    them (one rank per GPU, rendezvous on 127.0.0.1), started before this process has initialised anything on the GPU -- children,
    never an exec.  Rank 0's stdout (the JSON line) is drained by a reader thread while the ranks run (a full pipe must not block it);
    a rank that dies takes the others down (they would wait in a collective for ever), and so does the overall timeout; the exit
    code is that of the rank that failed FIRST on its own, not of a rank this launcher terminated."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    first_failure, t0 = 0, time.monotonic()
    while any(p.poll() is None for p in procs):
        failed = [p.returncode for p in procs if p.poll() not in (None, 0)]
        timed_out = time.monotonic() - t0 > timeout_s
        if failed or timed_out:
            first_failure = first_failure or (failed[0] if failed else 124)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=30)
    sys.stdout.write(b''.join(out).decode())
    sys.stdout.flush()
    return first_failure or next((p.returncode for p in procs if p.returncode), 0)


def newest_profile(kind):
    """profiles/r<NN>_<kind>.json of the latest round that has one (tools/profile_round.sh + tools/make_profiles.py write them)"""
    best = None
    for f in glob.glob(os.path.join(ROOT, 'profiles', f'r*_{kind}.json')):
        m = re.match(rf'r(\d+)_{kind}\.json$', os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--instructions', type=int, default=None, help='instructions per GPU per step (default: 1000 S2 / 10000 mixed / 200000 nveto / 10000 S1)')
    ap.add_argument('--workload', choices=sorted(DEFAULT_INSTRUCTIONS), default='s2',
                    help='s2: the headline batch (BASELINE configs[2]); s2map: the same under a position dependent pattern map; mixed: configs[3]; nveto: configs[4]; s1: configs[1]')
    ap.add_argument('--cpu-sample', type=int, default=120, help="S2 instructions timed on the CPU oracle, ~13 s on one thread + ~4 s on all cores (0: skip)")
    ap.add_argument('--pmt-afterpulses', action='store_true', help='s2 workload with PMT afterpulses on (synthetic tables): a side measurement, not the headline')
    ap.add_argument('--exact-currents', action='store_true', help="fused_multiply_add off: add_current with numpy's separately rounded product and sum "
                    '(currents bit-exact with the reference; side measurement)')
    ap.add_argument('--set', action='append', default=[], metavar='KEY=JSON', help='config override, e.g. --set enable_electron_afterpulses=true (side measurements)')
    ap.add_argument('--no-copy-ceiling', action='store_true', help='skip the 1 GiB device-copy measurement (counter passes: it is not part of the batch)')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the gathered measurement, report only the rate with every rank keeping its records (then `value`)')
    ap.add_argument('--gather', action='store_true', help='(the default since round 4; kept for old command lines)')
    ap.add_argument('--sync-gather', action='store_true', help='do not overlap the gather with the next batch')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1 and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (nothing has touched the GPU yet) and relay rank 0's line
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    cpu = None
    if args.instructions is None:
        args.instructions = DEFAULT_INSTRUCTIONS[args.workload]
    if args.cpu_sample > 0 and world == 1 and args.workload == 's2':          # the CPU baseline is timed on rank 0 of the 1-GPU run only, before the GPU is touched
        cpu = cpu_baseline(3, args.cpu_sample)
    import torch
    import torch.distributed as dist
    # WFS_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with ONE card (all ranks on device 0, the
    # collectives on host tensors); the driver's runs use RCCL, one rank per GPU
    backend = os.environ.get('WFS_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = 0
    coll = 'cuda' if backend == 'nccl' else 'cpu'
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from wfsim_amd.engine import Engine
    M = args.instructions
    overrides = {}
    for kv in args.set:
        k, v = kv.split('=', 1)
        overrides[k] = json.loads(v)
    if args.exact_currents:
        overrides['fused_multiply_add'] = False
    if args.workload == 'nveto':
        cfg = W.nveto_config(seed=31, **overrides)
        res = Resource(cfg)
        ins, channels, timings = W.optical_instructions(M, 1000.0, 3 + rank)
        order, key, cluster = schedule(ins, cfg)
        eng = Engine(cfg, res, device=local_rank)
        eng.load_optical(ins[order], (rank * M + order).astype(np.uint32), cluster, key, channels, timings, int(1e6))
    else:
        if args.workload == 's2':
            cfg, ins = bench_config(seed=3, pmt_afterpulses=args.pmt_afterpulses, **overrides), s2_batch(M, first_gid=rank * M)
        elif args.workload == 's2map':
            cfg, ins = W.s2map_config(seed=3, pmt_afterpulses=args.pmt_afterpulses, **overrides), s2_batch(M, first_gid=rank * M, spread_xy=True)
        elif args.workload == 's1':
            cfg, ins = W.xenonnt_test_config(seed=2, **overrides), W.s1_batch(M, first_gid=rank * M)
        else:
            cfg, ins = W.mixed_config(seed=3, **overrides), W.mixed_batch(M, first_gid=rank * M)
        res = Resource(cfg)
        M = len(ins)
        order, key, cluster = schedule(ins, cfg)
        s_ins = ins[order]
        gid = (rank * M + order).astype(np.uint32)          # run-wide instruction ids: streams do not depend on the sharding
        eng = Engine(cfg, res, device=local_rank)
        ip = instruction_params(s_ins, cfg, res, device_maps=eng.device_maps)
        eng.load_instructions(s_ins, gid, cluster, key, ip)   # inputs resident in HBM before the timed region

    pending = None        # gather of the previous step, still in flight

    def step(gather, profile=False):
        """compute one batch; the gather of the previous batch's records overlaps it (RCCL runs on its own stream)"""
        nonlocal pending
        eng.set_profiling(profile)
        counts = eng.run()
        if world > 1 and gather:
            if pending is not None:
                wait_gather(pending[1])
            mine = torch.empty(counts['n_records'] * 244, dtype=torch.uint8, device='cuda')
            eng.copy_records_to_device(mine.data_ptr(), counts['n_records'])
            pending = gather_records(mine if coll == 'cuda' else mine.cpu(), dst=0, async_op=not args.sync_gather)
            if args.sync_gather:
                pending = None
        return counts

    def drain():
        nonlocal pending
        if pending is not None:
            wait_gather(pending[1])
            pending = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(gather):
        """W untimed warmup steps, then exactly K steps between two barrier + synchronize pairs; the maximum over the ranks"""
        for _ in range(args.warmup):
            step(gather)
        drain()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            c = step(gather)
        drain()                     # the last gather completes inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=coll, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, c

    gathered = world > 1 and not args.no_gather
    elapsed, counts = timed(gathered)                      # the headline region: with the gather when N > 1
    elapsed_ng, gather_ms = None, None
    if gathered:
        elapsed_ng, _ = timed(False)                       # the same K steps, every rank keeping its records in its own HBM
        # the gather alone: one batch's records from every rank to rank 0, nothing else running
        mine = torch.empty(counts['n_records'] * 244, dtype=torch.uint8, device='cuda')
        eng.copy_records_to_device(mine.data_ptr(), counts['n_records'])
        src = mine if coll == 'cuda' else mine.cpu()
        gather_records(src, dst=0)
        barrier()
        tg = time.perf_counter()
        n_g = 5
        for _ in range(n_g):
            gather_records(src, dst=0)
        barrier()
        tg = torch.tensor([time.perf_counter() - tg], device=coll, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather_ms = 1e3 * float(tg.item()) / n_g
        del mine, src
    if world > 1:
        tot = torch.tensor([counts['n_pe'], counts['n_records'], counts['n_photons']], device=coll, dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_pe, total_rec, total_ph = (int(x) for x in tot.tolist())
    else:
        total_pe, total_rec, total_ph = counts['n_pe'], counts['n_records'], counts['n_photons']

    # PCIe-inclusive rate (never the headline value): steps whose records also travel to (pinned) host memory, the copy of
    # batch k on the engine's copy stream under the kernels of batch k + 1 -- what ChunkRawRecords does with its record buffer
    from wfsim_amd.ministrax import raw_record_dtype
    host_bufs = [np.empty(int(counts['n_records'] * 1.05) + 1024, dtype=raw_record_dtype()) for _ in range(2)]
    pinned = all([eng.pin(b) for b in host_bufs])
    barrier()
    n_pcie = max(2, min(args.steps, 6))
    t1 = time.perf_counter()
    for k in range(n_pcie):
        c = step(False)
        eng.wait_records()                      # the previous batch's copy (it ran under this batch's kernels)
        eng.records_into_async(host_bufs[k & 1], c['n_records'])
    eng.wait_records()
    pcie_ms = 1e3 * (time.perf_counter() - t1) / n_pcie
    eng.unpin_all()
    del host_bufs

    # one extra (untimed) profiled step: HIP-event duration of every kernel on the engine's stream
    counts = step(False, profile=True)
    ktimes = eng.kernel_times()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    # measured ceiling next to the spec peak: a plain device-to-device copy of 1 GiB (bytes read + written per second)
    copy_gbs = None
    if not args.no_copy_ceiling:
        src_buf = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); dst_buf = torch.empty_like(src_buf)
        for _ in range(2):
            dst_buf.copy_(src_buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst_buf.copy_(src_buf)
        e1.record(); torch.cuda.synchronize()
        copy_gbs = 2 * (1 << 30) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src_buf, dst_buf
    ms_per_step = 1e3 * elapsed / args.steps
    dom = max(ktimes, key=lambda k: ktimes[k][0])
    dom_ms, dom_launches = ktimes[dom]
    b_alg = algorithmic_bytes(counts)
    # HBM bytes per launch of the dominant kernel and what limits it, from the committed PMC profiles of this command (the newest round's
    # profiles/r<NN>_traffic.json / _counters.json, written by tools/profile_round.sh + tools/make_profiles.py); only for the profiled workload
    traffic, limiter, valu_util, prof_src = None, None, None, None
    if args.workload == 's2' and M == 1000 and not overrides and not args.pmt_afterpulses:
        for kind in ('traffic', 'counters'):
            f = newest_profile(kind)
            try:
                kc = json.load(open(f))['kernels'][dom]
            except (OSError, KeyError, TypeError):
                continue
            prof_src = os.path.basename(f)[:3]
            if kind == 'traffic':
                traffic = kc.get('hbm_bytes')
            else:
                limiter, valu_util = kc.get('limited_by'), kc.get('valu_busy')
    achieved = b_alg / (dom_ms / dom_launches * 1e-3) / 1e9
    # the dominant kernel's OWN algorithmic bytes (the pulse kernel reads every photon record once and adds into the raw
    # accumulators once; the generator writes every photon record once): what "achieved" would be without crediting one
    # kernel with the whole pipeline's bytes
    # (k_s2_tile makes its photons in registers: what it moves are the tile buffers' samples, written once)
    own = {'k_pulse_dense': 8 * counts['n_photons'] + 4 * counts['n_raw_samples'], 'k_photon_fill': 8 * counts['n_photons'],
           'k_s2_tile': 4 * counts['n_raw_samples']}.get(dom)
    desc = {'s2': f'{M} S2 instructions per GPU, 1e4 electrons each (~1e6 PE), 494 PMTs, z=-10 cm, s2_secondary_sc_gain=100, noise/afterpulses off (BASELINE configs[2])',
            's2map': f'{M} S2 instructions per GPU as configs[2], spread over the TPC under a synthetic position dependent S2 pattern map (device evaluated; the PMT above '
                     f'the event takes ~5 % of the light) -- SIDE MEASUREMENT',
            'mixed': f'{M} instructions per GPU: S1 (3000 quanta) + S2 (1500 e-) pairs over the TPC, PMT afterpulses and noise on (synthetic tables), 494 PMTs (BASELINE configs[3])',
            'nveto': f'{M} optical nVeto instructions per GPU at 1 MHz, ~10 photons each, 120 channels (BASELINE configs[4])',
            's1': f'{M} S1 instructions per GPU, ~200 PE each, 494 PMTs (BASELINE configs[1])'}[args.workload]
    if args.pmt_afterpulses and args.workload in ('s2', 's2map'):
        desc += ' -- SIDE MEASUREMENT: PMT afterpulses ON (synthetic tables)'
    if overrides:
        desc += f' -- SIDE MEASUREMENT: config overrides {json.dumps(overrides, sort_keys=True)}'
    gather_desc = 'none (one GPU)' if world == 1 else ('none: every rank keeps its records in its own HBM (--no-gather)' if not gathered else
                                                       ('rccl' if backend == 'nccl' else backend) + ' send/recv to rank 0' + (', synchronous' if args.sync_gather else ', overlapped with the next batch'))
    out = dict(
        metric='photoelectrons/sec + raw_records MB/s, 10^6-PE S2 batch' + (', raw_records gathered on rank 0' if gathered else ''),
        value=total_pe * args.steps / elapsed,
        unit='photoelectrons/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
        higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f64', data='synthetic',
        raw_records_MB_per_s=total_rec * 244 * args.steps / elapsed / 1e6,
        config=dict(workload=desc, arithmetic='add_current with fused multiply-adds (one rounding per term)' if cfg.get('fused_multiply_add', True) else "add_current with numpy's two roundings per term (exact currents)",
                    instructions_per_gpu=M, instructions_per_s=M * world * args.steps / elapsed, pe_per_step=total_pe, photons_per_step=total_ph, records_per_step=total_rec,
                    gather=gather_desc,
                    ms_per_step_incl_d2h_of_records=pcie_ms, d2h='pinned host buffers, copy of batch k overlapped with batch k + 1' if pinned else 'pageable host buffers'),
        # bound: the roofline these numbers are priced on (HBM bytes: there is no dense contraction for MFMA).  What actually
        # limits the dominant kernel is in `limited_by` (SQ counters: f64 VALU issue + exposed latency, not HBM).
        roofline=dict(bound='hbm', kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                      traffic=traffic, algorithmic_bytes_per_launch=b_alg, kernel_ms=dom_ms / dom_launches,
                      limited_by=limiter, valu_busy=valu_util, counters_from=prof_src,
                      kernel_own_bytes=own, kernel_own_frac=(own / (dom_ms / dom_launches * 1e-3) / 1e9 / HBM_PEAK_GBS) if own else None,
                      pipeline_frac=b_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      measured_copy_GBs=copy_gbs, frac_of_measured_copy=(achieved / copy_gbs) if copy_gbs else None,
                      kernels_ms={k: round(v[0], 4) for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1][0])}),
    )
    if gathered:
        out['value_no_gather'] = total_pe * args.steps / elapsed_ng
        out['config']['ms_per_step_no_gather'] = 1e3 * elapsed_ng / args.steps
        out['config']['gather_ms_per_step'] = gather_ms
        out['config']['gather_bytes_per_step'] = total_rec * 244
    if cpu is not None:
        out['cpu_baseline'] = cpu
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

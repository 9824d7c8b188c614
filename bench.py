#!/usr/bin/env python3
"""Benchmark of the distortion hot path on MI355X: BASELINE config 3.

Workload (per GPU): B independent 2048x2048 RGB page images, each with its own ``camera_cubic_curve`` state
(config from the reference-compatible generator at level 5, seed = image index), through
    image-grid remap  ->  gaussian_blur(sigma=1.0, k=5)  ->  color_shift(delta=37)  ->  gaussion_noise(std=10)
with the images resident in HBM before the timed region.  Since round 5 a step starts from the CONFIGS: the camera_cubic_curve states
of the batch (CameraModel, 2-D -> 3-D lift, projection, shift, rounding: the reference's generate_state) are built inside every
step, scalars in C on the host and vertices on the device (ChainBatch.add_config -> vkx_camera_states_dev); ``--states host`` keeps
the lattices of host-built states resident as in rounds 1 - 4 (reported beside as ``lattices_resident``).  The noise of image i is the
reference's: np.round(default_rng(5000 + i).normal(0, 10, shape)) -- drawn ON THE DEVICE from that numpy stream, value for
value, INSIDE every timed step (vkx_np_draw_batch_dev: PCG64 jump-ahead + ziggurat); no host-generated plane exists.
A "step" is one pass over the whole batch: build the states, draw the noise of every image (it stays in the generator's tile
slots), run the chain (k_chain_fused adds the noise, the chain's last member, from those slots).  Images shard across GPUs without any exchange
(one process per GPU, weak scaling: every GPU processes its own B images).

``python bench.py --gpus N`` from a plain shell launches its own ranks (torch.distributed.run on 127.0.0.1, the way the
reference's pool forks its own workers, vkit/utility/pool.py:153-243); under torch.distributed.run it is one of the ranks.

Prints ONE JSON line on rank 0 (see the driver contract in the task description):
  value      = source megapixels (H*W per image) processed per second by all ranks together
  roofline   = the kernel with the largest share of the step, algorithmic bytes / HIP-event duration against the 8 TB/s HBM
               peak; roofline.step prices the whole step (every kernel of it) against the same 3S + 3D, roofline.kernels
               lists every kernel of the step
  cpu_baseline = the CPU oracle (port of the reference arithmetic, 1 thread) on a bounded sample, rank 0, N=1 only
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes and VALU instructions per image from the PMC passes of the current kernel sources.  tools/record.sh rewrites this
# file together with the sha256 of the sources it profiled; a line measured on other sources reports traffic = null instead
# of a stale figure.
TRAFFIC_FILE = os.path.join('profiles', 'current_traffic.json')
KERNEL_SOURCES = ('vkit_amd/csrc/fused.hip', 'vkit_amd/csrc/nprand.hip', 'vkit_amd/csrc/vkx_cell.h', 'vkit_amd/csrc/vkx_color.h',
                  'vkit_amd/csrc/vkx_internal.h')


def kernel_source_digest():
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]

BLUR_SIGMA = 1.0
HUE_DELTA = 37
NOISE_STD = 10.0
LEVEL = 5


def _noise_plane(args):
    seed, shape = args
    return np.round(np.random.default_rng(seed).normal(0, NOISE_STD, shape)).astype(np.int16)


def make_config(index, size):
    from numpy.random import default_rng
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), LEVEL)
    return gen((size, size), default_rng(index))


def make_state(index, size):
    """The host operator's state (numpy): the CPU baseline, the resident-lattice leg and the check of the device-built lattices."""
    from vkit_amd.mechanism import distortion as D
    return D.camera_cubic_curve.generate_state(make_config(index, size), (size, size))


def _oracle_noise(O, seed, shape):
    """The int16 plane of the step, drawn by the oracle's restatement of numpy's stream (as fast as numpy itself)."""
    words = O.np_state_words(np.random.default_rng(seed))
    plane, _, _ = O.np_normal_i16(words, int(np.prod(shape)), NOISE_STD)
    return plane.reshape(shape)


def cpu_baseline(size, n_images):
    """The oracle (CPU restatement of the reference arithmetic) on the first ``n_images`` images, one thread; like the GPU
    step it draws the noise plane inside the timed region."""
    import oracle as O
    states = [make_state(i, size) for i in range(n_images)]
    images = [np.random.default_rng(1000 + i).integers(0, 256, (size, size, 3), dtype=np.uint8) for i in range(n_images)]
    O.lib()
    t0 = time.perf_counter()
    checksum = 0
    for i, (img, st) in enumerate(zip(images, states)):
        noise = _oracle_noise(O, 5000 + i, tuple(st.result_shape) + (3,))
        mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
        out = O.remap(img, mx, my)
        out = O.gaussian_blur(out, 5, BLUR_SIGMA)
        out = O.color_shift_rgb(out, HUE_DELTA)
        out = O.add_noise_i16(out, noise)
        checksum += int(out[::97, ::89].sum())
    dt = time.perf_counter() - t0
    return {
        'value': n_images * size * size / dt / 1e6,
        'unit': 'Mpixels/s',
        'cores': 1,
        'kind': 'port',
        'sample': f'{n_images} images of the same workload, {dt:.1f} s on 1 thread of {os.cpu_count()} host cores; like the GPU step '
                  f'the sample draws its noise plane from the numpy stream inside the timed region (since round 3; the round-2 figure '
                  f'of 17 Mpx/s took the planes as given), then grid->map, remap, blur, hue shift, noise add',
    }


def _cpu_worker(rank, n_procs, per_proc, size, barrier, queue):
    """One process of the all-cores CPU leg: prepares its images, meets the others at the barrier, runs the oracle."""
    import oracle as O
    idx = [rank * per_proc + j for j in range(per_proc)]
    states = [make_state(i, size) for i in idx]
    images = [np.random.default_rng(1000 + i).integers(0, 256, (size, size, 3), dtype=np.uint8) for i in idx]
    O.lib()
    barrier.wait()
    t0 = time.time()
    for i, img, st in zip(idx, images, states):
        noise = _oracle_noise(O, 5000 + i, tuple(st.result_shape) + (3,))
        mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
        O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(img, mx, my), 5, BLUR_SIGMA), HUE_DELTA), noise)
    queue.put((t0, time.time()))


def cpu_baseline_all_cores(size, n_procs, per_proc):
    """The same oracle, one single-threaded process per core, all processes timed between a common barrier and the
    last one to finish."""
    ctx = mp.get_context('spawn')
    barrier, queue = ctx.Barrier(n_procs), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, n_procs, per_proc, size, barrier, queue)) for r in range(n_procs)]
    for p in procs:
        p.start()
    spans, deadline = [], time.time() + 300
    while len(spans) < n_procs and time.time() < deadline:
        try:
            spans.append(queue.get(timeout=1.0))
        except Exception:                      # queue.Empty: keep waiting while every worker is alive or done cleanly
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(timeout=5)
        if p.is_alive():
            p.terminate()
    if len(spans) < n_procs:
        return None
    dt = max(e for _, e in spans) - min(b for b, _ in spans)
    return {'value': n_procs * per_proc * size * size / dt / 1e6, 'unit': 'Mpixels/s', 'cores': n_procs,
            'host_cores': os.cpu_count(),
            'sample': f'{n_procs} single-threaded processes (one per host core) x {per_proc} images, {dt:.1f} s'}


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """``python bench.py --gpus N`` without a launcher: start the N ranks (one per GPU) under torch.distributed.run on the
    loopback address and hand their output through -- rank 0 prints the JSON line; the exit code is the launcher's (non-zero
    when any rank fails)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


def _tool_json(name, device_index, timeout, extra=()):
    """A tools/ script in a process of its own (a clean HIP runtime); its last stdout line is JSON."""
    try:
        proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', name)] + list(extra), capture_output=True, text=True,
                              timeout=timeout, env=dict(os.environ, VKX_DEVICE=str(device_index)))
        if proc.returncode != 0:
            return {'error': (proc.stderr or proc.stdout)[-400:]}
        return json.loads(proc.stdout.strip().splitlines()[-1])
    except Exception as exc:                  # the headline must not depend on a side leg
        return {'error': repr(exc)}


def dropin_mode(args, group, rank, world, device_index, first, affinity):
    """The PCIe-inclusive leg at N ranks: every rank sends ``--batch`` pages per step through ``HostPipeline.submit_chain`` -- page-
    locked host arrays in, page-locked host views out, 8 lanes --, camera_cubic_curve remap + blur + hue + the numpy noise stream of
    the page drawn on the device (value-exact).  Weak scaling like the resident mode; the link and the host's memory bandwidth are
    what the ranks share."""
    from vkit_amd import _native, shard
    from vkit_amd.hostpipe import HostPipeline
    B, size = args.batch, args.size
    ctx = _native.Context(device_index)
    n_img = min(B, 8)
    pinned = []
    for j in range(n_img):
        a = ctx.pinned_empty((size, size, 3), np.uint8)
        a[...] = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
        pinned.append(a)
    states = [make_state(first + j, size) for j in range(n_img)]
    pipe = HostPipeline(ctx)
    result_bytes = [int(np.prod(st.result_shape)) * 3 for st in states]

    def step():
        tickets = []
        for k in range(B):
            i = k % n_img
            tickets.append(pipe.submit_chain(pinned[i], states[i], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD,
                                             noise_rng=np.random.default_rng(5000 + first + i)))
            if k >= 7:
                pipe.result(tickets[k - 7])
        pipe.drain()

    def sync():
        pipe.drain()
        ctx.sync()

    elapsed = shard.timed_steps(group, step, steps=args.steps, warmup=args.warmup, device_sync=sync)
    group.close()
    pipe.close()
    if rank != 0:
        return
    total_px = B * size * size * world * args.steps
    link_bytes = sum((size * size * 3 + result_bytes[k % n_img]) for k in range(B)) * world * args.steps
    print(json.dumps({
        'metric': 'Mpixels/s (2048^2 RGB, geo+photo chain), host arrays in and out (PCIe inclusive)',
        'mode': 'dropin', 'value': total_px / elapsed / 1e6, 'unit': 'Mpixels/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'config': {'workload': f'C3 chain per page through HostPipeline.submit_chain (8 lanes): page-locked {size}x{size}x3 page in, '
                               f'result out, noise drawn on the device from the page\'s numpy stream; {B} pages per step and GPU',
                   'batch_per_gpu': B, 'affinity': affinity},
        'link_gb_per_s_all_ranks': link_bytes / elapsed / 1e9,
        'note': 'never the headline: the resident mode (default) is; this leg exists so that the first 8-GPU run yields the '
                'PCIe-inclusive curve beside the resident one',
    }))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200,
                    help='timed passes over the batch (the default keeps the timed region above 2 s)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU')
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--cpu-sample', type=int, default=40, help='images timed on the CPU oracle (rank 0, N=1)')
    ap.add_argument('--cpu-procs', type=int, default=-1,
                    help='processes of the all-cores CPU leg (-1 = one per host core, 0 = skip)')
    ap.add_argument('--verify', type=int, default=4,
                    help='images of the batch checked against the oracle (spread over the batch, the last one included)')
    ap.add_argument('--extra-legs', type=int, default=1,
                    help='at N=1 also measure the other BASELINE configs (C2, C4, C5), the other noise modes and the drop-in paths '
                         '(reported beside, never as value)')
    ap.add_argument('--noise-planes', type=int, default=0,
                    help='how the device-drawn numpy stream reaches the chain: 0 = the generator\'s tile slots, read by k_chain_fused '
                         '(default); 1 = an int16 plane per image in HBM, added inside k_chain_fused (rounds 1 - 2); 2 = added to the '
                         'chain output by the generator\'s placement pass (round 3)')
    ap.add_argument('--states', default='device', choices=('device', 'host'),
                    help='device (default): the timed step starts from the CONFIGS -- every step builds the camera_cubic_curve states of '
                         'the batch on the device (ChainBatch.add_config -> vkx_camera_states_dev), as the reference builds the state '
                         'inside Distortion.distort; host: the lattices of host-built states are resident before the timed region '
                         '(rounds 1 - 4)')
    ap.add_argument('--lanes', type=int, default=1,
                    help='HIP streams the batch is dealt over (ChainLanes).  1: every kernel interval of the step is disjoint and '
                         'the per-kernel figures add up to the step; 2 hides the microsecond kernels of one lane under the other '
                         'lane\'s (-1 %% of the step)')
    ap.add_argument('--mode', default='resident', choices=('resident', 'dropin'),
                    help='resident (default, the headline): images resident in HBM.  dropin: the PCIe-inclusive curve -- every rank '
                         'sends its images through HostPipeline.submit_chain, host arrays in and host arrays out (page-locked), the '
                         'noise of every image drawn on the device from its numpy stream; never the headline value')
    ap.add_argument('--dry-run', action='store_true',
                    help='rendezvous, barriers and the MAX reduction of the timing protocol only, over gloo, no GPU: the N > 1 path on '
                         'a box without GPUs (tests/test_bench_launch.py)')
    ap.add_argument('--noise-workers', type=int, default=0, help='unused since round 3 (the planes are drawn on the device); kept for old command lines')
    args = ap.parse_args()

    from vkit_amd import shard
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank, local_rank, world = shard.world_from_env()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    if args.dry_run:
        group = shard.Group(backend='gloo' if (world > 1 or 'MASTER_ADDR' in os.environ) else None)
        first, count = shard.weak_span(args.batch, rank)
        elapsed = shard.timed_steps(group, lambda: time.sleep(0.01 * (1 + rank)), steps=args.steps, warmup=args.warmup,
                                    device_sync=lambda: None)
        units = group.sum_int(count * args.steps)
        group.close()
        if rank == 0:
            print(json.dumps({'dry_run': True, 'mode': args.mode, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': elapsed / args.steps * 1e3, 'units': units, 'first_image_of_last_rank': (world - 1) * args.batch}))
            sys.stdout.flush()
        return

    # The libraries are prebuilt in-tree (__graft_entry__.build()); build here only when they are missing, on rank 0,
    # while the other ranks wait -- never relink a library another rank may be loading.
    libs = [os.path.join(ROOT, 'vkit_amd', 'libvkx.so'), os.path.join(ROOT, 'oracle', '_build', 'libvkx_oracle.so')]
    if not all(os.path.exists(p) for p in libs):
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            deadline = time.time() + 900
            while not all(os.path.exists(p) for p in libs) and time.time() < deadline:
                time.sleep(1.0)
            time.sleep(2.0)

    B, size = args.batch, args.size
    first, _ = shard.weak_span(B, rank)  # global index of this rank's first image

    # ---- host-side setup (no GPU yet): states ---------------------------------------------------------------------
    t_setup = time.perf_counter()
    configs = [make_config(first + j, size) for j in range(B)]
    from_configs = args.states == 'device' and args.lanes == 1 and args.noise_planes == 0
    states = None if from_configs else [make_state(first + j, size) for j in range(B)]

    import torch
    # process -> GPU by the pool's rule (local_rank % visible GPUs): one rank per GPU on the driver's N-GPU node; on a
    # box with fewer GPUs than ranks (the 2-ranks-on-1-GPU rendezvous check, profiles/r4_selflaunch_2ranks_1gpu.log) the
    # ranks share devices and the collectives go through gloo, because RCCL refuses two ranks on one device
    n_dev = torch.cuda.device_count()
    device_index = shard.device_for(local_rank, n_dev)
    shared_devices = world > n_dev
    torch.cuda.set_device(device_index)
    backend = None
    if world > 1 or 'MASTER_ADDR' in os.environ:
        backend = os.environ.get('VKX_DIST_BACKEND') or ('gloo' if shared_devices else 'nccl')
    group = shard.Group(backend=backend, device=torch.device('cuda', device_index))

    from vkit_amd import _native
    from vkit_amd.batch import ChainBatch, ChainLanes
    # CPU placement next to the GPU: the cores of the GPU's NUMA node, shared out among the ranks whose GPUs hang off the same node
    try:
        bus_ids = [_native.device_pci_bus_id(d) for d in range(n_dev)]
        pos, sharing = shard.ranks_sharing_numa(bus_ids, device_index)
        bound = shard.bind_to_device_numa(bus_ids[device_index], pos, sharing if world > 1 else 1)
        affinity = {'pci_bus_id': bus_ids[device_index], 'cpus': len(bound) if bound else None,
                    'numa_share': f'{pos + 1} of {sharing}' if bound and world > 1 else None}
    except Exception as exc:                    # placement is an optimisation: never a reason to fail
        affinity = {'error': repr(exc)}
    if args.mode == 'dropin':
        return dropin_mode(args, group, rank, world, device_index, first, affinity)
    ctx = _native.Context(device_index)
    noise_mode = {0: 'tiles', 1: 'planes', 2: 'late'}[args.noise_planes]
    batch = ChainLanes(device_index, lanes=args.lanes, stream_noise_mode=noise_mode)
    images = []
    for j in range(B):
        image = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
        images.append(image)
        if from_configs:
            batch.lanes[0].add_config(image, configs[j], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD,
                                      noise_rng=np.random.default_rng(5000 + first + j))
            batch._where.append((0, j))
        else:
            batch.add(image, states[j], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD,
                      noise_rng=np.random.default_rng(5000 + first + j))
    t_setup = time.perf_counter() - t_setup

    def full_sync():
        batch.sync()
        ctx.sync()
        torch.cuda.synchronize()

    # ---- warmup, then K timed steps (barrier + device sync on both sides, MAX over ranks) -------------------------
    for _ in range(args.warmup):
        batch.run()
    full_sync()
    # HIP events on the launch streams, live in the timed region, around the two LARGE kernels of the step (k_np_draw,
    # k_chain_fused: timing level 2).  An event pair costs a few microseconds of stream time: around every one of the step's
    # dozen microsecond kernels the pairs themselves were 0.1 ms of a 16.6 ms step, so the complete per-kernel table comes from a
    # breakdown pass of the same step right after the timed region (timing level 1).
    batch.set_timing(2 if os.environ.get('VKX_BENCH_TIMING', '1') != '0' else False)
    elapsed = shard.timed_steps(group, batch.run, steps=args.steps, warmup=0, device_sync=full_sync)
    major_times = batch.timings()
    batch.set_timing(False)
    group.close()  # every rank is past the closing barrier and the MAX reduction: nothing collective is left
    breakdown_steps = max(1, min(args.steps, 20))
    kernel_times = {}
    if major_times:
        batch.set_timing(True)
        for _ in range(breakdown_steps):
            batch.run()
        full_sync()
        detail = batch.timings()
        batch.set_timing(False)
        # per step: the large kernels from the timed region, the others from the breakdown pass
        for name, (ms, launches) in detail.items():
            kernel_times[name] = (ms / breakdown_steps * args.steps, launches / breakdown_steps * args.steps, 'breakdown pass')
        for name, (ms, launches) in major_times.items():
            kernel_times[name] = (ms, launches, 'timed region')

    state_build = None
    if from_configs:
        cb = batch.lanes[0]
        state_build = {'host_ms_per_step': cb.state_build_s / max(cb.state_builds, 1) * 1e3, 'builds': cb.state_builds,
                       'note': 'host time of ChainBatch._build_states per step, INSIDE the timed region: 256 x vkx_camera_model_host '
                               '(C), one k_camera_states launch on the side stream, the wait for the result shapes, whole-array layout '
                               'of destinations / tile buffers / stream jobs; it runs while the compute stream still executes the '
                               'previous step'}
        # the host operator's states: CPU legs below, and how many device-built lattices equal them on THIS box
        states = [make_state(first + j, size) for j in range(B)]
        equal = 0
        for j in sorted({0, B // 3, (2 * B) // 3, B - 1}):
            sv, dv = cb.lattices(j)
            equal += int(np.array_equal(sv, states[j].src_image_grid.vertices) and np.array_equal(dv, states[j].dst_image_grid.vertices)
                         and tuple(states[j].result_shape) == cb._dst_shapes[j])
        state_build['lattices_equal_host_operator'] = f'{equal} of 4 checked'
    noise_jobs = [(5000 + first + j, tuple(states[j].result_shape) + (3,)) for j in range(B)]

    # ---- the chain alone on noise already in HBM (r2's headline mode, planes resident), for continuity --------------------
    planes_resident = None
    if world == 1 and args.extra_legs:
        rsteps = max(1, min(args.steps, 50))
        pbatch = ChainBatch(ctx, stream_noise_planes=True)
        for j in range(B):
            pbatch.add(images[j], states[j], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD,
                       noise_rng=np.random.default_rng(5000 + first + j))
        pbatch.run()
        full_sync()
        ctx.set_timing(True)
        ctx.reset_timings()
        t0 = time.perf_counter()
        for _ in range(rsteps):
            pbatch.run(draw_streams=False)
        full_sync()
        rdt = time.perf_counter() - t0
        pk = ctx.timings()
        ctx.set_timing(False)
        planes_resident = {'value': pbatch.source_pixels * rsteps / rdt / 1e6, 'unit': 'Mpixels/s', 'steps': rsteps,
                           'ms_per_step': rdt / rsteps * 1e3,
                           'kernels_ms_per_step': {k: round(v[0] / rsteps, 3) for k, v in sorted(pk.items())},
                           'note': 'the chain on int16 planes resident in HBM (drawn once, added inside k_chain_fused; no '
                                   'drawing inside the step): the mode the round-1 / round-2 headline was measured in'}
        pbatch.close()
    del images

    # ---- parity spot check of this very batch against the oracle (outside the timed region) -------------------------
    # the expected noise plane comes from numpy ITSELF (_noise_plane): the device-drawn samples have to equal it
    verified = 0
    if rank == 0 and args.verify > 0:
        import oracle as O
        picks = sorted({int(round(k * (B - 1) / max(args.verify - 1, 1))) for k in range(min(args.verify, B))} | {B - 1})
        for j in picks:
            st = states[j]
            img = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
            got = batch.result(j)
            if from_configs:      # the pixels against the oracle on the lattices the device built (those against the host operator: above)
                sv, dv = batch.lanes[0].lattices(j)
                shape = got.shape[:2]
            else:
                sv, dv, shape = st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape
            mx, my = O.grid_to_map(sv, dv, shape)
            want = O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(img, mx, my), 5, BLUR_SIGMA), HUE_DELTA),
                                   _noise_plane((5000 + first + j, tuple(shape) + (3,))))
            if not (got == want).all():
                raise SystemExit(f'bench: image {j} differs from the oracle ({int((got != want).sum())} bytes)')
            verified += 1

    if rank != 0:
        return

    src_px = batch.source_pixels            # per rank, per step
    dst_px = batch.result_pixels
    fallbacks = batch.stream_fallbacks
    lattices_resident = None
    # ---- round 4's headline mode: the same step with the lattices of host-built states resident before the timed region ------
    if world == 1 and args.extra_legs and from_configs:
        rsteps = max(1, min(args.steps, 30))
        main_ctx = batch.contexts[0]
        batch.lanes[0].close()            # the buffers of the timed batch; its context (streams, scratch) serves this leg
        lb = ChainBatch(main_ctx)
        for j in range(B):
            image = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
            lb.add(image, states[j], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD,
                   noise_rng=np.random.default_rng(5000 + first + j))
        lb.run()
        main_ctx.sync()
        main_ctx.set_timing(2)
        main_ctx.reset_timings()
        t0 = time.perf_counter()
        for _ in range(rsteps):
            lb.run()
        main_ctx.sync()
        ldt = time.perf_counter() - t0
        lk = main_ctx.timings()
        main_ctx.set_timing(False)
        lattices_resident = {'value': lb.source_pixels * rsteps / ldt / 1e6, 'unit': 'Mpixels/s', 'steps': rsteps,
                             'ms_per_step': ldt / rsteps * 1e3,
                             'kernels_ms_per_step': {k: round(v[0] / rsteps, 3) for k, v in sorted(lk.items())},
                             'note': 'the step of rounds 3 - 4: noise drawn inside the step, vertex lattices of host-built states '
                                     'resident in HBM before the timed region (state construction excluded)'}
        lb.close()

    batch.close()

    # ---- reported beside the headline, N=1 only -----------------------------------------------------------------------
    throughput_mode, dropin, other_configs = None, None, None
    if world == 1 and args.extra_legs:
        # (a) the same chain with the noise plane drawn on the device every step (vkx_noise_normal_i16_dev: the reference's
        #     distribution, not numpy's values) instead of the numpy stream
        tb = ChainBatch(ctx)
        for j in range(B):
            image = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
            tb.add(image, states[j], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD, noise_seed=5000 + first + j)
        tb.run()
        ctx.sync()
        ctx.set_timing(True)
        ctx.reset_timings()
        tsteps = max(1, min(args.steps, 50))
        t0 = time.perf_counter()
        for _ in range(tsteps):
            tb.run()
        ctx.sync()
        tdt = time.perf_counter() - t0
        tk = ctx.timings()
        ctx.set_timing(False)
        throughput_mode = {
            'value': tb.source_pixels * tsteps / tdt / 1e6, 'unit': 'Mpixels/s', 'steps': tsteps,
            'ms_per_step': tdt / tsteps * 1e3,
            'kernels_ms_per_step': {k: round(v[0] / tsteps, 3) for k, v in sorted(tk.items())},
            'note': 'noise planes drawn on the device every step (Philox2x32-10 + inverse-CDF table of round(N(0, std))): '
                    'the distribution of the reference, not the values of its numpy stream -- a separately labelled mode',
        }
        tb.close()
        # (b) host arrays in, host arrays out, and (c) the other BASELINE configs: tools/ scripts in processes of their own (a
        #     clean HIP runtime: the stream -> hardware-queue mapping the overlapped pipeline leans on is a property of the
        #     process)
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            out_path = os.path.join(tmp, 'dropin.json')
            try:
                proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dropin.py'), out_path, '--quick'],
                                      capture_output=True, text=True, timeout=240,
                                      env=dict(os.environ, VKX_DEVICE=str(device_index)))
                if proc.returncode == 0 and os.path.exists(out_path):
                    with open(out_path) as fin:
                        dropin = json.load(fin)
                else:
                    dropin = {'error': (proc.stderr or proc.stdout)[-400:]}
            except Exception as exc:          # the headline must not depend on this leg
                dropin = {'error': repr(exc)}
        other_configs = {
            'C2_similarity_mls_remap_2048_batch64': _tool_json('c2.py', device_index, 180),
            'C4_page_synth_1024_64_layers_batch64': _tool_json('c4.py', device_index, 240),
            'C5_shared_grid_4096_three_elements': _tool_json('c5.py', device_index, 180),
            'poisson_noise_1024': _tool_json('poisson_probe.py', device_index, 180, ('1024',)),
            'note': 'BASELINE.json configs[1], [3], [4] on this box and clock, device resident (tools/c2.py, c4.py, c5.py): never '
                    'the headline value; poisson_noise_1024: rng.poisson(image) drawn on the device against numpy itself (values and stream '
                    'position, tools/poisson_probe.py), host arrays in and out',
        }

    total_px = src_px * world * args.steps
    value = total_px / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline: every kernel of the step, the dominant one on top --------------------------------------------------
    S, D = src_px / B, dst_px / B           # mean source / result pixels per image
    chain_bytes = (3 * S + 3 * D) * B       # the fully fused geo + photo chain, per step
    samples = 3 * D * B                     # noise samples per step
    # algorithmic bytes per STEP of the kernels that have a figure; None: no mandatory HBM traffic worth a roofline
    # (the stream kernels are compute: 128-bit LCG + ziggurat)
    algorithmic = {
        'k_chain_fused': chain_bytes,
        'k_np_draw': 2 * samples if noise_mode != 'late' else None,          # the int16 sample it has to leave somewhere
        'k_np_apply': 2 * samples,                                           # late: 1 byte read + 1 byte written per sample
        'k_np_place': 2 * samples,
    }
    tj, traffic_source = None, None
    tpath = os.path.join(ROOT, TRAFFIC_FILE)
    if size == 2048 and os.path.exists(tpath):
        with open(tpath) as fin:
            tj = json.load(fin)
        if tj.get('kernel_source_digest') != kernel_source_digest():
            traffic_source = (f'{TRAFFIC_FILE} was measured on other kernel sources (digest '
                              f'{tj.get("kernel_source_digest")} != {kernel_source_digest()}): not used')
            tj = None
        elif tj.get('noise_mode', 'tiles') != noise_mode:
            traffic_source = f'{TRAFFIC_FILE} was measured in noise mode {tj.get("noise_mode")}: not used'
            tj = None
        else:
            traffic_source = f'{TRAFFIC_FILE} <- {tj.get("profile", "?")}'
    pmc = (tj or {}).get('kernels', {})
    kernels = {}
    for name, (ms, launches, where) in sorted(kernel_times.items()):
        per_step = ms / args.steps
        entry = {'ms_per_step': round(per_step, 4), 'launches_per_step': launches / args.steps,
                 'avg_launch_ms': ms / max(launches, 1), 'frac_of_step': per_step / ms_per_step, 'events_in': where}
        alg = algorithmic.get(name)
        if alg:
            entry['algorithmic_bytes_per_step'] = alg
            entry['achieved_gbs'] = alg / (per_step / 1e3) / 1e9 if per_step > 0 else 0.0
            entry['hbm_frac'] = entry['achieved_gbs'] / HBM_PEAK_GBS
        p = pmc.get(name)
        if p:
            # why a kernel sits where it sits: wavefront VALU instructions (PMC, same sources) at one issue per 4 cycles on
            # 1024 SIMDs (256 CUs x 4) at 2.4 GHz against the measured time; HBM bytes by FETCH_SIZE x 2 + WRITE_SIZE
            floor_ms = p['valu_insts_per_image'] * B / 1024 * 4 / 2.4e9 * 1e3
            entry['valu_issue_frac'] = floor_ms / per_step if per_step > 0 else None
            entry['valu_insts_per_wavefront'] = round(p.get('valu_insts_per_wavefront', 0))
            entry['hbm_traffic_bytes_per_step'] = p['hbm_bytes_per_image'] * B
        kernels[name] = entry
    if not kernels:      # VKX_BENCH_TIMING=0 (A/B runs of the step without the event pairs): no per-kernel figures
        print(json.dumps({'ms_per_step': ms_per_step, 'value': value, 'roofline': {'kernels_ms_per_step': {}},
                          'config': {'verified_against_oracle': verified}}))
        return
    dominant = max(kernels, key=lambda k: kernels[k]['ms_per_step'])
    dom = kernels[dominant]
    dom_alg = algorithmic.get(dominant) or chain_bytes
    dom_s = dom['ms_per_step'] / 1e3 / max(dom['launches_per_step'], 1e-9)          # average launch duration
    dom_bytes_per_launch = dom_alg / max(dom['launches_per_step'], 1e-9)
    achieved = dom_bytes_per_launch / dom_s / 1e9 if dom_s > 0 else 0.0
    kernel_sum_ms = sum(k['ms_per_step'] for k in kernels.values())
    np_ms = sum(v['ms_per_step'] for k, v in kernels.items() if k.startswith('k_np_'))
    chain_note = {
        'tiles': 'k_chain_fused is the whole fused geo + photo chain: remap, gaussian_blur, color_shift AND the gaussion_noise add, '
                 'whose samples it reads from the generator\'s tile slots (2 bytes per sample on top of the 3S + 3D numerator)',
        'planes': 'k_chain_fused is the whole fused geo + photo chain incl. the gaussion_noise add from an int16 plane (6 D read on top '
                  'of the 3S + 3D numerator)',
        'late': 'k_chain_fused ends with color_shift here: the noise member is added to its output by the generator (k_np_apply)',
    }[noise_mode]
    dom_traffic = dom.get('hbm_traffic_bytes_per_step')
    result = {
        'metric': 'Mpixels/s (2048^2 RGB, geo+photo chain)',
        'value': value,
        'unit': 'Mpixels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u8',
        'data': 'synthetic',
        'config': {
            'workload': f'C3 fused chain: camera_cubic_curve remap (level {LEVEL}) + gaussian_blur(sigma={BLUR_SIGMA}) + '
                        f'color_shift({HUE_DELTA}) + gaussion_noise(std={NOISE_STD}, the numpy stream default_rng(5000 + i) drawn on '
                        f'the device inside every step), ' + ('states built from the configs inside every step, ' if from_configs else '') +
                        f'{size}x{size}x3 uint8, batch {B} per GPU',
            'batch_per_gpu': B,
            'image': f'{size}x{size}x3',
            'mean_result_pixels': D,
            'noise_mode': noise_mode,
            'states': ('built on the device from the configs inside every step (vkx_camera_states_dev)' if from_configs
                       else 'host-built, lattices resident before the timed region'),
            'lanes': args.lanes,
            'sharding': f'{world} process(es), one per GPU, independent images, no collective' +
                        (f' (ranks share {n_dev} GPU(s): rendezvous over {backend})' if shared_devices else ''),
            'verified_against_oracle': verified,
            'affinity': affinity,
            'kernel_source_digest': kernel_source_digest(),
            'setup_s': round(t_setup, 1),
        },
        'roofline': {
            'bound': 'hbm',
            'bound_measured': 'valu',
            'kernel': dominant,
            'kernel_note': 'the kernel with the largest share of the step.  ' + chain_note,
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'avg_launch_ms': dom_s * 1e3,
            'algorithmic_bytes_per_launch': dom_bytes_per_launch,
            # the north star words the target as "HBM-read roofline": the bytes the chain must READ (the source, once)
            'read_frac': ((3 * S * B) / (kernels['k_chain_fused']['ms_per_step'] / 1e3) / 1e9 / HBM_PEAK_GBS
                          if 'k_chain_fused' in kernels else None),
            'traffic': dom_traffic / max(dom['launches_per_step'], 1e-9) if dom_traffic else None,
            'traffic_source': traffic_source,
            'traffic_note': 'HBM bytes per launch = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE of separate --pmc passes on the same '
                            'sources (digest checked), scaled to this batch',
            # the figure for "fused geo + photo chain" as a whole: every kernel of the step -- the stream kernels that draw the
            # noise included -- against the chain's 3S + 3D
            'step': {'algorithmic_bytes': chain_bytes, 'ms': ms_per_step, 'achieved': chain_bytes / (ms_per_step / 1e3) / 1e9,
                     'frac': chain_bytes / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS,
                     'kernel_ms': kernel_sum_ms,
                     'note': 'elapsed time of the step (barrier to barrier / steps), not a sum of kernel intervals: since round 5 the '
                             'microsecond kernels run on two side streams UNDER the large ones (vkx_chain_rgb_batch_np_dev), so '
                             'kernel_ms exceeds ms' +
                             ('' if args.lanes == 1 else '; with several lanes the kernel intervals overlap and kernel_ms exceeds ms')},
            'kernels': kernels,
            'kernels_ms_per_step': {k: round(v['ms_per_step'], 3) for k, v in kernels.items()},
        },
        'noise_stream': {
            'kernels_ms_per_step': {k: round(v['ms_per_step'], 3) for k, v in kernels.items() if k.startswith('k_np_')},
            'ms_per_step': np_ms,
            'samples_per_step': samples,
            'gsamples_per_s': samples / (np_ms / 1e3) / 1e9 if np_ms > 0 else None,
            'host_fallback_planes': fallbacks,
            'note': 'np.round(default_rng(5000 + i).normal(0, std, shape)).astype(int16) drawn on the device inside every step, value '
                    'for value numpy\'s (checked against numpy on the verified images): 128-bit LCG + ziggurat, VALU bound, not an '
                    'HBM-bound kernel',
        },
    }
    if state_build is not None:
        result['state_construction'] = state_build
    if lattices_resident is not None:
        result['lattices_resident'] = lattices_resident
    if planes_resident is not None:
        result['planes_resident'] = planes_resident
    if throughput_mode is not None:
        result['throughput_mode'] = throughput_mode
    if dropin is not None:
        result['dropin'] = dropin
    if other_configs is not None:
        result['other_configs'] = other_configs
    if world == 1:
        result['cpu_baseline'] = cpu_baseline(size, args.cpu_sample) if args.cpu_sample > 0 else None
        n_procs = (os.cpu_count() or 1) if args.cpu_procs < 0 else args.cpu_procs
        if n_procs > 1 and args.cpu_sample > 0:
            all_cores = cpu_baseline_all_cores(size, n_procs, 2)
            if all_cores is not None:
                result['cpu_baseline']['all_cores'] = all_cores
    else:
        result['cpu_baseline'] = None
    print(json.dumps(result))
    sys.stdout.flush()


if __name__ == '__main__':
    main()

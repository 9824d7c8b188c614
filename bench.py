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

# the CPUs this process was started on, before a rank binds itself to the cores next to its GPU (shard.bind_to_device_numa):
# the all-cores CPU leg runs on THESE
START_AFFINITY = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None

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


def host_cpu_facts():
    """What this process may actually run on: the affinity mask, the cgroup's CPU quota and cpuset (v2 and v1 paths), the load.
    os.cpu_count() counts the box's logical CPUs whether or not the container may use them."""
    facts = {'os_cpu_count': os.cpu_count(), 'sched_getaffinity_at_start': len(START_AFFINITY) if START_AFFINITY else None,
             'sched_getaffinity_now': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None}

    def _read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None

    cpu_max = _read('/sys/fs/cgroup/cpu.max')
    if cpu_max is None:
        quota, period = _read('/sys/fs/cgroup/cpu/cpu.cfs_quota_us'), _read('/sys/fs/cgroup/cpu/cpu.cfs_period_us')
        if quota is not None and period is not None:
            cpu_max = f'{quota} {period}' if int(quota) > 0 else f'max {period}'
    facts['cgroup_cpu_max'] = cpu_max
    if cpu_max and not cpu_max.startswith('max'):
        try:
            q, per = cpu_max.split()
            facts['cgroup_quota_cpus'] = int(q) / int(per)
        except ValueError:
            pass
    facts['cgroup_cpuset_effective'] = _read('/sys/fs/cgroup/cpuset.cpus.effective') or _read('/sys/fs/cgroup/cpuset/cpuset.effective_cpus')
    facts['loadavg'] = _read('/proc/loadavg')
    return facts


def _cpu_worker(rank, per_proc, size, points, barrier, queue, cpus):
    """One process of the all-cores CPU leg: prepares its images once, then for every sweep point ``k`` meets the others at
    the barrier and -- when its rank is below ``k`` -- runs the oracle on its images; the others sit the point out."""
    try:
        if cpus:
            os.sched_setaffinity(0, cpus)       # the launch-time mask, not the GPU-side NUMA binding the parent took later
        import oracle as O
        idx = [rank * per_proc + j for j in range(per_proc)]
        states = [make_state(i, size) for i in idx]
        images = [np.random.default_rng(1000 + i).integers(0, 256, (size, size, 3), dtype=np.uint8) for i in idx]
        O.lib()
        for k in points:
            barrier.wait()
            if rank >= k:
                continue
            t0 = time.time()
            for i, img, st in zip(idx, images, states):
                noise = _oracle_noise(O, 5000 + i, tuple(st.result_shape) + (3,))
                mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
                O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(img, mx, my), 5, BLUR_SIGMA), HUE_DELTA), noise)
            queue.put((k, rank, t0, time.time()))
    except Exception:                          # a broken barrier (the parent gave up) or a failed worker: leave quietly
        try:
            barrier.abort()
        except Exception:
            pass


def sweep_points(limit):
    """1, 2, 4, ... up to ``limit`` (the limit itself included)."""
    pts, k = [], 1
    while k < limit:
        pts.append(k)
        k *= 2
    pts.append(limit)
    return pts


def cpu_baseline_all_cores(size, n_procs, per_proc, single_thread_value=None, budget_s=150.0):
    """The same oracle as single-threaded processes, swept over 1, 2, 4, ... ``n_procs`` of them (one pool of processes, started
    and prepared once; each sweep point is timed from the common barrier to the last worker of the point).  Reports the best
    figure, the ``knee`` (fewest processes within 5 % of the best) and ``effective_cores`` = best / one process: what the box
    gives this workload however many logical CPUs it lists."""
    ctx = mp.get_context('spawn')
    points = sweep_points(n_procs)
    barrier, queue = ctx.Barrier(n_procs + 1), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, per_proc, size, points, barrier, queue, START_AFFINITY)) for r in range(n_procs)]
    for p in procs:
        p.start()
    sweep, t_begin = [], time.time()
    try:
        for k in points:
            if sweep and time.time() - t_begin > budget_s:
                break
            barrier.wait(timeout=300)            # every worker is prepared / done with the point before
            spans, deadline = [], time.time() + 300
            while len(spans) < k and time.time() < deadline:
                try:
                    spans.append(queue.get(timeout=1.0))
                except Exception:              # queue.Empty: keep waiting while every worker is alive
                    if any(p.exitcode not in (None, 0) for p in procs):
                        break
            if len(spans) < k:
                break
            dt = max(s[3] for s in spans) - min(s[2] for s in spans)
            sweep.append({'processes': k, 'value': round(k * per_proc * size * size / dt / 1e6, 2), 'seconds': round(dt, 2)})
    except Exception:
        pass
    finally:
        try:
            barrier.abort()                      # releases the workers still waiting for points that were not run
        except Exception:
            pass
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
    if not sweep:
        return None
    best = max(sweep, key=lambda e: e['value'])
    knee = min((e for e in sweep if e['value'] >= 0.95 * best['value']), key=lambda e: e['processes'])
    one = sweep[0]['value']
    out = {'value': best['value'], 'unit': 'Mpixels/s', 'cores': best['processes'], 'knee_processes': knee['processes'],
           'effective_cores': round(best['value'] / one, 1), 'one_process_value': one, 'sweep': sweep, 'host': host_cpu_facts(),
           'sample': f'single-threaded oracle processes x {per_proc} images each, swept over {[e["processes"] for e in sweep]} processes '
                     f'of one pool; value = the best point ({best["processes"]} processes, {best["seconds"]} s); effective_cores = best / '
                     f'the 1-process point: how many cores\' worth of this workload the box delivers (memory bandwidth, cgroup quota '
                     f'and SMT siblings included), whatever os.cpu_count() says'}
    if single_thread_value:
        out['one_process_vs_single_thread_leg'] = round(one / single_thread_value, 3)
    return out


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


def c4_layers(_native, page_index, size=1024, n_layers=64, lh=32, lw=512):
    """The layer list of synthetic page ``page_index``: gray background ``default_rng(i).integers(127, 256)``, 64 text-line
    layers of 32 x 512 at seeded positions, float32 alpha with ~30 % non-zero, glyph colour (10, 20, 30) -> (make_layer records, plan for the oracle)."""
    rng = np.random.default_rng(page_index)
    gray = int(rng.integers(127, 256))
    layers = [_native.make_layer((0, 0, size, size), 3, (gray, gray, gray))]
    plan = [((0, 0, size, size), (gray, gray, gray), None)]
    for _ in range(n_layers):
        alpha = (rng.random((lh, lw), dtype=np.float32) * (rng.random((lh, lw)) < 0.3)).astype(np.float32)
        box = (int(rng.integers(0, size - lh)), int(rng.integers(0, size - lw)), lh, lw)
        layers.append(_native.make_layer(box, 3, (10, 20, 30), alpha=alpha))
        plan.append((box, (10, 20, 30), alpha))
    return layers, plan


def _c4_api_worker(process_idx, num_processes, seed, device_index, size, seconds, ready, go, results):
    """One worker of the reference-API leg of --config c4: its own process, context and generator (the pool's rule), whole pages through
    the three step objects for ``seconds`` after the common start."""
    os.environ['VKX_DEVICE'] = str(device_index)
    from numpy.random import SeedSequence, default_rng
    from vkit_amd import _native
    from vkit_amd.pipeline import text_detection as T
    from vkit_amd.pipeline.text_detection.synthetic_page import synthetic_page_input
    step_input = synthetic_page_input(seed=3 + process_idx, size=size, n_lines=64)
    assembler = T.page_assembler_step_factory.create()
    distortion = T.page_distortion_step_factory.create()
    resizing = T.page_resizing_step_factory.create()
    rng = default_rng(SeedSequence(seed).spawn(num_processes)[process_idx])       # vkit/utility/pool.py:85-88

    def page():
        a = assembler.run(step_input, rng)
        d = distortion.run(T.PageDistortionStepInput(a), rng)
        r = resizing.run(T.PageResizingStepInput(d), rng)
        return int(r.page_image.mat[0, 0, 0])

    for _ in range(3):
        page()
    _native.default_ctx().sync()
    ready.wait()
    go.wait()
    w0, t0, n, lat = time.time(), time.perf_counter(), 0, []
    while time.perf_counter() - t0 < seconds:
        t1 = time.perf_counter()
        page()
        lat.append(time.perf_counter() - t1)
        n += 1
    _native.default_ctx().sync()
    results.put((n, lat, w0, time.time()))


def c4_mode(args, group, rank, world, device_index, affinity):
    """BASELINE configs[3] at N ranks: page synthesis at 1024^2, one process per GPU, pages sharded by index, no collective.
    Leg 1 (value): the resident batch -- per rank ``--batch`` pages (default 64), each = background + 64 text-line layers composited by ONE
    launch per batch (ChainBatch.set_layers -> vkx_fill_u8_batch_dev) and sent through the C3 chain (camera_cubic_curve + gaussian_blur +
    color_shift + gaussion_noise from the page's numpy stream, drawn on the device) without leaving HBM; a step = one pass over the batch.
    Leg 2 (reported beside): the reference's API -- PageAssemblerStep -> PageDistortionStep -> PageResizingStep, host objects in and out,
    a pool of ``--api-workers`` processes per GPU with the pool's rule for their generators (SeedSequence(seed).spawn(processes)[process_idx], vkit/utility/pool.py:85-88;
    vkit/pipeline/pool.py:44-48), for ``--api-seconds`` seconds between barriers."""
    from numpy.random import SeedSequence, default_rng
    from vkit_amd import _native, shard
    from vkit_amd.batch import ChainBatch
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    size = 1024
    B = args.batch if args.batch > 0 else 64
    first, _ = shard.weak_span(B, rank)
    ctx = _native.Context(device_index)
    gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), LEVEL)
    batch = ChainBatch(ctx)
    blank = np.zeros((size, size, 3), np.uint8)
    states, plans = [], []
    for j in range(B):
        g = first + j
        state = D.camera_cubic_curve.generate_state(gen((size, size), default_rng(g)), (size, size))
        batch.add(blank, state, blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD, noise_rng=default_rng(5000 + g))
        layers, plan = c4_layers(_native, g, size)
        batch.set_layers(j, layers)
        states.append(state)
        plans.append(plan)

    def sync():
        ctx.sync()

    elapsed = shard.timed_steps(group, batch.run, steps=args.steps, warmup=args.warmup, device_sync=sync)
    pages = group.sum_int(B * args.steps)
    # kernel times of the step (HIP events, outside the timed region)
    ctx.set_timing(True)
    ctx.reset_timings()
    ksteps = max(1, min(args.steps, 10))
    for _ in range(ksteps):
        batch.run()
    ctx.sync()
    kernels = {k: round(v[0] / ksteps, 4) for k, v in sorted(ctx.timings().items())}
    ctx.set_timing(False)
    # parity of this very batch: the rank's first and last page against the oracle (composite layer by layer, then the chain)
    verified = 0
    if args.verify > 0:
        import oracle as O
        for j in sorted({0, B - 1}):
            want_src = np.zeros((size, size, 3), np.uint8)
            for box, value, alpha in plans[j]:
                O.fill(want_src, box, value, mask=None, alpha=1.0 if alpha is None else alpha)
            st = states[j]
            mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
            want = O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(want_src, mx, my), 5, BLUR_SIGMA), HUE_DELTA),
                                   _noise_plane((5000 + first + j, tuple(st.result_shape) + (3,))))
            if not ((batch.source(j) == want_src).all() and (batch.result(j) == want).all()):
                raise SystemExit(f'bench c4: page {first + j} differs from the oracle')
            verified += 1
    batch.close()

    # ---- leg 2: the reference's step objects, a pool of --api-workers processes per GPU ------------------------------------------------
    api = None
    if args.api_seconds > 0:
        K = max(1, args.api_workers)
        ctx_mp = mp.get_context('spawn')
        ready, go, results = ctx_mp.Barrier(K + 1), ctx_mp.Barrier(K + 1), ctx_mp.Queue()
        procs = [ctx_mp.Process(target=_c4_api_worker, args=(rank * K + k, world * K, args.seed, device_index, size, args.api_seconds, ready, go, results))
                 for k in range(K)]
        for p_ in procs:
            p_.start()
        ready.wait(timeout=600)              # every worker of this rank is warm
        group.barrier()                      # ... and of every other rank
        go.wait(timeout=60)
        got = [results.get(timeout=args.api_seconds + 300) for _ in range(K)]
        group.barrier()
        # from the first worker's start to the last worker's last page (the workers' own clocks: one host), MAX over the ranks
        api_elapsed = group.max_float(max(g[3] for g in got) - min(g[2] for g in got))
        for p_ in procs:
            p_.join(timeout=30)
        n = sum(g[0] for g in got)
        lat = [x for g in got for x in g[1]]
        api_pages = group.sum_int(n)
        lats = sorted(x for part in group.all_gather_object(lat) for x in part)
        api = {'pages_per_s': api_pages / api_elapsed, 'pages': api_pages, 'seconds': api_elapsed, 'workers': world * K, 'workers_per_gpu': K,
               'latency_ms': {'mean': sum(lats) / len(lats) * 1e3, 'median': lats[len(lats) // 2] * 1e3, 'p90': lats[int(len(lats) * 0.9)] * 1e3},
               'note': 'PageAssemblerStep -> PageDistortionStep -> PageResizingStep through the step objects, host objects in and out, a pool '
                       f'of {K} worker process(es) per GPU (the reference scales by processes: vkit/utility/pool.py:153-243); worker process_idx '
                       '= rank x K + k draws its pages one after the other from default_rng(SeedSequence(seed).spawn(world x K)[process_idx]) as '
                       'PipelinePoolWorker.run does.  One worker is bound by its Python, a pool by the kernel time of a page (DESIGN.md section 5)'}
    dist_evidence = group.evidence(device_index=device_index, pci_bus_id=(affinity or {}).get('pci_bus_id'), first_page=first, pages=B)
    group.close()
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    print(json.dumps({
        'metric': 'pages/s (1024^2 page synth: 64 text layers composited + geo+photo chain)',
        'value': pages / elapsed, 'unit': 'pages/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8', 'data': 'synthetic',
        'mpixels_per_s': pages * size * size / elapsed / 1e6,
        'config': {'workload': f'C4 page synthesis, device resident: {B} pages of {size}x{size}x3 per GPU and step, each a gray background + 64 '
                               f'text-line layers (32x512, float32 alpha) composited by one batched launch, then camera_cubic_curve remap (level '
                               f'{LEVEL}) + gaussian_blur({BLUR_SIGMA}) + color_shift({HUE_DELTA}) + gaussion_noise({NOISE_STD}, numpy stream drawn on '
                               f'the device)',
                   'batch_per_gpu': B, 'verified_against_oracle': verified, 'affinity': affinity, 'distributed': dist_evidence,
                   'sharding': f'{world} process(es), one per GPU, pages [rank x {B}, (rank + 1) x {B}), no collective'},
        'kernels_ms_per_step': kernels,
        'reference_api': api,
    }))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200,
                    help='timed passes over the batch (the default keeps the timed region above 2 s)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=0, help='images (pages) per GPU; 0 = the config\'s own: 256 for c3, 64 for c4')
    ap.add_argument('--seed', type=int, default=0, help='--config c4: rng_seed of the worker pool (SeedSequence(seed).spawn(world)[rank])')
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--cpu-sample', type=int, default=40, help='images timed on the CPU oracle (rank 0, N=1)')
    ap.add_argument('--cpu-procs', type=int, default=-1,
                    help='largest process count of the all-cores CPU sweep (-1 = one per CPU of the affinity mask, 0 = skip)')
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
    ap.add_argument('--config', default='c3', choices=('c3', 'c4'),
                    help='c3 (default, the headline): BASELINE configs[2], the fused chain on 2048^2 images.  c4: BASELINE configs[3], page '
                         'synthesis at 1024^2 -- per rank a resident batch of --batch pages (64 text layers composited + the chain, one launch '
                         'each) AND the reference-API leg (PageAssemblerStep -> PageDistortionStep -> PageResizingStep, --api-workers processes per '
                         'GPU, the pool\'s process_idx -> rng rule); value = pages/s of the resident batch over all ranks')
    ap.add_argument('--api-seconds', type=float, default=4.0, help='--config c4: seconds of the reference-API leg per rank')
    ap.add_argument('--api-workers', type=int, default=4, help='--config c4: worker processes per GPU of the reference-API leg (the pool)')
    ap.add_argument('--dry-run', action='store_true',
                    help='rendezvous, barriers and the MAX reduction of the timing protocol only, over gloo, no GPU: the N > 1 path on '
                         'a box without GPUs (tests/test_bench_launch.py)')
    ap.add_argument('--noise-workers', type=int, default=0, help='unused since round 3 (the planes are drawn on the device); kept for old command lines')
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = 64 if args.config == 'c4' else 256

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
        dist_evidence = group.evidence(first_unit=first, units=count * args.steps)
        group.close()
        if rank == 0:
            print(json.dumps({'dry_run': True, 'mode': args.mode, 'config': args.config, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                              'ms_per_step': elapsed / args.steps * 1e3, 'units': units, 'first_image_of_last_rank': (world - 1) * args.batch,
                              'distributed': dist_evidence}))
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
    if args.config == 'c4':
        return c4_mode(args, group, rank, world, device_index, affinity)
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
    # what took part: the pixels every rank really processed (SUM all-reduce, not rank 0's count x world), the backend and world size
    # the process group reports, the PCI bus id of every rank's GPU
    total_px = group.sum_int(batch.source_pixels * args.steps)
    dist_evidence = group.evidence(device_index=device_index, pci_bus_id=(affinity or {}).get('pci_bus_id'), first_image=first, images=B)
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

    state_build, lattices_ok = None, True
    if from_configs:
        cb = batch.lanes[0]
        nb = max(cb.state_builds, 1)
        parts = {k: v / nb * 1e3 for k, v in cb.state_parts_s.items()}
        state_build = {'host_ms_per_step': parts['launch'] + parts['layout'], 'builds': cb.state_builds,
                       'wait_ms_per_step': {'lattice_set_free': parts['set_free_wait'], 'result_shapes': parts['shape_wait']},
                       'total_ms_per_step_in_build_states': cb.state_build_s / nb * 1e3,
                       'note': 'ChainBatch._build_states per step, INSIDE the timed region.  host_ms_per_step = the host WORK: 256 x '
                               'vkx_camera_model_host (C) + one k_camera_states launch on the side stream (launch) and the whole-array layout of '
                               'destinations / tile buffers / stream jobs (layout).  wait_ms_per_step = time the host is BLOCKED: '
                               'lattice_set_free = until the pixel kernel of step N - 2 has released the lattice set this step rebuilds (the host '
                               'runs ahead of the device and is held here: this is the device\'s step time showing through, not host work -- '
                               'rounds 5\'s "13.35 ms of Python / ctypes" was this wait, then inside the result_shapes sync), result_shapes = '
                               'until k_camera_states has delivered the shapes'}
        # the host operator's states: CPU legs below, and how many device-built lattices equal them on THIS box
        states = [make_state(first + j, size) for j in range(B)]
        equal = 0
        lattice_picks = sorted({0, B // 3, (2 * B) // 3, B - 1})
        for j in lattice_picks:
            sv, dv = cb.lattices(j)
            equal += int(np.array_equal(sv, states[j].src_image_grid.vertices) and np.array_equal(dv, states[j].dst_image_grid.vertices)
                         and tuple(states[j].result_shape) == cb._dst_shapes[j])
        state_build['lattices_equal_host_operator'] = f'{equal} of {len(lattice_picks)} checked'
        lattices_ok = equal == len(lattice_picks)
        if not lattices_ok:
            print(f'bench: WARNING: only {equal} of {len(lattice_picks)} device-built lattices equal the host operator\'s: the line is marked unverified', file=sys.stderr)
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
    verified, picks = 0, []
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
            # the reference's own scaling model on one GPU: a pool of worker PROCESSES (vkit/utility/pool.py:153-243), each running whole
            # C4 pages through PageAssemblerStep -> PageDistortionStep -> PageResizingStep, host objects in and out (tools/pool_scale.py)
            'C4_reference_api_worker_pool': _tool_json('pool_scale.py', device_index, 240, ('--workers', '1,8', '--seconds', '3', '--modes', 'pipeline')),
            'note': 'BASELINE.json configs[1], [3], [4] on this box and clock, device resident (tools/c2.py, c4.py, c5.py): never '
                    'the headline value; poisson_noise_1024: rng.poisson(image) drawn on the device against numpy itself (values and stream '
                    'position, tools/poisson_probe.py), host arrays in and out; C4_reference_api_worker_pool: pages/s through the reference\'s '
                    'step objects with 1 and 8 worker processes sharing this GPU (a worker is bound by its Python, a pool by the kernel time '
                    'of a page: DESIGN.md section 5, round 6)',
        }

    value = total_px / elapsed / 1e6         # total_px: summed over the ranks before the group closed
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
            # one flag for "this number is a checked number": every picked image equals the oracle AND (from configs) the lattices the
            # device built equal the host operator's on this box -- a lattice regression must not hide behind pixels checked on the
            # lattices the device itself produced
            'verified': bool(verified == len(picks) and verified > 0 and lattices_ok),
            'affinity': affinity,
            'distributed': dist_evidence,
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
        usable = len(START_AFFINITY) if START_AFFINITY else (os.cpu_count() or 1)
        n_procs = usable if args.cpu_procs < 0 else args.cpu_procs
        if result['cpu_baseline'] is not None:
            result['cpu_baseline']['host'] = host_cpu_facts()
        if n_procs > 1 and args.cpu_sample > 0:
            all_cores = cpu_baseline_all_cores(size, n_procs, 2, result['cpu_baseline']['value'])
            if all_cores is not None:
                result['cpu_baseline']['all_cores'] = all_cores
    else:
        result['cpu_baseline'] = None
    print(json.dumps(result))
    sys.stdout.flush()


if __name__ == '__main__':
    main()

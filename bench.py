#!/usr/bin/env python3
"""Benchmark of the distortion hot path on MI355X: BASELINE config 3.

Workload (per GPU): B independent 2048x2048 RGB page images, each with its own ``camera_cubic_curve`` state
(config from the reference-compatible generator at level 5, seed = image index), through
    image-grid remap  ->  gaussian_blur(sigma=1.0, k=5)  ->  color_shift(delta=37)  ->  gaussion_noise(std=10)
with the images and integer vertex lattices resident in HBM before the timed region.  The noise of image i is the
reference's: np.round(default_rng(5000 + i).normal(0, 10, shape)) -- drawn ON THE DEVICE from that numpy stream, value for
value, INSIDE every timed step (vkx_np_draw_batch_dev: PCG64 jump-ahead + ziggurat); no host-generated plane exists.
A "step" is one pass over the whole batch: draw the noise planes, run the chain.  Images shard across GPUs without
any exchange (one process per GPU, weak scaling: every GPU processes its own B images).

Prints ONE JSON line on rank 0 (see the driver contract in the task description):
  value      = source megapixels (H*W per image) processed per second by all ranks together
  roofline   = algorithmic bytes / HIP-event duration of the dominant kernel, against the 8 TB/s HBM peak
  cpu_baseline = the CPU oracle (port of the reference arithmetic, 1 thread) on a bounded sample, rank 0, N=1 only
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per image from the PMC passes of the current kernel source.  tools/record.sh rewrites this file together
# with the sha256 of the sources it profiled; a line measured on other sources reports traffic = null instead of a
# stale figure.
TRAFFIC_FILE = os.path.join('profiles', 'current_traffic.json')
KERNEL_SOURCES = ('vkit_amd/csrc/fused.hip', 'vkit_amd/csrc/vkx_cell.h', 'vkit_amd/csrc/vkx_color.h',
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


def make_state(index, size):
    from numpy.random import default_rng
    from vkit_amd.mechanism import distortion as D
    from vkit_amd.mechanism.distortion_policy.geometric import camera as P_cam
    gen = P_cam.CameraCubicCurveConfigGenerator(P_cam.CameraCubicCurveConfigGeneratorConfig(), LEVEL)
    cfg = gen((size, size), default_rng(index))
    return D.camera_cubic_curve.generate_state(cfg, (size, size))


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
        'sample': f'{n_images} images of the same workload (noise plane drawn from the numpy stream, grid->map, remap, blur, '
                  f'hue shift, noise add), {dt:.1f} s on 1 thread of {os.cpu_count()} host cores',
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
            'sample': f'{n_procs} processes x {per_proc} images, {dt:.1f} s'}


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
                    help='processes of the all-cores CPU leg (-1 = min(64, cores), 0 = skip)')
    ap.add_argument('--verify', type=int, default=4,
                    help='images of the batch checked against the oracle (spread over the batch, the last one included)')
    ap.add_argument('--extra-legs', type=int, default=1,
                    help='at N=1 also measure the throughput noise mode and the drop-in paths (reported beside, never as value)')
    ap.add_argument('--noise-planes', type=int, default=0,
                    help='1: keep an int16 noise plane per image in HBM and add it inside k_chain_fused (the form of rounds 1 - 2) '
                         'instead of letting the generator add its samples to the chain output')
    ap.add_argument('--lanes', type=int, default=2,
                    help='HIP streams the batch is dealt over (ChainLanes): the microsecond kernels of one lane run under the large '
                         'kernels of the other')
    ap.add_argument('--noise-workers', type=int, default=0, help='unused since round 3 (the planes are drawn on the device); kept for old command lines')
    args = ap.parse_args()

    from vkit_amd import shard
    rank, local_rank, world = shard.world_from_env()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')

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

    # ---- host-side setup (no GPU yet): states and noise planes -------------------------------------------------
    t_setup = time.perf_counter()
    states = [make_state(first + j, size) for j in range(B)]
    noise_jobs = [(5000 + first + j, tuple(states[j].result_shape) + (3,)) for j in range(B)]

    import torch
    # process -> GPU by the pool's rule (local_rank % visible GPUs): one rank per GPU on the driver's N-GPU node; on a
    # box with fewer GPUs than ranks (the 2-ranks-on-1-GPU rendezvous check, profiles/r2_torchrun_2ranks_1gpu.log) the
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
    ctx = _native.Context(device_index)
    # the noise member: the numpy stream of image i is drawn on the device every step and added to the chain's output by
    # the pass that puts the samples at their final index (ChainBatch's default); --noise-planes 1 keeps the int16 planes of
    # rounds 1 - 2 in HBM and lets k_chain_fused add them
    batch = ChainLanes(device_index, lanes=args.lanes, stream_noise_mode={0: 'tiles', 1: 'planes', 2: 'late'}[args.noise_planes])
    images = []
    for j in range(B):
        image = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
        images.append(image)
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
    batch.set_timing(True)
    elapsed = shard.timed_steps(group, batch.run, steps=args.steps, warmup=0, device_sync=full_sync)
    kernel_times = batch.timings()
    batch.set_timing(False)
    group.close()  # every rank is past the closing barrier and the MAX reduction: nothing collective is left

    # ---- the chain alone on the planes the last step drew (r2's headline mode: planes resident in HBM), for continuity --
    planes_resident = None
    if world == 1 and args.extra_legs:
        rsteps = max(1, min(args.steps, 50))
        pbatch = None
        if True:
            pbatch = ChainBatch(ctx, stream_noise_planes=True)
            for j in range(B):
                pbatch.add(images[j], states[j], blur_sigma=BLUR_SIGMA, hue_delta=HUE_DELTA, noise_std=NOISE_STD,
                           noise_rng=np.random.default_rng(5000 + first + j))
            pbatch.run()
        full_sync()
        t0 = time.perf_counter()
        for _ in range(rsteps):
            pbatch.run(draw_streams=False)
        full_sync()
        rdt = time.perf_counter() - t0
        planes_resident = {'value': pbatch.source_pixels * rsteps / rdt / 1e6, 'unit': 'Mpixels/s', 'steps': rsteps,
                           'ms_per_step': rdt / rsteps * 1e3,
                           'note': 'the chain on int16 planes resident in HBM (drawn once, added inside k_chain_fused; no '
                                   'drawing inside the step): the mode the round-1 / round-2 headline was measured in'}
        pbatch.close()
    del images

    # ---- parity spot check of this very batch against the oracle (outside the timed region) -------------------------
    # the expected noise plane comes from numpy ITSELF (_noise_plane): the device-drawn plane has to equal it
    verified = 0
    if rank == 0 and args.verify > 0:
        import oracle as O
        picks = sorted({int(round(k * (B - 1) / max(args.verify - 1, 1))) for k in range(min(args.verify, B))} | {B - 1})
        for j in picks:
            st = states[j]
            img = np.random.default_rng(1000 + first + j).integers(0, 256, (size, size, 3), dtype=np.uint8)
            mx, my = O.grid_to_map(st.src_image_grid.vertices, st.dst_image_grid.vertices, st.result_shape)
            want = O.add_noise_i16(O.color_shift_rgb(O.gaussian_blur(O.remap(img, mx, my), 5, BLUR_SIGMA), HUE_DELTA),
                                   _noise_plane(noise_jobs[j]))
            got = batch.result(j)
            if not (got == want).all():
                raise SystemExit(f'bench: image {j} differs from the oracle ({int((got != want).sum())} bytes)')
            verified += 1

    if rank != 0:
        return

    # ---- reported beside the headline, N=1 only: the throughput noise mode and the drop-in paths -----------------------
    throughput_mode, dropin = None, None
    if world == 1 and args.extra_legs:
        # (a) the same chain with the noise plane drawn on the device every step (vkx_noise_normal_i16_dev: the reference's
        #     distribution, not numpy's values) instead of a resident numpy plane: no host generation, no 6 D upload
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
        # (b) host arrays in, host arrays out: tools/dropin.py in a process of its own (a clean HIP runtime: the stream ->
        #     hardware-queue mapping the overlapped pipeline leans on is a property of the process)
        import subprocess
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

    src_px = batch.source_pixels            # per rank, per step
    dst_px = batch.result_pixels
    total_px = src_px * world * args.steps
    value = total_px / elapsed / 1e6

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / mean HIP-event duration -------------------
    S, D = src_px / B, dst_px / B           # mean source / result pixels per image
    algorithmic = {                         # bytes a launch has to move at the very least
        'k_owner_remap': 3 * S + 3 * D,     # per-image launches ...
        # ... and the batch-wide launches: one launch covers all B images.  SURVEY 8(d)'s figure for the fully
        # fused geo+photo chain is 3S + 3D per image (~6.2 B per source pixel); that is the numerator used here.
        'k_chain_fused': (3 * S + 3 * D) * B,
        'k_gaussian_blur': 3 * D + 3 * D,
        'k_hsv': 3 * D + 3 * D,
        'k_add_noise': 3 * D + 6 * D + 3 * D,
        'k_cell_raster': 4 * D,             # the ownership plane it produces
        'k_cell_setup': 0,
        'k_chain_setup': 0,
    }
    # The int16 noise plane is an API input of this workload (host numpy Generator stream, SURVEY 8(d): "+6 D when
    # host-generated int16 noise is an input"); the kernel has to read it, so it is reported next to the strict figure.
    noise_input_bytes = 6 * D * B if args.noise_planes else 0
    # the kernel the roofline is quoted for is the fused geo+photo remap north_star names; the stream kernels are priced
    # beside it (noise_stream) and the whole step against 3S + 3D (chain_frac)
    dominant = 'k_chain_fused' if 'k_chain_fused' in kernel_times else max(kernel_times, key=lambda k: kernel_times[k][0])
    dom_ms, dom_n = kernel_times[dominant]
    avg_s = dom_ms / 1e3 / max(dom_n, 1)
    achieved = algorithmic.get(dominant, 0) / avg_s / 1e9 if avg_s > 0 else 0.0
    achieved_with_noise = ((algorithmic.get(dominant, 0) + (noise_input_bytes if dominant == 'k_chain_fused' else 0))
                           / avg_s / 1e9 if avg_s > 0 else 0.0)
    chain_bytes = (3 * S + 3 * D) * B       # the fully fused figure for the whole chain, per step
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), measured in separate
    # rocprofv3 --pmc passes (profiles/) on the same workload and scaled by the number of images of this launch.
    traffic, traffic_source, valu_issue = None, None, None
    tpath = os.path.join(ROOT, TRAFFIC_FILE)
    tj = None
    if dominant == 'k_chain_fused' and size == 2048 and os.path.exists(tpath):
        with open(tpath) as fin:
            tj = json.load(fin)
        if tj.get('kernel_source_digest') != kernel_source_digest():
            traffic_source = (f'{TRAFFIC_FILE} was measured on other kernel sources (digest '
                              f'{tj.get("kernel_source_digest")} != {kernel_source_digest()}): not used')
            tj = None
        elif int(tj.get('noise_planes', 1)) != int(bool(args.noise_planes)):
            traffic_source = f'{TRAFFIC_FILE} was measured in the other noise mode (--noise-planes): not used'
            tj = None
    if tj is not None:
        traffic = tj['hbm_bytes_per_image'] * B
        traffic_source = f'{TRAFFIC_FILE} <- {tj.get("profile", "?")}'
        if 'valu_insts_per_image' in tj and avg_s > 0:
            # why the HBM fraction is what it is: wavefront VALU instructions (PMC, same profile) at one issue per 4 cycles
            # on 1024 SIMDs (256 CUs x 4) at 2.4 GHz, against the measured launch time
            floor_s = tj['valu_insts_per_image'] * B / 1024 * 4 / 2.4e9
            valu_issue = {'insts_per_wavefront': round(tj['valu_insts_per_wavefront']), 'floor_ms': floor_s * 1e3,
                          'frac_of_launch': floor_s / avg_s}
    kernel_sum_s = sum(v[0] for v in kernel_times.values()) / 1e3 / args.steps
    np_ms = {k: v[0] / args.steps for k, v in kernel_times.items() if k.startswith('k_np_')}
    samples = 3 * D * B
    noise_stream = {
        'kernels_ms_per_step': {k: round(v, 3) for k, v in sorted(np_ms.items())},
        'ms_per_step': elapsed / args.steps * 1e3 - sum(v[0] for k, v in kernel_times.items() if k.startswith('k_chain')) / args.steps,
        'kernel_intervals_ms_per_step': sum(np_ms.values()),
        'samples_per_step': samples,
        'host_fallback_planes': batch.stream_fallbacks,
        'note': 'ms_per_step = the step minus the chain kernels; the kernel intervals overlap (a call runs in chunks of 32 planes: the '
                'draw pass of a chunk on the ctx stream, the resolve / place / walk passes of the chunk before it on a second stream). '
                'np.round(default_rng(5000 + i).normal(0, std, shape)).astype(int16) drawn on the device inside every step, '
                'value for value numpy\'s (checked against numpy on the verified images): 128-bit LCG + ziggurat, VALU bound, '
                'not an HBM-bound kernel' + (' -- its only mandatory traffic is the 2-byte sample it writes' if args.noise_planes else
                '; the samples are added to the chain output in place by the placement pass (clip(uint8 + int16), the '
                'gaussion_noise operator itself): 1 byte read + 1 byte written per sample, no plane'),
    }
    result = {
        'metric': 'Mpixels/s (2048^2 RGB, geo+photo chain)',
        'value': value,
        'unit': 'Mpixels/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'u8',
        'data': 'synthetic',
        'config': {
            'workload': f'C3 fused chain: camera_cubic_curve remap (level {LEVEL}) + gaussian_blur(sigma={BLUR_SIGMA}) + '
                        f'color_shift({HUE_DELTA}) + gaussion_noise(std={NOISE_STD}, the numpy stream default_rng(5000 + i) drawn on '
                        f'the device inside every step), '
                        f'{size}x{size}x3 uint8, batch {B} per GPU',
            'batch_per_gpu': B,
            'image': f'{size}x{size}x3',
            'mean_result_pixels': D,
            'sharding': f'{world} process(es), one per GPU, independent images, no collective' +
                        (f' (ranks share {n_dev} GPU(s): rendezvous over {backend})' if shared_devices else ''),
            'verified_against_oracle': verified,
            'kernel_source_digest': kernel_source_digest(),
            'setup_s': round(t_setup, 1),
        },
        'roofline': {
            'bound': 'hbm',
            'bound_measured': 'valu',
            'kernel': dominant,
            'achieved': achieved,
            'peak': HBM_PEAK_GBS,
            'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            # the north star words the target as "HBM-read roofline": the bytes the kernel must READ (source once, and the
            # noise plane of this workload) over the same launch time
            'read_frac': (3 * S * B) / avg_s / 1e9 / HBM_PEAK_GBS if avg_s > 0 else 0.0,
            'read_frac_incl_noise_input': (3 * S * B + noise_input_bytes) / avg_s / 1e9 / HBM_PEAK_GBS if avg_s > 0 else 0.0,
            'traffic': traffic,
            'traffic_source': traffic_source,
            'valu_issue': valu_issue,
            'avg_launch_ms': avg_s * 1e3,
            'algorithmic_bytes_per_launch': algorithmic.get(dominant, 0),
            'noise_input_bytes_per_launch': noise_input_bytes if dominant == 'k_chain_fused' else 0,
            'achieved_incl_noise_input': achieved_with_noise,
            'frac_incl_noise_input': achieved_with_noise / HBM_PEAK_GBS,
            'traffic_note': 'traffic = FETCH_SIZE x2 + WRITE_SIZE of separate --pmc passes' +
                            ('; it contains the 6 D noise input that the strict 3S+3D numerator leaves out' if args.noise_planes else
                             '; k_chain_fused ends with color_shift here: the noise member is added to its output by the '
                             'generator\'s placement pass (k_np_place, priced in noise_stream), so no noise plane is read'),
            'chain_frac': chain_bytes / kernel_sum_s / 1e9 / HBM_PEAK_GBS if kernel_sum_s > 0 else 0.0,
            'kernels_ms_per_step': {k: round(v[0] / args.steps, 3) for k, v in sorted(kernel_times.items())},
        },
    }
    noise_stream['gsamples_per_s'] = samples / (noise_stream['ms_per_step'] / 1e3) / 1e9 if noise_stream['ms_per_step'] > 0 else None
    result['noise_stream'] = noise_stream
    if planes_resident is not None:
        result['planes_resident'] = planes_resident
    if throughput_mode is not None:
        result['throughput_mode'] = throughput_mode
    if dropin is not None:
        result['dropin'] = dropin
    if world == 1:
        result['cpu_baseline'] = cpu_baseline(size, args.cpu_sample)
        n_procs = min(64, os.cpu_count() or 1) if args.cpu_procs < 0 else args.cpu_procs
        if n_procs > 1 and args.cpu_sample > 0:
            all_cores = cpu_baseline_all_cores(size, n_procs, 4)
            if all_cores is not None:
                result['cpu_baseline']['all_cores'] = all_cores
    else:
        result['cpu_baseline'] = None
    print(json.dumps(result))
    sys.stdout.flush()


if __name__ == '__main__':
    main()

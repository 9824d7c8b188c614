"""Device-resident batch API: the distortion chain over a ragged batch of independent page images that stay in
HBM between the geometric and the photometric members (BASELINE config 3).

``ChainBatch`` owns the device buffers of its images (sources, destinations, vertex lattices, optional noise
planes) and issues the whole batch with ONE ``vkx_chain_rgb_batch_dev`` call.  Images shard across GPUs by
giving every process (one per GPU) its own ``ChainBatch``; there is no exchange step.
"""
import ctypes
import os
import time
from typing import List, Optional, Tuple

import numpy as np

from vkit_amd import _native
from vkit_amd.mechanism.distortion.geometric.grid_rendering.interface import DistortionStateImageGridBased
from vkit_amd.mechanism.distortion.photometric.blur import _estimate_gaussian_kernel_size


class ChainBatch:

    def __init__(self, ctx: Optional[_native.Context] = None, stream_noise_planes: bool = False,
                 stream_noise_mode: Optional[str] = None):
        """How the noise of a ``noise_rng`` image (the caller's numpy stream, drawn on the device) reaches the chain --
        the same pixels in every mode:

        ``'tiles'`` (default): the generator leaves its samples in its own tile slots (``VKX_NP_NORMAL_TILES``) and the chain
        kernel looks them up there (``vkx_chain_item.noise_tiled``): no placement pass, one 2-byte read per sample.
        ``'planes'`` (= ``stream_noise_planes=True``): an int16 plane in HBM that the chain kernel adds (rounds 1 - 2).
        ``'late'``: the generator adds its samples to the chain's output in place, after the chain (``VKX_NP_NORMAL_ADD_U8``,
        round 3); images with a ``streak`` stage take the tile form instead (the streak is drawn over the noise).
        ``run(draw_streams=False)`` re-reads what the previous run drew (tiles or planes)."""
        self.ctx = ctx or _native.default_ctx()
        if stream_noise_mode is None:
            stream_noise_mode = 'planes' if stream_noise_planes else 'tiles'
        if stream_noise_mode not in ('tiles', 'planes', 'late'):
            raise ValueError(f'stream_noise_mode={stream_noise_mode!r}')
        self.stream_noise_mode = stream_noise_mode
        self.stream_noise_planes = stream_noise_mode == 'planes' 
        self._items: List[_native.VkxChainItem] = []
        self._owned: List[int] = []
        self._dst_shapes: List[Tuple[int, int]] = []
        self._array = None
        self._device_noise = []      # (item index, std, seed, dh, dw) of the throughput-mode items
        self._stream_noise = []      # (item index, std, (state, inc), samples, late): the caller's numpy stream drawn on the device
        self._stream_jobs = None     # [(VkxNpJob array, VkxNpResult array, [item index])] in chunks, built on the first run
        self._late_jobs = None       # the same for the images whose noise the generator adds after the chain
        self.stream_chunk = 4096     # planes per vkx_np_draw_batch_dev call (the library pipelines a call in chunks of 32 planes)
        self.stream_fallbacks = 0    # planes the device declared ambiguous and the host drew instead
        self._runs = 0
        self.joint_call = os.environ.get('VKX_CHAIN_JOINT', '1') != '0'       # tile-buffer streams + chain through vkx_chain_rgb_batch_np_dev (False: two calls, as in round 4)
        self._marked = False
        self._cam = []               # (item index, VkxCameraConfig, noise std or None, stream): items whose state is built on the device
        self._cam_state = None       # _CameraStates: records, lattice sets, page-locked results
        self.state_build_s = 0.0     # host time spent in _build_states (C scalars, launch, the wait for the shapes, layout)
        self.state_builds = 0
        self.state_parts_s = {'set_free_wait': 0.0, 'launch': 0.0, 'shape_wait': 0.0, 'layout': 0.0}   # ... and where it goes
        self._set_events = [None, None]   # the point of the compute stream after which lattice set 0 / 1 is no longer read
        self.state_stream = _native.STREAM_COPY_OUT   # where the states are built: a side stream, ahead of the compute stream
        self._page_layers = {}       # item index -> [VkxLayer with device planes]: assembled into the source before the chain
        self._layer_tables = None
        self.debug_kind_flags = {}   # tests: item index -> bits or-ed into its stream job's kind (VKX_NP_DEBUG_WIDE_MARGIN forces the fallback)

    def _put(self, array: np.ndarray) -> int:
        array = np.ascontiguousarray(array)
        ptr = self.ctx.malloc(array.nbytes)
        self._owned.append(ptr)
        self.ctx.upload(ptr, array)
        return ptr

    def add(self, image: np.ndarray, state: DistortionStateImageGridBased, blur_sigma: Optional[float] = None,
            hue_delta: Optional[int] = None, noise: Optional[np.ndarray] = None, streak=None,
            noise_std: Optional[float] = None, noise_seed: Optional[int] = None, noise_rng=None):
        """Registers one HxWx3 uint8 image with its image-grid state and per-image photometric parameters (stage order:
        remap, gaussian_blur, color_shift, gaussion_noise, line_streak; ``None`` skips a stage).

        ``noise``: the caller's int16 plane (parity mode: the reference's values, from the caller's numpy stream).
        ``noise_std`` + ``noise_rng`` (a numpy Generator over PCG64, left untouched): parity mode without a host plane --
        every ``run`` draws ``np.round(rng.normal(0, std, (dh, dw, 3)))`` from that generator's stream ON THE DEVICE, value for
        value what numpy would have drawn (``vkx_np_draw_batch_dev``); the same plane every run, like a resident one.
        ``noise_std`` + ``noise_seed``: throughput mode -- every ``run`` draws a fresh plane of the same distribution on
        the device (``vkx_noise_normal_i16_dev``, seed advanced per run), nothing crosses the link."""
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError('ChainBatch takes HxWx3 uint8 images')
        sh, sw = image.shape[:2]
        dh, dw = state.result_shape
        sv = _native._vertices(state.src_image_grid.vertices)   # int32 [rows, cols, 2], whatever the caller holds
        dv = _native._vertices(state.dst_image_grid.vertices)
        if sv.shape != dv.shape:
            raise ValueError('source / destination grids differ in shape')
        item = _native.VkxChainItem()
        item.src = self._put(image)
        item.dst = self.ctx.malloc(dh * dw * 3)
        self._owned.append(item.dst)
        item.src_stride, item.dst_stride = sw * 3, dw * 3
        item.sh, item.sw, item.dh, item.dw = sh, sw, dh, dw
        item.src_vertices, item.dst_vertices = self._put(sv), self._put(dv)
        item.rows, item.cols = sv.shape[0], sv.shape[1]
        if noise_std is not None:
            if noise is not None:
                raise ValueError('pass either a noise plane or noise_std with noise_seed / noise_rng')
            late = noise_rng is not None and streak is None and self.stream_noise_mode == 'late'
            tiled = noise_rng is not None and not late and self.stream_noise_mode != 'planes'
            if tiled:
                item.noise = self.ctx.malloc(_native.np_tiles_layout(dh * dw * 3)[4])
                self._owned.append(item.noise)
                item.noise_tiled = 1
            elif not late:
                item.noise = self.ctx.malloc(dh * dw * 3 * 2)
                self._owned.append(item.noise)
                item.noise_stride_el = dw * 3
            if noise_rng is not None:
                stream = _native.np_stream(noise_rng)
                if stream is None:
                    raise ValueError('noise_rng must be a numpy Generator over PCG64 (numpy.random.default_rng)')
                self._stream_noise.append((len(self._items), float(noise_std), stream, dh * dw * 3, late))
                self._stream_jobs = self._late_jobs = None
            else:
                self._device_noise.append((len(self._items), float(noise_std), int(noise_seed or 0), dh, dw))
        if noise is not None:
            if noise.shape != (dh, dw, 3):
                raise ValueError(f'noise plane must be {(dh, dw, 3)}, got {noise.shape}')
            item.noise = self._put(noise.astype(np.int16, copy=False))
            item.noise_stride_el = dw * 3
        if blur_sigma is not None:
            item.blur_sigma = float(blur_sigma)
            item.blur_ksize = _estimate_gaussian_kernel_size(blur_sigma)
        if hue_delta is not None:
            item.hue_delta, item.hue_enabled = int(hue_delta), 1
        if streak is not None:   # a LineStreakConfig (or any object with its fields)
            item.streak_enabled = 1
            item.streak_thickness, item.streak_gap = int(streak.thickness), int(streak.gap)
            item.streak_dash_thickness, item.streak_dash_gap = int(streak.dash_thickness), int(streak.dash_gap)
            item.streak_enable_vert, item.streak_enable_hori = int(streak.enable_vert), int(streak.enable_hori)
            for c in range(3):
                item.streak_color[c] = int(streak.color[c])
            item.streak_alpha = float(streak.alpha)
        self._items.append(item)
        self._dst_shapes.append((dh, dw))
        self._array = None
        self._marked = False
        return len(self._items) - 1

    def add_config(self, image: np.ndarray, config, blur_sigma: Optional[float] = None, hue_delta: Optional[int] = None, streak=None,
                   noise_std: Optional[float] = None, noise_rng=None):
        """Like ``add``, from the CONFIG of a ``camera_plane_only`` / ``camera_cubic_curve`` distortion instead of its state: every
        ``run`` builds the vertex lattices of all such items on the device (``vkx_camera_states_dev``: the reference's
        ``generate_state`` -- CameraModel, 2-D -> 3-D lift, cv.projectPoints, the shift by the rounded minimum -- scalars in C on
        the host, vertices on the GPU), reads the result shapes back and lays out destinations and noise buffers for them.  The
        state of the reference is built inside ``Distortion.distort`` (mechanism/distortion/interface.py:318-347); here it is
        inside ``run``.  ``noise_std`` + ``noise_rng``: as in ``add`` (the stream is drawn for the shape the state turns out to
        have); a caller's plane cannot be given before the shape is known."""
        if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] != 3:
            raise ValueError('ChainBatch takes HxWx3 uint8 images')
        if (noise_std is None) != (noise_rng is None):
            raise ValueError('add_config: noise_std together with noise_rng')
        if self.stream_noise_mode != 'tiles' and noise_rng is not None:
            raise ValueError("add_config draws its noise into tile buffers (stream_noise_mode='tiles')")
        sh, sw = image.shape[:2]
        item = _native.VkxChainItem()
        item.src = self._put(image)
        item.src_stride = sw * 3
        item.sh, item.sw = sh, sw
        stream = None
        if noise_rng is not None:
            stream = _native.np_stream(noise_rng)
            if stream is None:
                raise ValueError('noise_rng must be a numpy Generator over PCG64 (numpy.random.default_rng)')
            item.noise_tiled = 1
        if blur_sigma is not None:
            item.blur_sigma = float(blur_sigma)
            item.blur_ksize = _estimate_gaussian_kernel_size(blur_sigma)
        if hue_delta is not None:
            item.hue_delta, item.hue_enabled = int(hue_delta), 1
        if streak is not None:
            item.streak_enabled = 1
            item.streak_thickness, item.streak_gap = int(streak.thickness), int(streak.gap)
            item.streak_dash_thickness, item.streak_dash_gap = int(streak.dash_thickness), int(streak.dash_gap)
            item.streak_enable_vert, item.streak_enable_hori = int(streak.enable_vert), int(streak.enable_hori)
            for c in range(3):
                item.streak_color[c] = int(streak.color[c])
            item.streak_alpha = float(streak.alpha)
        self._cam.append((len(self._items), self._state_record(config, (sh, sw)), None if noise_std is None else float(noise_std), stream))
        self._cam_state = None
        self._items.append(item)
        self._dst_shapes.append((0, 0))
        self._array = None
        self._stream_jobs = self._late_jobs = None
        return len(self._items) - 1

    def set_config(self, index: int, config):
        """Replaces the config of an ``add_config`` item (the next ``run`` builds its state)."""
        for k, (i, _rec, std, stream) in enumerate(self._cam):
            if i == index:
                it = self._items[index]
                self._cam[k] = (i, self._state_record(config, (int(it.sh), int(it.sw))), std, stream)
                if self._cam_state is not None:        # rebuilt on the next run (new lattice buffers, new layout)
                    self.ctx.sync()
                    self._cam_state.close(self.ctx)
                    self._cam_state = None
                return
        raise KeyError(f'item {index} was not added with add_config')

    @staticmethod
    def _state_record(config, shape):
        """The C record of a config whose state the device builds: ('camera', VkxCameraConfig) for camera_plane_only /
        camera_cubic_curve, ('mls', VkxMlsConfig, keepalive) for similarity_mls."""
        if hasattr(config, 'src_handle_points'):
            rec, keep = _native.mls_config(config, shape)
            return ('mls', rec, keep)
        if hasattr(config, 'camera_model_config') and not hasattr(config, 'fold_point') and not hasattr(config, 'curve_point'):
            return ('camera', _native.camera_config(config, shape))
        raise TypeError(f'{type(config).__name__}: states built on the device exist for camera_plane_only, camera_cubic_curve and '
                        'similarity_mls (build the state with the operator and use add())')

    def _ensure_array(self):
        """The contiguous descriptor array of the batch; ``self._items`` become views of its records (one copy of every field)."""
        if self._array is None:
            self._array = (_native.VkxChainItem * max(len(self._items), 1))(*self._items)
            self._items = [self._array[i] for i in range(len(self._items))]
        return self._array

    def _build_states(self):
        """States of the ``add_config`` items for this run, on the side stream: lattices (one of two sets: the previous run's may
        still be read by its pixel kernel), shapes back, destinations / noise buffers / stream jobs laid out for them -- whole-batch
        numpy operations on views of the descriptor arrays, no Python loop over the images."""
        t0 = time.perf_counter()
        cs = self._cam_state
        if cs is None:
            cs = self._cam_state = _DeviceStates(self.ctx, [rec for _i, rec, _s, _st in self._cam])
            cs.index = np.asarray([i for i, _r, _s, _st in self._cam], np.int64)
            cs.position = {int(i): k for k, i in enumerate(cs.index)}
            cs.noisy = np.asarray([st is not None for _i, _r, _s, st in self._cam], bool)
        which = self._runs & 1
        # the pixel kernel of run N - 2 read this set: it has to be past it before the set is rebuilt (the host may be several runs
        # ahead of the device; with immutable configs the rewrite would store the same values, but the buffering must not lean on that)
        if self._set_events[which] is not None:
            self.ctx.event_wait(self._set_events[which])
            self._set_events[which] = None
        t1 = time.perf_counter()
        states = cs.build(self.ctx, which, self.state_stream, self.state_parts_s)   # synchronises the side stream; raises like the reference
        t2 = time.perf_counter()
        self.state_parts_s['set_free_wait'] += t1 - t0
        self._ensure_array()
        view = _native.struct_view(self._array)
        idx = cs.index
        dh, dw = states['dh'].astype(np.int64), states['dw'].astype(np.int64)
        if cs.last_shapes is None or not (np.array_equal(cs.last_shapes[0], dh) and np.array_equal(cs.last_shapes[1], dw)):
            # new shapes: destinations and tile buffers from one arena, 256-byte aligned.  A stream the device declared ambiguous for
            # the OLD shape (its item then reads an owned plane, _verify_streams) is a stream again: the verdict belongs to the sample
            # count, and the first run on the new shapes asks the device anew
            cs.noisy = np.asarray([st is not None for _i, _r, _s, st in self._cam], bool)
            view['noise_tiled'][idx] = cs.noisy
            view['noise_stride_el'][idx] = 0
            n = dh * dw * 3
            dst_bytes = (n + 255) & ~255
            tiles = (n + n // 45 + 4096 + cs.tile_draws - 1) // cs.tile_draws      # (csrc/nprand.hip np_tiles_for_n)
            slots_off = (16 + 8 * (tiles + 1) + 255) & ~255
            tile_bytes = np.where(cs.noisy, slots_off + tiles * cs.slot_elems * 2, 0)
            sizes = np.stack([dst_bytes, (tile_bytes + 255) & ~255], axis=1).reshape(-1)
            offsets = np.concatenate([[0], np.cumsum(sizes)])
            base = cs.arena(self.ctx, int(offsets[-1]))
            cs.dst_ptr = (base + offsets[0:-1:2]).astype(np.uint64)
            cs.noise_ptr = np.where(cs.noisy, base + offsets[1::2], 0).astype(np.uint64)
            cs.last_shapes = (dh, dw)
            for k, i in enumerate(idx):
                self._dst_shapes[int(i)] = (int(dh[k]), int(dw[k]))
            # the stream entries (index, std, stream, samples, late) of these items follow their shapes
            cam_items = set(int(i) for i in idx)
            self._stream_noise = [e for e in self._stream_noise if e[0] not in cam_items]
            for k, (i, _rec, std, stream) in enumerate(self._cam):
                if stream is not None:
                    self._stream_noise.append((i, std, stream, int(n[k]), False))
            self._stream_noise.sort(key=lambda e: e[0])
            self._stream_jobs = self._late_jobs = None
        view['dh'][idx], view['dw'][idx] = dh, dw
        view['dst_stride'][idx] = dw * 3
        view['rows'][idx], view['cols'][idx] = states['rows'], states['cols']
        view['src_vertices'][idx], view['dst_vertices'][idx] = cs.sv_ptr[which], cs.dv_ptr[which]
        view['dst'][idx] = cs.dst_ptr
        view['noise'][idx] = cs.noise_ptr
        t3 = time.perf_counter()
        self.state_parts_s['layout'] += t3 - t2
        self.state_build_s += t3 - t0
        self.state_builds += 1
        return states

    def lattices(self, index: int):
        """(source, destination) vertex lattices of item ``index`` as the last run used them: int32 [rows, cols, 2] host arrays."""
        self.ctx.sync()
        it = self._items[index]
        shape = (int(it.rows), int(it.cols), 2)
        sv, dv = np.empty(shape, np.int32), np.empty(shape, np.int32)
        self.ctx.download(it.src_vertices, sv)
        self.ctx.download(it.dst_vertices, dv)
        return sv, dv

    def set_layers(self, index: int, layers):
        """The text / image layers of page ``index`` (``_native.make_layer`` records for its source shape, in paint order:
        ``PageAssemblerStep.run``, pipeline/text_detection/page_assembler.py:155-236).  Their planes are uploaded once; every
        ``run`` composites the layers of ALL pages onto the sources with one ``vkx_fill_u8_batch_dev`` launch -- a first layer
        covering the page re-initialises it -- and then sends the pages through the chain: assembling and distorting a batch
        of pages never leaves HBM (BASELINE config 3).  All pages of a batch that carry layers must share one source shape."""
        item = self._items[index]
        dev_layers = []
        for layer, _keep in layers:
            if not isinstance(layer, _native.VkxLayer):
                raise TypeError('layers for uint8 pages: make_layer(..., dtype=np.uint8)')
            rec = _native.VkxLayer()
            ctypes.pointer(rec)[0] = layer
            for field, nbytes in (('mask', layer.height * layer.width), ('alpha', layer.height * layer.width * 4),
                                  ('value', layer.height * layer.width * 3)):
                host = getattr(layer, field)
                if host:
                    plane = np.ctypeslib.as_array(ctypes.cast(host, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
                    setattr(rec, field, self._put(plane))
            dev_layers.append(rec)
        for other in self._page_layers:
            if (self._items[other].sh, self._items[other].sw) != (item.sh, item.sw):
                raise ValueError('pages with layers must share one source shape')
        for old in self._page_layers.get(index, ()):      # the planes of the list this call replaces
            for field in ('mask', 'alpha', 'value'):
                ptr = getattr(old, field)
                if ptr and ptr in self._owned:
                    self.ctx.sync()
                    self._owned.remove(ptr)
                    self.ctx.free(ptr)
        self._page_layers[index] = dev_layers
        self._layer_tables = None

    def _composite(self):
        if self._layer_tables is None:
            order = sorted(self._page_layers)
            total = sum(len(self._page_layers[i]) for i in order)
            table = (_native.VkxLayer * max(total, 1))()
            begin = np.zeros(len(order) + 1, np.int32)
            k = 0
            for n, i in enumerate(order):
                for rec in self._page_layers[i]:
                    table[k] = rec
                    k += 1
                begin[n + 1] = k
            ptrs = (ctypes.c_void_p * len(order))(*[self._items[i].src for i in order])
            first = self._items[order[0]]
            self._layer_tables = (ptrs, len(order), int(first.sh), int(first.sw), table, begin)
        ptrs, n, sh, sw, table, begin = self._layer_tables
        _native.check(_native.lib().vkx_fill_u8_batch_dev(self.ctx.handle, ptrs, n, sh, sw, 3, sw * 3, table, begin.ctypes.data))

    def source(self, index: int) -> np.ndarray:
        """The (assembled) source page ``index`` as it sits on the device."""
        self.ctx.sync()
        item = self._items[index]
        out = np.empty((int(item.sh), int(item.sw), 3), np.uint8)
        return self.ctx.download(item.src, out)

    def __len__(self):
        return len(self._items)

    @property
    def source_pixels(self) -> int:
        return sum(int(it.sh) * int(it.sw) for it in self._items)

    @property
    def result_pixels(self) -> int:
        return sum(h * w for h, w in self._dst_shapes)

    def _job(self, entry):
        index, std, stream, n, late = entry
        item = self._items[index]
        if late:
            return _native.np_job(_native.NP_NORMAL_ADD_U8, stream, n, std, src=item.dst, dst=item.dst)
        if item.noise_tiled:
            return _native.np_job(_native.NP_NORMAL_TILES | self.debug_kind_flags.get(index, 0), stream, n, std, dst=item.noise)
        return _native.np_job(_native.NP_NORMAL_I16, stream, n, std, dst=item.noise)

    def _build_jobs(self, entries):
        chunks = []
        step = max(1, self.stream_chunk)
        # the jobs of one call share a kind
        tiled = [e for e in entries if self._items[e[0]].noise_tiled]
        plain = [e for e in entries if not self._items[e[0]].noise_tiled]
        for group in (tiled, plain):
            for k in range(0, len(group), step):
                part = group[k:k + step]
                jobs = (_native.VkxNpJob * len(part))()
                for t, entry in enumerate(part):
                    jobs[t] = self._job(entry)
                chunks.append((jobs, _native.NpResults(self.ctx, len(part)), part))
        return chunks

    def _launch(self, chunks):
        lib = _native.lib()
        for jobs, results, _part in chunks:
            _native.check(lib.vkx_np_draw_batch_dev(self.ctx.handle, jobs, len(jobs), results.array))

    def _verify_streams(self):
        """After the first run: the streams are fixed, so is the device's verdict on them.  A stream it declared ambiguous in
        the last bits of exp / log1p (expected < 1e-6 per plane) is drawn by numpy once and stays resident as a plane the
        chain adds, like a caller's.  Returns True when an image changed sides (its output has to be produced again)."""
        self.ctx.sync()
        flagged = set()
        for chunks in (self._stream_jobs, self._late_jobs):
            for _jobs, results, part in chunks:
                for t, entry in enumerate(part):
                    if results[t].flags:
                        flagged.add(entry[0])
        if not flagged:
            return False
        kept = []
        for entry in self._stream_noise:
            index, std, (state, inc), n, late = entry
            if index not in flagged:
                kept.append(entry)
                continue
            rng = np.random.default_rng()
            st = rng.bit_generator.state
            st['state'] = {'state': state, 'inc': inc}
            rng.bit_generator.state = st
            plane = np.round(rng.normal(0, std, n)).astype(np.int16)
            item = self._items[index]
            if late or item.noise_tiled:
                item.noise = self.ctx.malloc(plane.nbytes)
                self._owned.append(item.noise)
                item.noise_stride_el = int(item.dw) * 3
                item.noise_tiled = 0
                cs = self._cam_state
                if cs is not None and index in cs.position:
                    # an add_config item: _build_states rewrites item.noise from cs.noise_ptr on EVERY run -- the owned plane has
                    # to be what it writes, and the item no longer owns tile slots of the arena
                    k = cs.position[index]
                    cs.noisy[k] = False
                    cs.noise_ptr[k] = item.noise
            self.ctx.upload(item.noise, plane)
            self.stream_fallbacks += 1
        self._stream_noise = kept
        self._array = None
        self._ensure_array()
        self._stream_jobs = self._build_jobs([e for e in kept if not e[4]])
        self._late_jobs = self._build_jobs([e for e in kept if e[4]])
        return True

    def run(self, draw_streams: bool = True):
        """Enqueues the chain for every image on the ctx stream (asynchronous).  ``draw_streams=False`` leaves the PLANES of
        the ``noise_rng`` items as the previous run drew them (they are the same every run); noise that the generator adds
        after the chain is always drawn."""
        if self._cam:
            self._build_states()
        self._ensure_array()
        lib = _native.lib()
        first = self._stream_noise and self._stream_jobs is None
        if first:
            self._stream_jobs = self._build_jobs([e for e in self._stream_noise if not e[4]])
            self._late_jobs = self._build_jobs([e for e in self._stream_noise if e[4]])
        # the tile-buffer streams and the chain as ONE call (vkx_chain_rgb_batch_np_dev): the library knows which image waits for
        # which stream and hides the small kernels of either under the large kernels of the other
        joint = None
        if self._stream_noise and (draw_streams or first) and self._stream_jobs:
            jobs, _results, part = self._stream_jobs[0]
            if self._items[part[0][0]].noise_tiled and not self._device_noise and self.joint_call:
                joint = self._stream_jobs[0]
                self._launch(self._stream_jobs[1:])
            else:
                self._launch(self._stream_jobs)
        if self._device_noise:
            # the planes that share a deviation in one launch (its inverse-CDF table is staged once per workgroup)
            by_std = {}
            for index, std, seed, dh, dw in self._device_noise:
                by_std.setdefault(std, []).append((index, seed, dh, dw))
            for std, members in by_std.items():
                planes = (_native.VkxNoisePlane * len(members))()
                for t, (index, seed, dh, dw) in enumerate(members):
                    pl = planes[t]
                    pl.dst, pl.stride_el, pl.h, pl.w, pl.cn = self._items[index].noise, dw * 3, dh, dw, 3
                    pl.seed = (seed + self._runs * 0x9E3779B97F4A7C15) & 0xffffffffffffffff
                _native.check(lib.vkx_noise_normal_i16_batch_dev(self.ctx.handle, planes, len(members), std))
        self._runs += 1
        if self._page_layers:
            if not self._cam:        # the lattices were uploaded by add(): the chain's cell setup need not wait for the composite
                _native.check(lib.vkx_chain_lattices_ready(self.ctx.handle))      # (one chain call per mark)
            self._composite()
        if joint is not None:
            jobs, results, _part = joint
            _native.check(lib.vkx_chain_rgb_batch_np_dev(self.ctx.handle, self._array, len(self._items), jobs, len(jobs), results.array))
        else:
            _native.check(lib.vkx_chain_rgb_batch_dev(self.ctx.handle, self._array, len(self._items)))
        if self._stream_noise and self._late_jobs:
            self._launch(self._late_jobs)
        if self._cam:            # the lattice set this run read is free once the compute stream is past this point
            which = (self._runs - 1) & 1
            if self._set_events[which] is not None:
                self.ctx.event_wait(self._set_events[which])
            self._set_events[which] = self.ctx.event_record(_native.STREAM_COMPUTE)
        if first and self._verify_streams():
            self._runs -= 1
            self.run(draw_streams=True)

    def result(self, index: int) -> np.ndarray:
        self.ctx.sync()
        dh, dw = self._dst_shapes[index]
        out = np.empty((dh, dw, 3), np.uint8)
        return self.ctx.download(self._items[index].dst, out)

    def close(self):
        for ptr in self._owned:
            self.ctx.free(ptr)
        self._owned.clear()
        self._items.clear()
        self._device_noise.clear()
        self._stream_noise.clear()
        self._stream_jobs = self._late_jobs = None
        self._page_layers.clear()
        self._layer_tables = None
        self._array = None
        if self._cam_state is not None:
            self._cam_state.close(self.ctx)
            self._cam_state = None
        self._cam.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DeviceStates:
    """The device side of ``ChainBatch.add_config``: the config records (camera and similarity_mls kinds, each kind one call), two sets
    of lattice buffers, the page-locked result records and the arena the destinations / tile buffers of the batch live in."""

    def __init__(self, ctx, records):
        n = len(records)
        self.n = n
        self.kinds = [r[0] for r in records]
        self.keep = [r[2:] for r in records]
        self.groups = {}         # kind -> (positions in the batch, ctypes record array)
        for kind, cls in (('camera', _native.VkxCameraConfig), ('mls', _native.VkxMlsConfig)):
            pos = [k for k, r in enumerate(records) if r[0] == kind]
            if pos:
                self.groups[kind] = (pos, (cls * len(pos))(*[records[k][1] for k in pos]))
        self._out_ptr = ctx.host_alloc(n * ctypes.sizeof(_native.VkxGridState))
        self.out = (_native.VkxGridState * n).from_address(self._out_ptr)
        self.states = _native.struct_view(self.out)
        self.sets = [None, None]
        self.sv_ptr = [None, None]
        self.dv_ptr = [None, None]
        self._ptr_c = {}
        self._arena = 0
        self._arena_cap = 0
        self.last_shapes = None
        self.dst_ptr = self.noise_ptr = None
        shape = [ctypes.c_int64() for _ in range(5)]
        _native.check(_native.lib().vkx_np_tiles_layout(1 << 20, *[ctypes.byref(v) for v in shape]))
        self.slot_elems = int(shape[1].value)
        self.tile_draws = 3072          # raw draws per generator tile (csrc/nprand.hip kTile; tests/test_camera_states.py checks the layout formula)
        # lattice buffers: the states of a kind are contiguous in the result records (camera first), `order` maps batch position -> record
        cells = np.empty(n, np.int64)
        for k, r in enumerate(records):
            rec = r[1]
            rows, cols = _native.lattice_shape(rec.height, rec.width, rec.grid_size)
            cells[k] = rows * cols
        lattice_bytes = (cells * 8 + 255) & ~255
        offsets = np.concatenate([[0], np.cumsum(np.repeat(lattice_bytes, 2))])
        total = int(offsets[-1])
        self.order = np.empty(n, np.int64)
        first = 0
        for kind in ('camera', 'mls'):
            if kind in self.groups:
                pos = self.groups[kind][0]
                self.order[pos] = first + np.arange(len(pos))
                first += len(pos)
        for which in (0, 1):
            base = self.sets[which] = ctx.malloc(max(total, 256))
            self.sv_ptr[which] = (base + offsets[0:-1:2]).astype(np.uint64)
            self.dv_ptr[which] = (base + offsets[1::2]).astype(np.uint64)
            for kind, (pos, _recs) in self.groups.items():
                self._ptr_c[(which, kind)] = ((ctypes.c_void_p * len(pos))(*[int(self.sv_ptr[which][k]) for k in pos]),
                                              (ctypes.c_void_p * len(pos))(*[int(self.dv_ptr[which][k]) for k in pos]))

    def build(self, ctx, which, stream, parts=None):
        lib = _native.lib()
        t0 = time.perf_counter()
        first = 0
        size = ctypes.sizeof(_native.VkxGridState)
        for kind, fn in (('camera', lib.vkx_camera_states_dev), ('mls', lib.vkx_mls_states_dev)):
            if kind not in self.groups:
                continue
            pos, recs = self.groups[kind]
            sv, dv = self._ptr_c[(which, kind)]
            _native.check(fn(ctx.handle, recs, len(pos), sv, dv, ctypes.c_void_p(self._out_ptr + first * size), int(stream)))
            first += len(pos)
        t1 = time.perf_counter()
        ctx.sync_stream(stream)
        if parts is not None:
            parts['launch'] += t1 - t0
            parts['shape_wait'] += time.perf_counter() - t1
        states = self.states[self.order]          # in batch order
        flags = states['flags']
        if flags.any():
            # the reference builds a Point per vertex and fails in its round() (element/point.py:31-47); similarity_mls divides by a
            # zero distance under np.errstate(divide='raise') when a vertex sits on an integer handle position
            if (flags & _native.GRID_STATE_DIVIDE).any():
                raise FloatingPointError('divide by zero encountered in divide')
            if (flags & _native.GRID_STATE_NAN).any():
                raise ValueError('cannot convert float NaN to integer')
            if (flags & _native.GRID_STATE_INF).any():
                raise OverflowError('cannot convert float infinity to integer')
            raise OverflowError('destination lattice outside the int32 range')
        return states

    def arena(self, ctx, nbytes):
        if nbytes > self._arena_cap:
            if self._arena:
                ctx.free(self._arena)          # (synchronises the compute stream)
            self._arena_cap = nbytes + nbytes // 16
            self._arena = ctx.malloc(self._arena_cap)
        return self._arena

    def close(self, ctx):
        for which in (0, 1):
            if self.sets[which]:
                ctx.free(self.sets[which])
                self.sets[which] = None
        if self._arena:
            ctx.free(self._arena)
            self._arena = 0
        if self._out_ptr:
            ctx.host_free(self._out_ptr)
            self._out_ptr = 0


class ChainLanes:
    """``lanes`` ``ChainBatch`` objects, each on a context (HIP stream + scratch) of its own, behind the interface of one: images
    are dealt round robin, ``run`` enqueues every lane.  The lanes share the device: while one lane is between its large
    kernels -- the carry resolution of its numpy streams, its cell setup: microsecond kernels of a few workgroups -- the other
    lane's draw pass or chain kernel has the CUs, so the step costs the sum of the large kernels instead of the sum of all."""

    def __init__(self, device: Optional[int] = None, lanes: int = 2, **kwargs):
        base = _native.default_ctx() if device is None else None
        dev = base.device if base is not None else int(device)
        self.contexts = [_native.Context(dev) for _ in range(max(1, int(lanes)))]
        self.lanes = [ChainBatch(ctx, **kwargs) for ctx in self.contexts]
        self._where = []

    def add(self, *args, **kwargs):
        lane = len(self._where) % len(self.lanes)
        self._where.append((lane, self.lanes[lane].add(*args, **kwargs)))
        return len(self._where) - 1

    def run(self, **kwargs):
        for lane in self.lanes:
            if len(lane):
                lane.run(**kwargs)

    def sync(self):
        for ctx in self.contexts:
            ctx.sync()

    def result(self, index: int) -> np.ndarray:
        lane, k = self._where[index]
        return self.lanes[lane].result(k)

    def __len__(self):
        return len(self._where)

    @property
    def source_pixels(self) -> int:
        return sum(lane.source_pixels for lane in self.lanes)

    @property
    def result_pixels(self) -> int:
        return sum(lane.result_pixels for lane in self.lanes)

    @property
    def stream_fallbacks(self) -> int:
        return sum(lane.stream_fallbacks for lane in self.lanes)

    def set_timing(self, on: bool):
        for ctx in self.contexts:
            ctx.set_timing(on)
            if on:
                ctx.reset_timings()

    def timings(self):
        """kernel -> (total ms, launches) summed over the lanes (their intervals overlap on the device)."""
        out = {}
        for ctx in self.contexts:
            for name, (ms, count) in ctx.timings().items():
                t, c = out.get(name, (0.0, 0))
                out[name] = (t + ms, c + count)
        return out

    def close(self):
        for lane in self.lanes:
            lane.close()
        for ctx in self.contexts:
            try:
                ctx.close()
            except Exception:
                pass

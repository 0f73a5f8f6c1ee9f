"""Row-strip domain decomposition of one grid over the GPUs of a node, one process per GPU.

New functionality relative to the reference (it has no spatial decomposition, SURVEY.md section
8e); the semantics are those of a single StencilUpdate on the whole grid.

Every rank owns `rows_owned` consecutive rows and keeps them in a buffer with `g_max` ghost rows on
each side.  One pass = one kernel launch of T generations (temporal blocking) and consumes
g = T * radius * n_subiterations ghost rows per side, so ghosts are exchanged once per pass.

Inside a rank the owned rows are split once more into V sub-strips (V = 1 or 2, the rule of the
single-GPU pass driver), each advancing on its own stream.  A sub-strip sweeps its two g-row
boundary bands first and its interior afterwards; neighbouring sub-strips are coupled only through
"the bands of the previous pass are done" events, so the ragged tail of one sub-strip's launch
overlaps the next launches of the other.  The outermost bands are the rows the neighbouring ranks
need:

    sub-strip v   : wait(neighbour bands of pass p-1 | ghosts of pass p) -> bands of pass p -> [event] -> interior of pass p
    comm stream   : wait(outermost bands of pass p) -> send/recv g rows with rank-1 / rank+1 (RCCL p2p over xGMI) -> [ghosts of pass p+1]

so the exchange for pass p+1 runs concurrently with the interiors of pass p.  There is no
collective on the data path.  The sweep itself is libststhip.so (ststhip_app_sweep); tests inject a
CPU sweep to exercise this logic with the gloo backend.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi


def split_rows(total_rows, world):
    """Row range [begin, end) of every rank: as even as possible, earlier ranks get the remainder."""
    base, extra = divmod(total_rows, world)
    bounds, at = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        bounds.append((at, at + n))
        at += n
    return bounds


def pass_depths(n_generations, max_generations):
    """Greedy over max_generations and its repeated halvings, largest first (the compiled depths)."""
    out, remaining = [], int(n_generations)
    while remaining > 0:
        t = int(max_generations)
        while t > 1 and t > remaining:
            t //= 2
        out.append(t)
        remaining -= t
    return out


class HipSweep:
    """Default sweep backend: the HIP kernels behind the C ABI."""

    def __init__(self, app, tf_params, halo_bytes):
        self.app, self.tf_params, self.halo_bytes = app, tf_params, halo_bytes
        # Jacobi5General with five equal positive coefficients and a +0 halo has a bit-identical form with
        # 5 instead of 9 flops per cell (Jacobi5Uniform, stencilstream_amd/csrc/apps/jacobi.hpp); which of
        # its four kernels a launch needs depends on the first / last generation of the run
        self.uniform = None
        self.first_iteration = self.last_iteration = None
        if app == "jacobi5general" and bytes(halo_bytes) == b"\0\0\0\0":
            coef = [tf_params.coef[i] for i in range(5)]
            if coef[0] > 0 and all(np.float32(c).tobytes() == np.float32(coef[0]).tobytes() for c in coef):
                self.uniform = capi.JacobiUniformParams(coef[0])
        info = capi.app_info("jacobi5uniform" if self.uniform is not None else app)
        self.n_planes = info.n_planes
        self.plane_elem_size = [info.plane_elem_size[i] for i in range(info.n_planes)]
        # the depth the family's rule trusts (a family may be compiled deeper: ststhip_app_info.default_generations)
        self.max_generations = info.default_generations or info.max_generations
        self.halo_per_generation = info.halo_depth_per_generation

    def begin_run(self, iteration_offset, n_generations):
        self.first_iteration = int(iteration_offset)
        self.last_iteration = int(iteration_offset) + int(n_generations) - 1

    def suggest_sub_strips(self, rows, width, n_passes):
        return capi.suggest_row_strips("jacobi5uniform" if self.uniform is not None else self.app, rows, width,
                                       n_passes)

    def side_by_side(self, n):
        capi.set_launch_concurrency(n)

    def __call__(self, src, dst, dom, out_begin, out_end, iteration, depth, stream):
        app, params = self.app, self.tf_params
        if self.uniform is not None and self.first_iteration is not None:
            has_first = iteration == self.first_iteration
            has_last = iteration + depth - 1 == self.last_iteration
            app = "jacobi5uniform" + {(False, False): "", (True, False): "_first", (False, True): "_last",
                                      (True, True): "_only"}[(has_first, has_last)]
            params = self.uniform
        capi.app_sweep(app, params, self.halo_bytes, dom, [t.data_ptr() for t in src],
                       [t.data_ptr() for t in dst], out_begin, out_end, iteration, depth,
                       stream.cuda_stream if stream is not None else 0)


class StripDomain:
    """The rows of one rank plus ghost rows, double buffered."""

    def __init__(self, app, tf_params, halo_bytes, total_rows, width, rank, world, device, cell_dtype=None,
                 sweep=None, group=None, min_rows_check=True, exchange_via_host=False, sub_strips=None):
        self.sweep = sweep if sweep is not None else HipSweep(app, tf_params, halo_bytes)
        self.rank, self.world, self.device, self.group = rank, world, torch.device(device), group
        self.total_rows, self.width = int(total_rows), int(width)
        self.bounds = split_rows(self.total_rows, world)
        self.row_begin, self.row_end = self.bounds[rank]
        self.g_max = self.sweep.max_generations * self.sweep.halo_per_generation
        if min_rows_check and world > 1 and min(b - a for a, b in self.bounds) < 2 * self.g_max:
            raise ValueError("strips are thinner than two halo depths; use fewer ranks or a larger grid")
        self.row_origin = self.row_begin - self.g_max
        self.local_rows = (self.row_end - self.row_begin) + 2 * self.g_max
        self.on_gpu = self.device.type == "cuda"
        # debugging aid: ghost rows travel through host memory, so that the multi-process GPU path can
        # run where the process group cannot move device memory (gloo; several ranks on one GPU)
        self.exchange_via_host = bool(exchange_via_host)
        self.planes = [
            [torch.zeros((self.local_rows, self.width * es), dtype=torch.uint8, device=self.device)
             for es in self.sweep.plane_elem_size]
            for _ in range(2)
        ]
        self.current = 0
        if self.on_gpu:
            self.compute_stream = torch.cuda.Stream(self.device)
            self.comm_stream = torch.cuda.Stream(self.device)
        else:
            self.compute_stream = self.comm_stream = None
        # sub-strips of the owned rows (None: the pass driver's rule decides per advance())
        self.sub_strips = sub_strips
        self.side_streams = []
        self.dom = capi.Domain(self.total_rows, self.width, self.row_origin, self.local_rows, self.width)
        self.n_launches = 0

    # ---- data in / out -------------------------------------------------------------------
    def _owned_slice(self):
        return slice(self.row_begin - self.row_origin, self.row_end - self.row_origin)

    def load_owned(self, *plane_rows):
        """Set the owned rows from one 2-D tensor per plane (any dtype of the right row size)."""
        for dst, rows in zip(self.planes[self.current], plane_rows):
            raw = rows.contiguous().view(torch.uint8).reshape(rows.shape[0], -1)
            dst[self._owned_slice()].copy_(raw)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()

    def owned(self, plane=0, dtype=torch.float32):
        if self.on_gpu:
            self.compute_stream.synchronize()
        return self.planes[self.current][plane][self._owned_slice()].view(dtype)

    # ---- ghost exchange ------------------------------------------------------------------
    def _exchange(self, planes, g):
        """Fill the g ghost rows next to the owned rows of `planes` from the neighbours."""
        if self.world == 1 or g == 0:
            return
        o = self._owned_slice()
        ops, landed = [], []
        up, down = self.rank - 1, self.rank + 1

        def outgoing(rows):
            return rows.cpu() if self.exchange_via_host else rows

        def incoming(rows):
            if not self.exchange_via_host:
                return rows
            staged = torch.empty(rows.shape, dtype=rows.dtype)
            landed.append((rows, staged))
            return staged

        for p in planes:
            if up >= 0:
                ops.append(dist.P2POp(dist.isend, outgoing(p[o.start:o.start + g]), self._peer(up), self.group))
                ops.append(dist.P2POp(dist.irecv, incoming(p[o.start - g:o.start]), self._peer(up), self.group))
            if down < self.world:
                ops.append(dist.P2POp(dist.isend, outgoing(p[o.stop - g:o.stop]), self._peer(down), self.group))
                ops.append(dist.P2POp(dist.irecv, incoming(p[o.stop:o.stop + g]), self._peer(down), self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for rows, staged in landed:
            rows.copy_(staged, non_blocking=False)

    def warm_up_exchange(self):
        """One exchange of the deepest halo outside any timed region: RCCL sets up its point-to-point
        channels on first use (tens of milliseconds).  Ghost rows are refreshed by every advance(), so
        this changes nothing."""
        if self.world == 1:
            return
        if self.on_gpu:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self._exchange(self.planes[self.current], self.g_max)
            self.comm_stream.synchronize()
        else:
            self._exchange(self.planes[self.current], self.g_max)

    def _peer(self, strip_rank):
        return strip_rank if self.group is None else dist.get_global_rank(self.group, strip_rank)

    # ---- time stepping -------------------------------------------------------------------
    def _sub_strip_bounds(self, n_passes):
        """Row boundaries of the sub-strips of the owned rows, and the stream each one runs on."""
        a, b = self.row_begin, self.row_end
        v = self.sub_strips
        if v is None:
            suggest = getattr(self.sweep, "suggest_sub_strips", None)
            v = suggest(b - a, self.width, n_passes) if (suggest and self.on_gpu) else 1
        v = max(1, min(int(v), 8))
        while v > 1 and (b - a) < v * 2 * max(self.g_max, 1):
            v -= 1  # every sub-strip keeps at least its two bands
        if v == 2:
            # unequal strips drift out of phase, so one strip's tail meets the other's bulk
            bounds = [a, a + (b - a) * 2 // 5, b]
        else:
            bounds = [a + (b - a) * i // v for i in range(v + 1)]
        lanes = [self.compute_stream] * v
        if self.on_gpu:
            while len(self.side_streams) < v - 1:
                self.side_streams.append(torch.cuda.Stream(self.device))
            lanes = [self.compute_stream] + self.side_streams[:v - 1]
        return bounds, lanes

    def advance(self, iteration_offset, n_generations):
        """Advance the whole (distributed) grid by n_generations generations.

        Per pass p (buffers ping-pong; `g` = halo rows of the pass):
            sub-strip v (own stream): wait(bands of v-1, v+1 in pass p-1; outermost: ghosts of pass p)
                                      -> top band, bottom band -> [bands event] -> interior
            comm stream            : wait(bands events of the outermost sub-strips) -> exchange for pass p+1
        A band reads rows up to 2g into its own strip and g into the neighbour, all of the previous
        pass; bands and interior of one pass write disjoint rows.
        """
        depths = pass_depths(n_generations, self.sweep.max_generations)
        if not depths:
            return None
        iteration = int(iteration_offset)
        cs, ms = self.compute_stream, self.comm_stream
        gpu = self.on_gpu
        hpg = self.sweep.halo_per_generation
        if hasattr(self.sweep, "begin_run"):
            self.sweep.begin_run(iteration_offset, n_generations)
        bounds, lanes = self._sub_strip_bounds(len(depths))
        n_sub = len(bounds) - 1
        has_up, has_down = self.rank > 0, self.rank + 1 < self.world
        if hasattr(self.sweep, "side_by_side"):
            self.sweep.side_by_side(n_sub)

        def on_comm():
            return torch.cuda.stream(ms) if gpu else _NullContext()

        if gpu:
            ms.wait_stream(cs)  # everything queued so far (previous advance, load_owned)
            for lane in lanes[1:]:
                lane.wait_stream(cs)
        with on_comm():
            self._exchange(self.planes[self.current], depths[0] * hpg)
            ghosts_ready = ms.record_event() if gpu and self.world > 1 else None
        bands_done = [None] * n_sub
        for i, depth in enumerate(depths):
            g = depth * hpg
            src, dst = self.planes[self.current], self.planes[self.current ^ 1]
            bands_now = [None] * n_sub
            for v in range(n_sub):
                va, vb, lane = bounds[v], bounds[v + 1], lanes[v]
                up = v > 0 or has_up            # somebody above needs (and feeds) my top rows
                down = v + 1 < n_sub or has_down
                if gpu:
                    if v > 0 and bands_done[v - 1] is not None:
                        lane.wait_event(bands_done[v - 1])
                    if v + 1 < n_sub and bands_done[v + 1] is not None:
                        lane.wait_event(bands_done[v + 1])
                    if ghosts_ready is not None and ((v == 0 and has_up) or (v == n_sub - 1 and has_down)):
                        lane.wait_event(ghosts_ready)
                top_end = min(va + g, vb) if up else va
                bot_begin = max(vb - g, top_end) if down else vb
                if va < top_end:
                    self.sweep(src, dst, self.dom, va, top_end, iteration, depth, lane)
                    self.n_launches += 1
                if bot_begin < vb:
                    self.sweep(src, dst, self.dom, bot_begin, vb, iteration, depth, lane)
                    self.n_launches += 1
                bands_now[v] = lane.record_event() if gpu else None
                if top_end < bot_begin:
                    self.sweep(src, dst, self.dom, top_end, bot_begin, iteration, depth, lane)
                    self.n_launches += 1
            if i + 1 < len(depths) and self.world > 1:
                with on_comm():
                    if gpu:
                        ms.wait_event(bands_now[0])
                        ms.wait_event(bands_now[n_sub - 1])
                    self._exchange(dst, depths[i + 1] * hpg)
                    ghosts_ready = ms.record_event() if gpu else None
            bands_done = bands_now
            self.current ^= 1
            iteration += depth
        if gpu:
            for lane in lanes[1:]:
                cs.wait_stream(lane)
            cs.wait_stream(ms)  # the compute stream is the one callers synchronise with
        if hasattr(self.sweep, "side_by_side"):
            self.sweep.side_by_side(1)
        return None


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def host_exchange_callback(rank, world, first="chain", second="chain"):
    """An `exchange` callable for capi.Strip (contract of ststhip_comm_exchange_rows) that stages the ghost rows
    through host memory over the default process group (gloo): for hosts whose ranks RCCL cannot join -- several
    processes sharing one GPU (tests/test_strip_native_gpu.py, `bench.py --debug-host-exchange`).
    first / second: the ranks on the "up" and "down" side (default: rank - 1 and rank + 1, a chain of strips; None =
    no neighbour) -- the row and column callables of a capi.Block name their mesh neighbours here."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from . import capi

    lib = capi.load()

    def exchange(n_planes, send_up, send_down, recv_up, recv_down, row_bytes, n_rows, stream):
        capi.check(lib.ststhip_stream_synchronize(C.c_void_p(stream)), "sync")
        ops, landing = [], []
        for p in range(n_planes):
            n = row_bytes[p] * n_rows
            peer_first = rank - 1 if first == "chain" else first
            peer_second = rank + 1 if second == "chain" else second
            for peer, send, recv in ((peer_first, send_up[p], recv_up[p]), (peer_second, send_down[p], recv_down[p])):
                if peer is None or peer < 0 or peer >= world:
                    continue
                out = torch.empty(n, dtype=torch.uint8)
                capi.check(lib.ststhip_memcpy_d2h(C.c_void_p(out.data_ptr()), C.c_void_p(send), n, C.c_void_p(stream)), "d2h")
                inc = torch.empty(n, dtype=torch.uint8)
                landing.append((recv, inc, n))
                ops.append((peer, out, inc))
        capi.check(lib.ststhip_stream_synchronize(C.c_void_p(stream)), "sync")
        reqs = dist.batch_isend_irecv([x for peer, out, inc in ops
                                       for x in (dist.P2POp(dist.isend, out, peer), dist.P2POp(dist.irecv, inc, peer))])
        for r in reqs:
            r.wait()
        for recv, inc, n in landing:
            capi.check(lib.ststhip_memcpy_h2d(C.c_void_p(recv), C.c_void_p(inc.data_ptr()), n, C.c_void_p(stream)), "h2d")
        capi.check(lib.ststhip_stream_synchronize(C.c_void_p(stream)), "sync")

    return exchange

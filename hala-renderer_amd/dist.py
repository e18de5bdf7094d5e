"""Multi-GPU glue (BASELINE.json north_star: "frames shard by pixel-tile across the 8 GPUs of one node with RCCL
all-gather of tiles over xGMI").  One process per GPU; the scene, BVH and env tables are replicated, pixels are the
only sharded unit, and the single exchange step is one all-gather per AOV after the last sample of a frame.

`TileLayout` is the host-side statement of docs/RENDER_SPEC.md §9 (the same mapping the HIP kernels evaluate in
slot_to_pixel); it is pure integer arithmetic and is what the CPU (gloo) tests exercise.
"""
import math

import numpy as np


class TileLayout:
    """tile t (row-major over the tile grid) sits at position k = (t*A + B) mod n of the dealing order; rank k % world
    owns it and stores it as its (k // world)-th tile; A = first integer >= 0x9E3779B1 % n coprime to n, B = 7."""

    def __init__(self, width, height, world, tile_size=32):
        self.width, self.height, self.world, self.tile_size = width, height, world, tile_size
        self.tiles_x = (width + tile_size - 1) // tile_size
        self.tiles_y = (height + tile_size - 1) // tile_size
        n = self.tiles_x * self.tiles_y
        self.n_tiles = n
        self.tiles_per_rank = (n + world - 1) // world
        a = 0x9E3779B1 % n
        if a == 0:
            a = 1
        while math.gcd(a, n) != 1:
            a += 1
        self.a, self.b = a, 7
        t = np.arange(n, dtype=np.int64)
        k = (t * a + self.b) % n
        self.owner = (k % world).astype(np.int64)
        self.slot = (k // world).astype(np.int64)
        self.pixels_per_rank = self.tiles_per_rank * tile_size * tile_size

    def rank_pixel_map(self, rank):
        """for every local pixel slot of `rank`: (py, px) or (-1, -1) for padding / out-of-frame slots"""
        ts = self.tile_size
        out = np.full((self.pixels_per_rank, 2), -1, dtype=np.int64)
        ly, lx = np.meshgrid(np.arange(ts), np.arange(ts), indexing="ij")
        for t in np.nonzero(self.owner == rank)[0]:
            ty, tx = divmod(int(t), self.tiles_x)
            py = ty * ts + ly.reshape(-1)
            px = tx * ts + lx.reshape(-1)
            ok = (py < self.height) & (px < self.width)
            base = int(self.slot[t]) * ts * ts
            sel = np.arange(ts * ts)[ok]
            out[base + sel, 0] = py[ok]
            out[base + sel, 1] = px[ok]
        return out

    def shard(self, image, rank):
        """row-major [H, W, C] -> this rank's tile-major buffer [pixels_per_rank, C] (zeros in padding slots)"""
        m = self.rank_pixel_map(rank)
        buf = np.zeros((self.pixels_per_rank, image.shape[2]), dtype=image.dtype)
        ok = m[:, 0] >= 0
        buf[ok] = image[m[ok, 0], m[ok, 1]]
        return buf

    def unshard(self, gathered, channels=4):
        """[world * pixels_per_rank, C] (all-gather output) -> row-major [H, W, C]"""
        g = np.asarray(gathered).reshape(self.world, self.pixels_per_rank, channels)
        img = np.zeros((self.height, self.width, channels), dtype=g.dtype)
        for r in range(self.world):
            m = self.rank_pixel_map(r)
            ok = m[:, 0] >= 0
            img[m[ok, 0], m[ok, 1]] = g[r][ok]
        return img


class _DeviceView:
    """zero-copy torch view of a device allocation owned by libhalart.so"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class TileGather:
    """RCCL all-gather of a sharded renderer's AOVs + de-interleave on every rank.

    Usage (one process per GPU, torch.distributed initialised with backend "nccl" == RCCL):
        r.set_tile_shard(rank, world, 32); r.set_scene(...); r.commit()
        g = TileGather(r, device_index)
        for _ in range(spp): r.update()
        g.gather()            # accum, albedo, normal are now complete row-major images on every rank
    """

    def __init__(self, renderer, device_index, aovs=(0, 1, 2), group=None):
        import torch
        import torch.distributed as dist
        self.r, self.dist, self.group, self.torch = renderer, dist, group, torch
        self.world = dist.get_world_size(group)
        self.bufs = []
        for which in aovs:
            ptr, nbytes = renderer.tile_buffer(which)
            src = torch.as_tensor(_DeviceView(ptr, nbytes // 4), device=f"cuda:{device_index}")
            if src.data_ptr() != ptr:
                raise RuntimeError("TileGather: torch copied the tile buffer instead of aliasing it")
            dst = torch.empty(self.world * (nbytes // 4), dtype=torch.float32, device=f"cuda:{device_index}")
            self.bufs.append((which, src, dst, nbytes))

    # ---- pipelined form: the all-gather of frame k runs while frame k + 1 is rendered -------------------------------
    # xGMI is point-to-point: a ring all-gather of N x 33 MB (1080p RGBA32F per rank) is bound by ONE link per hop, i.e.
    # it can take as long as rendering the frame.  begin() snapshots the rank's tile buffer (the next frame overwrites
    # it) and starts the collective on a side stream; finish() — called by the next begin(), or explicitly — makes the
    # renderer's stream wait for it and de-interleaves.  Every frame is still gathered and de-interleaved in full.
    # With RCCL nothing here blocks the host: the hand-overs are stream dependencies between the renderer's HIP stream R
    # (hala_rt_get_stream) and the side stream S,
    #   finish(k-1): S waits for the collective; S: de-interleave(k-1), beside frame k on R; R's later work waits for S
    #   begin(k):    S waits for R (frame k rendered, receive buffer read out); S: staging <- tiles; R waits for that copy
    #                (frame k+1 may overwrite the tiles); S: all-gather(receive <- staging)
    # so the host can keep enqueueing frames (hala_rt_render bounds them to two in flight).  gloo (CPU collectives; the
    # 1-GPU rehearsal) synchronises instead.
    def _streams(self):
        torch = self.torch
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream()
            self._stage = [torch.empty_like(src) for _, src, _, _ in self.bufs]
            self._rstream = torch.cuda.ExternalStream(self.r.stream_handle())
            self._gloo = self.dist.get_backend(self.group) == "gloo"
        return self._side, self._rstream

    def begin(self):
        torch = self.torch
        side, rstream = self._streams()
        self.finish()  # frame k - 1 must have left the staging / receive buffers
        if self._gloo:
            self.r.wait_idle()
        side.wait_stream(rstream)  # frame k is complete in the rank's tile buffer, de-interleave k - 1 has read the receive buffer
        self._works = []
        with torch.cuda.stream(side):
            for (_, src, dst, _), stage in zip(self.bufs, self._stage):
                stage.copy_(src, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record(side)
            rstream.wait_event(copied)  # the next frame overwrites the tile buffer
            for (_, src, dst, _), stage in zip(self.bufs, self._stage):
                if self._gloo:
                    side.synchronize()
                    self._works.append(self.dist.all_gather(list(dst.view(self.world, -1).unbind(0)), stage, group=self.group, async_op=True))
                else:
                    self._works.append(self.dist.all_gather_into_tensor(dst, stage, group=self.group, async_op=True))
        self._pending = True

    def finish(self):
        if not getattr(self, "_pending", False):
            return
        side, rstream = self._streams()
        with self.torch.cuda.stream(side):
            for w in self._works:
                w.wait()  # RCCL: the side stream waits for the collective; gloo: the host does
        if self._gloo:
            side.synchronize()
        self._pending = False
        if self.world > 1:
            for which, _, dst, nbytes in self.bufs:  # on the side stream: the de-interleave of frame k runs beside the rendering of k + 1
                self.r.scatter_gathered_tiles(which, dst.data_ptr(), nbytes * self.world, stream=side.cuda_stream)
        rstream.wait_stream(side)  # whatever the renderer's stream does next (and whoever waits for it) sees the images complete

    def gather(self):
        self.finish()
        self.r.wait_idle()  # the renderer works on its own HIP stream
        for _, src, dst, _ in self.bufs:
            if self.dist.get_backend(self.group) == "gloo":  # rehearsal on a 1-GPU box; same buffer layout
                self.dist.all_gather(list(dst.view(self.world, -1).unbind(0)), src, group=self.group)
            else:
                self.dist.all_gather_into_tensor(dst, src, group=self.group)
        self.torch.cuda.synchronize()
        if self.world == 1:
            return  # an unsharded frame is already row-major
        for which, _, dst, nbytes in self.bufs:
            self.r.scatter_gathered_tiles(which, dst.data_ptr(), nbytes * self.world)

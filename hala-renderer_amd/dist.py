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
    owns it and stores it as its (k // world)-th tile; A = first integer >= 0x9E3779B1 % n coprime to n, B = 7.  Inside a tile whose
    size is a multiple of 8 the pixels are stored in 8 x 8 blocks (row-major over the blocks, row-major inside), else row-major."""

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
        if ts % 8 == 0:  # RENDER_SPEC §9: inside a tile the pixels come in 8 x 8 blocks (one block = one wave), row-major over the blocks
            w = np.arange(ts * ts)
            blk, j = w // 64, w % 64
            ly, lx = (blk // (ts // 8)) * 8 + j // 8, (blk % (ts // 8)) * 8 + j % 8
        else:
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
    """zero-copy torch view of a device allocation owned by libhalart.so (gloo rehearsal only)"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class TileGather:
    """All-gather of a sharded renderer's AOVs + de-interleave on every rank.

    The exchange itself lives in libhalart.so (`hala_rt_tile_allgather*`: RCCL on a side stream, stream-ordered against the
    renderer's stream, nothing blocks the host); this class only sets the communicator up.  `torch.distributed` is used for ONE
    thing: handing rank 0's 128-byte ncclUniqueId to the other ranks (any host channel would do — a C / Rust host uses its own).

    Usage (one process per GPU, torch.distributed initialised):
        r.set_tile_shard(rank, world, 32); r.set_scene(...); r.commit()
        g = TileGather(r, device_index)
        for _ in range(spp): r.update()
        g.gather()            # accum, albedo, normal are now complete row-major images on every rank
    or pipelined, one frame deep: g.begin() after frame k's updates, g.finish() before the images of frame k are read.

    backend "gloo" (CPU collectives; several ranks rehearsed on ONE GPU, which RCCL refuses): the library's own pipeline runs
    unchanged — staging copy, event order, de-interleave of the receive buffer in finish() — and only the exchange itself is swapped:
    hala_rt_tile_allgather_begin_external leaves it to the caller, who moves the staging buffers through
    torch.distributed.all_gather on the host into the library's receive buffer, on the library's exchange stream.
    """

    def __init__(self, renderer, device_index, aovs=(0, 1, 2), group=None, exchange=None):
        import torch
        import torch.distributed as dist
        self.r, self.dist, self.group, self.torch, self.aovs = renderer, dist, group, torch, tuple(aovs)
        self.device = f"cuda:{device_index}"
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._gloo = dist.get_backend(group) == "gloo"
        self._pending = False
        # exchange: "library" = RCCL inside libhalart.so (the design); "torch" = the same pipeline with the exchange handed to
        # torch.distributed.all_gather_into_tensor on the library's exchange stream (device to device, no host copy); "host" = gloo.
        # "torch" is what every rank falls back to when ANY rank could not create the library's communicator: a frame must not be lost
        # to a loader or version problem of the RCCL the process happens to find.
        self.exchange = "host" if self._gloo else (exchange or "library")
        self.fallback_reason = None
        if self.exchange == "library":
            import hala_renderer_amd as H
            failed = None
            try:
                ident = [renderer.comm_unique_id() if self.rank == 0 else None]
            except H.HalaRendererError as e:  # RCCL cannot be loaded: the same on every rank
                ident, failed = [None], str(e)
            dist.broadcast_object_list(ident, src=0, group=group)
            if ident[0] is None:
                failed = failed or "rank 0 could not create the ncclUniqueId"
            else:
                try:
                    renderer.comm_init_rank(ident[0], self.rank, self.world)
                except H.HalaRendererError as e:
                    failed = str(e)
            flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            if int(flag.item()):
                if not failed:
                    renderer.comm_destroy()
                self.exchange, self.fallback_reason = "torch", failed or "another rank could not create the library's communicator"

    def begin(self):
        if self.exchange == "library":
            self.r.tile_allgather_begin(self.aovs)
            return
        self.finish()
        self.r.tile_allgather_begin_external(self.aovs)  # stream-ordered snapshot of the tile buffers: the next frame may overwrite them
        self._pending = True
        if self.exchange == "torch":  # enqueued right away, like the library's own collective: it runs beside the next frame
            self._exchange_external()

    def _exchange_external(self):
        torch = self.torch
        for which in self.aovs:
            sp, sn, rp, rn, stream = self.r.exchange_buffers(which)
            if rn != sn * self.world:
                raise RuntimeError("TileGather: the receive buffer is not world x the staging buffer")
            ext = torch.cuda.ExternalStream(stream, device=self.device)
            with torch.cuda.stream(ext):  # the library's exchange stream: behind the staging copy, before finish()'s de-interleave
                staged = torch.as_tensor(_DeviceView(sp, sn // 4), device=self.device)
                recv = torch.as_tensor(_DeviceView(rp, rn // 4), device=self.device)
                if staged.data_ptr() != sp or recv.data_ptr() != rp:
                    raise RuntimeError("TileGather: torch copied an exchange buffer instead of aliasing it")
                if self.exchange == "torch":
                    self.dist.all_gather_into_tensor(recv, staged, group=self.group)  # stream-ordered on `ext`; nothing blocks the host
                    continue
                mine = staged.to("cpu")  # waits for the staging copy on this stream
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                self.dist.all_gather(parts, mine, group=self.group)
                recv.copy_(torch.cat(parts).to(self.device, non_blocking=False))
            if self.exchange == "host":
                ext.synchronize()

    def finish(self):
        if self.exchange == "library":
            self.r.tile_allgather_finish()
            return
        if not self._pending:
            return
        self._pending = False
        if self.exchange == "host":
            self._exchange_external()
        self.r.tile_allgather_finish()

    def gather(self):
        self.begin()
        self.finish()

    def close(self):
        if self.exchange == "library":
            self.r.comm_destroy()

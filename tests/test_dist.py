"""N > 1 path.  CPU tier: the tile layout (docs/RENDER_SPEC.md §9) against the oracle's statement of it, and a
world_size-2 gloo all-gather of tile-major shards + de-interleave.  GPU tier: a sharded renderer (several fake ranks
on one GPU) must reproduce the unsharded image bit for bit, through the same all-gather buffer layout."""
import os
import socket

import numpy as np
import pytest

from hala_renderer_amd import scenes
from hala_renderer_amd.dist import TileLayout


@pytest.mark.parametrize("w,h,world,ts", [(1920, 1080, 8, 32), (1920, 1080, 2, 32), (100, 70, 4, 16), (64, 64, 8, 32), (3840, 2160, 8, 32)])
def test_tile_layout_matches_oracle(oracle, w, h, world, ts):
    L = TileLayout(w, h, world, ts)
    owner, slot = oracle.tile_assignment(L.tiles_x, L.tiles_y, world)
    assert np.array_equal(owner, L.owner) and np.array_equal(slot, L.slot)
    counts = np.bincount(L.owner, minlength=world)
    assert counts.max() - counts.min() <= 1


def test_shard_unshard_roundtrip():
    rng = np.random.RandomState(0)
    for w, h, world, ts in [(100, 70, 4, 16), (64, 48, 3, 32), (33, 17, 8, 8)]:
        L = TileLayout(w, h, world, ts)
        img = rng.rand(h, w, 4).astype(np.float32)
        gathered = np.concatenate([L.shard(img, r) for r in range(world)])
        assert np.array_equal(L.unshard(gathered), img)
        # every real pixel is owned exactly once
        seen = np.zeros((h, w), dtype=np.int32)
        for r in range(world):
            m = L.rank_pixel_map(r)
            ok = m[:, 0] >= 0
            np.add.at(seen, (m[ok, 0], m[ok, 1]), 1)
        assert np.all(seen == 1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, w, h, ts, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = TileLayout(w, h, world, ts)
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        img = np.stack([yy, xx, yy * w + xx, np.ones_like(xx)], -1).astype(np.float32)  # every pixel carries its own id
        local = torch.from_numpy(L.shard(img, rank).reshape(-1).copy())
        out = torch.empty(world * local.numel(), dtype=torch.float32)
        dist.all_gather_into_tensor(out, local)  # same call and buffer layout as TileGather.gather() on RCCL
        full = L.unshard(out.numpy())
        q.put((rank, bool(np.array_equal(full, img))))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_all_gather_deinterleave():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 100, 70, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, True), (1, True)]


def _gathered(r, which, shape):
    """the receive buffer of the library's last collective, read through a zero-copy torch view"""
    import torch
    from hala_renderer_amd.dist import _DeviceView
    ptr, nbytes = r.gathered_buffer(which)
    torch.cuda.synchronize()
    return torch.as_tensor(_DeviceView(ptr, nbytes // 4), device="cuda:0").cpu().numpy().reshape(shape)


@pytest.mark.gpu
def test_rccl_all_gather_inside_the_library(halart):
    """the RCCL leg with the ranks this box has (one), through the C ABI only — no torch.distributed: ncclGetUniqueId, ncclCommInitRank,
    ncclAllGather on the library's side stream; the receive buffer must hold the rank's accumulated tiles, and a second communicator
    on the same renderer, a wrong rank and a gather without communicator must be refused"""
    r = halart.HalaRenderer("rccl", 96, 64, 4, 2, False, False, False, 0)
    r.set_scene(scenes.cornell_box(aspect=1.5))
    r.commit()
    with pytest.raises(halart.HalaRendererError, match="no communicator"):
        r.tile_allgather((r.ACCUM,))
    ident = halart.HalaRenderer.comm_unique_id()
    assert len(ident) == 128 and any(ident)
    with pytest.raises(halart.HalaRendererError, match="differ from the renderer's tile shard"):
        r.comm_init_rank(ident, 1, 2)
    r.comm_init_rank(ident, 0, 1)
    with pytest.raises(halart.HalaRendererError, match="already has a communicator"):
        r.comm_init_rank(ident, 0, 1)
    r.update_batch(2)
    r.tile_allgather((r.ACCUM, r.ALBEDO))
    r.wait_idle()
    assert np.array_equal(_gathered(r, r.ACCUM, (64, 96, 4)), r.read_image(r.ACCUM))
    assert np.array_equal(_gathered(r, r.ALBEDO, (64, 96, 4)), r.read_image(r.ALBEDO))
    r.comm_destroy()
    r.close()


@pytest.mark.gpu
def test_rccl_pipelined_gather_is_stream_ordered(halart):
    """begin() / finish() never block the host: frame k + 1 is enqueued on the renderer's stream before frame k's gather is finished;
    the receive buffer must still hold frame k (snapshot taken before the tiles are overwritten) — one rank, which is what this box
    has"""
    w, h = 480, 270
    scene = scenes.cornell_box(aspect=w / h)
    r = halart.HalaRenderer("rccl-pipe", w, h, 5, 3, False, False, False, 0)
    ref = halart.HalaRenderer("ref", w, h, 5, 3, False, False, False, 0)
    for x in (r, ref):
        x.set_scene(scene); x.commit()
    r.comm_init_rank(halart.HalaRenderer.comm_unique_id(), 0, 1)
    want = []
    spps = (1, 3, 2, 4)
    for spp in spps:
        ref.reset_accumulation(); ref.update_batch(spp)
        want.append(ref.read_image(ref.ACCUM))
    r.reset_accumulation(); r.update_batch(spps[0])
    for k in range(4):
        r.tile_allgather_begin((r.ACCUM,))  # (finishes gather k - 1,) snapshots frame k, starts its gather; the host is not blocked
        r.render()
        if k + 1 < 4:                        # frame k + 1 goes to the renderer's stream while gather k may still be running
            r.reset_accumulation(); r.update_batch(spps[k + 1])
        r.tile_allgather_finish()
        r.wait_idle()
        assert np.array_equal(_gathered(r, r.ACCUM, (h, w, 4)), want[k]), k
    r.close(); ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["library", "torch", "fallback"])
def test_tile_gather_over_torch_distributed_rccl(halart, exchange, monkeypatch):
    """what bench.py --gpus N does per rank, with the ranks this box has (one).  "library": torch.distributed (backend nccl = RCCL) only
    carries rank 0's ncclUniqueId; TileGather then drives the library's own communicator.  "torch": the library's pipeline with the
    exchange handed to torch.distributed.all_gather_into_tensor on the library's exchange stream (device to device).  "fallback": the
    library's communicator cannot be created (forced here) -> every rank agrees on "torch" and the frame still arrives."""
    import torch
    import torch.distributed as dist
    from hala_renderer_amd.dist import TileGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        r = halart.HalaRenderer("rccl", 96, 64, 4, 2, False, False, False, 0)
        r.set_scene(scenes.cornell_box(aspect=1.5))
        r.commit()
        if exchange == "fallback":
            def refuse(*a, **k):
                raise halart.HalaRendererError("forced: no communicator")
            monkeypatch.setattr(type(r), "comm_init_rank", refuse)
        g = TileGather(r, 0, aovs=(r.ACCUM,), exchange=None if exchange == "fallback" else exchange)
        assert g.exchange == ("library" if exchange == "library" else "torch")
        assert (g.fallback_reason is not None) == (exchange == "fallback")
        for spp in (2, 3):  # two frames through the pipelined form: a stale receive buffer would show
            r.reset_accumulation()
            r.update_batch(spp)
            g.begin(); r.render(); g.finish()
            r.wait_idle()
            assert np.array_equal(_gathered(r, r.ACCUM, (64, 96, 4)), r.read_image(r.ACCUM))
        g.close()
        r.close()
    finally:
        dist.destroy_process_group()


def _pipelined_worker(rank, world, port, q):
    """two real processes (gloo: RCCL refuses two ranks on one device) sharing cuda:0: sharded renders of three different
    frames through the pipelined begin()/finish() gather; every frame's gathered image must equal the unsharded render"""
    import torch
    import torch.distributed as dist
    import hala_renderer_amd as H
    from hala_renderer_amd.dist import TileGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = 160, 96
        scene = scenes.cornell_box(aspect=w / h)
        ref = H.HalaRenderer("ref", w, h, 4, 2, False, False, False, 0)
        ref.set_scene(scene); ref.commit()
        r = H.HalaRenderer("shard", w, h, 4, 2, False, False, False, 0)
        r.set_tile_shard(rank, world, 32)
        r.set_scene(scene); r.commit()
        g = TileGather(r, 0, aovs=(r.ACCUM,))
        ok = True
        want_prev = None
        for frame, spp in enumerate((1, 2, 3)):  # three different frames: a stale buffer would show
            r.reset_accumulation(); r.update_batch(spp)
            g.begin()                        # finishes the previous frame's gather first
            if want_prev is not None:
                ok = ok and np.array_equal(r.read_image(r.ACCUM), want_prev)
            ref.reset_accumulation(); ref.update_batch(spp)
            want_prev = ref.read_image(ref.ACCUM)
        g.finish()
        ok = ok and np.array_equal(r.read_image(r.ACCUM), want_prev)
        q.put((rank, bool(ok)))
        r.close(); ref.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_pipelined_gather_two_processes(halart):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_render_equals_unsharded(halart, world):
    """RNG is keyed by the global pixel id, so the union of the ranks' tiles is bit-identical to a 1-GPU render.
    The ranks are emulated one after another on the same GPU; the gathered buffer is assembled exactly as
    all_gather_into_tensor lays it out and de-interleaved by the library's HIP kernel."""
    import torch
    w, h, ts, spp = 200, 120, 32, 2
    scene = scenes.cornell_box(aspect=w / h)

    def render(rank, n):
        r = halart.HalaRenderer("shard", w, h, 5, 3, False, False, False, 0)
        if n > 1:
            r.set_tile_shard(rank, n, ts)
        r.set_scene(scene)
        r.commit()
        for _ in range(spp):
            r.update()
        r.render()
        r.wait_idle()  # torch reads the tile buffer on its own stream below; render() bounds the frames in flight, it does not flush
        return r

    ref = render(0, 1)
    ref_imgs = [ref.read_image(k) for k in range(3)]
    ref.close()
    L = TileLayout(w, h, world, ts)
    shards = {k: [] for k in range(3)}
    last = None
    for rank in range(world):
        r = render(rank, world)
        for k in range(3):
            ptr, nbytes = r.tile_buffer(k)
            assert nbytes == L.pixels_per_rank * 16
            t = torch.as_tensor(halart.dist._DeviceView(ptr, nbytes // 4), device="cuda:0").clone()
            shards[k].append(t)
        if last is not None:
            last.close()
        last = r
    for k in range(3):
        gathered = torch.cat(shards[k]).contiguous()
        # host-side statement of the layout
        assert np.array_equal(L.unshard(gathered.cpu().numpy()), ref_imgs[k])
        # the library's de-interleave kernel
        last.scatter_gathered_tiles(k, gathered.data_ptr(), gathered.numel() * 4)
        assert np.array_equal(last.read_image(k), ref_imgs[k])
    last.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_emulated_ranks_through_the_librarys_own_exchange_pipeline(halart, world):
    """the world > 1 leg of hala_rt_tile_allgather_begin / _finish on a one-GPU box: every emulated rank renders its share; on ONE of
    them the library's own pipeline runs — begin_external (stream-ordered snapshot into the staging buffer, receive buffer sized
    world x n), the exchange done by the test (device-to-device copies of every rank's staging buffer into the receive buffer, on the
    library's exchange stream), finish() (de-interleave of the receive buffer by k_scatter_tiles on that stream, full_valid, the
    renderer's stream waits) — pipelined: the next frame is enqueued before finish().  read_image must equal the unsharded frame."""
    import torch
    w, h, ts = 200, 120, 32
    scene = scenes.cornell_box(aspect=w / h)
    ref = halart.HalaRenderer("ref", w, h, 5, 3, False, False, False, 0)
    ref.set_scene(scene); ref.commit()
    ranks = []
    for rank in range(world):
        r = halart.HalaRenderer("shard", w, h, 5, 3, False, False, False, 0)
        r.set_tile_shard(rank, world, ts)
        r.set_scene(scene); r.commit()
        ranks.append(r)
    me = ranks[world - 1]
    want = []
    for frame, spp in enumerate((2, 1, 3)):
        ref.reset_accumulation(); ref.update_batch(spp)
        want.append([ref.read_image(k) for k in (ref.ACCUM, ref.NORMAL)])
    for frame, spp in enumerate((2, 1, 3)):
        for r in ranks:
            r.reset_accumulation(); r.update_batch(spp)
        aovs = (me.ACCUM, me.NORMAL)
        for r in ranks:
            r.tile_allgather_begin_external(aovs)  # every rank snapshots; `me` is the one whose receive buffer is filled
        if frame + 1 < 3:  # the next frame overwrites the tile buffers while the exchange is still open
            me.reset_accumulation(); me.update_batch(1)
        for which in aovs:
            _, sn, rp, rn, stream = me.exchange_buffers(which)
            assert rn == sn * world
            ext = torch.cuda.ExternalStream(stream, device="cuda:0")
            recv = torch.as_tensor(halart.dist._DeviceView(rp, rn // 4), device="cuda:0")
            for k, r in enumerate(ranks):
                sp, sn_k, _, _, stream_k = r.exchange_buffers(which)
                torch.cuda.ExternalStream(stream_k, device="cuda:0").synchronize()  # rank k's snapshot is complete
                staged = torch.as_tensor(halart.dist._DeviceView(sp, sn_k // 4), device="cuda:0")
                with torch.cuda.stream(ext):
                    recv[k * (sn // 4):(k + 1) * (sn // 4)].copy_(staged)
        for r in ranks:
            r.tile_allgather_finish()
        for j, which in enumerate(aovs):
            assert np.array_equal(me.read_image(which), want[frame][j]), (frame, which)
    with pytest.raises(halart.HalaRendererError, match="No exchange of this image is in flight"):
        me.exchange_buffers(me.ACCUM)
    for r in ranks:
        r.close()
    ref.close()


@pytest.mark.gpu
def test_tile_shard_change_with_a_communicator_or_a_gather_in_flight(halart):
    """hala_rt_set_tile_shard: a communicator is bound to (rank, world) — changing them under it is refused (the receive buffer is sized
    by the communicator, the de-interleave indexes it by the shard); a gather in flight is completed before the buffers are re-laid"""
    r = halart.HalaRenderer("reshard", 96, 64, 4, 2, False, False, False, 0)
    r.set_scene(scenes.cornell_box(aspect=1.5)); r.commit()
    r.comm_init_rank(halart.HalaRenderer.comm_unique_id(), 0, 1)
    with pytest.raises(halart.HalaRendererError, match="call hala_rt_comm_destroy before changing the tile shard"):
        r.set_tile_shard(0, 2, 32)
    r.set_tile_shard(0, 1, 16)  # same rank / world: allowed
    r.commit()
    r.update()
    r.tile_allgather_begin((r.ACCUM,))
    r.comm_destroy()
    r.set_tile_shard(1, 2, 32)  # completes / drops what was pending, re-lays the buffers
    r.commit()
    r.update()
    r.tile_allgather_begin_external((r.ACCUM,))
    _, sn, _, rn, _ = r.exchange_buffers(r.ACCUM)
    assert rn == 2 * sn
    r.set_tile_shard(0, 4, 32)  # finishes the open exchange first (its receive buffer belongs to the old world size)
    with pytest.raises(halart.HalaRendererError, match="No exchange of this image is in flight"):
        r.exchange_buffers(r.ACCUM)
    r.close()

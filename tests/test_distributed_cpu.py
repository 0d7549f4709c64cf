"""CPU, world_size 2 and 8 over gloo: ray sharding + the all-gather of rendered maps (the N > 1 path of bench.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intrinsicnerf_amd import distributed as idist
    b, e = idist.shard_bounds(n_total, rank, world)
    idx = torch.arange(b, e, dtype=torch.float32)
    # a deterministic "render": map values are functions of the global ray index
    maps = {"rgb_map": torch.stack([idx, idx + 0.25, idx + 0.5], 1), "disp_map": -idx, "acc_map": idx * 2,
            "albedo_map": torch.stack([idx * 3, idx * 3 + 1, idx * 3 + 2], 1), "shading_map": idx + 100,
            "residual_map": torch.stack([idx, idx, idx], 1) * 0.5}
    full = idist.gather_maps(maps, n_total)
    torch.save({k: v.clone() for k, v in full.items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _ssr_worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from intrinsicnerf_amd import distributed as idist
    layout = idist.ssr_map_layout(5, endpoint_feat=True)
    rays = torch.arange(n_total * 11, dtype=torch.float32).reshape(n_total, 11)        # the same frame on every rank

    def render(band):                        # a deterministic "render": every map a function of the ray's first float
        t = band[:, 0]
        return {k: (t[:, None] * (i + 1) + torch.arange(w)[None, :]).squeeze(-1) if w > 1 else t * (i + 1)
                for i, (k, w) in enumerate(layout)} | {"raw_fine": band}                # raw_* is not in the layout: stays local

    full = idist.render_sharded(render, rays, layout)
    torch.save({k: v.clone() for k, v in full.items()}, os.path.join(out_dir, f"ssr{rank}.pt"))
    dist.destroy_process_group()


def _frame_worker(rank, world, port, n_total, ssr_classes, out_dir):
    """One rank of a full-size frame through distributed.render_sharded: the band's maps are cheap functions of the global ray
    index (the first float of the ray row), so the gathered frame can be checked exactly without rendering anything."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from intrinsicnerf_amd import distributed as idist
    layout = idist.OBJECT_MAP_LAYOUT if ssr_classes < 0 else idist.ssr_map_layout(ssr_classes)
    rays = torch.zeros(n_total, 11)
    rays[:, 0] = torch.arange(n_total, dtype=torch.float32)
    seen = {}

    def render(band):
        seen["rows"] = (int(band[0, 0]), band.shape[0])
        t = band[:, 0]
        return {k: (t[:, None] + 0.125 * (i + 1) + torch.arange(w)[None, :]) if w > 1 else t + 0.125 * (i + 1) for i, (k, w) in enumerate(layout)}

    full = idist.render_sharded(render, rays, layout)
    t = rays[:, 0]
    ok = all(torch.equal(full[k], (t[:, None] + 0.125 * (i + 1) + torch.arange(w)[None, :]) if w > 1 else t + 0.125 * (i + 1))
             for i, (k, w) in enumerate(layout))
    torch.save({"ok": ok, "band": seen["rows"], "keys": sorted(full)}, os.path.join(out_dir, f"frame{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,ssr_classes", [(640000, -1), (76800, 28), (76801, 28)])
def test_eight_rank_full_frames(tmp_path, n_total, ssr_classes):
    """BASELINE configs[2] / [4] at the driver's largest scale: the 800x800 object frame (12 floats per ray) and the 320x240 SSR
    frame (26 + 2C floats per ray, C = 28) tiled over EIGHT ranks - contiguous bands that cover the frame, every rank ends up
    with the identical full frame; 76 801 rays: the ragged path (bands differ by one ray)."""
    from intrinsicnerf_amd import distributed as idist
    world = 8
    mp.spawn(_frame_worker, args=(world, _free_port(), n_total, ssr_classes, str(tmp_path)), nprocs=world, join=True)
    covered = 0
    for r in range(world):
        rec = torch.load(os.path.join(tmp_path, f"frame{r}.pt"))
        assert rec["ok"], f"rank {r}: gathered frame differs"
        b, e = idist.shard_bounds(n_total, r, world)
        assert rec["band"] == (b, e - b)
        covered += e - b
    assert covered == n_total


@pytest.mark.parametrize("n_total", [12, 13])
def test_two_rank_ssr_frame(tmp_path, n_total):
    """SSR frame (BASELINE configs[4]): coarse + fine maps, C logits, z_std and the endpoint feature in one all-gather."""
    from intrinsicnerf_amd import distributed as idist
    layout = idist.ssr_map_layout(5, endpoint_feat=True)
    assert sum(w for _, w in layout) == 2 * (13 + 5) + 1 + 128 and "raw_fine" not in dict(layout)
    assert [k for k, _ in idist.ssr_map_layout(3, n_importance=0)] == [
        "rgb_coarse", "disp_coarse", "acc_coarse", "depth_coarse", "albedo_coarse", "shading_coarse", "residual_coarse", "sem_logits_coarse"]
    world = 2
    mp.spawn(_ssr_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    t = torch.arange(n_total, dtype=torch.float32) * 11
    for r in range(world):
        full = torch.load(os.path.join(tmp_path, f"ssr{r}.pt"))
        assert set(full) == {k for k, _ in layout}
        for i, (k, w) in enumerate(layout):
            want = (t[:, None] * (i + 1) + torch.arange(w)[None, :]) if w > 1 else t * (i + 1)
            assert torch.equal(full[k], want), k


@pytest.mark.parametrize("n_total", [10, 11])       # even and ragged bands
def test_two_rank_gather(tmp_path, n_total):
    import __graft_entry__
    __graft_entry__.build()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    idx = torch.arange(n_total, dtype=torch.float32)
    for r in range(world):
        full = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        assert torch.equal(full["rgb_map"], torch.stack([idx, idx + 0.25, idx + 0.5], 1))
        assert torch.equal(full["disp_map"], -idx) and torch.equal(full["acc_map"], idx * 2)
        assert torch.equal(full["shading_map"], idx + 100)
        assert torch.equal(full["albedo_map"][:, 2], idx * 3 + 2) and torch.equal(full["residual_map"][:, 1], idx * 0.5)


def test_shard_bounds_cover_everything():
    from intrinsicnerf_amd.distributed import shard_bounds
    for n in (0, 1, 7, 640000, 76800):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1

"""Multi-GPU rendering: rays are independent, so a frame is split into contiguous row bands, one per
rank (one process per GPU), each rank renders its band with the single-GPU path, and the rendered
maps are exchanged with ONE all-gather over RCCL/xGMI.  No other collective is on the data path and
the networks are replicated (2 x 2.7 MB).

The reference has no distributed code at all; this is the new capability
BASELINE.json's config 5 asks for.  Payload per ray is 12 floats for the object-level maps
(rgb3 disp acc albedo3 shading residual3) - 31 MB for an 800x800 frame - so the collective is
latency-bound; raw tensors are never gathered.
"""
import torch
import torch.distributed as dist

OBJECT_MAP_LAYOUT = (("rgb_map", 3), ("disp_map", 1), ("acc_map", 1), ("albedo_map", 3), ("shading_map", 1),
                     ("residual_map", 3))


def ssr_map_layout(n_classes, enable_semantic=True, n_importance=128, endpoint_feat=False):
    """Layout of what ``SSRTrainer.render_rays`` returns per ray (trainer.py:776-797; ``raw_*`` never travels): the
    coarse and fine maps, the semantic logits and ``z_std`` - 26 + 2C floats per ray with a fine pass (8.4 MB per
    320x240 frame at C = 28)."""
    base = (("rgb", 3), ("disp", 1), ("acc", 1), ("depth", 1), ("albedo", 3), ("shading", 1), ("residual", 3))
    out = []
    for level in ("coarse", "fine") if n_importance > 0 else ("coarse",):
        out += [(k + "_" + level, w) for k, w in base]
        if enable_semantic:
            out.append(("sem_logits_" + level, int(n_classes)))
    if n_importance > 0:
        out.append(("z_std", 1))
        if endpoint_feat:
            out.append(("feat_map_fine", 128))
    return tuple(out)


def render_sharded(render_fn, rays, layout, group=None):
    """Render this rank's band of ``rays`` ([N, 11], the same tensor on every rank) with ``render_fn(band) -> dict`` and
    all-gather the maps named in ``layout``; every rank returns the full-frame dict."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, end = shard_bounds(rays.shape[0], rank, world)
    local = render_fn(rays[begin:end].contiguous())
    return gather_maps(local, rays.shape[0], layout, group)


def shard_bounds(n_rays, rank, world_size):
    """[begin, end) of this rank's contiguous band; bands differ in size by at most one ray."""
    base, extra = divmod(n_rays, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pack_maps(maps, layout=OBJECT_MAP_LAYOUT):
    """dict of per-ray maps -> one [n_local, sum(widths)] tensor (a single collective instead of six)."""
    cols = [maps[k].reshape(maps[k].shape[0], -1) for k, _ in layout]
    for c, (k, w) in zip(cols, layout):
        if c.shape[1] != w:
            raise ValueError(f"{k} has width {c.shape[1]}, layout says {w}")
    return torch.cat(cols, dim=1).contiguous()


def unpack_maps(packed, layout=OBJECT_MAP_LAYOUT):
    out, c = {}, 0
    for k, w in layout:
        out[k] = packed[:, c] if w == 1 else packed[:, c:c + w]
        c += w
    return out


def gather_maps(local_maps, n_rays_total, layout=OBJECT_MAP_LAYOUT, group=None):
    """All-gather every rank's band of rendered maps; returns the full-frame dict on every rank.

    Equal bands (``n_rays_total`` divisible by the world size - every frame size of BASELINE.json at 2, 4 and 8 GPUs):
    the collective writes straight into the final ``[n_rays_total, width]`` tensor and the returned maps are views
    of it - one pack, one collective, nothing else.  Otherwise bands differ by one ray: each rank pads its block to
    the largest band before the (fixed-size) ``all_gather_into_tensor`` and the padding rows are dropped afterwards.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    packed = pack_maps(local_maps, layout)
    if world == 1:
        return unpack_maps(packed, layout)
    width = packed.shape[1]
    if n_rays_total % world == 0:
        if packed.shape[0] * world != n_rays_total:
            raise ValueError(f"this rank rendered {packed.shape[0]} rays, expected {n_rays_total // world}")
        full = packed.new_empty(n_rays_total, width)
        dist.all_gather_into_tensor(full, packed, group=group)
        return unpack_maps(full, layout)
    biggest = (n_rays_total + world - 1) // world
    block = packed
    if packed.shape[0] < biggest:
        block = torch.cat([packed, packed.new_zeros(biggest - packed.shape[0], width)], 0)
    full = packed.new_empty(world * biggest, width)
    dist.all_gather_into_tensor(full, block.contiguous(), group=group)
    rows = []
    for r in range(world):
        b, e = shard_bounds(n_rays_total, r, world)
        rows.append(full[r * biggest: r * biggest + (e - b)])
    return unpack_maps(torch.cat(rows, 0), layout)

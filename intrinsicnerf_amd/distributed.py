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

    Bands may differ by one ray, so each rank pads its block to the largest band before the
    (fixed-size) ``all_gather_into_tensor`` and the padding rows are dropped afterwards.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    packed = pack_maps(local_maps, layout)
    if world == 1:
        return unpack_maps(packed, layout)
    width = packed.shape[1]
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

"""Image-level loop around the render path: N poses -> N frames of maps on the host.

The reference's loops (``render_path``: object_level/run_nerf.py:142-212, SSR/training/trainer.py:1221-1389) render one
image, then pull each map to the host with its own ``.cpu().numpy()`` - six blocking device->host copies per image, each
of which first waits for the whole render.  Here a frame's maps are packed into ONE device tensor, copied into one of two
pinned host buffers on a side stream, and the next frame's kernels are enqueued while that copy runs; the host touches a
frame only when the frame after it has been enqueued.  ``to8b`` happens on the device when only 8-bit images are wanted
(``inerf_frame_to_u8``), a quarter of the bytes.  The ray batch of a pose comes from ``inerf_gen_rays``.
"""
import os
import struct
import zlib

import numpy as np
import torch

from . import kernels


class FrameStreamer:
    """Double-buffered device->host transfer of per-frame packs.

    ``push(pack)`` enqueues the copy of a device tensor (any dtype; every call must use the same shape / dtype) on a side
    stream and returns the numpy array of the frame pushed BEFORE the previous one, if it is done - frames come out in
    order, two calls late; ``drain()`` returns the rest.  The caller's stream never waits for a copy."""

    def __init__(self, device, depth=2):
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.depth = depth
        self.slots = []          # [host buffer, event, device tensor kept alive until the copy is done]
        self.pending = []        # slot indices in flight, oldest first
        self.next = 0

    def _collect(self, slot):
        host, event, _ = self.slots[slot]
        event.synchronize()
        self.slots[slot][2] = None
        return host.numpy().copy()

    def push(self, pack):
        out = None
        if len(self.slots) < self.depth:
            self.slots.append([torch.empty(pack.shape, dtype=pack.dtype, pin_memory=True), torch.cuda.Event(), None])
            slot = len(self.slots) - 1
        else:
            slot = self.pending.pop(0)
            out = self._collect(slot)
        host = self.slots[slot][0]
        if host.shape != pack.shape or host.dtype != pack.dtype:
            raise ValueError("FrameStreamer: every frame must have the same shape and dtype")
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))       # the pack is complete on the render stream here
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            host.copy_(pack, non_blocking=True)
            self.slots[slot][1].record(self.copy_stream)
        self.slots[slot][2] = pack                                  # keep the device tensor alive until then
        self.pending.append(slot)
        return out

    def drain(self):
        out = [self._collect(s) for s in self.pending]
        self.pending = []
        return out


def pack_maps(maps, keys):
    """[n, sum(widths)] fp32 tensor of the named per-ray maps side by side (one copy instead of len(keys))."""
    cols = [maps[k].reshape(maps[k].shape[0], -1).float() for k in keys]
    return torch.cat(cols, 1), [c.shape[1] for c in cols]


def unpack_frame(frame, widths, keys, image_shape):
    out, c = {}, 0
    for k, w in zip(keys, widths):
        a = frame[:, c:c + w]
        out[k] = a.reshape(*image_shape, w) if w > 1 else a.reshape(*image_shape)
        c += w
    return out


def write_png(path, image):
    """Minimal PNG encoder (8-bit grey / RGB, 16-bit grey) for hosts without imageio - the reference writes its frames
    with ``imageio.imwrite`` (run_nerf.py:193-211), which is used instead when importable."""
    a = np.ascontiguousarray(image)
    if a.dtype != np.uint16:          # 16-bit maps (disp / depth in mm, trainer.py:1359,1365: format="png", prefer_uint8=False) always
        try:                          # take the encoder below: what imageio does with uint16 depends on its version and plugin
            import imageio
            if hasattr(imageio, "imwrite"):
                imageio.imwrite(path, image)
                return
        except ImportError:
            pass
    if a.dtype == np.uint16:
        depth, a = 16, a.astype(">u2")
    elif a.dtype == np.uint8:
        depth = 8
    else:
        raise ValueError(f"write_png: dtype {a.dtype}")
    if a.ndim == 2:
        colour, h, w = 0, *a.shape
    elif a.ndim == 3 and a.shape[2] == 3:
        colour, (h, w) = 2, a.shape[:2]
    else:
        raise ValueError(f"write_png: shape {a.shape}")
    raw = b"".join(b"\x00" + a[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def to8b(x):
    """(255 * clip(x, 0, 1)).astype(uint8) - run_nerf_helpers.py:13: numpy arrays on the host, device tensors through
    ``inerf_frame_to_u8``."""
    if isinstance(x, torch.Tensor):
        return kernels.frame_to_u8(x.float().contiguous()) if x.is_cuda else (255 * torch.clamp(x, 0, 1)).to(torch.uint8)
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def ensure_dir(path):
    if path is not None:
        os.makedirs(path, exist_ok=True)

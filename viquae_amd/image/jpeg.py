"""Host side of the split JPEG decoder (csrc/jpeg.hip): what `load_image` gets from ``Image.open(path).convert('RGB')``
(meerqat/data/loading.py:108-124), for the files the library covers, with the Huffman scan decoded on the host and the inverse
DCT / chroma upsampling / colour conversion on the GPU -- the same bytes as Pillow (oracle/jpeg.py is pinned against it).

``probe`` / ``stage`` need no GPU (the decode workers of viquae_amd/image/decode_pool.py call them); ``decode_files`` is the
one-call form (tests, small jobs): file bytes -> uint8 [H, W, 3] device tensors.  A file the library declines is NOT decoded
here by other means: the caller falls back to Pillow, which keeps the reference's errors and warnings."""
import ctypes
import os

import numpy as np

from .. import _lib

HEADER = 512            # MQ_JPEG_HEADER_BYTES
MAGIC_RGB = 0x20424752  # MQ_JPEG_MAGIC_RGB


def enabled():
    """MQ_IMAGE_DEVICE_JPEG=0: every file is decoded by Pillow on the host, as in rounds 3-5."""
    return os.environ.get("MQ_IMAGE_DEVICE_JPEG", "1") != "0"


def probe(data):
    """File bytes -> None (not a JPEG this library decodes) or (height, width, components, blocks, staging bytes)."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        return None
    info = np.zeros(6, dtype=np.int64)
    if _lib.load().mq_jpeg_probe(data, len(data), info.ctypes.data) != 0:
        return None
    return int(info[0]), int(info[1]), int(info[2]), int(info[3]), int(info[4])


def stage(data, address, capacity):
    """Huffman-decode ``data`` into the staging area at ``address`` (header + coefficient blocks) -> True, or False when the
    scan is irregular in any way (the caller then decodes the file with Pillow and stores it with :func:`stage_rgb`)."""
    return _lib.load().mq_jpeg_read_coefficients(data, len(data), ctypes.c_void_p(address), capacity) == 0


def stage_rgb(rgb, buf, off):
    """An image decoded by other means, uint8 [H, W, 3], stored where :func:`stage` would have put the file: the device copies
    its bytes to the image's place in the packed source buffer."""
    h, w = rgb.shape[:2]
    buf[off:off + HEADER] = 0
    buf[off:off + 12].view(np.int32)[:] = (MAGIC_RGB, h, w)
    buf[off + HEADER:off + HEADER + h * w * 3] = rgb.reshape(-1)


def strips(h, w):
    """16 x 2 pixel strips of an h x w image: the colour kernel's threads."""
    return ((h + 1) // 2) * ((w + 15) // 16)


def plan_layout(geom, totals, jpeg_rows):
    r"""The packed source buffer of a batch with split-decoded JPEG files in it.  ``geom`` / ``totals`` come from
    ``mq_image_plan`` (every image's RGB bytes one after the other); ``jpeg_rows`` = {row of geom: (blocks, staging bytes)}.
    Rewrites ``geom[:, 0]`` and ``totals[0]`` IN PLACE for the layout

        [ RGB of the host-decoded images | staging areas of the JPEG files | RGB of the JPEG files ]
          \_______________ copied from the host ________________________/   \__ written by the GPU __/

    -> dict(h2d_bytes, staging = {row: offset}, items int64 [n, 2], max_blocks, max_strips)."""
    rows = sorted(jpeg_rows)
    off = 0
    for r in range(len(geom)):
        if r not in jpeg_rows:
            geom[r, 0] = off
            off += (int(geom[r, 1]) * int(geom[r, 2]) * 3 + 15) & ~15
    staging = {}
    for r in rows:
        staging[r] = off
        off += int(jpeg_rows[r][1])
    h2d = off
    for r in rows:
        geom[r, 0] = off
        off += (int(geom[r, 1]) * int(geom[r, 2]) * 3 + 15) & ~15
    totals[0] = off
    return {"h2d_bytes": h2d, "staging": staging,
            "items": np.array([[staging[r], int(geom[r, 0])] for r in rows], dtype=np.int64).reshape(-1, 2),
            "max_blocks": max(int(jpeg_rows[r][0]) for r in rows), "max_strips": max(strips(int(geom[r, 1]), int(geom[r, 2])) for r in rows)}


def decode_staged(buf_dev, items, max_blocks, max_strips, stream=None):
    """items int64 [n, 2] (header offset, RGB offset) inside the uint8 device tensor ``buf_dev``; enqueues the kernels."""
    import torch
    items = np.ascontiguousarray(items, dtype=np.int64).reshape(-1, 2)
    if not len(items):
        return None
    idev = torch.from_numpy(items).to(buf_dev.device, non_blocking=True)
    st = stream if stream is not None else torch.cuda.current_stream(buf_dev.device).cuda_stream
    _lib.check(_lib.load().mq_jpeg_decode_rgb_u8(buf_dev.data_ptr(), idev.data_ptr(), len(items), int(max_blocks), int(max_strips), st),
               "mq_jpeg_decode_rgb_u8")
    return idev


def decode_files(datas, device=None):
    """[file bytes] -> [uint8 [H, W, 3] device tensor]; raises ValueError for a file the library does not decode."""
    import torch
    _lib.require_gpu()
    dev = torch.device(device if device is not None else "cuda")
    infos = []
    for n, d in enumerate(datas):
        p = probe(d)
        if p is None:
            raise ValueError(f"file {n}: not a JPEG the device decoder covers")
        infos.append(p)
    st_off, off = [], 0
    for p in infos:
        st_off.append(off)
        off += p[4]
    h2d = off
    rgb_off = []
    for p in infos:
        rgb_off.append(off)
        off += (p[0] * p[1] * 3 + 15) & ~15
    host = torch.empty(max(h2d, 16), dtype=torch.uint8, pin_memory=True)
    hnp = host.numpy()
    for n, (d, p, o) in enumerate(zip(datas, infos, st_off)):
        if not stage(d, hnp.ctypes.data + o, p[4]):
            raise ValueError(f"file {n}: irregular entropy-coded data")
    with torch.cuda.device(dev):
        buf = torch.empty(max(off, 16), dtype=torch.uint8, device=dev)
        buf[:h2d].copy_(host[:h2d], non_blocking=True)
        decode_staged(buf, np.array(list(zip(st_off, rgb_off)), dtype=np.int64), max(p[3] for p in infos),
                      max(strips(p[0], p[1]) for p in infos))
        torch.cuda.current_stream(dev).synchronize()
    return [buf[o:o + p[0] * p[1] * 3].view(p[0], p[1], 3) for p, o in zip(infos, rgb_off)]

"""Decode workers of the image-embedding pipeline: PROCESSES that turn image files into RGB bytes inside a shared, page-locked
staging buffer (viquae_amd/pipeline.py; the reference's own answer to slow decoding is a process pool too, `processes` in
meerqat/image/embedding.py:169-183 -- but its workers pickle every decoded image back through a pipe).

Pillow's per-file Python (open, plugin probing, convert, export) holds the GIL, so decode THREADS stop scaling at ~2x; here W
forked workers each own a contiguous chunk of a batch's files and work in two phases:

  sizes    open every file (lazy: header only), report (height, width) or the error; the opened images stay with the worker
  decode   given each image's byte offset in staging slot s (from mq_image_plan, computed by the parent between the phases),
           decode + convert to RGB and write the H x W x 3 bytes straight into the slot -- nothing crosses a pipe but offsets

JPEG files the library's split decoder covers (viquae_amd/image/jpeg.py, csrc/jpeg.hip; MQ_IMAGE_DEVICE_JPEG=0 switches it off) are
only Huffman-decoded here: the worker writes their quantised coefficients into the slot and the GPU does the inverse DCT,
the chroma upsampling and the colour conversion -- the same bytes as Pillow's, for about a third of the host time per file.  A
file whose scan turns out irregular is decoded by Pillow after all and stored as RGB in the same place.

The slots are anonymous shared mappings created BEFORE the fork (so every worker has them) and registered with the HIP
runtime as page-locked, which makes the host -> device copy of a packed batch a plain asynchronous DMA.  Workers never touch the
GPU.  A worker reproduces `meerqat.data.loading.load_image` (:108-119): unreadable or empty images are reported and become
`None` in the output, with the same warning text issued by the parent."""
import errno
import mmap
import multiprocessing as mp
import os
import warnings

import numpy as np

_SLOTS = []   # the staging mappings; inherited by the forked workers


_MAX_OPEN = 128  # lazily opened images a worker keeps between the `sizes` and the `decode` phase
_MAX_HELD = 64 << 20  # bytes of JPEG files a worker keeps between the phases (beyond that: read again in `decode`)


def _jpeg_bytes(path):
    """The whole file if it starts like a JPEG, else None (any error: None -- the Pillow path reports it in its own words)."""
    try:
        with open(path, "rb") as f:
            head = f.read(2)
            if head != b"\xff\xd8":
                return None
            return head + f.read()
    except OSError:
        return None


def _worker(conn, slots):
    import io
    from PIL import Image
    from . import jpeg as dj
    device_jpeg = dj.enabled()
    opened = {}
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            return
        kind = msg[0]
        if kind == "quit":
            return
        if kind == "sizes":
            _, base, paths = msg
            opened.clear()
            out = []
            held = 0
            for n, path in enumerate(paths):
                try:
                    data = _jpeg_bytes(path) if device_jpeg else None
                    info = dj.probe(data) if data is not None else None
                    if info is not None:   # (height, width, components, blocks, staging bytes): the scan is decoded in `decode`
                        held += len(data)
                        opened[base + n] = ("jpeg", data if held <= _MAX_HELD else None, path, info)
                        out.append(((info[0], info[1]), None, (info[3], info[4])))
                        continue
                    del data
                    # Lazily opened images keep their file descriptor until they are decoded.  Only a bounded number stays open
                    # between the two phases (a worker with a 1000-image chunk would pass the usual RLIMIT_NOFILE of 1024, and
                    # the EMFILE would be reported as an unreadable image): the rest is closed here and reopened in `decode`.
                    im = Image.open(path)
                    w, h = im.size
                    if w < 1 or h < 1:
                        im.close()
                        out.append((None, f"Empty image '{path}'", None))
                        continue
                    if len(opened) >= _MAX_OPEN:
                        im.close()
                        im = None
                    opened[base + n] = (im, path)
                    out.append(((h, w), None, None))
                except OSError as e:
                    if e.errno in (errno.EMFILE, errno.ENFILE, errno.ENOMEM):
                        raise  # resource exhaustion is not "this image is unreadable"
                    out.append((None, f"Caught exception '{e}' with image '{path}'", None))
                except Exception as e:  # noqa: BLE001 - load_image catches everything too
                    out.append((None, f"Caught exception '{e}' with image '{path}'", None))
            conn.send(out)
        elif kind == "call":
            # a small per-image host computation spread over the workers (e.g. the faces' alignment matrices): the named function
            # is applied to every item's argument; an exception travels back and is raised in the parent
            _, module, function, items = msg
            try:
                import importlib
                fn = getattr(importlib.import_module(module), function)
                conn.send([(i, fn(arg)) for i, arg in items])
            except Exception as e:  # noqa: BLE001
                conn.send(RuntimeError(f"{module}.{function} failed in a decode worker: {e!r}"))
        elif kind == "decode":
            _, slot, items = msg   # items: (index in the batch, byte offset in the slot, offset is a JPEG staging area)
            buf = np.frombuffer(slots[slot], dtype=np.uint8)
            failed = []
            for idx, off, staged in items:
                entry = opened.pop(idx)
                held = None
                if entry[0] == "jpeg" and not staged:
                    held, entry = entry[1], (None, entry[2])   # decoded by Pillow below, like the files of every other format
                if entry[0] == "jpeg":
                    _, data, path, info = entry
                    try:
                        if data is None:   # (not kept between the phases: _MAX_HELD)
                            with open(path, "rb") as f:
                                data = f.read()
                        if not dj.stage(data, buf.ctypes.data + off, info[4]):
                            # an irregular scan (truncated, damaged, a marker inside): Pillow decides what this file is
                            im = Image.open(io.BytesIO(data))
                            a = np.asarray(im if im.mode == "RGB" else im.convert("RGB"))
                            if a.shape[:2] != (info[0], info[1]):
                                raise RuntimeError(f"Pillow decodes {a.shape[:2]}, the frame header says {info[:2]}")
                            dj.stage_rgb(a, buf, off)
                    except Exception as e:  # noqa: BLE001
                        failed.append((idx, f"Caught exception '{e}' with image '{path}'"))
                    continue
                im, path = entry
                try:
                    if im is None:
                        im = Image.open(path if held is None else io.BytesIO(held))
                    # load_image's `.convert('RGB')` is the identity on an RGB file: skip its full-size copy there
                    a = np.asarray(im if im.mode == "RGB" else im.convert("RGB"))
                    n = a.shape[0] * a.shape[1] * 3
                    buf[off:off + n] = a.reshape(-1)
                except Exception as e:  # noqa: BLE001
                    failed.append((idx, f"Caught exception '{e}' with image '{path}'"))
            opened.clear()
            conn.send(failed)


class DecodePool:
    def __init__(self, n_procs, slot_bytes, n_slots=2):
        import torch
        self.slot_bytes = int(slot_bytes)
        self.maps = [mmap.mmap(-1, self.slot_bytes) for _ in range(n_slots)]
        self.tensors = [torch.frombuffer(m, dtype=torch.uint8) for m in self.maps]
        ctx = mp.get_context("fork")
        self.conns, self.procs = [], []
        for _ in range(max(1, int(n_procs))):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(child, self.maps), daemon=True)
            p.start()
            child.close()
            self.conns.append(parent)
            self.procs.append(p)
        # page-lock the slots AFTER the fork (MQ_IMAGE_PIN=0: leave them pageable): the children only ever see plain shared
        # memory, and the fork never meets a registered range of THIS mapping
        self.pinned = []
        for t in self.tensors:
            ok = False
            if torch.cuda.is_available() and os.environ.get("MQ_IMAGE_PIN", "1") != "0":
                try:
                    ok = int(torch.cuda.cudart().cudaHostRegister(t.data_ptr(), t.numel(), 0)) == 0
                except Exception:  # noqa: BLE001 - unpinned staging still works (a blocking copy in the prefetch thread)
                    ok = False
            self.pinned.append(ok)
        self.next_slot = 0
        self._chunks = None
        self.last_jpeg = {}

    def _recv(self, conn, what):
        if not conn.poll(600):
            raise RuntimeError(f"image decode worker did not answer ({what})")
        return conn.recv()

    def sizes(self, paths):
        """-> [(h, w) or None] per file; warnings for the unreadable ones (load_image's wording).  ``self.last_jpeg`` names the
        files that go through the split JPEG decoder."""
        n, w = len(paths), len(self.conns)
        per = -(-n // w)
        self._chunks = [(c, c * per, min(n, (c + 1) * per)) for c in range(w) if c * per < n]
        for c, lo, hi in self._chunks:
            self.conns[c].send(("sizes", lo, paths[lo:hi]))
        out = []
        self.last_jpeg = {}   # index in the batch -> (blocks, staging bytes) of the files whose scan the workers will decode
        for c, lo, hi in self._chunks:
            for size, err, jpeg in self._recv(self.conns[c], "sizes"):
                if err:
                    warnings.warn(err)
                if jpeg is not None:
                    self.last_jpeg[len(out)] = jpeg
                out.append(size)
        return out

    def take_slot(self):
        s = self.next_slot
        self.next_slot = (s + 1) % len(self.maps)
        return s

    def decode(self, slot, offsets, staged=None):
        """offsets: {index in the batch: byte offset in `slot`} for the images to decode (RGB bytes, or the staging area of a
        ``last_jpeg`` file) -> set of indices that failed.  ``staged``: the indices whose offset IS a staging area (default: every
        ``last_jpeg`` file); a ``last_jpeg`` file outside it is decoded by Pillow to RGB bytes like any other file."""
        self.decode_start(slot, offsets, staged)
        return self.decode_finish()

    def decode_start(self, slot, offsets, staged=None):
        """The first half of :meth:`decode`: the workers get their items and start; the caller does something else meanwhile."""
        staged = set(self.last_jpeg) if staged is None else staged
        for c, lo, hi in self._chunks:
            self.conns[c].send(("decode", slot, [(i, int(offsets[i]), i in staged) for i in range(lo, hi) if i in offsets]))

    def call_start(self, module, function, items):
        """``items``: {index in the batch (of the last :meth:`sizes`): argument}.  Every worker applies ``module.function`` to the
        arguments of its share of the batch; collect with :meth:`call_finish` (replies come in the order the requests were sent:
        a ``call_start`` before a ``decode_start`` is finished before it)."""
        for c, lo, hi in self._chunks:
            self.conns[c].send(("call", module, function, [(i, items[i]) for i in range(lo, hi) if i in items]))

    def call_finish(self):
        out = {}
        for c, lo, hi in self._chunks:
            got = self._recv(self.conns[c], "call")
            if isinstance(got, Exception):
                raise got
            out.update(got)
        return out

    def decode_finish(self):
        failed = set()
        for c, lo, hi in self._chunks:
            for idx, err in self._recv(self.conns[c], "decode"):
                warnings.warn(err)
                failed.add(idx)
        return failed

    def close(self):
        import torch
        for conn in self.conns:
            try:
                conn.send(("quit",))
                conn.close()
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        for t, ok in zip(self.tensors, self.pinned):
            if ok:
                try:
                    torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
                except Exception:  # noqa: BLE001
                    pass
        self.tensors = []
        for m in self.maps:
            try:
                m.close()
            except Exception:  # noqa: BLE001 - exported buffers may still be referenced
                pass


def slot_bytes(batch_size):
    """Bytes of one staging slot for batches of ``batch_size`` images: MQ_IMAGE_SLOT_KB (default 768 KB = a 512 x 512 RGB image)
    per image; a batch that needs more takes the thread path."""
    return max(64 << 20, int(batch_size) * int(os.environ.get("MQ_IMAGE_SLOT_KB", "768")) * 1024)


def early_pool(processes, batch_size):
    """The decode workers, forked BEFORE the caller loads the model or allocates page-locked memory (see
    ImageEmbedPipeline: the first device operation after a fork pays for what the process has pinned) -- or None when the
    pipeline / the processes are switched off or there is no GPU."""
    import torch
    if os.environ.get("MQ_EMBED_PIPELINE", "1") == "0" or not hasattr(os, "fork") or not torch.cuda.is_available():
        return None
    procs = int(processes) if processes else default_procs()
    if procs <= 0:
        return None
    try:
        return DecodePool(procs, slot_bytes(batch_size), n_slots=2)
    except Exception as e:  # noqa: BLE001
        warnings.warn(f"image decode processes unavailable ({e!r}): decoding in threads")
        return None


def default_procs():
    """MQ_IMAGE_DECODE_PROCS, else a quarter of the hardware threads (at most 32); 0 = decode in threads."""
    env = os.environ.get("MQ_IMAGE_DECODE_PROCS")
    if env is not None:
        return max(0, int(env))
    return max(0, min(32, (os.cpu_count() or 1) // 4))

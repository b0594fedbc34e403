"""Software pipeline behind ``Dataset.map`` for the encode jobs (VERDICT r2 "What's missing" 1).

The reference's ``embed`` (meerqat/ir/embedding.py:197-246, meerqat/image/embedding.py:125-166) is strictly serial per batch:
tokenise / decode on the CPU -> pageable synchronous ``.to(device)`` -> forward -> ``.cpu().numpy()`` -> Arrow write, so the
GPU idles during every host stage.  ``Dataset.map`` stays the driver here (same call, same batches, same output dataset), but
the function it calls is a :class:`Lookahead` step:

* a WORKER THREAD prepares batch j + 1 while the GPU runs batch j: it reads the batch's rows straight from the Arrow table,
  does the host work (tokenisation into PINNED buffers / image decoding and packing), issues the host -> device copies with
  ``non_blocking=True`` on a SIDE STREAM, builds whatever the forward would otherwise have to read back from the device (the
  packed-forward plan from the tokenizer's lengths) and records an event;
* the map function for batch i first LAUNCHES batch i + 1 on the compute stream (which waits for the worker's event on the
  device, not on the host), then waits for ONE event -- batch i's result landing in pinned memory -- and returns it.  The
  Arrow write of batch i (done by ``Dataset.map`` after the function returns) therefore overlaps the forward of batch i + 1.

Results are identical to the serial path: the same tokenizer backend (validated against ``tokenizer(texts, **kwargs)`` on the
first batch, else the tokenizer itself is called in the worker), the same forward.  Anything the prefetcher cannot know ahead
of ``Dataset.map`` (query expansion, KB features, shuffled / filtered datasets, ``num_proc``) keeps the serial path.
"""
import itertools
import os
import queue
import threading
import time

import numpy as np
import torch


class _Failure:
    def __init__(self, exc):
        self.exc = exc


class Lookahead:
    """``prepare(j)`` (worker thread) -> prepared inputs of batch j; ``launch(prepared)`` (caller's thread) enqueues the
    device work and returns a handle whose ``result()`` blocks until batch j's output is on the host.  ``step(i)`` returns
    batch i's result after making sure batch i + 1 is already running."""

    def __init__(self, n_batches, prepare, launch, depth=2):
        self.n, self._prepare, self._launch = int(n_batches), prepare, launch
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._handles = {}
        self._launched = 0
        self._stop = False
        self.wait_prepared_s = 0.0   # time the caller spent waiting for the worker (host-bound when this grows)
        self._thread = threading.Thread(target=self._work, name="viquae-amd-prefetch", daemon=True)
        self._thread.start()

    def _work(self):
        try:
            for j in range(self.n):
                if self._stop:
                    return
                item = self._prepare(j)
                while not self._stop:
                    try:
                        self._q.put(item, timeout=0.2)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:  # noqa: BLE001 - handed to the caller's thread
            self._q.put(_Failure(e))

    def _launch_next(self):
        t0 = time.perf_counter()
        item = self._q.get()
        self.wait_prepared_s += time.perf_counter() - t0
        if isinstance(item, _Failure):
            self._stop = True
            raise item.exc
        self._handles[self._launched] = self._launch(item)
        self._launched += 1

    def step(self, i):
        if i < self._launched - len(self._handles) or i >= self.n:
            raise RuntimeError(f"batch {i} was requested out of order (batches must be taken in sequence)")
        while self._launched <= min(i + 1, self.n - 1):
            self._launch_next()
        return self._handles.pop(i).result()

    def close(self):
        self._stop = True
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5.0)
        self._handles.clear()


class _Handle:
    def __init__(self, event, pinned, rows, extra=None):
        self.event, self.pinned, self.rows, self.extra = event, pinned, rows, extra

    def result(self):
        if self.event is not None:
            self.event.synchronize()
        return np.array(self.pinned[: self.rows].numpy()), self.extra  # a copy: the pinned slot is reused two batches later


class _PinnedRing:
    """Page-locked staging slots, reused round-robin; a slot is handed out again only after the copy that read it is done."""

    def __init__(self, slots):
        self.bufs = [None] * slots
        self.events = [None] * slots
        self.next = 0

    def take(self, nbytes):
        s = self.next
        self.next = (s + 1) % len(self.bufs)
        if self.events[s] is not None:
            self.events[s].synchronize()
            self.events[s] = None
        if self.bufs[s] is None or self.bufs[s].numel() < nbytes:
            self.bufs[s] = torch.empty(int(nbytes * 5 // 4) + 64, dtype=torch.uint8, pin_memory=True)
        return s, self.bufs[s]


# ----------------------------------------------------------------------------------------------------------------------
# text
# ----------------------------------------------------------------------------------------------------------------------
class FastBatchTokenizer:
    """``tokenizer(texts, **tokenization_kwargs)`` for the plain case -- a list of single texts, ``return_tensors="pt"``,
    ``padding`` "max_length" / True / "longest", optional truncation -- computed from the tokenizer's own Rust backend
    (``tokenizer.backend_tokenizer``: same normaliser, pre-tokeniser, model and post-processor) without the per-element Python
    passes of ``BatchEncoding.convert_to_tensors`` (transformers 5: 0.86 s of a 0.95 s call for 2048 passages padded to 256).
    The backend releases the GIL and encodes in parallel; the padded int64 matrices are then built with numpy.  ``check()``
    compares with the tokenizer's own output and must pass before the fast path is trusted."""

    SUPPORTED = {"return_tensors", "padding", "truncation", "max_length"}

    def __init__(self, tokenizer, tokenization_kwargs):
        self.tokenizer, self.kwargs = tokenizer, dict(tokenization_kwargs)
        self.ok = False
        self.left = getattr(tokenizer, "padding_side", "right") == "left"
        backend = getattr(tokenizer, "backend_tokenizer", None) or getattr(tokenizer, "_tokenizer", None)
        kw = self.kwargs
        if backend is None or set(kw) - self.SUPPORTED or kw.get("return_tensors") != "pt":
            return
        padding = kw.get("padding", False)
        if padding not in ("max_length", True, "longest"):
            return
        truncation = kw.get("truncation", False)
        if truncation not in (True, False, "longest_first"):
            return
        max_length = kw.get("max_length")
        if max_length is None and (truncation or padding == "max_length"):
            mml = getattr(tokenizer, "model_max_length", None)
            max_length = int(mml) if mml is not None and mml < 1_000_000 else None
        if padding == "max_length" and max_length is None:
            return
        try:
            import tokenizers
            self.backend = tokenizers.Tokenizer.from_str(backend.to_str())  # a private copy: truncation / padding are state
        except Exception:
            return
        self.backend.no_padding()
        if truncation and max_length is not None:
            self.backend.enable_truncation(int(max_length))
        else:
            self.backend.no_truncation()
        self.pad_to = int(max_length) if padding == "max_length" else None
        self.pad_id = int(tokenizer.pad_token_id if tokenizer.pad_token_id is not None else 0)
        self.left = getattr(tokenizer, "padding_side", "right") == "left"
        self.names = list(getattr(tokenizer, "model_input_names", ["input_ids", "attention_mask"]))
        if not set(self.names) <= {"input_ids", "token_type_ids", "attention_mask"}:
            return
        self.ok = True

    def encode(self, texts):
        """-> (lengths int64 [B], flat ids int64 [sum lengths])"""
        enc = self.backend.encode_batch_fast(texts, add_special_tokens=True) if hasattr(self.backend, "encode_batch_fast") \
            else self.backend.encode_batch(texts, add_special_tokens=True)
        ids = [e.ids for e in enc]
        lens = np.fromiter(map(len, ids), dtype=np.int64, count=len(ids))
        flat = np.fromiter(itertools.chain.from_iterable(ids), dtype=np.int64, count=int(lens.sum()))
        return lens, flat

    def fill(self, lens, flat, out):
        """out: dict name -> int64 numpy [B, L] views to fill (the tokenizer's padded matrices)."""
        B, L = out["input_ids"].shape
        cols = np.arange(L, dtype=np.int64)[None, :]
        mask = (cols >= L - lens[:, None]) if self.left else (cols < lens[:, None])
        ids = out["input_ids"]
        ids.fill(self.pad_id)
        ids[mask] = flat
        if "attention_mask" in out:
            np.copyto(out["attention_mask"], mask, casting="unsafe")
        if "token_type_ids" in out:
            out["token_type_ids"].fill(0)  # single texts: every token belongs to segment 0 (pad_token_type_id is 0 too)

    def width(self, lens):
        return self.pad_to if self.pad_to is not None else int(lens.max()) if lens.size else 0

    def __call__(self, texts):
        lens, flat = self.encode(texts)
        L = self.width(lens)
        out = {n: np.empty((len(texts), L), dtype=np.int64) for n in self.names}
        self.fill(lens, flat, out)
        return {n: torch.from_numpy(a) for n, a in out.items()}, lens

    def check(self, texts):
        """True when the fast path reproduces ``tokenizer(texts, **kwargs)`` exactly (keys, order, dtypes, values)."""
        if not self.ok:
            return False
        try:
            want = self.tokenizer(texts, **self.kwargs)
            got, _ = self(texts)
            same = list(want.keys()) == list(got.keys()) and all(
                want[k].dtype == got[k].dtype and want[k].shape == got[k].shape and bool(torch.equal(want[k], got[k])) for k in got)
        except Exception:
            same = False
        self.ok = bool(same)
        return self.ok


def _arrow_strings(dataset, key):
    """The Arrow column ``key`` of a dataset without an indices mapping, or None."""
    import pyarrow as pa
    if getattr(dataset, "_indices", None) is not None or key not in dataset.column_names:
        return None
    col = dataset.data.column(key)
    if not (pa.types.is_string(col.type) or pa.types.is_large_string(col.type)):
        return None
    return col


class TextEmbedPipeline:
    """``embed`` for ``Dataset.map(..., batched=True, with_indices=True)`` with the host stages of batch i + 1 (Arrow ->
    str, tokenisation, pinned staging, H2D, packed-forward plan) hidden behind the forward of batch i.  Plain passages /
    questions only (no query expansion, no KB features, no per-layer dump): see :func:`text_pipeline_or_none`."""

    def __init__(self, dataset, model, tokenizer, tokenization_kwargs, key, save_as, output_key, forward_kwargs, call, batch_size,
                 depth=2):
        self.model, self.key, self.save_as, self.output_key = model, key, save_as, output_key
        self.forward_kwargs, self.call = dict(forward_kwargs or {}), call
        self.tokenizer, self.tokenization_kwargs = tokenizer, dict(tokenization_kwargs or {})
        self.column = _arrow_strings(dataset, key)
        n = len(dataset)
        self.bounds = [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]
        self.device = next(iter(model.parameters()), None)
        self.device = (self.device if self.device is not None else next(iter(model.buffers()))).device
        self.main = torch.cuda.current_stream(self.device)
        self.side = torch.cuda.Stream(device=self.device)
        self.fast = FastBatchTokenizer(tokenizer, self.tokenization_kwargs)
        self._checked = False
        self.use_plan = bool(getattr(model, "supports_pack_plan", False)) and call is None and not self.fast.left
        self.in_ring = _PinnedRing(depth + 2)
        self.out_ring = _PinnedRing(3)
        self.stats = {"batches": 0, "tokenize_s": 0.0, "prepare_s": 0.0, "launch_s": 0.0, "wait_result_s": 0.0,
                      "fast_tokenizer": None, "pack_plan": self.use_plan}
        self.look = Lookahead(len(self.bounds), self._prepare, self._launch, depth=depth)

    # ---- worker thread ---------------------------------------------------------------------------------------------
    def _prepare(self, j):
        t0 = time.perf_counter()
        s, e = self.bounds[j]
        texts = self.column.slice(s, e - s).to_pylist()
        if not self._checked:
            self._checked = True
            self.stats["fast_tokenizer"] = self.fast.check(texts[: min(len(texts), 256)])
        lens = None
        tt = time.perf_counter()
        if self.fast.ok:
            lens, flat = self.fast.encode(texts)
            L = self.fast.width(lens)
            names = self.fast.names
            nbytes = len(names) * len(texts) * L * 8
            plan_bytes = int(lens.sum()) * 12 + len(texts) * 24 + 1024   # keep i64 + pos i32 per token; cu, classes, cls rows
            slot, buf = self.in_ring.take(nbytes + plan_bytes)
            host = {}
            for n_, name in enumerate(names):
                host[name] = buf[n_ * len(texts) * L * 8:(n_ + 1) * len(texts) * L * 8].view(torch.int64).view(len(texts), L)
            self.fast.fill(lens, flat, {k: v.numpy() for k, v in host.items()})
        else:  # the tokenizer itself (whatever its kwargs mean), staged through pinned memory all the same
            enc = self.tokenizer(texts, **self.tokenization_kwargs)
            tensors = {k: v for k, v in enc.items()}
            nbytes = sum(v.numel() * v.element_size() + 64 for v in tensors.values())
            slot, buf = self.in_ring.take(nbytes)
            host, off = {}, 0
            for k, v in tensors.items():
                v = v.contiguous()
                nb = v.numel() * v.element_size()
                dst = buf[off:off + nb].view(v.dtype).view(v.shape)
                dst.copy_(v)
                host[k] = dst
                off += (nb + 63) // 64 * 64
        self.stats["tokenize_s"] += time.perf_counter() - tt
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            inputs = {k: torch.empty(v.shape, dtype=v.dtype, device=self.device).copy_(v, non_blocking=True) for k, v in host.items()}
            plan = None
            if self.use_plan and lens is not None and "attention_mask" in inputs:
                from .encoders import pack_plan_from_lengths
                cursor = [(nbytes + 63) // 64 * 64]

                def stage(a):  # the plan's host arrays go through the same pinned slot as the token ids
                    nb = a.nbytes
                    dst = buf[cursor[0]:cursor[0] + nb].view(torch.from_numpy(a).dtype).view(a.shape)
                    cursor[0] += (nb + 63) // 64 * 64
                    dst.numpy()[...] = a
                    return dst

                plan = pack_plan_from_lengths(lens, inputs["input_ids"].shape[1], self.device, stage=stage)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.in_ring.events[slot] = ev
        self.stats["prepare_s"] += time.perf_counter() - t0
        return {"texts": texts, "inputs": inputs, "plan": plan, "event": ev, "rows": len(texts)}

    # ---- caller's thread -------------------------------------------------------------------------------------------
    def _launch(self, p):
        t0 = time.perf_counter()
        with torch.cuda.device(self.device):
            self.main.wait_event(p["event"])
            for t in p["inputs"].values():
                t.record_stream(self.main)
            if p["plan"] is not None:
                keep, pos, cu, classes, cls_rows = p["plan"]
                for t in [keep, pos, cu, cls_rows] + [c[0] for c in classes]:
                    t.record_stream(self.main)
            method = self.model if self.call is None else getattr(self.model, self.call)
            kw = dict(self.forward_kwargs)
            if p["plan"] is not None:
                kw["pack_plan"] = p["plan"]
            with torch.no_grad():
                outputs = method(**p["inputs"], **kw)
            output = select_output(outputs, self.output_key)
            output = output.to(torch.float32) if output.dtype != torch.float32 else output
            output = output.contiguous()
            slot, buf = self.out_ring.take(output.numel() * 4)
            pinned = buf[: output.numel() * 4].view(torch.float32).view(output.shape)
            pinned.copy_(output, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.out_ring.events[slot] = ev
        self.stats["launch_s"] += time.perf_counter() - t0
        return _Handle(ev, pinned, p["rows"], extra=p["texts"])

    def embed(self, batch, indices):
        """The function ``Dataset.map`` calls (batched, with_indices): ``batch[save_as]`` = float32 [B, H] of THIS batch."""
        i = self.stats["batches"]
        s, e = self.bounds[i] if i < len(self.bounds) else (-1, -1)
        if len(indices) != e - s or int(indices[0]) != s or int(indices[-1]) != e - 1:
            raise RuntimeError("Dataset.map handed over batches in another order than the prefetcher prepared them "
                               f"(batch {i}: rows {indices[0]}..{indices[-1]}, expected {s}..{e - 1}); set MQ_EMBED_PIPELINE=0")
        t0 = time.perf_counter()
        out, texts = self.look.step(i)
        self.stats["wait_result_s"] += time.perf_counter() - t0
        if texts != batch[self.key]:
            raise RuntimeError(f"the prefetched texts of batch {i} differ from what Dataset.map decoded; set MQ_EMBED_PIPELINE=0")
        self.stats["batches"] = i + 1
        self.stats.setdefault("returned_at", []).append(time.perf_counter())  # per batch: lets a caller separate start-up from pace
        batch[self.save_as] = out
        return batch

    def close(self):
        self.stats["wait_prepared_s"] = self.look.wait_prepared_s
        self.look.close()


def select_output(outputs, output_key):
    """meerqat/ir/embedding.py:227-236: a tensor, or ``outputs[output_key]`` of a dict / list / tuple."""
    if isinstance(outputs, torch.Tensor):
        return outputs
    if isinstance(outputs, (dict, list, tuple)):
        if output_key is None:
            raise ValueError(f"You should set output_key to choose from the model's outputs (got {output_key})")
        return outputs[output_key]
    raise TypeError(f"Invalid type '{type(outputs)}' for model's outputs:\\n{outputs}")


def pipeline_enabled():
    return os.environ.get("MQ_EMBED_PIPELINE", "1") != "0"


def _plain_dataset(dataset, map_kwargs):
    from datasets import Dataset
    return (isinstance(dataset, Dataset) and getattr(dataset, "_indices", None) is None and len(dataset) > 0
            and not map_kwargs.get("num_proc") and map_kwargs.get("batched", True) and not map_kwargs.get("drop_last_batch")
            and not map_kwargs.get("with_rank") and not map_kwargs.get("input_columns") and not map_kwargs.get("with_indices"))


def _map_batch_size(map_kwargs):
    """``Dataset.map``'s batch size, or None when the pipeline does not apply: ``batch_size=None`` means "the whole dataset as one
    batch" there (nothing to pipeline), and a non-positive value is the caller's business."""
    bs = map_kwargs.get("batch_size", 1000)
    if bs is None or isinstance(bs, bool):
        return None
    try:
        bs = int(bs)
    except (TypeError, ValueError):
        return None
    return bs if bs > 0 else None


def text_pipeline_or_none(dataset, map_kwargs, model=None, tokenizer=None, tokenization_kwargs={}, key="passage",
                          save_as="text_embedding", output_key=None, forward_kwargs={}, layers=None, kb=None, call=None, run=None,
                          qe_predictions_key=None, **other):
    """A :class:`TextEmbedPipeline` when the job is the plain one the prefetcher can reproduce ahead of ``Dataset.map`` --
    a CUDA model, a plain ``datasets.Dataset`` walked in order, texts = ``batch[key]`` as they are -- else None (the caller
    then maps the serial ``embed``)."""
    if not pipeline_enabled() or other or layers is not None or kb is not None or run is not None or qe_predictions_key is not None:
        return None
    if _map_batch_size(map_kwargs) is None or getattr(tokenizer, "truncation_side", "right") != "right":
        return None  # (the fast tokenizer path truncates on the right like the tokenizer's default; anything else: the serial embed)
    if model is None or tokenizer is None or not torch.cuda.is_available() or not _plain_dataset(dataset, map_kwargs):
        return None
    from .ir.embedding import is_multimodal
    if is_multimodal(model) or not isinstance(model, torch.nn.Module):
        return None
    first = next(iter(model.parameters()), None)
    first = first if first is not None else next(iter(model.buffers()), None)
    if first is None or not first.is_cuda or _arrow_strings(dataset, key) is None:
        return None
    return TextEmbedPipeline(dataset, model, tokenizer, tokenization_kwargs, key, save_as, output_key, forward_kwargs, call,
                             _map_batch_size(map_kwargs))


# ----------------------------------------------------------------------------------------------------------------------
# images
# ----------------------------------------------------------------------------------------------------------------------
class ImageEmbedPipeline:
    """The image job (meerqat/image/embedding.py:125-166) with the host stages of batch i + 1 -- file decoding, packing the
    decoded RGB images into one pinned buffer, H2D and the Pillow-exact resize / crop / normalise kernels on a side stream --
    hidden behind the CLIP forward of batch i.  Needs the device-side transform (``transform.on_device``)."""

    def __init__(self, dataset, model, transform, save_as, image_key, call, pool, batch_size, depth=2, decode_procs=None,
                 decode_pool=None):
        self.model, self.transform, self.save_as, self.image_key, self.call, self.pool = model, transform, save_as, image_key, call, pool
        self.column = _arrow_strings(dataset, image_key)
        n = len(dataset)
        self.bounds = [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]
        first = next(iter(model.parameters()), None)
        self.device = (first if first is not None else next(iter(model.buffers()))).device
        self.main = torch.cuda.current_stream(self.device)
        self.side = torch.cuda.Stream(device=self.device)
        self.out_ring = _PinnedRing(3)
        self.stats = {"batches": 0, "decode_s": 0.0, "prepare_s": 0.0, "launch_s": 0.0, "wait_result_s": 0.0}
        # Decoding.  Default: forked decode PROCESSES that write the RGB bytes straight into a shared page-locked staging slot
        # (viquae_amd/image/decode_pool.py: Pillow's per-file Python holds the GIL, threads stop scaling at ~2x; `processes`
        # of the reference's config sets their number, MQ_IMAGE_DECODE_PROCS the default, 0 disables).  Otherwise the caller's
        # multiprocessing pool (arrays pickled back), or a few threads.
        from .image import decode_pool as dp
        self.decode, self.threads = decode_pool, None
        if self.decode is not None and not hasattr(transform, "run_packed"):
            self.decode.close()
            self.decode = None
        procs = decode_procs if decode_procs is not None else (dp.default_procs() if pool is None else 0)
        if self.decode is None and procs and hasattr(transform, "run_packed") and hasattr(os, "fork"):
            # (created here, the workers are forked from a process that already holds the model: the FIRST device operation
            # after a fork pays for the page-locked memory the process owns -- 4 s with the CLIP weights loaded, 22 s with 2 GB
            # of pinned buffers, measured; image.embedding.dataset_embed therefore forks them before it loads anything)
            try:
                self.decode = dp.DecodePool(procs, dp.slot_bytes(batch_size), n_slots=2)
            except Exception as e:  # noqa: BLE001 - e.g. not enough lockable memory: decode in threads instead
                import warnings
                warnings.warn(f"image decode processes unavailable ({e!r}): decoding in threads")
        if pool is None:
            # also beside the decode processes: a batch whose packed bytes exceed the staging slots (images larger than the slots
            # were sized for) is decoded here, by a few threads -- not one image after the other inside the prefetch thread
            from concurrent.futures import ThreadPoolExecutor
            self.threads = ThreadPoolExecutor(max(1, int(os.environ.get("MQ_IMAGE_DECODE_THREADS", min(8, os.cpu_count() or 1)))))
        self._slot_busy = {}
        self.stats["decode"] = (f"{len(self.decode.procs)} processes -> shared pinned staging" if self.decode is not None else
                                "caller's process pool" if pool is not None else f"{self.threads._max_workers} threads")
        self.look = Lookahead(len(self.bounds), self._prepare, self._launch, depth=depth)

    def _prepare_shared(self, names, t0):
        """Decode workers: sizes -> plan -> decode into the planned offsets of a shared slot -> device-side transform."""
        from .data import loading
        paths = [str(loading.IMAGE_PATH / n) for n in names]
        sizes = self.decode.sizes(paths)
        kept = [i for i, sz in enumerate(sizes) if sz is not None]
        if not kept:
            return kept, set(), None, None
        geom, totals = self.transform.plan(np.array([sizes[i] for i in kept], dtype=np.int64))
        # JPEG files whose scans the workers decode (decode_pool.last_jpeg): staging areas instead of RGB bytes in the slot, their
        # RGB produced on the device (viquae_amd/image/jpeg.py)
        jrows = {row: self.decode.last_jpeg[k] for row, k in enumerate(kept) if k in self.decode.last_jpeg}
        layout = None
        if jrows:
            from .image import jpeg as dj
            layout = dj.plan_layout(geom, totals, jrows)
        if layout is not None and int(layout["h2d_bytes"]) > self.decode.slot_bytes:
            # coefficients take 2 bytes per sample: a batch of 4:4:4 files can outgrow a slot its RGB bytes fit -- the workers
            # then decode this batch's files with Pillow, as they do for every other format
            geom, totals = self.transform.plan(np.array([sizes[i] for i in kept], dtype=np.int64))
            jrows, layout = {}, None
        if int(layout["h2d_bytes"] if layout else totals[0]) > self.decode.slot_bytes:
            return None  # larger images than the slots were sized for: this batch takes the thread (or caller's pool) path
        slot = self.decode.take_slot()
        busy = self._slot_busy.pop(slot, None)
        if busy is not None:      # the copy that last read this slot (two batches ago) must be over before it is rewritten
            busy[0].synchronize()
        failed = self.decode.decode(slot, {k: int(layout["staging"][row] if row in jrows else g[0]) for row, (k, g) in enumerate(zip(kept, geom))},
                                    staged={k for row, k in enumerate(kept) if row in jrows})
        self.stats["jpeg_scans_decoded_by_workers"] = self.stats.get("jpeg_scans_decoded_by_workers", 0) + len(jrows)
        self.stats["decode_s"] += time.perf_counter() - t0
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            # enqueued only: the DMA of this slot and the resize kernels run while the workers decode the NEXT batch into the
            # other slot
            got = self.transform.run_packed(self.decode.tensors[slot], geom, totals, len(kept), sync=False, jpeg=layout)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self._slot_busy[slot] = (ev, got.pop("_keep", None))
        return kept, failed, dict(got), ev

    def _prepare(self, j):
        from .data.loading import load_image_array, load_image_batch
        t0 = time.perf_counter()
        s, e = self.bounds[j]
        names = self.column.slice(s, e - s).to_pylist()
        if self.decode is not None:
            got = self._prepare_shared(names, t0)
            if got is not None:
                kept, failed, inputs, ev = got
                self.stats["prepare_s"] += time.perf_counter() - t0
                return {"names": names, "kept": kept, "failed": failed, "inputs": inputs, "event": ev, "rows": len(names)}
        if self.threads is not None:
            images = list(self.threads.map(load_image_array, names))
        else:
            images = load_image_batch(names, pool=self.pool, as_arrays=True)
        kept = [i for i, im in enumerate(images) if im is not None]
        self.stats["decode_s"] += time.perf_counter() - t0
        inputs = ev = None
        if kept:
            with torch.cuda.device(self.device), torch.cuda.stream(self.side):
                inputs = dict(self.transform([images[i] for i in kept], return_tensors="pt"))  # packs, copies, runs its kernels HERE
                ev = torch.cuda.Event()
                ev.record(self.side)
        self.stats["prepare_s"] += time.perf_counter() - t0
        return {"names": names, "kept": kept, "failed": set(), "inputs": inputs, "event": ev, "rows": len(names)}

    def _launch(self, p):
        t0 = time.perf_counter()
        if not p["kept"]:
            return _Handle(None, torch.empty((0, 1)), 0, extra=(p["names"], p["kept"], p["failed"], p["rows"]))
        with torch.cuda.device(self.device):
            self.main.wait_event(p["event"])
            for t in p["inputs"].values():
                t.record_stream(self.main)
            method = self.model if self.call is None else getattr(self.model, self.call)
            with torch.no_grad():
                out = method(**p["inputs"])
            if not isinstance(out, torch.Tensor):  # transformers >= 5 returns a ModelOutput
                out = out.pooler_output
            out = out.reshape(len(p["kept"]), -1).to(torch.float32).contiguous()
            slot, buf = self.out_ring.take(out.numel() * 4)
            pinned = buf[: out.numel() * 4].view(torch.float32).view(out.shape)
            pinned.copy_(out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.out_ring.events[slot] = ev
        self.stats["launch_s"] += time.perf_counter() - t0
        return _Handle(ev, pinned, len(p["kept"]), extra=(p["names"], p["kept"], p["failed"], p["rows"]))

    def embed(self, batch, indices):
        i = self.stats["batches"]
        s, e = self.bounds[i] if i < len(self.bounds) else (-1, -1)
        if len(indices) != e - s or int(indices[0]) != s or int(indices[-1]) != e - 1:
            raise RuntimeError("Dataset.map handed over batches in another order than the prefetcher prepared them; "
                               "set MQ_EMBED_PIPELINE=0")
        t0 = time.perf_counter()
        found, (names, kept, failed, rows) = self.look.step(i)
        self.stats["wait_result_s"] += time.perf_counter() - t0
        if names != batch[self.image_key]:
            raise RuntimeError(f"the prefetched file names of batch {i} differ from what Dataset.map decoded; set MQ_EMBED_PIPELINE=0")
        self.stats["batches"] = i + 1
        self.stats.setdefault("returned_at", []).append(time.perf_counter())
        if len(kept) == rows and not failed:
            batch[self.save_as] = found   # every image was readable: ONE [B, H] array (Arrow ingests it without a per-row pass)
            return batch
        output = [None] * rows
        if len(failed) == len(kept):
            return output  # (sic) the reference returns the bare list here: meerqat/image/embedding.py:134-135
        for row, k in enumerate(kept):
            if k not in failed:   # a file that opened but did not decode: None, like an unreadable one
                output[k] = found[row]
        batch[self.save_as] = output
        return batch

    def close(self):
        self.stats["wait_prepared_s"] = self.look.wait_prepared_s
        self.look.close()
        if self.threads is not None:
            self.threads.shutdown(wait=False)
        if self.decode is not None:
            self.decode.close()


def image_pipeline_or_none(dataset, map_kwargs, model=None, transform=None, save_as="image_embedding", image_key="image", call=None,
                           pool=None, decode_procs=None, decode_pool=None, **other):
    if not pipeline_enabled() or other or model is None or transform is None or not getattr(transform, "on_device", False):
        return None
    if not torch.cuda.is_available() or not _plain_dataset(dataset, map_kwargs) or _arrow_strings(dataset, image_key) is None:
        return None
    if _map_batch_size(map_kwargs) is None:
        return None
    return ImageEmbedPipeline(dataset, model, transform, save_as, image_key, call, pool, _map_batch_size(map_kwargs),
                              decode_procs=decode_procs, decode_pool=decode_pool)


# ----------------------------------------------------------------------------------------------------------------------
# faces
# ----------------------------------------------------------------------------------------------------------------------
class FaceEmbedPipeline:
    """The face job (meerqat/image/face_recognition.py:72-112: per batch, load every image that has ``face_landmarks``, align its
    first ``max_n_faces`` faces, run ArcFace) with the host stages of batch i + 1 -- file decoding by the decode workers (JPEG
    files: Huffman scan only, viquae_amd/image/jpeg.py), the 2 x 3 alignment matrices, H2D, the rest of the JPEG decoding and the
    alignment kernel on a side stream -- hidden behind the ArcFace forward of batch i.  Same outputs as the serial
    ``compute_face_embedding`` (the same faces in the same order through the same kernels)."""

    def __init__(self, dataset, model, max_n_faces, image_key, batch_size, decode_pool, depth=2):
        from .image import face_recognition as fr
        self.fr, self.model, self.max_n_faces, self.image_key, self.decode = fr, model, int(max_n_faces), image_key, decode_pool
        self.names = _arrow_strings(dataset, image_key)
        self.landmarks = dataset.data.column("face_landmarks")
        n = len(dataset)
        self.bounds = [(s, min(s + batch_size, n)) for s in range(0, n, batch_size)]
        first = next(iter(model.parameters()), None)
        self.device = (first if first is not None else next(iter(model.buffers()))).device
        self.main = torch.cuda.current_stream(self.device)
        self.side = torch.cuda.Stream(device=self.device)
        self.out_ring = _PinnedRing(3)
        self._slot_busy = {}
        self.tform = fr.SimilarityTransform()
        self.stats = {"batches": 0, "decode_s": 0.0, "prepare_s": 0.0, "launch_s": 0.0, "wait_result_s": 0.0, "faces": 0,
                      "decode": f"{len(decode_pool.procs)} processes -> shared pinned staging"}
        self.look = Lookahead(len(self.bounds), self._prepare, self._launch, depth=depth)

    def _prepare(self, j):
        from .data import loading
        from .image import jpeg as dj
        from . import _lib
        t0 = time.perf_counter()
        s, e = self.bounds[j]
        names = self.names.slice(s, e - s).to_pylist()
        marks = self.landmarks.slice(s, e - s).to_pylist()
        wanted = [i for i, lm in enumerate(marks) if lm is not None]      # only images with detected faces are loaded (:81-83)
        empty = {"names": names, "rows": len(names), "images": [], "counts": [], "inputs": None, "event": None}
        if not wanted:
            return empty
        sizes = self.decode.sizes([str(loading.IMAGE_PATH / names[i]) for i in wanted])
        kept = [k for k, sz in enumerate(sizes) if sz is not None]         # positions in `wanted` of the readable files
        if not kept:
            return empty
        geom = np.zeros((len(kept), 3), dtype=np.int64)                    # (offset, height, width) like mq_image_plan's first columns
        geom[:, 1:] = [sizes[k] for k in kept]
        totals = np.zeros(1, dtype=np.int64)
        jrows = {row: self.decode.last_jpeg[k] for row, k in enumerate(kept) if k in self.decode.last_jpeg}
        layout = None
        if jrows:
            layout = dj.plan_layout(geom, totals, jrows)
        if not jrows or int(layout["h2d_bytes"]) > self.decode.slot_bytes:   # (coefficients can outgrow a slot the RGB bytes fit)
            off = 0
            for row in range(len(kept)):
                geom[row, 0] = off
                off += (int(geom[row, 1]) * int(geom[row, 2]) * 3 + 15) & ~15
            totals[0] = off
            jrows, layout = {}, None
        h2d = int(layout["h2d_bytes"] if layout else totals[0])
        if h2d > self.decode.slot_bytes:
            raise RuntimeError(f"a batch of {len(kept)} images needs {h2d} bytes of staging, the slots hold {self.decode.slot_bytes}: "
                               "raise MQ_IMAGE_SLOT_KB or set MQ_EMBED_PIPELINE=0")
        slot = self.decode.take_slot()
        busy = self._slot_busy.pop(slot, None)
        if busy is not None:      # the copy that last read this slot (two batches ago) must be over before it is rewritten
            busy[0].synchronize()
        # the faces' alignment matrices (Umeyama's estimate per face: ~50 us of small numpy calls each) are the workers' job too
        self.decode.call_start("viquae_amd.image.face_recognition", "face_matrices", {k: (marks[wanted[k]], self.max_n_faces) for k in kept})
        self.decode.decode_start(slot, {k: int(layout["staging"][row] if row in jrows else geom[row, 0]) for row, k in enumerate(kept)},
                                 staged={k for row, k in enumerate(kept) if row in jrows})
        mats = self.decode.call_finish()
        per_image = [mats[k] for k in kept]
        failed = self.decode.decode_finish()
        self.stats["decode_s"] += time.perf_counter() - t0
        # faces of the images that decoded, in batch order: (row of geom, inverted 2 x 3 matrix)
        images, counts, owner, minv = [], [], [], []
        for row, k in enumerate(kept):
            if k in failed:
                continue
            images.append(wanted[k])
            counts.append(per_image[row][0])
            for m in per_image[row][1]:
                owner.append(row)
                minv.append(m)
        out = dict(empty, images=images, counts=counts)
        if not owner:
            return out
        lib = _lib.load()
        with torch.cuda.device(self.device), torch.cuda.stream(self.side):
            dev = self.device
            src = torch.empty(int(totals[0]), dtype=torch.uint8, device=dev)
            src[:h2d].copy_(self.decode.tensors[slot][:h2d], non_blocking=True)
            keep = [src]
            if layout is not None:
                keep.append(dj.decode_staged(src, layout["items"], layout["max_blocks"], layout["max_strips"]))
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev, non_blocking=True)  # noqa: E731
            off_d, hw_d = t(geom[:, 0]), t(geom[:, 1:].astype(np.int32).reshape(-1))
            own_d, minv_d = t(np.array(owner, np.int32)), t(np.stack(minv).reshape(-1))
            px = torch.empty((len(owner), 3, self.fr.IMAGE_SIZE, self.fr.IMAGE_SIZE), dtype=torch.float32, device=dev)
            _lib.check(lib.mq_warp_affine_faces_f32(src.data_ptr(), off_d.data_ptr(), hw_d.data_ptr(), own_d.data_ptr(), minv_d.data_ptr(),
                                                    len(owner), self.fr.IMAGE_SIZE, px.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                       "mq_warp_affine_faces_f32")
            ev = torch.cuda.Event()
            ev.record(self.side)
            keep += [off_d, hw_d, own_d, minv_d]
        self._slot_busy[slot] = (ev, keep)
        self.stats["prepare_s"] += time.perf_counter() - t0
        self.stats["faces"] += len(owner)
        out.update(inputs=px, event=ev)
        return out

    def _launch(self, p):
        t0 = time.perf_counter()
        extra = (p["names"], p["images"], p["counts"], p["rows"])
        if p["inputs"] is None:
            return _Handle(None, torch.empty((0, 1)), 0, extra=extra)
        with torch.cuda.device(self.device):
            self.main.wait_event(p["event"])
            p["inputs"].record_stream(self.main)
            with torch.no_grad():
                out = self.model(p["inputs"]).to(torch.float32).contiguous()
            slot, buf = self.out_ring.take(out.numel() * 4)
            pinned = buf[: out.numel() * 4].view(torch.float32).view(out.shape)
            pinned.copy_(out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.main)
            self.out_ring.events[slot] = ev
        self.stats["launch_s"] += time.perf_counter() - t0
        return _Handle(ev, pinned, out.shape[0], extra=extra)

    def embed(self, batch, indices):
        i = self.stats["batches"]
        s, e = self.bounds[i] if i < len(self.bounds) else (-1, -1)
        if len(indices) != e - s or int(indices[0]) != s or int(indices[-1]) != e - 1:
            raise RuntimeError("Dataset.map handed over batches in another order than the prefetcher prepared them; set MQ_EMBED_PIPELINE=0")
        t0 = time.perf_counter()
        found, (names, images, counts, rows) = self.look.step(i)
        self.stats["wait_result_s"] += time.perf_counter() - t0
        if names != batch[self.image_key]:
            raise RuntimeError(f"the prefetched file names of batch {i} differ from what Dataset.map decoded; set MQ_EMBED_PIPELINE=0")
        self.stats["batches"] = i + 1
        self.stats.setdefault("returned_at", []).append(time.perf_counter())
        output = [None] * rows
        j = 0
        if sum(counts):   # (a batch without a single face: every entry None, like the reference's early return)
            for at, n_faces in zip(images, counts):
                output[at] = found[j: j + n_faces]
                j += n_faces
        batch["face_embedding"] = output
        return batch

    def close(self):
        self.stats["wait_prepared_s"] = self.look.wait_prepared_s
        self.look.close()
        self.decode.close()


def face_pipeline_or_none(dataset, map_kwargs, model=None, max_n_faces=1, image_key="image", decode_pool=None, **other):
    """The pipelined face job, or None when it does not apply (then ``compute_face_embedding`` is mapped as the reference does)."""
    if not pipeline_enabled() or other or model is None or decode_pool is None or not torch.cuda.is_available():
        return None
    if not _plain_dataset(dataset, map_kwargs) or _arrow_strings(dataset, image_key) is None or "face_landmarks" not in dataset.column_names:
        return None
    if _map_batch_size(map_kwargs) is None:
        return None
    first = next(iter(model.parameters()), None) if hasattr(model, "parameters") else None
    if first is None and hasattr(model, "buffers"):
        first = next(iter(model.buffers()), None)
    if first is None or not first.is_cuda:
        return None
    return FaceEmbedPipeline(dataset, model, max_n_faces, image_key, _map_batch_size(map_kwargs), decode_pool)

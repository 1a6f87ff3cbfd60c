"""Hidden-state ingest: pre-captured ``.ckpt`` samples -> pinned host -> HBM, double-buffered.

SURVEY.md section 8f rank 1 -- the step *before* the hot path.  At ~60 k tokens/s/GPU the trainer consumes
~2 GB/s of bf16 hidden states per GPU; the reference's ``torch.load`` + Python collate + blocking
``.to(device)`` on the compute stream (specforge/runtime/data_plane/feature_store.py:235-240,
feature_dataloader.py:255-295, strategies/base.py:270-289) would serialise 0.54 GB of PCIe per micro-step
(~9 ms at 63 GB/s) in front of every step.  Here a loader thread fills a pinned staging slot while the GPU
works, a dedicated HIP copy stream moves it to a device slot, and the compute stream only waits on an event.

Semantics mirrored from the reference (bit-for-bit on the produced tensors, see tests/test_ingest.py):
* file format: ``torch.save`` dict with ``input_ids [S]``, ``loss_mask [S]``, ``hidden_state [1,S,Ht]``,
  ``aux_hidden_state [1,S,3Ht]`` (scripts/prepare_hidden_states.py:446-480, tests/test_runtime/_fixtures.py:131-149)
* sample normalisation (algorithms/eagle3/data.py:10-27): ``hidden_state <- aux_hidden_state[:max_len]``,
  ``target <- hidden_state[:max_len]``, ``loss_mask[-1] = 0``, ``attention_mask = 1``
* collation (data/utils.py:106-196, sp_degree 1): right-pad with zeros to the longest sample of the batch and
  no further -- ``ploss_k`` is a mean over ALL B*S rows including masked ones (core/loss.py:20,201) and
  ``metric_loss_denoms`` is B*S (eagle3/model.py:186-190), so any extra padding would rescale every loss.
  (``pad_multiple`` > 1 exists for experiments only and is NOT the reference's semantics.)
  tests/golden/ingest_collate.pt pins the batches to the reference's own normaliser + ``DataCollatorWithPadding``.
* sharding (launch.py:174-239): ``distributed_sampler_indices`` (== torch DistributedSampler), ``drop_last``
  batches (trainer.py:151)
"""
from __future__ import annotations

import os
import pickle
import queue
import struct
import threading
import zipfile
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence

import torch

from .eagle3 import TrainBatch, loss_mask_suffix_counts
from .training import distributed_sampler_indices


def normalize_offline_sample(raw: Dict[str, torch.Tensor], max_len: int) -> Dict[str, torch.Tensor]:
    """the reference's sample normalisation, restated (specforge/algorithms/eagle3/data.py:10-27: the statements are forced
    by the file format; tests/golden/ingest_collate.pt pins the result to the reference's own function).  Only the fallback
    path of the loader uses it -- the fast path reads the same bytes straight into the staging slot (``_read_direct``)."""
    hidden_state = raw["aux_hidden_state"].squeeze(0)[:max_len].unsqueeze(0)
    target = raw["hidden_state"].squeeze(0)[:max_len].unsqueeze(0)
    input_ids = raw["input_ids"][:max_len].unsqueeze(0)
    loss_mask = raw["loss_mask"][:max_len].clone().unsqueeze(0)
    if loss_mask.numel() > 0:
        loss_mask[0, -1] = 0
    return dict(attention_mask=torch.ones_like(loss_mask, dtype=torch.long), loss_mask=loss_mask, target=target,
                hidden_state=hidden_state, input_ids=input_ids)


# ---- direct reader ---------------------------------------------------------------------------------------------------
# ``torch.save`` writes an uncompressed zip: ``<name>/data.pkl`` (a pickle whose tensors are persistent-id references) and one
# 64-byte-aligned record ``<name>/data/<key>`` per storage.  For the ingest that is all that is needed: the byte range of every
# tensor.  ``torch.load(mmap=True)`` + normalise + ``copy_`` went through three Python-level tensor passes per sample and an
# OpenMP team per copy; eight loader processes on one host (one per rank) reached 7.5 GB/s together where the ranks need 20
# (tools/ingest_procs.py, profiles/old/r3_ingest_procs.json).  Here the file's bytes go from the page cache into the pinned staging
# slot with ONE ``preadv`` per tensor (the GIL is released for its duration, so a small thread pool reads the samples of a
# batch concurrently), and nothing else touches them.
_STORAGE_DTYPES = {"BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "FloatStorage": torch.float32,
                   "HalfStorage": torch.float16, "IntStorage": torch.int32, "BoolStorage": torch.bool,
                   "ByteStorage": torch.uint8, "DoubleStorage": torch.float64, "ShortStorage": torch.int16,
                   "CharStorage": torch.int8}


class _TensorRef:
    __slots__ = ("key", "dtype", "offset", "shape", "stride")

    def __init__(self, key, dtype, offset, shape, stride):
        self.key, self.dtype, self.offset, self.shape, self.stride = key, dtype, offset, tuple(shape), tuple(stride)

    def contiguous(self) -> bool:
        exp, acc = [], 1
        for n in reversed(self.shape):
            exp.append(acc)
            acc *= n
        return all(n == 1 or st == e for n, st, e in zip(self.shape, self.stride, reversed(exp)))


class _MetaUnpickler(pickle.Unpickler):
    """reads data.pkl without touching a storage: tensors come back as _TensorRef"""

    def find_class(self, module, name):
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return lambda storage, offset, size, stride, *a: _TensorRef(storage[0], storage[1], offset, size, stride)
        if module == "torch" and name in _STORAGE_DTYPES:
            return _STORAGE_DTYPES[name]
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        raise pickle.UnpicklingError(f"unsupported global {module}.{name}")

    def persistent_load(self, pid):
        if pid[0] != "storage":
            raise pickle.UnpicklingError("unsupported persistent id")
        return (pid[2], pid[1])          # (record key, dtype)


def _sample_layout(path: str) -> Optional[Dict[str, tuple]]:
    """{tensor name: (file offset of its first byte, dtype, shape)} or None when the file is not a plain torch zip"""
    try:
        with zipfile.ZipFile(path) as zf:
            names = zf.namelist()
            pkl = next(n for n in names if n.endswith("/data.pkl") or n == "data.pkl")
            prefix = pkl[: -len("data.pkl")]
            obj = _MetaUnpickler(zf.open(pkl)).load()
            if not isinstance(obj, dict):
                return None
            out = {}
            with open(path, "rb") as f:
                for name, ref in obj.items():
                    if not isinstance(ref, _TensorRef) or not ref.contiguous():
                        return None
                    zi = zf.getinfo(f"{prefix}data/{ref.key}")
                    if zi.compress_type != zipfile.ZIP_STORED:
                        return None
                    f.seek(zi.header_offset)
                    hdr = f.read(30)
                    fn_len, extra_len = struct.unpack("<HH", hdr[26:30])
                    data0 = zi.header_offset + 30 + fn_len + extra_len
                    out[name] = (data0 + ref.offset * torch.empty((), dtype=ref.dtype).element_size(), ref.dtype, ref.shape)
            return out
    except Exception:   # anything unusual about the file (subclassed tensors, odd persistent ids, compressed / encrypted
        return None     # members, a truncated zip ...) means "not the plain layout": the generic torch.load path reads it


def _bytes_view(t: torch.Tensor):
    """writable byte view of a contiguous CPU tensor (numpy has no bfloat16: go through uint8)"""
    return memoryview(t.view(torch.uint8).numpy()).cast("B")


class _Slot:
    """one staging slot: pinned host tensors + device tensors sized for the largest possible batch"""

    def dview(self, k, L):
        h = self.h[k]
        shape = (h.shape[0], L) + tuple(h.shape[2:])
        n = 1
        for x in shape:
            n *= x
        return self.d[k][:n].view(shape)

    def __init__(self, B, L, Ht, device, pin):
        def host(*shape, dtype):
            t = torch.zeros(*shape, dtype=dtype)
            return t.pin_memory() if pin else t

        self.h = dict(input_ids=host(B, L, dtype=torch.int64), attention_mask=host(B, L, dtype=torch.int64),
                      loss_mask=host(B, L, dtype=torch.int64), hidden_state=host(B, L, 3 * Ht, dtype=torch.bfloat16),
                      target=host(B, L, Ht, dtype=torch.bfloat16))
        # device side: flat buffers, viewed as a CONTIGUOUS [B, L, ...] tensor of the batch's actual padded length
        self.d = {k: torch.zeros(v.numel(), dtype=v.dtype, device=device) for k, v in self.h.items()} if device.type == "cuda" else None
        self.copied = None      # event on the copy stream: H2D of this slot finished
        self.released = None    # event on the compute stream: the consumer moved past this slot


_LAYOUT_CACHE_MAX = 1 << 16     # entries; a multi-million-file dataset must not grow the cache without bound


class HiddenStateIngest:
    def __init__(self, files: Sequence[str], *, batch_size: int, max_len: int, target_hidden_size: Optional[int] = None, device="cuda",
                 dp_rank: int = 0, dp_size: int = 1, seed: int = 0, shuffle: bool = True, pad_multiple: int = 1,
                 slots: int = 2, reader_threads: int = 4, direct: bool = True):
        """``target_hidden_size=None``: taken from the first file read (the reference's loader does not know it either)."""
        self.files = list(files)
        self.B, self.max_len, self.Ht = batch_size, max_len, target_hidden_size
        self.device = torch.device(device)
        self.dp_rank, self.dp_size, self.seed, self.shuffle = dp_rank, dp_size, seed, shuffle
        self.pad_multiple = pad_multiple
        cuda = self.device.type == "cuda"
        self._Lcap = (max_len + pad_multiple - 1) // pad_multiple * pad_multiple
        self._nslots = max(2, slots)
        self._slots: List[_Slot] = []
        self._copy_stream = torch.cuda.Stream(device=self.device) if cuda else None
        self._direct = bool(direct)
        self._layouts: "OrderedDict[str, Optional[Dict[str, tuple]]]" = OrderedDict()
        self._layout_lock = threading.Lock()
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(reader_threads)), thread_name_prefix="sf-ingest")
        if target_hidden_size is not None:
            self._ensure_slots(None)

    def _ensure_slots(self, first_path: Optional[str]) -> None:
        if self._slots:
            return
        if self.Ht is None:
            lay = self._layout(first_path) if self._direct else None
            if lay is not None and "hidden_state" in lay:
                self.Ht = int(lay["hidden_state"][2][-1])
            else:
                self.Ht = int(torch.load(first_path, mmap=True, weights_only=True)["hidden_state"].shape[-1])
        cuda = self.device.type == "cuda"
        self._slots = [_Slot(self.B, self._Lcap, self.Ht, self.device, pin=cuda) for _ in range(self._nslots)]

    def batches_per_epoch(self) -> int:
        n = len(distributed_sampler_indices(len(self.files), dp_rank=self.dp_rank, dp_size=self.dp_size, seed=self.seed,
                                            epoch=0, shuffle=self.shuffle))
        return n // self.B

    # what each staged tensor is in the file (algorithms/eagle3/data.py:10-27): name in the slot <- name in the file
    _SOURCE = (("hidden_state", "aux_hidden_state"), ("target", "hidden_state"), ("input_ids", "input_ids"), ("loss_mask", "loss_mask"))

    def _layout(self, path: str):
        with self._layout_lock:
            if path in self._layouts:
                return self._layouts[path]
        lay = _sample_layout(path)
        with self._layout_lock:
            self._layouts[path] = lay
            while len(self._layouts) > _LAYOUT_CACHE_MAX:
                self._layouts.popitem(last=False)
        return lay

    def _read_direct(self, slot: _Slot, b: int, path: str) -> int:
        """sample -> row b of the slot, straight from the file; returns its (truncated) length, or -1 if the file needs the
        generic path (not a plain torch zip, other dtypes, unexpected shapes, a read that came back short)"""
        lay = self._layout(path)
        if lay is None or any(src not in lay for _, src in self._SOURCE):
            return -1
        n = None
        plan = []
        for dst, src in self._SOURCE:
            off, dtype, shape = lay[src]
            buf = slot.h[dst]
            hidden = buf.dim() == 3      # hidden states are [1, S, features] in the file, ids / mask are [S]
            if dtype != buf.dtype or (hidden and not (len(shape) == 3 and shape[0] == 1 and shape[2] == buf.shape[-1])) \
                    or (not hidden and len(shape) != 1):
                return -1
            rows, feat = (shape[1], shape[2]) if hidden else (shape[0], 1)
            if n is not None and min(rows, self.max_len) != n:
                return -1
            n = min(rows, self.max_len)
            plan.append((buf, off, n * feat * buf.element_size()))
        fd = os.open(path, os.O_RDONLY)
        try:
            for buf, off, nbytes in plan:
                if not nbytes:
                    continue
                view, done = _bytes_view(buf[b, :n]), 0
                while done < nbytes:       # preadv may legally return fewer bytes (network / FUSE filesystems, signals)
                    got = os.preadv(fd, [view[done:]], off + done)
                    if got <= 0:
                        return -1          # truncated file: let torch.load produce the error (or read it)
                    done += got
        finally:
            os.close(fd)
        if n > 0:
            slot.h["loss_mask"][b, n - 1] = 0
        slot.h["attention_mask"][b, :n] = 1
        return n

    def _read_generic(self, slot: _Slot, b: int, path: str) -> int:
        s = normalize_offline_sample(torch.load(path, mmap=True, weights_only=True), self.max_len)
        n = s["input_ids"].shape[1]
        for k, buf in slot.h.items():
            buf[b, :n].copy_(s[k][0])
        return n

    def _read_one(self, slot: _Slot, b: int, path: str) -> int:
        n = self._read_direct(slot, b, path) if self._direct else -1
        return n if n >= 0 else self._read_generic(slot, b, path)

    def _fill(self, slot: _Slot, paths: Sequence[str]) -> int:
        lens = list(self._pool.map(lambda bp: self._read_one(slot, bp[0], bp[1]), enumerate(paths)))
        L = max(lens)
        L = (L + self.pad_multiple - 1) // self.pad_multiple * self.pad_multiple
        for b, n in enumerate(lens):          # right-pad with zeros to the longest sample: only the tails are written
            if n < L:
                for buf in slot.h.values():
                    buf[b, n:L].zero_()
        return L

    def collate_paths(self, paths: Sequence[str]) -> Dict[str, torch.Tensor]:
        """host-side batch of the given sample files (normalise + right-pad), as fresh CPU tensors -- what ``stream`` stages"""
        self._ensure_slots(paths[0])
        slot = _Slot(len(paths), self._Lcap, self.Ht, torch.device("cpu"), pin=False)
        L = self._fill(slot, list(paths))
        return {k: v[:, :L].clone() for k, v in slot.h.items()}

    def collate_indices(self, idxs: Sequence[int]) -> Dict[str, torch.Tensor]:
        return self.collate_paths([self.files[i] for i in idxs])

    def epoch(self, epoch: int = 0) -> Iterator[TrainBatch]:
        """One epoch of this rank's shard (``DistributedSampler`` order, ``drop_last`` batches) as device-resident batches."""
        idx = distributed_sampler_indices(len(self.files), dp_rank=self.dp_rank, dp_size=self.dp_size, seed=self.seed,
                                          epoch=epoch, shuffle=self.shuffle)
        groups = [idx[i:i + self.B] for i in range(0, len(idx) - self.B + 1, self.B)]  # drop_last
        for batch, g in zip(self.stream([[self.files[i] for i in g] for g in groups]), groups):
            batch.metadata["sample_indices"] = list(g)
            yield batch

    def stream(self, groups: Sequence[Sequence[str]]) -> Iterator[TrainBatch]:
        """Device-resident batches of the given groups of sample files, in order (a group may be shorter than ``batch_size``:
        the reference's eval loader keeps the last partial batch).  A batch's tensors are views of a staging slot: they stay
        valid until the NEXT batch is requested (by then the consumer has enqueued all work that reads them; the slot is
        refilled only after an event on the consumer's stream has passed)."""
        groups = [list(g) for g in groups]
        if not groups:
            return
        # an earlier stream() whose iterator was abandoned: its loader must be gone and the copies it queued must have left the
        # pinned slots before this call refills them (a filled-but-never-consumed slot has no `released` event to wait on)
        old = getattr(self, "_loader_thread", None)
        if old is not None and old.is_alive():
            # a suspended-but-referenced iterator (kept in a variable, held by a traceback) leaves its loader blocked in free.get(), not
            # reading: tell it to stop and wake it; what can still be alive after that is a loader in the middle of a file read
            ctl = getattr(self, "_loader_ctl", None)
            if ctl is not None:
                ctl[0].set()
                ctl[1].put(None)
            old.join(timeout=120.0)
            if old.is_alive():
                raise RuntimeError("HiddenStateIngest.stream: the loader thread of an abandoned iterator is still reading (slow "
                                   "storage?); a new stream would overwrite the staging slots under it")
        self._ensure_slots(groups[0][0])
        if self._copy_stream is not None:
            self._copy_stream.synchronize()
        nslots = len(self._slots)
        free: "queue.Queue[int]" = queue.Queue()
        ready: "queue.Queue" = queue.Queue()
        stop = threading.Event()
        self._loader_ctl = (stop, free)
        for i in range(nslots):
            free.put(i)

        def loader():
            try:
                for g in groups:
                    si = free.get()
                    if si is None or stop.is_set():
                        return
                    slot = self._slots[si]
                    if slot.released is not None:      # the consumer's stream must be past this slot's tensors
                        slot.released.synchronize()
                    L = self._fill(slot, g)
                    counts = loss_mask_suffix_counts(slot.h["loss_mask"][:len(g), :L])   # (host side: the engine's loss-row compaction)
                    if self._copy_stream is not None:
                        with torch.cuda.stream(self._copy_stream):
                            for k in slot.h:
                                slot.dview(k, L)[:len(g)].copy_(slot.h[k][:len(g), :L], non_blocking=True)
                            slot.copied = torch.cuda.Event()
                            slot.copied.record()
                    ready.put((si, L, g, counts))
                ready.put(None)
            except BaseException as e:  # surface loader failures in the consumer
                ready.put(e)

        t = self._loader_thread = threading.Thread(target=loader, daemon=True)
        t.start()
        prev = None
        try:
            while True:
                item = ready.get()
                if prev is not None:  # the consumer asked for the next batch: everything it enqueued on `prev` is ordered before this
                    if self._copy_stream is not None:
                        self._slots[prev].released = torch.cuda.Event()
                        self._slots[prev].released.record()
                    free.put(prev)
                    prev = None
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                si, L, g, counts = item
                slot = self._slots[si]
                nb = len(g)
                if self._copy_stream is not None:
                    torch.cuda.current_stream().wait_event(slot.copied)
                    tensors = {k: slot.dview(k, L)[:nb] for k in slot.d}
                else:
                    tensors = {k: v[:nb, :L].clone() for k, v in slot.h.items()}
                prev = si
                yield TrainBatch(tensors, {"target_repr": "hidden_state", "sample_files": list(g), "loss_mask_suffix_counts": counts})
        finally:
            # the consumer may abandon the iterator mid-epoch (max_steps reached): release the loader thread
            stop.set()
            if prev is not None and self._copy_stream is not None:
                self._slots[prev].released = torch.cuda.Event()
                self._slots[prev].released.record()
            free.put(None)
            t.join(timeout=30.0)       # (still alive after that: the next stream() waits for it, or refuses to start over it)


class PinnedStager:
    """CPU batch -> HBM without a pageable copy on the compute stream.

    What still arrives as CPU tensors (the reference's own ``FeatureDataLoader`` batches -- strategies/base.py:270-289 moves
    them with a blocking pageable ``.to(device)`` on the compute stream -- a ``.ckpt.gz`` dataset, an online ``mem://`` store)
    is copied into one of two pinned slots, sent on a dedicated HIP copy stream, and the compute stream only waits on the
    copy's event.  The pageable form stalls the host until the previous step has drained and then moves 0.54 GB through the
    runtime's bounce buffers in front of the step; here the host-side copy runs while the GPU is still busy with the previous
    step.  A slot is reused only after an event on the consumer's stream has passed (recorded when the next batch is staged)."""

    def __init__(self, device, slots: int = 2):
        self.device = torch.device(device)
        assert self.device.type == "cuda"
        self._stream = torch.cuda.Stream(device=self.device)
        self._slots = [dict(h={}, d={}, released=None) for _ in range(max(2, slots))]
        self._next = 0
        self._prev = None

    @staticmethod
    def _buf(store, key, t, **kw):
        n = t.numel()
        cur = store.get(key)
        if cur is None or cur.dtype != t.dtype or cur.numel() < n:
            cur = store[key] = torch.empty(max(n, 1), dtype=t.dtype, **kw)
        return cur[:n].view(t.shape)

    def stage(self, tensors: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self._prev is not None:    # the consumer came back for another batch: its work on the previous one is enqueued
            ev = torch.cuda.Event()
            ev.record()
            self._slots[self._prev]["released"] = ev
        slot = self._slots[self._next]
        if slot["released"] is not None:
            slot["released"].synchronize()
        out = {}
        with torch.cuda.stream(self._stream):
            for k, t in tensors.items():
                if t is None or t.is_cuda:
                    out[k] = t
                    continue
                t = t.contiguous()
                h = self._buf(slot["h"], k, t, pin_memory=True)
                h.copy_(t)
                d = self._buf(slot["d"], k, t, device=self.device)
                d.copy_(h, non_blocking=True)
                out[k] = d
            done = torch.cuda.Event()
            done.record()
        torch.cuda.current_stream().wait_event(done)
        self._prev, self._next = self._next, (self._next + 1) % len(self._slots)
        return out

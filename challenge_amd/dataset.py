"""A small lazy Dataset with the tf.data surface the reference actually uses
(pipeline.py:143-174, sj_train.py:103-130, pipeline_test.py:72): from_generator,
repeat, shuffle, padded_batch, zip, map, batch, prefetch, take, iteration.
Elements are torch tensors or (nested) tuples of them.  Stages are Python
generators; heavy work happens inside the mapped functions (HIP kernels / torch
device ops), so this is glue, not a hot path."""
from __future__ import annotations

import atexit
import queue
import random
import threading
from typing import Callable, Iterable, Optional

import numpy as np
import torch

AUTOTUNE = -1


def _to_tensor(x):
    if isinstance(x, torch.Tensor):
        return x
    if isinstance(x, (tuple, list)):
        return tuple(_to_tensor(v) for v in x)
    return torch.as_tensor(np.asarray(x))


def _map_structure(fn, x):
    if isinstance(x, (tuple, list)):
        return tuple(_map_structure(fn, v) for v in x)
    return fn(x)


# prefetch workers still running: stopped and joined before the interpreter (and with it the HIP
# runtime) shuts down - a daemon thread caught inside a HIP call at teardown aborts the process
_live_workers: list = []


def _reap_workers(wait: bool = False) -> None:
    for stop, t in list(_live_workers):
        if wait:
            stop.set()
            t.join(timeout=5.0)
        if not t.is_alive():
            try:
                _live_workers.remove((stop, t))
            except ValueError:
                pass


atexit.register(_reap_workers, True)


class Dataset:
    def __init__(self, make_iter: Callable[[], Iterable], infinite: bool = False):
        self._make_iter = make_iter
        self._infinite = infinite

    # ---- sources ---------------------------------------------------------
    @staticmethod
    def from_generator(generator: Callable[[], Iterable], output_types=None, output_shapes=None) -> "Dataset":
        def it():
            for item in generator():
                yield _to_tensor(item)
        return Dataset(it)

    @staticmethod
    def from_tensor_slices(tensors) -> "Dataset":
        def it():
            first = tensors[0] if isinstance(tensors, (tuple, list)) else tensors
            for i in range(len(first)):
                yield _map_structure(lambda t: _to_tensor(t[i]), tensors) if isinstance(tensors, (tuple, list)) \
                    else _to_tensor(tensors[i])
        return Dataset(it)

    @staticmethod
    def zip(datasets) -> "Dataset":
        datasets = tuple(datasets)
        return Dataset(lambda: zip(*[iter(d) for d in datasets]), all(d._infinite for d in datasets))

    # ---- transformations ---------------------------------------------------
    def repeat(self, count: Optional[int] = None) -> "Dataset":
        def it():
            n = 0
            while count is None or n < count:
                empty = True
                for item in self._make_iter():
                    empty = False
                    yield item
                if empty:
                    return
                n += 1
        return Dataset(it, count is None)

    def shuffle(self, buffer_size: int, seed: Optional[int] = None) -> "Dataset":
        def it():
            rng = random.Random(seed)
            buf = []
            for item in self._make_iter():
                buf.append(item)
                if len(buf) >= buffer_size:
                    yield buf.pop(rng.randrange(len(buf)))
            while buf:
                yield buf.pop(rng.randrange(len(buf)))
        return Dataset(it, self._infinite)

    def map(self, fn: Callable, num_parallel_calls=None) -> "Dataset":
        def it():
            for item in self._make_iter():
                yield fn(*item) if isinstance(item, tuple) else fn(item)
        return Dataset(it, self._infinite)

    def batch(self, batch_size: int, drop_remainder: bool = False) -> "Dataset":
        def it():
            buf = []
            for item in self._make_iter():
                buf.append(item)
                if len(buf) == batch_size:
                    yield _stack(buf)
                    buf = []
            if buf and not drop_remainder:
                yield _stack(buf)
        return Dataset(it, self._infinite)

    def padded_batch(self, batch_size: int, padded_shapes=None, drop_remainder: bool = False) -> "Dataset":
        """Batch with zero padding of ragged axes to the longest item in the batch
        (padded_shapes entries that are None / -1), as pipeline.py:155-166 uses it."""
        def it():
            buf = []
            for item in self._make_iter():
                buf.append(item)
                if len(buf) == batch_size:
                    yield _stack(buf, pad=True)
                    buf = []
            if buf and not drop_remainder:
                yield _stack(buf, pad=True)
        return Dataset(it, self._infinite)

    def take(self, count: int) -> "Dataset":
        def it():
            if count <= 0:
                return
            for i, item in enumerate(self._make_iter()):
                yield item
                if i + 1 >= count:
                    return
        return Dataset(it)

    def prefetch(self, buffer_size: int = AUTOTUNE) -> "Dataset":
        """Run the upstream stages in a background thread, `buffer_size` items ahead
        (AUTOTUNE -> 2).  GPU work is enqueued asynchronously on the producer's stream
        of the current device; consumers receive tensors whose kernels are ordered on
        that same (default) stream."""
        depth = 2 if buffer_size in (AUTOTUNE, None) or buffer_size < 1 else int(buffer_size)

        def it():
            q: "queue.Queue" = queue.Queue(maxsize=depth)
            stop = threading.Event()
            sentinel = object()
            dev = torch.cuda.current_device() if torch.cuda.is_available() else None

            def worker():
                try:
                    if dev is not None:
                        torch.cuda.set_device(dev)
                    for item in self._make_iter():
                        while not stop.is_set():
                            try:
                                q.put(item, timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                    q.put(sentinel)
                except BaseException as e:  # surface errors in the consumer
                    q.put(e)

            t = threading.Thread(target=worker, daemon=True)
            _live_workers.append((stop, t))
            t.start()
            try:
                while True:
                    item = q.get()
                    if item is sentinel:
                        return
                    if isinstance(item, BaseException):
                        raise item
                    yield item
            finally:
                stop.set()
                _reap_workers()
        return Dataset(it, self._infinite)

    def __iter__(self):
        return iter(self._make_iter())


def _stack(items, pad: bool = False):
    first = items[0]
    if isinstance(first, tuple):
        return tuple(_stack([it[i] for it in items], pad) for i in range(len(first)))
    if not pad or all(tuple(t.shape) == tuple(first.shape) for t in items):
        return torch.stack(items)
    shape = [max(t.shape[d] for t in items) for d in range(first.dim())]
    out = first.new_zeros((len(items), *shape))
    for i, t in enumerate(items):
        out[(i, *[slice(0, s) for s in t.shape])] = t
    return out

"""Runner / Sampler / task split with the reference's interfaces (callers of the encode seam).

  * `Sampler`        clip_retrieval/clip_inference/runner.py:7-14 -- every count-th key/shard
  * `Runner`         runner.py:17-64 -- per partition: build reader, writer, mapper, logger; for each batch
                     read -> map -> write -> log the same seven stat keys (runner.py:50-61)
  * `get_task_list`  clip_retrieval/clip_inference/slurm_worker.py:16-37 -- contiguous task ranges per rank,
                     remainder to the lowest ranks; used here to deal output partitions to the 8 GPUs
  * `LoggerWriter`   clip_retrieval/clip_inference/logger.py:13-62 -- summed stats as JSON files per partition
With the reference's own mapper the loop is the reference's serial loop; with this package's mapper it is pipelined one
batch deep on the asynchronous tickets of the C ABI (upload of batch n+1 under the kernels of batch n).
"""

import sys
import time


class Sampler:
    """Keep element i of a list when i % output_partition_count == output_partition_id."""

    def __init__(self, output_partition_id, output_partition_count):
        self.output_partition_id = output_partition_id
        self.output_partition_count = output_partition_count

    def __call__(self, l):
        return list(l[self.output_partition_id::self.output_partition_count])


def get_task_list(num_tasks, world_size, global_rank, local_rank=-1):  # pylint: disable=unused-argument
    """Tasks of `global_rank`: floor(num_tasks / world_size) each, one extra for the first
    (num_tasks % world_size) ranks, contiguous."""
    base, extra = divmod(num_tasks, world_size)
    start = global_rank * base + min(global_rank, extra)
    return list(range(start, start + base + (1 if global_rank < extra else 0)))


class _Prefetcher:
    """Iterates `it` on a background thread, `depth` items ahead (the reader's decode pool keeps working while the GPU
    runs the previous batch).  Exceptions of the producer are re-raised in the consumer."""

    _END = object()

    def __init__(self, it, depth=2):
        import queue  # pylint: disable=import-outside-toplevel
        import threading  # pylint: disable=import-outside-toplevel

        self._q = queue.Queue(maxsize=depth)
        self._done = False
        self._stop = threading.Event()
        # torch's current device is per thread: the producer (which collates into pinned memory) must page-lock against the
        # GPU its creator is bound to, not GPU 0 (one process per GPU: worker.gpu_worker)
        self._device = None
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            self._device = torch.cuda.current_device()
        self._t = threading.Thread(target=self._run, args=(iter(it),), daemon=True)
        self._t.start()

    def _put(self, item):
        import queue  # pylint: disable=import-outside-toplevel

        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _run(self, it):
        try:
            if self._device is not None:
                sys.modules["torch"].cuda.set_device(self._device)
            for item in it:
                if not self._put(("ok", item)):
                    return
            self._put(("end", None))
        except BaseException as e:  # pylint: disable=broad-except
            self._put(("err", e))

    def __iter__(self):
        return self

    def __next__(self):
        if self._done:
            raise StopIteration
        kind, item = self._q.get()
        if kind == "ok":
            return item
        self._done = True
        if kind == "err":
            self._stop.set()
            raise item
        raise StopIteration

    def close(self):
        self._stop.set()


def _batch_rows(batch):
    if "image_tensor" in batch:
        return batch["image_tensor"].shape[0]
    if "image_raw" in batch:  # decoded sources, resized on the GPU (reader.decode_rgb_u8)
        return batch["image_raw"]["hw"].shape[0]
    return batch["text_tokens"].shape[0]


class Runner:
    """Runs one output partition end to end.

    Same builders, same call and the same seven stat keys per batch as the reference (runner.py:17-64).  The reference
    loop is strictly serial (read; map; write).  When the mapper offers `submit(batch)` / `collect(handle)` (ours: the
    asynchronous tickets of the C ABI, clipx_encode_*_async / clipx_wait) the loop is software-pipelined one batch deep:
    batch n+1 is read (background thread) and SUBMITTED -- its upload overlaps batch n's kernels -- before batch n is
    collected and written.  Output order, files and stats keys are unchanged; a mapper without submit/collect (the
    reference's own ClipMapper, the tests' fakes) runs the reference's serial loop.
    """

    def __init__(self, reader_builder, mapper_builder, writer_builder, logger_builder, output_partition_count, prefetch=2):
        self.reader_builder = reader_builder
        self.mapper_builder = mapper_builder
        self.writer_builder = writer_builder
        self.logger_builder = logger_builder
        self.output_partition_count = output_partition_count
        self.prefetch = prefetch

    def __call__(self, i):
        reader = self.reader_builder(Sampler(i, self.output_partition_count))
        writer = self.writer_builder(i)
        mapper = self.mapper_builder()
        logger = self.logger_builder(i)
        logger.start()
        pipelined = hasattr(mapper, "submit") and hasattr(mapper, "collect")
        source = _Prefetcher(reader, self.prefetch) if (pipelined and self.prefetch > 0) else None
        batches = iter(source if source is not None else reader)
        pending = None  # (handle, batch sample count, wall0, read_duration, submit_duration)
        handle = inflight = None  # inflight: the handle being collected (its tickets may be half waited for when collect raises)
        exhausted = False
        last_end = None
        try:
            while True:
                wall0 = time.time()
                t0 = time.perf_counter()
                batch = next(batches, None) if not exhausted else None
                exhausted = batch is None
                t1 = time.perf_counter()
                if batch is None and pending is None:
                    break
                handle = None
                if batch is not None and pipelined:
                    handle = mapper.submit(batch)
                t2 = time.perf_counter()
                if pipelined:
                    done, todo = pending, None
                    if batch is not None:
                        todo = (handle, _batch_rows(batch), wall0, t1 - t0, t2 - t1)
                    pending = todo
                    if done is None:
                        continue
                    h, count, w0, read_d, sub_d = done
                    t3 = time.perf_counter()
                    inflight = h
                    embeddings = mapper.collect(h)
                    inflight = None
                    t4 = time.perf_counter()
                    writer(embeddings)
                    t5 = time.perf_counter()
                    wall1 = time.time()
                    # batches overlap in the pipelined loop: a batch is charged the wall time since the previous batch
                    # finished, so that the summed total_duration is the partition's wall time like in the serial loop
                    start = w0 if last_end is None else max(w0, last_end)
                    last_end = wall1
                    logger({"start_time": start, "end_time": wall1, "read_duration": read_d,
                            "inference_duration": sub_d + (t4 - t3), "write_duration": t5 - t4,
                            "total_duration": wall1 - start, "sample_count": count})
                else:
                    if batch is None:
                        break
                    embeddings = mapper(batch)
                    t3 = time.perf_counter()
                    writer(embeddings)
                    t4 = time.perf_counter()
                    wall1 = time.time()
                    logger({"start_time": wall0, "end_time": wall1, "read_duration": t1 - t0,
                            "inference_duration": t3 - t1, "write_duration": t4 - t3,
                            "total_duration": wall1 - wall0, "sample_count": _batch_rows(batch)})
        except BaseException:
            # A submitted ticket owns one of the encoder's staging slots (and keeps its pinned input referenced) until it is
            # collected; the encoder outlives this partition (cached per model and device), so a ticket leaked here would
            # starve every later partition of this process.  Collect and drop whatever is still in flight.
            if pipelined:
                drop = getattr(mapper, "discard", None) or mapper.collect
                seen = []
                for h in (inflight, pending[0] if pending else None, handle):
                    if h is not None and not any(h is s for s in seen):
                        seen.append(h)
                        try:
                            drop(h)
                        except Exception:  # pylint: disable=broad-except
                            pass
            raise
        finally:
            if source is not None:
                source.close()
        logger.end()
        writer.flush()


class NullLogger:
    """Logger with the LoggerWriter call surface (logger.py:20-62) that keeps the stats in memory."""

    def __init__(self, partition_id=0):
        self.partition_id = partition_id
        self.records = []

    def start(self):
        self.records = []

    def __call__(self, stats):
        self.records.append(stats)

    def end(self):
        pass


class LoggerWriter:
    """Per-partition stats file of the reference (logger.py:13-62): the seven stat keys are SUMMED over the batches and
    written as JSON to `<stats_folder>/wip_<partition>.json` while the partition runs (at most every 5 s) and to
    `<stats_folder>/<partition>.json` at the end (the wip file is removed), through fsspec like the reference.  The
    reference feeds a spawned updater process through a queue; here the caller's thread does the (tiny) work."""

    def __init__(self, partition_id, stats_folder):
        self.partition_id = partition_id
        self.stats_folder = stats_folder
        self._stats = None
        self._last = None

    def start(self):
        import collections  # pylint: disable=import-outside-toplevel

        import fsspec  # pylint: disable=import-outside-toplevel

        self._stats = collections.defaultdict(lambda: 0)
        self._fs, self._path = fsspec.core.url_to_fs(self.stats_folder)
        self._last = None

    def _write(self, wip):
        import json  # pylint: disable=import-outside-toplevel

        self._fs.makedirs(self._path, exist_ok=True)
        if not wip and self._fs.exists(self._path + f"/wip_{self.partition_id}.json"):
            self._fs.rm(self._path + f"/wip_{self.partition_id}.json")
        with self._fs.open(self._path + f"/{'wip_' if wip else ''}{self.partition_id}.json", "w") as f:
            f.write(json.dumps(self._stats))

    def __call__(self, stats):
        for k in stats:
            self._stats[k] += stats[k]
        if self._last is None or time.time() - self._last > 5:
            self._write(True)
            self._last = time.time()

    def end(self):
        self._write(False)

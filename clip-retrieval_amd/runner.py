"""Runner / Sampler / task split with the reference's interfaces (callers of the encode seam).

  * `Sampler`        clip_retrieval/clip_inference/runner.py:7-14 -- every count-th key/shard
  * `Runner`         runner.py:17-64 -- per partition: build reader, writer, mapper, logger; for each batch
                     read -> map -> write -> log the same seven stat keys (runner.py:50-61)
  * `get_task_list`  clip_retrieval/clip_inference/slurm_worker.py:16-37 -- contiguous task ranges per rank,
                     remainder to the lowest ranks; used here to deal output partitions to the 8 GPUs
The loop stays serial at this level, exactly like the reference; H2D / compute / D2H overlap happens below
the mapper (csrc/clipx_api.hip host_pipeline).
"""

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


class Runner:
    """Runs one output partition end to end."""

    def __init__(self, reader_builder, mapper_builder, writer_builder, logger_builder, output_partition_count):
        self.reader_builder = reader_builder
        self.mapper_builder = mapper_builder
        self.writer_builder = writer_builder
        self.logger_builder = logger_builder
        self.output_partition_count = output_partition_count

    def __call__(self, i):
        reader = self.reader_builder(Sampler(i, self.output_partition_count))
        writer = self.writer_builder(i)
        mapper = self.mapper_builder()
        logger = self.logger_builder(i)
        logger.start()
        batches = iter(reader)
        while True:
            wall0 = time.time()
            t0 = time.perf_counter()
            batch = next(batches, None)
            if batch is None:
                break
            t1 = time.perf_counter()
            embeddings = mapper(batch)
            t2 = time.perf_counter()
            writer(embeddings)
            t3 = time.perf_counter()
            wall1 = time.time()
            key = "image_tensor" if "image_tensor" in batch else "text_tokens"
            logger({
                "start_time": wall0,
                "end_time": wall1,
                "read_duration": t1 - t0,
                "inference_duration": t2 - t1,
                "write_duration": t3 - t2,
                "total_duration": wall1 - wall0,
                "sample_count": batch[key].shape[0],
            })
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

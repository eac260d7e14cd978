"""Per-GPU encode worker (encode half: "replicas only").

Same call and the same builders as the reference worker (clip_retrieval/clip_inference/worker.py:22-127): a worker is
handed a list of output-partition ids (`tasks`) and runs `Runner(task)` for each -- reader, mapper, writer and logger
are built per partition exactly like there.  `gpu_worker()` is the counterpart of slurm_worker.py:40-61 for a
single-node launch with one process per GPU (torch.distributed.run / any launcher that sets RANK, LOCAL_RANK,
WORLD_SIZE): the partitions are dealt with `get_task_list`, the process binds to GPU `LOCAL_RANK`; there is no
collective anywhere on this path and no rank ever waits for another.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m clip_retrieval_amd.worker \
        --input_dataset '/data/{000..999}.tar' --input_format webdataset --output_folder /out --output_partition_count 1000
"""

import functools
import json
import os
import re

from .encoder import load_clip
from .mapper import ClipMapper
from .reader import DecodeRgbU8, FilesReader, WebdatasetReader, clip_preprocess_u8
from .runner import LoggerWriter, Runner, get_task_list
from .writer import NumpyWriter


def braceexpand(pattern):
    """`{000..127}` numeric ranges and `{a,b}` lists of a shard pattern (the subset of the braceexpand package the
    reference's datasets use, worker.py:47-48); nested and multiple groups expand as a cartesian product."""
    m = re.search(r"\{([^{}]*)\}", pattern)
    if not m:
        return [pattern]
    body, out = m.group(1), []
    rng = re.fullmatch(r"(-?\d+)\.\.(-?\d+)", body)
    if rng:
        a, b = rng.group(1), rng.group(2)
        width = max(len(a), len(b)) if (a.startswith("0") and len(a) > 1) or (b.startswith("0") and len(b) > 1) else 0
        step = 1 if int(a) <= int(b) else -1
        alts = [str(v).zfill(width) for v in range(int(a), int(b) + step, step)]
    else:
        alts = body.split(",")
    for alt in alts:
        out.extend(braceexpand(pattern[: m.start()] + alt + pattern[m.end():]))
    return out


def worker(
    tasks,
    input_dataset,
    output_folder,
    output_partition_count,
    input_format="files",
    cache_path=None,
    batch_size=256,
    num_prepro_workers=4,
    enable_text=True,
    enable_image=True,
    enable_metadata=False,
    wds_image_key="jpg",
    wds_caption_key="txt",
    clip_model="ViT-B/32",
    mclip_model="sentence-transformers/clip-ViT-B-32-multilingual-v1",
    use_mclip=False,
    use_jit=True,
    clip_cache_path=None,
    device=0,
    gpu_normalise=True,
    gpu_resize=False,
):
    """Start a worker.  gpu_normalise (not in the reference): the readers resize / crop on the host and hand uint8 pixels to
    the mapper, which normalises on the GPU; False reproduces the reference's float32 `image_tensor` batches.  gpu_resize (not in
    the reference): the readers only DECODE; resize + centre crop run on the GPU too, bit-identical to Pillow (csrc/preprocess.hip)."""
    print("Starting the worker", flush=True)
    if input_format == "webdataset" and not isinstance(input_dataset, list):
        input_dataset = braceexpand(input_dataset)
    print(f"dataset is {len(input_dataset)}", flush=True)

    def reader_builder(sampler):
        model, preprocess, tokenizer = load_clip(clip_model=clip_model, use_jit=use_jit, warmup_batch_size=0,
                                                 clip_cache_path=clip_cache_path, device=device)
        if gpu_resize:  # decode only; geometry and normalisation on the GPU (non-RGB modes and oversized sources: host crop)
            preprocess = DecodeRgbU8(model._enc.arch.image_size)  # pylint: disable=protected-access
        elif gpu_normalise:
            size = model._enc.arch.image_size  # pylint: disable=protected-access
            preprocess = functools.partial(clip_preprocess_u8, size=size)  # picklable: travels to the decode processes
        if input_format == "files":
            return FilesReader(sampler, preprocess, tokenizer, input_dataset, batch_size, num_prepro_workers,
                               enable_text=enable_text, enable_image=enable_image, enable_metadata=enable_metadata)
        if input_format == "webdataset":
            return WebdatasetReader(sampler, preprocess, tokenizer, input_dataset, batch_size, num_prepro_workers,
                                    enable_text=enable_text, enable_image=enable_image, enable_metadata=enable_metadata,
                                    wds_image_key=wds_image_key, wds_caption_key=wds_caption_key, cache_path=cache_path)
        raise ValueError(f"Unknown input_format: {input_format}")

    def mapper_builder():
        return ClipMapper(enable_image=enable_image, enable_text=enable_text, enable_metadata=enable_metadata,
                          use_mclip=use_mclip, clip_model=clip_model, use_jit=use_jit, mclip_model=mclip_model,
                          clip_cache_path=clip_cache_path, warmup_batch_size=batch_size, device=device)

    def writer_builder(i):
        return NumpyWriter(partition_id=i, output_folder=output_folder, enable_text=enable_text, enable_image=enable_image,
                           enable_metadata=enable_metadata, output_partition_count=output_partition_count)

    def logger_builder(i):
        return LoggerWriter(partition_id=i, stats_folder=output_folder + "/stats")

    runner = Runner(reader_builder=reader_builder, mapper_builder=mapper_builder, writer_builder=writer_builder,
                    logger_builder=logger_builder, output_partition_count=output_partition_count)
    for task in tasks:
        print(f"Starting work on task {task}", flush=True)
        runner(task)


def gpu_worker(**worker_args):
    """One process per GPU: this rank's share of the output partitions on GPU LOCAL_RANK (slurm_worker.py:40-61 with the
    launcher's RANK / LOCAL_RANK / WORLD_SIZE in place of SLURM's variables).  WORKER_ARGS_PATH may hold the json of the
    worker arguments like the reference's slurm distributor writes it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", os.environ.get("SLURM_PROCID", "0")))
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("SLURM_LOCALID", "0")))
    if os.environ.get("WORKER_ARGS_PATH"):
        # explicit keyword arguments win over the file; options the caller left unset (None) take the file's value
        with open(os.environ["WORKER_ARGS_PATH"], "r", encoding="utf-8") as f:
            worker_args = {**json.load(f), **{k: v for k, v in worker_args.items() if v is not None}}
    worker_args = {k: v for k, v in worker_args.items() if v is not None}
    missing = [k for k in ("input_dataset", "output_folder", "output_partition_count") if k not in worker_args]
    if missing:
        raise ValueError(f"gpu_worker: {missing} given neither as arguments nor in WORKER_ARGS_PATH")
    num_tasks = int(os.environ.get("NUM_TASKS", worker_args["output_partition_count"]))
    tasks = get_task_list(num_tasks, world, rank, local_rank)
    print(f"worker global rank:{rank}\tlocal rank: {local_rank}\tprocessing tasks {tasks}", flush=True)
    _bind_torch_to_device(local_rank)
    worker(tasks, device=local_rank, **worker_args)


def _bind_torch_to_device(local_rank):
    """One process per GPU: the HIP library is bound to LOCAL_RANK through `device=`; torch (pinned collation buffers of the
    readers, torch.cuda calls of user code) must follow, or every rank opens a context and page-locks memory against GPU 0."""
    try:
        import torch  # pylint: disable=import-outside-toplevel

        if torch.cuda.is_available() and local_rank < torch.cuda.device_count():
            torch.cuda.set_device(local_rank)
    except ImportError:
        pass


def _main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    # every option defaults to None = "not given": gpu_worker fills those from WORKER_ARGS_PATH (the json the reference's slurm
    # distributor writes), then from worker()'s own defaults
    truth = lambda v: v.lower() in ("1", "true")  # noqa: E731
    ap.add_argument("--input_dataset")
    ap.add_argument("--output_folder")
    ap.add_argument("--output_partition_count", type=int)
    ap.add_argument("--input_format")
    ap.add_argument("--cache_path")
    ap.add_argument("--batch_size", type=int)
    ap.add_argument("--num_prepro_workers", type=int)
    ap.add_argument("--enable_text", type=truth)
    ap.add_argument("--enable_image", type=truth)
    ap.add_argument("--enable_metadata", type=truth)
    ap.add_argument("--wds_image_key")
    ap.add_argument("--wds_caption_key")
    ap.add_argument("--clip_model")
    ap.add_argument("--clip_cache_path")
    ap.add_argument("--gpu_normalise", type=truth, help="uint8 crops to the GPU, /255 - mean / std there (default true)")
    ap.add_argument("--gpu_resize", type=truth, help="decode only on the host; resize + centre crop on the GPU too, bit-identical to Pillow (default false)")
    gpu_worker(**vars(ap.parse_args(argv)))


if __name__ == "__main__":
    _main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native clip-retrieval hot path.

Metric (BASELINE.json): images/sec embedded (ViT-L/14 bs=256) + QPS@top-40 over a flat fp16 index.
A "step" = one pass of the encode hot path over one batch of 256 synthetic samples already resident in HBM:
image tower on f32 [256,3,224,224] (the reference's `image_tensor`) + text tower on int32 [256,77], each through
L2-normalise + fp16 (what ClipMapper.__call__ does per batch, reference mapper.py:49-78).  `value` = whole-job
samples (image+text pairs)/s; image-only and text-only rates and the kNN scan are reported in extra fields.

Multi-GPU: encode = replicas only (no collective; weak scaling: every rank encodes its own batches).
kNN = row-sharded index, one all-gather of per-shard top-k + merge (RCCL), also weak (fixed rows per GPU).

    python bench.py [--gpus N --steps K --warmup W]          # N>1: launched by torch.distributed.run
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_PEAK_TFLOPS = 2500.0  # dense MFMA bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0      # HBM3E spec


def pmc_traffic(kernel):
    """HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) from the rocprofv3 --pmc passes of this same command, collected
    by tools/gpu_round.sh (separate passes, kernel-trace only) and summarised by tools/traffic_summary.py into
    profiles/traffic.json; None when that file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            k = json.load(f)["kernels"][kernel]
        return int(k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--knn-rows", type=int, default=-1, help="index rows per GPU (-1: 100M, 125M at 8 GPUs; 0: skip)")
    ap.add_argument("--knn-queries", type=int, default=64, help="queries per batch (64 = one wide scan; 32 = one exact scan)")
    ap.add_argument("--knn-scans", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0: skip)")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="torch threads of the CPU baseline (all 256 cores of the GPU box oversubscribe torch's CPU GEMMs: 0.07 samples/s)")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, random_blob
    from clip_retrieval_amd.knn import Mi355xIndex

    arch = ARCHS[args.model]
    B = args.batch
    single = world == 1 and rank == 0
    want_cpu = single and args.cpu_seconds > 0
    want_parity = single and not args.no_parity

    # ---- weights: random init of the named architecture (no checkpoints exist offline).  At N=1 the CPU-baseline
    # leg loads the SAME blob into the oracle (checker) so it can both be timed and gate parity of this very run.
    blob = random_blob(arch, seed=0)
    enc = ClipEncoder(arch, blob, local_rank)
    oracle = None
    if want_cpu or want_parity:
        from oracle.clip_oracle import ARCHS as OARCHS, HFClipOracle

        cpu_threads = min(os.cpu_count() or 1, args.cpu_threads)
        oracle = HFClipOracle(OARCHS[args.model], seed=0, threads=cpu_threads)
        oracle.load_blob(blob)
    del blob

    from clip_retrieval_amd.synth import normalise_u8_nhwc, perturbed_queries, synth_pixels_u8, synth_tokens, tower_gflop

    gf_img, gf_txt = tower_gflop(arch)
    pix_host = normalise_u8_nhwc(synth_pixels_u8(B, arch.image_size, seed=1 + rank))
    ids_host = synth_tokens(B, arch.ctx_len, arch.vocab, seed=2 + rank)
    pix = torch.from_numpy(pix_host).to(dev)
    ids = torch.from_numpy(ids_host).to(dev)
    out_i = torch.empty(B, arch.embed_dim, dtype=torch.float16, device=dev)
    out_t = torch.empty(B, arch.embed_dim, dtype=torch.float16, device=dev)
    o32_i = torch.empty(B, arch.embed_dim, dtype=torch.float32, device=dev)
    o32_t = torch.empty(B, arch.embed_dim, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step(img=True, txt=True):
        if img:
            enc.encode_image_device(pix.data_ptr(), B, 0, out_i.data_ptr(), o32_i.data_ptr(), stream)
        if txt:
            enc.encode_text_device(ids.data_ptr(), B, out_t.data_ptr(), o32_t.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    # timed region: only the dominant kernel family (GEMM, kind 0) is bracketed by hipEvents; the other kinds are
    # timed in an extra, untimed pass below (event records are not free: ~2 % of the step when every launch has two)
    PROF_GEMM, PROF_REST = 2, 4 | 8 | 16  # clipx_profile_enable masks: bit (kind + 1)
    enc.profile(PROF_GEMM)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    enc.profile(0)
    value = world * args.steps * B / dt

    # per-kernel live timing (hipEvents on the launch stream).  GEMM: recorded inside the timed region above.
    kinds = {"gemm": 0, "attention": 1, "layernorm": 2, "other": 3}
    prof = {}
    n, ms, fl = enc.profile_get(0)
    prof["gemm"] = {"launches": n, "ms": round(ms, 3), "tflops": (fl / (ms * 1e-3) / 1e12) if ms > 0 and fl > 0 else None, "steps": args.steps}
    enc.profile(PROF_REST)
    extra_steps = max(2, args.steps // 2)
    for _ in range(extra_steps):
        step()
    barrier()
    enc.profile(0)
    for name, kind in kinds.items():
        if name == "gemm":
            continue
        n, ms, fl = enc.profile_get(kind)
        prof[name] = {"launches": n, "ms": round(ms, 3), "tflops": (fl / (ms * 1e-3) / 1e12) if ms > 0 and fl > 0 else None, "steps": extra_steps}
    g = prof["gemm"]
    gemm_tflops = g["tflops"] or 0.0
    roofline = {"bound": "mfma", "kernel": "gemm256sp_kernel (+ gemm_bf16_kernel on the peeled 257th m-tile)", "achieved": round(gemm_tflops, 1), "peak": BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(gemm_tflops / BF16_PEAK_TFLOPS, 4), "traffic": pmc_traffic("gemm"),
                "launches": g["launches"], "avg_launch_ms": round(g["ms"] / max(g["launches"], 1), 4)}

    # image-only / text-only rates (untimed extras)
    extras = {}
    for key, kw in (("images_per_s", dict(img=True, txt=False)), ("texts_per_s", dict(img=False, txt=True))):
        barrier()
        t1 = time.perf_counter()
        for _ in range(max(2, args.steps // 2)):
            step(**kw)
        barrier()
        extras[key] = round(world * max(2, args.steps // 2) * B / max_over_ranks(time.perf_counter() - t1), 1)
    e2e_tflops = value / world * (gf_img + gf_txt) / 1e3
    extras["end_to_end_tflops_per_gpu"] = round(e2e_tflops, 1)
    extras["end_to_end_frac_of_mfma_peak"] = round(e2e_tflops / BF16_PEAK_TFLOPS, 4)
    extras["kernel_ms_per_step"] = {k: round(v["ms"] / v["steps"], 3) for k, v in prof.items()}

    # ---- parity gate on the benchmark's own weights and inputs (oracle = checker)
    parity = None
    if want_parity:
        from oracle.clip_oracle import mapper_semantics

        nchk = 2
        _, wi = mapper_semantics(oracle.encode_image(torch.from_numpy(pix_host[:nchk])))
        _, wt = mapper_semantics(oracle.encode_text(torch.from_numpy(ids_host[:nchk])))
        gi, gt = out_i[:nchk].float().cpu().numpy().astype(np.float64), out_t[:nchk].float().cpu().numpy().astype(np.float64)
        ci = (gi * wi).sum(-1) / (np.linalg.norm(gi, axis=-1) * np.linalg.norm(wi, axis=-1))
        ct = (gt * wt).sum(-1) / (np.linalg.norm(gt, axis=-1) * np.linalg.norm(wt, axis=-1))
        parity = {"checked": nchk, "image_cos_min": float(ci.min()), "text_cos_min": float(ct.min()), "bar": 1 - 1e-3,
                  "ok": bool(ci.min() >= 1 - 1e-3 and ct.min() >= 1 - 1e-3)}

    # ---- CPU baseline: the oracle (transformers.CLIPModel fp32 = the reference's hf_clip backend + mapper.py's
    # normalise/fp16) on this box's host cores, on a bounded sample of the same workload
    cpu = None
    if want_cpu:
        from oracle.clip_oracle import mapper_semantics

        cb = 4
        done, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_seconds and done < B:
            mapper_semantics(oracle.encode_image(torch.from_numpy(pix_host[done:done + cb])))
            mapper_semantics(oracle.encode_text(torch.from_numpy(ids_host[done:done + cb])))
            done += cb
        el = time.perf_counter() - t1
        cpu = {"value": round(done / el, 3), "unit": "samples/s", "cores": cpu_threads, "kind": "port",
               "sample": f"{done} of the {B} image+text pairs of one step, fp32, torch CPU ({el:.1f} s)"}
    del oracle

    # ---- kNN: flat fp16 index resident in HBM, top-40
    knn = None
    rows = args.knn_rows
    if rows < 0:
        rows = 125_000_000 if world == 8 else 100_000_000
    if rows > 0:
        from clip_retrieval_amd.distributed import ShardedIndex

        d, k, nq, seed = 768, 40, args.knn_queries, 3
        enc_free = torch.cuda.mem_get_info(dev)[0]
        rows = int(min(rows, (enc_free - (8 << 30)) // (d * 2)))
        ix = Mi355xIndex(d, device=local_rank, id_base=rank * rows)
        # every shard is the same synthetic generator with a per-rank seed: rows of rank r are synth(seed + r)
        ix.synth_fill(rows, seed + rank)
        # queries = perturbed copies of rows of rank 0's shard (planted neighbours -> self-check at full scale)
        rng = np.random.default_rng(7)
        planted_local = np.sort(rng.choice(rows, nq, replace=False))
        q = torch.empty(nq, d, dtype=torch.float32, device=dev)
        if rank == 0:
            q.copy_(torch.from_numpy(perturbed_queries(ix.reconstruct_batch(planted_local))))
        if world > 1:
            dist.broadcast(q, src=0)
        sh = ShardedIndex(ix)
        D, I = sh.search_device(q, k)  # warm-up + correctness
        torch.cuda.synchronize()
        hit = bool((I[:, 0].cpu().numpy() == planted_local).all())
        ix.profile(True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.knn_scans):
            D, I = sh.search_device(q, k)
        barrier()
        dk = max_over_ranks(time.perf_counter() - t1)
        ix.profile(False)
        nl, ms = ix.profile_get()
        scan_ms = ms / max(nl, 1)
        scan_gbs = rows * d * 2 / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        knn = {"metric": "QPS@top-40, flat IP, fp16 rows in HBM", "qps": round(args.knn_scans * nq / dk, 1),
               "rows_per_gpu": rows, "total_rows": rows * world, "d": d, "k": k, "queries_per_scan": nq,
               "ms_per_batch": round(dk / args.knn_scans * 1e3, 3), "planted_neighbour_top1": hit,
               "roofline": {"bound": "hbm", "kernel": "knn_scan_kernel", "achieved": round(scan_gbs, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(scan_gbs / HBM_PEAK_GBS, 4),
                            "traffic": pmc_traffic("knn_scan_kernel") if rows == 100_000_000 else None,
                            "launches": nl, "avg_launch_ms": round(scan_ms, 4),
                            "algorithmic_bytes_per_launch": rows * d * 2}}
        ix.close()

    if rank == 0:
        line = {
            "metric": "images/sec embedded (ViT-L/14 bs=256; each sample = image + caption through both towers)",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} image+text encode, bs={B} per GPU, f32 NCHW pixels + int32 tokens resident in HBM "
                                   "(BASELINE.json configs[1]); random-init weights",
                       "global_batch": B * world, "parallelism": f"replicas x{world} (no collective)"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "knn": knn, **extras,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native clip-retrieval hot path.

Metric (BASELINE.json): images/sec embedded (ViT-L/14 bs=256) + QPS@top-40 over a flat fp16 index.
A "step" = one pass of the encode hot path over one batch of 256 synthetic samples already resident in HBM:
image tower on f32 [256,3,224,224] (the reference's `image_tensor`) + text tower on int32 [256,77], each through
L2-normalise + fp16 (what ClipMapper.__call__ does per batch, reference mapper.py:49-78).  `value` = whole-job
samples (image+text pairs)/s; image-only and text-only rates and the kNN scan are reported in extra fields.

Multi-GPU: encode = replicas only (no collective; weak scaling: every rank encodes its own batches).
kNN = row-sharded index, one all-gather of per-shard top-k + merge (RCCL), also weak (fixed rows per GPU).

The line is only printed when every parity gate holds; otherwise the process exits non-zero (fail closed):
  * encode: all 256 image and text rows of the timed batch against the fp32 CPU oracle (cosine >= 1 - 1e-3);
  * kNN:    planted neighbours are the top hit at full scale, exact id lists against the numpy oracle on the first 1 M
            rows, id sets against a chunked torch fp32 matmul + topk over the whole index.

    python bench.py [--gpus N --steps K --warmup W]          # N>1: launched by torch.distributed.run
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_PEAK_TFLOPS = 2500.0  # dense MFMA bf16 / f16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0      # HBM3E spec


def refuse_debug_environment():
    """A timed region must run the product kernels: refuse any switch that changes what the hot path computes."""
    bad = [k for k in os.environ if (k.startswith("CLIPX_") and ("DBG" in k or "CFG" in k)) or k in ("CLIPX_GEMM_VARIANT", "KNNX_WIDE", "KNNX_RQ", "KNNX_RQ_MIN_ROWS", "KNNX_GRID", "KNNX_NT")]
    if bad:
        raise SystemExit(f"bench.py refuses to run with debug/ablation switches set: {sorted(bad)}")


def counters_match(j):
    """A committed counter file belongs beside this run's timings only if it was measured on THIS tree's kernels: the summaries are
    stamped with a digest of csrc/ + include/ (tools/source_digest.py); a file of another tree (or an unstamped one) is ignored and
    the line says traffic / mfma_busy_frac null rather than quoting stale counters."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_digest import source_digest  # pylint: disable=import-outside-toplevel

        return j.get("source_digest") == source_digest()
    except Exception:  # pylint: disable=broad-except
        return False


def pmc_traffic(kernel):
    """HBM bytes per STEP (encode kernels) / per scan (kNN) from the rocprofv3 --pmc passes of this same command
    (tools/gpu_round.sh -> tools/traffic_summary.py -> profiles/traffic.json: FETCH_SIZE x2 + WRITE_SIZE, separate passes,
    kernel-trace only).  Counters cannot be read from inside the run that is being timed; None when the file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            j = json.load(f)
        if not counters_match(j):
            return None, None
        k = j["kernels"][kernel]
        return int(k["bytes_per_unit"]), k["unit"]
    except (OSError, KeyError, ValueError):
        return None, None


def pmc_mfma_busy(family):
    """Measured matrix-pipe utilisation of a kernel family: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) from the
    rocprofv3 --pmc pass of this same command (tools/gpu_round.sh -> tools/mfma_busy_summary.py -> profiles/mfma_busy.json); None when
    the file is absent.  The `frac` beside it is arithmetic (flops / time / peak): the two differ by the clock the part sustained."""
    try:
        with open(os.path.join(ROOT, "profiles", "mfma_busy.json")) as f:
            j = json.load(f)
        if not counters_match(j):
            return None, None
        k = j["families"][family]
        return k["mfma_busy_frac"], k["effective_clock_ghz"]
    except (OSError, KeyError, ValueError):
        return None, None


def headline_numbers(line):
    """(flat dict of the numbers README / DESIGN quote, <= 12 stderr lines with the same numbers) from the finished result line.
    Key names carry the value class: knn_* = configs[2] (100 M x 768 isotropic), knn125m_* = the per-GPU shard of the headline
    configuration (125 M x 768), knn_aniso_* = the corpus with dominant columns, ivf_* = configs[4]'s shard, pipeline_* = configs[3]'s."""
    h = {"encode_samples_per_s": line["value"], "encode_gemm_frac": line["roofline"]["frac"], "encode_ms_per_step": line["ms_per_step"]}
    out = [f"HEADLINE encode {line['value']} samples/s ({line['ms_per_step']} ms/step, GEMM {line['roofline']['achieved']} TF = frac {line['roofline']['frac']})"
           + (f"; incl H2D/D2H {line['value_incl_h2d_d2h']}" if line.get("value_incl_h2d_d2h") else "")]
    par = line.get("parity")
    if par:
        h["encode_parity_ok"] = bool(par["ok"])
        h["encode_cos_min"] = round(min(par["image_cos_min"], par["text_cos_min"]), 7)
        out.append(f"HEADLINE parity ok={par['ok']} rows={par['checked']} image_cos_min={par['image_cos_min']:.7f} text_cos_min={par['text_cos_min']:.7f}")
    if line.get("cpu_baseline"):
        h["cpu_encode_samples_per_s"] = line["cpu_baseline"]["value"]

    def rows_of(leg, prefix, label):
        if not leg:
            return
        parts = []
        for r in leg["by_batch"]:
            b = r["B"]
            h[f"{prefix}_qps_b{b}"] = r["qps"]
            h[f"{prefix}_pass_ms_b{b}"] = r["scan_ms"]
            h[f"{prefix}_hbm_frac_b{b}"] = r["hbm_frac"]
            parts.append(f"B={b}: {r['qps']} QPS, batch {r['ms_per_batch']} ms, pass {r['scan_ms']} ms, hbm {r['hbm_frac']}")
        h[f"{prefix}_proof_failures"] = leg["wide_fallbacks"]
        h[f"{prefix}_planted_top1"] = bool(leg["planted_neighbour_top1"])
        out.append(f"HEADLINE {label} " + " | ".join(parts) + f" | fallbacks {leg['wide_fallbacks']}")

    knn = line.get("knn")
    if knn:
        rows_of(knn, "knn", f"kNN {knn['rows_per_gpu'] // 1_000_000}Mx768 isotropic")
        for sub in knn.get("by_shard_size") or []:
            rows_of(sub, "knn125m" if sub["rows_per_gpu"] == 125_000_000 else f"knn{sub['rows_per_gpu'] // 1_000_000}m", f"kNN {sub['rows_per_gpu'] // 1_000_000}Mx768 shard")
        rows_of(knn.get("anisotropic_corpus"), "knn_aniso", "kNN 100Mx768 dominant-columns")
        ck = knn.get("checks") or {}
        c1, c2 = ck.get("first_1M_rows_vs_numpy_oracle"), ck.get("full_index_vs_torch_matmul_topk")
        if c1:
            h["knn_check_first_1m_vs_numpy_oracle"] = bool(c1["id_lists_identical"] or c1["id_sets_equal_up_to_2e-6_ties"])
        if c2:
            h["knn_check_full_index_vs_torch"] = bool(c2["id_sets_equal_up_to_1e-5_ties"])
        h["knn_check_planted_top1"] = bool(knn["planted_neighbour_top1"])
        cb, hs = knn.get("cpu_baseline"), knn.get("host_search_and_reconstruct")
        if cb:
            h["cpu_knn_qps_extrapolated"] = cb["value"]
        if hs:
            for r in hs:
                h[f"knn_host_search_and_reconstruct_ms_b{r['B']}"] = r["ms_per_call"]
        out.append("HEADLINE kNN checks: planted_top1=%s first_1M_vs_numpy=%s full_index_vs_torch=%s | cpu %s QPS | host search_and_reconstruct %s" % (
            h["knn_check_planted_top1"], h.get("knn_check_first_1m_vs_numpy_oracle"), h.get("knn_check_full_index_vs_torch"),
            cb["value"] if cb else None, ", ".join(f"B={r['B']} {r['ms_per_call']} ms" for r in hs) if hs else None))
    ivf = line.get("ivf")
    if ivf:
        parts = []
        for r in ivf.get("by_nprobe", []):
            npb = r["nprobe"]
            b32, b256 = r.get("batch32") or {}, r.get("batch256") or {}
            served = max((x["qps"] for x in r.get("served", []) if not x.get("deduplicate")), default=None)
            h[f"ivf_np{npb}_recall40"] = r["recall_at_40_vs_exact_whole_shard"]
            h[f"ivf_np{npb}_qps_b32"] = b32.get("qps")
            h[f"ivf_np{npb}_ms_b256"] = b256.get("ms_per_batch")
            h[f"ivf_np{npb}_served_qps"] = served
            h[f"ivf_np{npb}_b256_scanned_over_union"] = b256.get("scanned_over_union")
            parts.append(f"np{npb}: recall {r['recall_at_40_vs_exact_whole_shard']}, B32 {b32.get('qps')} QPS, B256 {b256.get('ms_per_batch')} ms "
                         f"({b256.get('passes')} pass, {b256.get('scanned_over_union')} x union), served {served}")
        out.append(f"HEADLINE IVF {ivf['rows'] // 1_000_000}Mx{ivf['d']} nlist {ivf['nlist']} " + " | ".join(parts))
    pl = line.get("pipeline")
    if pl:
        for k_ in ("samples_per_s_u8", "samples_per_s_jpeg", "frac_of_value"):
            if pl.get(k_) is not None:
                h["pipeline_" + k_] = pl[k_]
        out.append(f"HEADLINE pipeline (tar shards -> reader -> runner -> writer) u8 {pl.get('samples_per_s_u8')} samples/s, jpeg {pl.get('samples_per_s_jpeg')}, frac_of_value {pl.get('frac_of_value')}")
    return h, [ln[:330] for ln in out[:12]]


def pipeline_leg(model, value, device, shards_n=8, per_shard=2048, workers=16, log=lambda m: None):
    """BASELINE configs[3]'s single-GPU share: synthetic tar shards in /dev/shm (seeded, untimed) ->
    `worker()` -> WebdatasetReader (decode processes) -> pipelined Runner -> ClipMapper (asynchronous tickets of the C ABI) ->
    NumpyWriter, timed wall to wall -- the reference's own loop (runner.py:27-64, reader.py:125-205) around the hot path.
    Two variants: members that are PRE-DECODED uint8 224 x 224 x 3 (binary PPM: a 15-byte header + the pixels; what is left of the
    reader is tar iteration, a memcpy and batching) and JPEG members of 256 x 256 with decode on the host, resize + centre crop on
    the GPU (`gpu_resize=True`).  Partition 0 of 2 is the untimed start-up (model build, decode processes, warm-up), partition 1 is
    timed; outputs are checked for the writer's file names, shapes and dtypes (writer.py:67-106)."""
    import glob
    import gzip
    import io
    import shutil
    import tarfile
    import tempfile

    import numpy as np
    from PIL import Image

    from clip_retrieval_amd.reader import _DecodePool
    from clip_retrieval_amd.worker import worker

    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (8 << 30) else None
    tmp = tempfile.mkdtemp(prefix="clipx_pipeline_", dir=base)
    out = {"shards": shards_n, "members_per_shard": per_shard, "decode_processes": workers, "model": model, "where": tmp,
           "timed": f"partition 1 of 2 ({shards_n // 2} shards x {per_shard} samples) through worker(); partition 0 is the untimed start-up"}
    old_bpe = os.environ.get("CLIP_BPE_PATH")
    try:
        rng = np.random.default_rng(11)
        bpe = os.path.join(tmp, "bpe_simple_vocab_16e6.txt.gz")  # a stand-in merges file (the real vocabulary is not available offline)
        with gzip.open(bpe, "wt", encoding="utf-8") as f:
            f.write("#version: 0.2\nt h\nth e</w>\no f</w>\np h\n")
        os.environ["CLIP_BPE_PATH"] = bpe
        n_distinct = 256
        ppm, jpg = [], []
        for i in range(n_distinct):
            a = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
            ppm.append(b"P6\n224 224\n255\n" + a.tobytes())
            g = np.linspace(0, 255, 256, dtype=np.float32)
            b = (g[None, :, None] * 0.5 + g[:, None, None] * 0.5 + rng.normal(0, 12, (256, 256, 3))).clip(0, 255).astype(np.uint8)
            buf = io.BytesIO()
            Image.fromarray(np.roll(b, 7 * i, axis=1)).save(buf, format="JPEG", quality=90)
            jpg.append(buf.getvalue())
        for variant, ext, payloads, kw in (("u8", "ppm", ppm, dict(gpu_normalise=True)), ("jpeg", "jpg", jpg, dict(gpu_resize=True))):
            shards = []
            for sh in range(shards_n):
                path = os.path.join(tmp, f"{variant}_{sh:03d}.tar")
                with tarfile.open(path, "w") as tf:
                    for i in range(per_shard):
                        k = sh * per_shard + i
                        for e, data in ((ext, payloads[k % n_distinct]), ("txt", f"a photo number {k} of something".encode())):
                            ti = tarfile.TarInfo(f"{sh:03d}{i:06d}.{e}")
                            ti.size = len(data)
                            tf.addfile(ti, io.BytesIO(data))
                shards.append(path)
            folder = os.path.join(tmp, "out_" + variant)
            args = dict(input_dataset=shards, output_folder=folder, output_partition_count=2, input_format="webdataset", batch_size=256,
                        num_prepro_workers=workers, enable_text=True, enable_image=True, enable_metadata=False, wds_image_key=ext,
                        clip_model="random:" + model, device=device, **kw)
            import contextlib

            with contextlib.redirect_stdout(sys.stderr):  # worker() prints its progress like the reference: keep stdout for the line
                worker([0], **args)
                t0 = time.perf_counter()
                worker([1], **args)
                el = time.perf_counter() - t0
            n_timed = len(range(1, shards_n, 2)) * per_shard  # Sampler: partition 1 of 2 takes the odd shards (runner.py:13-14)
            names = sorted(os.path.relpath(f, folder) for f in glob.glob(folder + "/*/*") if "stats" not in f)
            want = sorted([f"img_emb/img_emb_{i}.npy" for i in range(2)] + [f"text_emb/text_emb_{i}.npy" for i in range(2)]
                          + [f"metadata/metadata_{i}.parquet" for i in range(2)])
            img, txt = np.load(folder + "/img_emb/img_emb_1.npy", mmap_mode="r"), np.load(folder + "/text_emb/text_emb_1.npy", mmap_mode="r")
            with open(folder + "/stats/1.json") as f:
                stats = json.load(f)
            ok = bool(names == want and img.shape == txt.shape == (n_timed, img.shape[1]) and img.dtype == np.float16 == txt.dtype
                      and stats["sample_count"] == n_timed and np.isfinite(np.asarray(img[:256], dtype=np.float32)).all())
            out[variant] = {"samples_per_s": round(n_timed / el, 1), "seconds": round(el, 3), "samples": n_timed, "member": ext,
                            "member_bytes": len(payloads[0]), "outputs_ok": ok,
                            "stats_sum": {k_: round(float(v), 3) for k_, v in stats.items() if k_.endswith("_duration")}}
            log(f"pipeline {variant}: {n_timed} samples in {el:.3f} s = {n_timed / el:.0f} samples/s; outputs ok = {ok}; stats {out[variant]['stats_sum']}")
            for f in shards:
                os.remove(f)
            shutil.rmtree(folder, ignore_errors=True)
        out["samples_per_s_u8"] = out["u8"]["samples_per_s"]
        out["samples_per_s_jpeg"] = out["jpeg"]["samples_per_s"]
        out["frac_of_value"] = round(out["samples_per_s_u8"] / value, 4) if value else None
        out["jpeg_frac_of_value"] = round(out["samples_per_s_jpeg"] / value, 4) if value else None
        # reader-bound: the summed read_duration (time the loop waited for a batch) is a sizeable share of the partition's wall time
        out["reader_bound"] = {v: bool(out[v]["stats_sum"].get("read_duration", 0.0) > 0.15 * out[v]["seconds"]) for v in ("u8", "jpeg")}
    finally:
        _DecodePool.shutdown()
        shutil.rmtree(tmp, ignore_errors=True)
        if old_bpe is None:
            os.environ.pop("CLIP_BPE_PATH", None)
        else:
            os.environ["CLIP_BPE_PATH"] = old_bpe
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--model", default="ViT-L/14")
    ap.add_argument("--knn-rows", type=int, default=-1, help="index rows per GPU (-1: 100M, 125M at 8 GPUs; 0: skip)")
    ap.add_argument("--knn-batches", default="1,32,64,256", help="query batch sizes to time (SURVEY config 3: 1, 32, 256)")
    ap.add_argument("--knn-scans", type=int, default=5)
    ap.add_argument("--cpu-seconds", type=float, default=1.0, help="0 skips the CPU baseline legs (the encode baseline is timed on the parity pass)")
    ap.add_argument("--cpu-threads", type=int, default=32,
                    help="torch threads of the CPU baseline (all 256 cores of the GPU box oversubscribe torch's CPU GEMMs: 0.07 samples/s)")
    ap.add_argument("--parity-rows", type=int, default=64, help="rows of the timed batch the CPU oracle checks (and is timed on); all rows are checked for finite / unit norm")
    ap.add_argument("--no-parity", action="store_true", help="profiling runs only: the line is marked parity=null")
    ap.add_argument("--no-host-path", dest="host_path", action="store_false", help="skip the pinned-host-buffer leg (value_incl_h2d_d2h)")
    ap.add_argument("--no-ab", action="store_true", help="profiling runs only: skip the rectangular-text A/B (its extra steps would enter the per-step counter averages)")
    ap.add_argument("--profile-run", action="store_true", help="counter / kernel-trace runs (tools/gpu_round.sh): only the legs whose steps "
                    "tools/traffic_summary.py counts -- no parity, no CPU legs, no rectangular-text A/B, no host-buffer leg, no extra kNN legs, no ivf")
    ap.add_argument("--no-knn-extra", dest="knn_extra", action="store_false",
                    help="skip the two extra kNN legs (125 M x 768 = the per-GPU shard of the headline configuration; the anisotropic corpus)")
    ap.add_argument("--no-ivf", dest="ivf", action="store_false", help="skip BASELINE config 5's per-GPU shard (tools/config5.py: IVF-Flat "
                    "125 M x 1024, nlist 65 536, built on the device, served; embedded as `ivf`, ~1 minute)")
    ap.add_argument("--ivf-rows", type=int, default=125_000_000)
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="skip BASELINE config 4's single-GPU share (tar shards -> "
                    "worker() -> reader -> pipelined Runner -> writer, pre-decoded uint8 and JPEG members; embedded as `pipeline`, ~30 s)")
    args = ap.parse_args()
    if args.profile_run:
        args.no_parity, args.no_ab, args.knn_extra, args.ivf, args.cpu_seconds, args.host_path = True, True, False, False, 0.0, False
        args.pipeline = False
    refuse_debug_environment()

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
    # Under the launcher (torch.distributed.run sets RANK / MASTER_ADDR / MASTER_PORT) the process group is created even at world size 1,
    # so that `--nproc-per-node 1` executes what N ranks execute: init_process_group("nccl") = RCCL, the barrier, the MAX all-reduce of
    # the timings, the broadcast of the planted queries (tests/test_launcher_gpu.py runs exactly that on the one GPU of a test box).
    use_dist = world > 1 or all(k in os.environ for k in ("RANK", "MASTER_ADDR", "MASTER_PORT"))
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    from clip_retrieval_amd.encoder import ARCHS, ClipEncoder, random_blob
    from clip_retrieval_amd.knn import Mi355xIndex

    arch = ARCHS[args.model]
    B = args.batch
    single = world == 1 and rank == 0
    want_parity = single and not args.no_parity
    want_cpu = single and args.cpu_seconds > 0
    failures = []

    # ---- weights: random init of the named architecture (no checkpoints exist offline).  At N=1 the SAME blob is loaded
    # into the oracle (checker), which gates parity of this very run and is timed as the CPU baseline.
    blob = random_blob(arch, seed=0)
    enc = ClipEncoder(arch, blob, local_rank)
    oracle = None
    cpu_threads = min(os.cpu_count() or 1, args.cpu_threads)
    if want_parity:
        from oracle.clip_oracle import ARCHS as OARCHS, HFClipOracle

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
            # the tokens exist on the host too (the reference's batch holds them there, mapper.py:63-65): no read-back in the timed region
            enc.encode_text_device(ids.data_ptr(), B, out_t.data_ptr(), o32_t.data_ptr(), stream, ids_host=ids_host)

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
    # the committed counter passes were taken on the default workload only
    traffic, traffic_unit = pmc_traffic("gemm") if (args.model == "ViT-L/14" and B == 256) else (None, None)
    # one denominator throughout: everything below is PER STEP (one batch of 256 through both towers)
    roofline = {"bound": "mfma", "kernel": "gemm256w4_kernel (4-wave 256x256, the 16-bit-output and fp16-residual forms) + gemm256sp_kernel / gemm_bf16_kernel on the other forms and ragged rows; bf16 / fp16 operands, f32 accumulate", "achieved": round(gemm_tflops, 1),
                "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gemm_tflops / BF16_PEAK_TFLOPS, 4),
                "per": "step", "launches_per_step": g["launches"] // max(args.steps, 1), "ms_per_step": round(g["ms"] / max(args.steps, 1), 3),
                "algorithmic_tflop_per_step": round(gemm_tflops * g["ms"] / max(args.steps, 1) / 1e3, 3),
                "mfma_busy_frac": pmc_mfma_busy("gemm")[0] if (args.model == "ViT-L/14" and B == 256) else None,
                "mfma_busy_clock_ghz": pmc_mfma_busy("gemm")[1] if (args.model == "ViT-L/14" and B == 256) else None,
                "mfma_busy_source": "profiles/mfma_busy.json: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass of this command (tools/gpu_round.sh, separate run); busy cycles / (1024 SIMDs x kernel cycles)",
                "traffic": traffic, "traffic_unit": traffic_unit, "traffic_run": "separate --pmc pass" if traffic else None,
                "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) + WRITE_SIZE passes of this command (tools/gpu_round.sh); counters cannot be read inside the timed run" if traffic else None}

    # image-only / text-only rates (untimed extras)
    extras = {}
    for key, kw in (("images_per_s", dict(img=True, txt=False)), ("texts_per_s", dict(img=False, txt=True))):
        barrier()
        t1 = time.perf_counter()
        for _ in range(max(2, args.steps // 2)):
            step(**kw)
        barrier()
        extras[key] = round(world * max(2, args.steps // 2) * B / max_over_ranks(time.perf_counter() - t1), 1)
    # the headline depends on the caption lengths (the text tower runs on the rows up to each caption's EOT only): print the length
    # distribution and, in the same process, the figure with every row computed (CLIPX_OPT_RAGGED_TEXT = 0)
    tok_len = (ids_host.argmax(axis=1) + 1).astype(np.int64)
    extras["text_rows"] = {"tokens_incl_sot_eot": {"min": int(tok_len.min()), "mean": round(float(tok_len.mean()), 2), "max": int(tok_len.max())},
                           "rows_run": int(tok_len.sum()), "rows_rectangular": int(B * arch.ctx_len)}
    if enc.get_option(enc.OPT_RAGGED_TEXT) == 1 and not args.no_ab:
        # same-process A/B, both legs timed the same way (no per-launch events: the headline region above carries two hipEvents per
        # GEMM launch for `roofline`, which costs ~0.5 % -- its `value` is the conservative one)
        ab = {}
        for name, opt in (("ragged", 1), ("rectangular", 0)):
            enc.set_option(enc.OPT_RAGGED_TEXT, opt)
            step()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            barrier()
            ab[name] = max_over_ranks(time.perf_counter() - t1)
        enc.set_option(enc.OPT_RAGGED_TEXT, 1)
        step()
        barrier()
        extras["value_rectangular_text"] = round(world * args.steps * B / ab["rectangular"], 1)
        extras["ms_per_step_rectangular_text"] = round(ab["rectangular"] / args.steps * 1e3, 3)
        extras["value_ragged_text_same_timing"] = round(world * args.steps * B / ab["ragged"], 1)
    enc.check_range(stream)  # CLIPX_E_RANGE: no launch of this run may have overflowed the fp16 residual stream

    if args.host_path:
        # What ClipMapper.__call__ includes and `value` does not ("device-only AND including H2D/D2H"): the same batches from
        # pinned HOST buffers -- f32 NCHW pixels + int32 tokens in, fp16 rows out on the host -- through the asynchronous tickets
        # (clipx_encode_*_async / clipx_wait: upload, both towers, download; two steps in flight so that a step's copies overlap its
        # neighbour's kernels, as a pipelined Runner would drive it).  Never `value`.
        pin_pix, pin_ids = torch.from_numpy(pix_host).pin_memory(), torch.from_numpy(ids_host).pin_memory()
        def host_step():
            return enc.submit_image(pin_pix.numpy()), enc.submit_text(pin_ids.numpy())
        prev = host_step()
        for h in prev:
            enc.collect(h)
        barrier()
        t1 = time.perf_counter()
        prev = host_step()
        for _ in range(args.steps - 1):
            cur = host_step()
            for h in prev:
                enc.collect(h)
            prev = cur
        emb_host = [enc.collect(h) for h in prev]
        barrier()
        dt_host = max_over_ranks(time.perf_counter() - t1)
        extras["value_incl_h2d_d2h"] = round(world * args.steps * B / dt_host, 1)
        extras["ms_per_step_incl_h2d_d2h"] = round(dt_host / args.steps * 1e3, 3)
        extras["incl_h2d_d2h_note"] = ("pinned host f32 NCHW pixels (154 MB per batch) + int32 tokens in, fp16 embeddings out on the host, "
                                       "clipx_encode_image_async / _text_async + clipx_wait, two steps in flight; `value` keeps its definition (inputs resident in HBM)")
        dev_rows = (out_i.cpu().numpy(), out_t.cpu().numpy())
        extras["host_path_bitwise_equal_to_device_path"] = bool(np.array_equal(emb_host[0], dev_rows[0]) and np.array_equal(emb_host[1], dev_rows[1]))
        if not all(np.allclose(a.astype(np.float32), b.astype(np.float32), rtol=0, atol=1e-3) for a, b in zip(emb_host, dev_rows)):
            failures.append("the host-buffer path returned other embeddings than the device path on the same batch")
        del pin_pix, pin_ids

    # FLOPs that RUN per step (counters of the launches: GEMMs + attention), not the model formula: the last block's out-proj / MLP
    # (and, in the image tower, all but the first query block of its attention) are computed on the pooled rows only
    a = prof["attention"]
    run_tflop = gemm_tflops * g["ms"] / max(args.steps, 1) / 1e3 + ((a["tflops"] or 0.0) * a["ms"] / max(a["steps"], 1) / 1e3)
    model_tflop = B * (gf_img + gf_txt) / 1e3
    e2e_tflops = run_tflop / (dt / args.steps)
    extras["end_to_end_tflops_per_gpu"] = round(e2e_tflops, 1)
    extras["end_to_end_frac_of_mfma_peak"] = round(e2e_tflops / BF16_PEAK_TFLOPS, 4)
    extras["tflop_per_step"] = {"run": round(run_tflop, 3), "full_model_formula": round(model_tflop, 3),
                                "note": "the embedding reads one row per sample of the last block: its out-proj / MLP run on those rows only (bit-identical; CLIPX_FULL_LAST_BLOCK=1 runs every row)"}
    extras["kernel_ms_per_step"] = {k: round(v["ms"] / v["steps"], 3) for k, v in prof.items()}

    # ---- parity gate on the benchmark's own weights and inputs, ALL rows of the timed batch (oracle = checker), and
    # the CPU baseline: the same pass of the oracle (transformers.CLIPModel fp32 = the reference's hf_clip backend +
    # mapper.py's normalise/fp16) is timed on this box's host cores
    parity, cpu = None, None
    if want_parity:
        from oracle.clip_oracle import mapper_semantics

        gi = out_i.float().cpu().numpy().astype(np.float64)
        gt = out_t.float().cpu().numpy().astype(np.float64)
        # every row of the timed batch must be finite and unit-norm (non-finite activations make the chip clock higher and
        # the run "faster"); `--parity-rows` of them, spread over the batch, go through the oracle (~0.37 s per pair on this
        # box's 32 threads: 64 rows keep the CPU leg near 25 s; --parity-rows 256 checks them all)
        for name, g in (("image", gi), ("text", gt)):
            nrm = np.linalg.norm(g, axis=-1)
            if not (np.isfinite(g).all() and np.abs(nrm - 1).max() < 2e-3):
                failures.append(f"encode sanity: {name} embeddings not finite / not unit norm (max |norm - 1| {np.abs(nrm - 1).max()})")
        from oracle.clip_oracle import NORTH_STAR_BAR, TIGHT_BAR, CENTRED_BAR, parity_report

        cb = 8
        n_par = max(cb, min(B, args.parity_rows)) // cb * cb
        starts = [int(round(j * (B - cb) / max(1, n_par // cb - 1))) // cb * cb for j in range(n_par // cb)] if n_par < B else list(range(0, B, cb))
        starts = sorted(set(starts))
        live_rows = np.concatenate([np.arange(o, o + cb) for o in starts])
        wi_l, wt_l = [], []
        t1 = time.perf_counter()
        for o in starts:
            wi_l.append(mapper_semantics(oracle.encode_image(torch.from_numpy(pix_host[o:o + cb])))[1])
            wt_l.append(mapper_semantics(oracle.encode_text(torch.from_numpy(ids_host[o:o + cb])))[1])
        el = time.perf_counter() - t1
        n_checked = len(live_rows)
        wi_l, wt_l = np.concatenate(wi_l), np.concatenate(wt_l)
        # the gate: raw cosine >= 1 - 1e-4, centred cosine >= 0.99, and every row of ours is nearest to ITS
        # OWN oracle row -- over the live-oracle rows, and over ALL rows against the oracle's stored embeddings of this very batch
        # (tests/golden/make_golden_bench.py; default workload only), which the live rows must reproduce
        parity = {"bar": TIGHT_BAR, "north_star_bar": NORTH_STAR_BAR, "centred_bar": CENTRED_BAR, "finite_unit_norm_rows": B, "checked": n_checked, "ok": True}
        sets = [("live", live_rows, wi_l, wt_l)]
        gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "bench_oracle_" + args.model.replace("/", "-") + f"_b{B}.npz")
        if os.path.exists(gold_path) and rank == 0:
            import hashlib

            gold = np.load(gold_path)
            if (bytes(gold["pixel_sha"]) == hashlib.sha256(pix_host.tobytes()).digest() and bytes(gold["token_sha"]) == hashlib.sha256(ids_host.tobytes()).digest()):
                gwi, gwt = gold["image_embs"].astype(np.float32), gold["text_embs"].astype(np.float32)
                drift = min(parity_report(gwi[live_rows], wi_l)["cos"].min(), parity_report(gwt[live_rows], wt_l)["cos"].min())
                parity["stored_vs_live_oracle_cos_min"] = float(drift)
                if drift < 1 - 1e-6:
                    failures.append(f"stored oracle rows ({gold_path}) disagree with the live oracle: cosine {drift}")
                sets.append(("stored", np.arange(B), gwi, gwt))
                parity["checked"] = B
            else:
                parity["stored"] = "fixture is for other inputs: ignored"
        for tag, rows_, wi_, wt_ in sets:
            for name, g_, w_ in (("image", gi[rows_], wi_), ("text", gt[rows_], wt_)):
                rep = parity_report(g_, w_)
                ok = bool(rep["cos"].min() >= TIGHT_BAR and rep["centred"].min() >= CENTRED_BAR and (rep["nearest"] == np.arange(len(rows_))).all())
                parity[f"{name}_{tag}"] = {"rows": int(len(rows_)), "cos_min": float(rep["cos"].min()), "centred_cos_min": float(rep["centred"].min()),
                                           "nearest_oracle_row_is_own_row": bool((rep["nearest"] == np.arange(len(rows_))).all()),
                                           "closest_wrong_row_cos": float(rep["other"].max()), "ok": ok}
                parity["ok"] = parity["ok"] and ok
        parity["image_cos_min"] = min(parity[k]["cos_min"] for k in parity if k.startswith("image_") and isinstance(parity[k], dict))
        parity["text_cos_min"] = min(parity[k]["cos_min"] for k in parity if k.startswith("text_") and isinstance(parity[k], dict))
        parity["within_north_star_1e-3"] = bool(min(parity["image_cos_min"], parity["text_cos_min"]) >= NORTH_STAR_BAR)
        if not parity["ok"]:
            failures.append(f"encode parity: {parity}")
        cpu = {"value": round(n_checked / el, 3), "unit": "samples/s", "cores": cpu_threads, "kind": "port",
               "sample": f"{n_checked} of the {B} image+text pairs of one step, fp32, torch CPU ({el:.1f} s; the same pass gates parity)"}
    del oracle

    # ---- kNN: flat fp16 index resident in HBM, top-40
    knn = None
    rows = args.knn_rows
    if rows < 0:
        rows = 125_000_000 if world == 8 else 100_000_000
    if rows > 0:
        from clip_retrieval_amd.distributed import ShardedIndex
        from clip_retrieval_amd.knn import synth_rows_device

        d, k, seed = 768, 40, 3
        batches = [int(x) for x in args.knn_batches.split(",") if x]

        def knn_leg(rows, kind, batches, scans, full_checks, cpu_leg, host_leg):
            """One flat index of `rows` x 768 fp16 generated on the device (corpus kind 0: isotropic, 2: three dominant columns), timed
            at every batch size, planted neighbours checked at full scale; full_checks adds the numpy-oracle and whole-index torch
            cross-checks, cpu_leg the CPU baseline, host_leg the reference's own call (host float32 queries in, D / I / R out)."""
            nq_max = max(batches)
            free_b = torch.cuda.mem_get_info(dev)[0]
            rows = int(min(rows, (free_b - (16 << 30)) // (d * 2)))
            # the arena is a torch tensor the index borrows, so that the full-scale cross-check below can read the same bytes
            X = torch.empty((rows, d), dtype=torch.float16, device=dev)
            ix = Mi355xIndex(d, device=local_rank, id_base=rank * rows)
            if kind == 0:
                ix.attach_device_rows(X.data_ptr(), rows)
                ix.synth_fill(rows, seed + rank)  # shard r = synth(seed + r): every shard is the same generator with its own seed
            else:
                synth_rows_device(X.data_ptr(), 0, rows, d, seed + rank, kind=kind, device=local_rank)
                ix.attach_device_rows(X.data_ptr(), rows)
            # queries = perturbed copies of rows of rank 0's shard (planted neighbours -> self-check at full scale)
            rng = np.random.default_rng(7)
            planted_local = np.sort(rng.choice(rows, nq_max, replace=False))
            q = torch.empty(nq_max, d, dtype=torch.float32, device=dev)
            if rank == 0:
                q.copy_(torch.from_numpy(perturbed_queries(ix.reconstruct_batch(planted_local))))
            if use_dist:
                dist.broadcast(q, src=0)
            sh = ShardedIndex(ix)  # (CLIPX_FORCE_GATHER=1: the all-gather + merge also at world size 1 -- tests only)
            by_batch = []
            checks = {}
            for nq in batches:
                qq = q[:nq]
                D, I = sh.search_device(qq, k)  # warm-up + correctness
                torch.cuda.synchronize()
                hit = bool((I[:, 0].cpu().numpy() == planted_local[:nq]).all())
                if not hit:
                    failures.append(f"kNN rows={rows} kind={kind} B={nq}: a planted neighbour is not the top hit")
                s0 = ix.stats()
                i8_0 = ix.i8_served()
                ix.profile(True)
                barrier()
                t1 = time.perf_counter()
                for _ in range(scans):
                    D, I = sh.search_device(qq, k)
                barrier()
                dk = max_over_ranks(time.perf_counter() - t1)
                ix.profile(False)
                nl, ms = ix.profile_get()
                s1 = ix.stats()
                scan_ms = ms / max(nl, 1)
                passes = nl / max(scans, 1)
                i8 = ix.i8_served() - i8_0 >= scans * nq  # every timed query went through the int8 first stage
                n8 = ix.i8_rows() if i8 else 0          # rows the pass read as int8 (a partial copy: the rest as fp16, same launch bracket)
                planes = ix.i8_planes() if i8 else 0
                ndom = len(ix.i8_dominant()) if i8 else 0
                pass_bytes = n8 * d + (rows - n8) * d * 2   # ALGORITHMIC bytes of one pass: every row read once, 1 or 2 bytes per element
                scan_gbs = pass_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
                qpp = nq / max(passes, 1)
                kern = ((f"knn_rq8_scan_kernel over {n8} rows (int8 first stage: tile-ordered int8 copy, 1..256 register-stationary queries per "
                         f"pass, {planes} query plane(s)" + (f" + {ndom} dominant columns as 14-bit digits on v_dot4c_i32_i8" if ndom else "")
                         + ", v_mfma_i32_16x16x64_i8; hits re-scored exactly from the fp16 rows)"
                         + (f" + knn_rq_scan_kernel over the {rows - n8} rows the copy does not hold (fp16, same hit lists)" if n8 < rows else ""))
                        if i8 else
                        "knn_rq_scan_kernel (register-stationary queries, up to 256 per pass)" if qpp > 64 else
                        "knn_scan_kernel<QB=2> (64-query wide scan + exactness proof)" if qpp > 32 else "knn_scan_kernel<QB=1> (32-query exact scan)")
                # a pass is HBM-bound below ~312 queries per pass (2.5 PF / 8 TB/s); every row carries both fractions
                # (int8: the pass multiplies all its 128 / 256 query slots whatever the batch; its matrix-pipe roof is the dense int8 one, 2 x bf16)
                slots = ((128 if nq <= 128 else 256) * planes) if i8 else qpp  # (int8: 4 or 8 waves x 32 query slots, launch_rq8_scan)
                mfma_tf = 2.0 * rows * d * slots / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
                mfma_peak = 2.0 * BF16_PEAK_TFLOPS if (i8 and n8 == rows) else BF16_PEAK_TFLOPS
                by_batch.append({"B": nq, "qps": round(scans * nq / dk, 1), "ms_per_batch": round(dk / scans * 1e3, 3),
                                 "roofline": {"bound": "hbm", "kernel": kern, "achieved": round(scan_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(scan_gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(scan_ms, 3),
                                              "algorithmic_bytes_per_launch": pass_bytes, "mfma_frac": round(mfma_tf / mfma_peak, 4)},
                                 "int8_first_stage": i8, "int8_rows": n8, "int8_query_planes": planes, "int8_dominant_columns": ndom,
                                 "passes_over_hbm": round(passes, 2), "scan_ms": round(scan_ms, 3), "scan_GBps": round(scan_gbs, 1),
                                 "hbm_frac": round(scan_gbs / HBM_PEAK_GBS, 4),
                                 "scan_mfma_tflops": round(mfma_tf, 1) if scan_ms > 0 else None,
                                 "qps_ceiling_at_8TBps": round(nq / (pass_bytes / (HBM_PEAK_GBS * 1e9)), 1),
                                 "planted_neighbour_top1": hit, "proof_served": s1[0] - s0[0], "proof_failures": s1[1] - s0[1]})
            best = max(by_batch, key=lambda r: r["qps"])
            head = best  # knn.roofline describes the kernel that produced knn.qps; every by_batch row has its own

            if single and want_parity and full_checks:
                # (a) exact id lists vs the numpy oracle on the first 1 M rows (SURVEY config 3): a second, small index filled by
                # the same generator holds exactly rows [0, 1 M) of the big one
                from oracle.knn_oracle import FlatIPOracle, topk_sets_equal

                n_small = min(1_000_000, rows)
                small = Mi355xIndex(d, device=local_rank)
                small.synth_fill(n_small, seed)
                ora = FlatIPOracle(d)
                for o in range(0, n_small, 1 << 18):
                    ora.add(small.reconstruct_batch(np.arange(o, min(o + (1 << 18), n_small), dtype=np.int64)).astype(np.float16))
                qs = q[:64].cpu().numpy()
                Ds, Is = small.search(qs, k)
                Do, Io = ora.search(qs, k)
                same = bool(np.array_equal(Is, Io))
                near = not topk_sets_equal(Is, Ds, Io, Do)
                checks["first_1M_rows_vs_numpy_oracle"] = {"rows": n_small, "queries": 64, "id_lists_identical": same,
                                                           "id_sets_equal_up_to_2e-6_ties": near,
                                                           "max_score_err": float(np.abs(Ds - Do).max())}
                if not (same or near) or np.abs(Ds - Do).max() > 1e-5:
                    failures.append(f"kNN exact-id check on the first 1M rows: {checks['first_1M_rows_vs_numpy_oracle']}")
                small.close()
                del ora
            if single and want_parity:
                # (b) the whole index against chunked torch fp32 matmul + topk (independent arithmetic on the same bytes), ALL queries:
                # at B = 256 this is the scan that produces the row's qps
                from oracle.knn_oracle import topk_sets_equal

                nqc = nq_max
                D, I = sh.search_device(q[:nqc], k)
                bs_, bi_ = None, None
                CH = 2_000_000
                for o in range(0, rows, CH):
                    sc = q[:nqc] @ X[o:o + CH].float().T
                    ts, ti = torch.topk(sc, min(k, sc.shape[1]), dim=1)
                    ti = ti + o
                    bs_ = ts if bs_ is None else torch.cat([bs_, ts], 1)
                    bi_ = ti if bi_ is None else torch.cat([bi_, ti], 1)
                    if bs_.shape[1] > 4 * k:
                        ts, sel = torch.topk(bs_, k, dim=1)
                        bs_, bi_ = ts, torch.gather(bi_, 1, sel)
                ts, sel = torch.topk(bs_, k, dim=1)
                ti = torch.gather(bi_, 1, sel)
                bad = topk_sets_equal(I.cpu().numpy(), D.cpu().numpy(), ti.cpu().numpy(), ts.cpu().numpy(), tol=1e-5)
                checks["full_index_vs_torch_matmul_topk"] = {"rows": rows, "queries": nqc, "id_sets_equal_up_to_1e-5_ties": not bad,
                                                             "max_score_err": float((D - ts).abs().max().item())}
                if bad:
                    failures.append(f"kNN full-scale cross-check (rows={rows}, kind={kind}): {bad[:3]}")
                del sc, bs_, bi_

            # the reference's own call (clip_back.py:362): host float32 queries in, host D / I / R out through knnx_search -- pinned
            # staging, H2D, scan, gather of the k stored vectors, D2H: what `index.search_and_reconstruct(query, k)` costs a service
            host = None
            if host_leg and single:
                host = []
                for nq in (1, 32):
                    qh = q[:nq].cpu().numpy()
                    ix.search_and_reconstruct(qh, k)
                    t1 = time.perf_counter()
                    for _ in range(scans):
                        Dh, Ih, Rh = ix.search_and_reconstruct(qh, k)
                    el = (time.perf_counter() - t1) / scans
                    ok = bool((Ih[:, 0] == planted_local[:nq]).all() and Rh.shape == (nq, k, d))
                    if not ok:
                        failures.append(f"kNN host search_and_reconstruct B={nq}: wrong top hit / R shape")
                    host.append({"B": nq, "ms_per_call": round(el * 1e3, 3), "qps": round(nq / el, 1), "returns": "D f32 [B,40], I i64 [B,40], R f32 [B,40,768] on the host"})

            # kNN CPU baseline: BLAS q @ X.T + running top-k over 1 M-row blocks (what faiss' IndexFlatIP does; faiss is not
            # installed) on the first 10 M rows in fp32, B in {1, 32, 256}, extrapolated linearly to the index size and labelled so
            cpu_knn = None
            if want_cpu and cpu_leg:
                n_cpu = min(10_000_000, rows)
                blk = 1_000_000
                xc = torch.empty((n_cpu, d), dtype=torch.float32)
                for o in range(0, n_cpu, blk):
                    xc[o:o + blk] = X[o:o + blk].float().cpu()
                torch.set_num_threads(cpu_threads)
                by_b = []
                for nq_c in (1, 32, 256):
                    qc = q[:min(nq_c, nq_max)].cpu()
                    t1 = time.perf_counter()
                    best_s, best_i = None, None
                    for o in range(0, n_cpu, blk):
                        ts, ti = torch.topk(qc @ xc[o:o + blk].T, k, dim=1)
                        ti = ti + o
                        if best_s is None:
                            best_s, best_i = ts, ti
                        else:
                            cs, ci = torch.cat([best_s, ts], 1), torch.cat([best_i, ti], 1)
                            best_s, sel = torch.topk(cs, k, dim=1)
                            best_i = torch.gather(ci, 1, sel)
                    el = time.perf_counter() - t1
                    by_b.append({"B": int(qc.shape[0]), "ms_per_batch_on_sample": round(el * 1e3, 1), "qps_extrapolated": round(qc.shape[0] / (el * rows / n_cpu), 3)})
                bestc = max(by_b, key=lambda r: r["qps_extrapolated"])
                cpu_knn = {"value": bestc["qps_extrapolated"], "unit": f"QPS@top-{k} over {rows} x {d} (extrapolated linearly from {n_cpu} rows)",
                           "cores": cpu_threads, "kind": "port", "by_batch": by_b,
                           "sample": f"torch CPU fp32 matmul + running top-k over 1 M-row blocks, first {n_cpu} rows of the index, B = 1 / 32 / 256 (value = the best, B = {bestc['B']})"}
                del xc
            # the counter bytes of the kernel the roofline object describes (the best row's)
            hk = head["roofline"].get("kernel", "")
            kfam = "knn_rq8_scan_kernel" if "knn_rq8_scan_kernel" in hk else ("knn_rq_scan_kernel" if "knn_rq_scan_kernel" in hk else "knn_scan_kernel")
            ktraffic, _ = pmc_traffic(kfam) if (rows == 100_000_000 and kind == 0) else (None, None)
            out = {"metric": f"QPS@top-{k}, flat IP, fp16 rows in HBM" + (" (+ int8 copy for the first-stage scan; exact results)" if best.get("int8_first_stage") else ""),
                   "corpus": {0: "isotropic unit vectors (knnx_synth_fill)", 2: "unit vectors with three dominant columns (knnx_synth_rows_device kind 2: CLIP-like anisotropy)"}[kind],
                   "qps": best["qps"], "qps_batch": best["B"], "int8_first_stage": bool(best.get("int8_first_stage")),
                   "rows_per_gpu": rows, "total_rows": rows * world, "d": d, "k": k,
                   "queries_per_scan": head["B"], "ms_per_batch": head["ms_per_batch"],
                   "planted_neighbour_top1": all(r["planted_neighbour_top1"] for r in by_batch),
                   "wide_fallbacks": sum(r["proof_failures"] for r in by_batch),
                   "roofline": {**head["roofline"], "traffic": ktraffic, "traffic_run": "separate --pmc pass" if ktraffic else None},
                   "by_batch": by_batch, "checks": checks or None, "cpu_baseline": cpu_knn, "host_search_and_reconstruct": host,
                   "exchange": ("dist.all_gather_into_tensor of B*k*12 bytes per rank on nccl (RCCL) + knnx_merge_topk_device" if (world > 1 or sh.force_gather)
                                else "none (one shard: its top-k is the answer)")}
            ix.close()
            del X, sh, ix
            torch.cuda.empty_cache()
            return out

        knn = knn_leg(rows, 0, batches, args.knn_scans, True, True, True)
        if single and args.knn_extra and rows == 100_000_000:
            # the per-GPU shard of the HEADLINE configuration (BASELINE metric: 1 B x 768 over 8 GPUs = 125 M rows = 192 GB of fp16 per GPU):
            # the int8 copy does not fit whole next to it, the library keeps a partial one (include/knnx.h)
            knn["by_shard_size"] = [knn_leg(125_000_000, 0, [1, 64, 256], 3, False, False, False)]
            # ... and the default size on a corpus with dominant columns (three columns 6 x the rest plus an offset, as CLIP embeddings
            # have), where one plain int8 plane admits 100 x more rows: the first stage keeps those columns as 14-bit digits (round 5;
            # two planes before -- DESIGN 4.3).  The isotropic corpus above is the favourable case.
            knn["anisotropic_corpus"] = knn_leg(rows, 2, [1, 64, 256], 3, False, False, False)

    # ---- BASELINE config 4's single-GPU share: the reference's whole loop around the hot path, wall to wall
    pipeline = None
    if args.pipeline and single:
        pipeline = pipeline_leg(args.model, value, local_rank, log=lambda m: sys.stderr.write(m + "\n"))
        if not (pipeline["u8"]["outputs_ok"] and pipeline["jpeg"]["outputs_ok"]):
            failures.append(f"pipeline leg: written files / shapes / dtypes / row counts are not the writer's: {pipeline}")

    # ---- BASELINE config 5 (opt-in: minutes): one GPU's IVF-Flat shard at its stated size, built on the device, served
    ivf = None
    if args.ivf and single:
        import importlib.util

        enc.close()
        torch.cuda.empty_cache()
        spec = importlib.util.spec_from_file_location("config5", os.path.join(ROOT, "tools", "config5.py"))
        c5 = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(c5)
        ivf = c5.run(rows=args.ivf_rows, device=local_rank, threads=(1, 64), seconds=1.0, log=lambda m: sys.stderr.write(m + "\n"))
        rec = [r["recall_at_40_vs_exact_whole_shard"] for r in ivf["by_nprobe"]]
        if any(b < a - 1e-3 for a, b in zip(rec, rec[1:])) or ivf["exact"]["planted_top1"] < 0.999:
            failures.append(f"IVF: recall must not fall with nprobe and the exact scan must find every planted row: {rec}, {ivf['exact']}")

    if failures:
        sys.stderr.write("bench.py: parity gate FAILED, no result line is printed:\n  " + "\n  ".join(failures) + "\n")
        if use_dist:
            dist.destroy_process_group()
        sys.exit(1)
    if rank == 0:
        line = {
            "metric": f"images/sec embedded ({args.model} bs={B}; each sample = image + caption through both towers)",
            "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "dtype_detail": "bf16 MFMA operands (IEEE fp16 where the operand is the fp16 residual stream), f32 accumulate, f32 softmax / LayerNorm statistics", "data": "synthetic",
            "config": {"workload": f"{args.model} image+text encode, bs={B} per GPU, f32 NCHW pixels + int32 tokens resident in HBM "
                                   "(BASELINE.json configs[1]); random-init weights"
                                   + (f" | kNN: flat IP top-40 over {knn['rows_per_gpu']} x 768 fp16 rows per GPU resident in HBM, query batches "
                                      f"{args.knn_batches} (configs[2]; reported under `knn`)" if knn else "")
                                   + (" | tar shards -> reader -> runner -> writer (configs[3]'s per-GPU share) under `pipeline`" if pipeline else "")
                                   + (" | IVF-Flat shard of configs[4] under `ivf`" if ivf else ""),
                       "global_batch": B * world, "parallelism": f"replicas x{world} (no collective)",
                       "process_group": ("nccl (RCCL), world %d" % world) if use_dist else None,
                       "knn_exchange": knn.get("exchange") if knn else None},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "knn": knn, "pipeline": pipeline, "ivf": ivf, **extras,
        }
        # The whole headline, driver-visible: the driver keeps the standard keys, the NAMES of the extra ones, the
        # last ~10 KB of stdout and the last ~2 KB of stderr.  So: scalars whose names carry the value class at the top level, a
        # compact `headline` object as the LAST key of the line, and the same numbers as <= 12 short stderr lines that end the run.
        scal, summary = headline_numbers(line)
        line.update(scal)
        line["headline"] = scal
        sys.stdout.flush()
        print(json.dumps(line))
        sys.stdout.flush()
        sys.stderr.write("\n".join(summary) + "\n")
        sys.stderr.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

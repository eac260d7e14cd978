"""One hardware execution of the code path N > 1 ranks take, on the ONE GPU of a test box.

Every box this repository is built and judged on before the driver's 8-GPU run has a single MI355X, so the multi-GPU path is
correct by construction + world-2 `gloo` tests on the CPU (tests/test_distributed_gloo.py).  What CAN run here is everything but the
second device: `python -m torch.distributed.run --nproc-per-node 1` around `bench.py` (init_process_group("nccl") = RCCL, the
barrier, the MAX all-reduce of the timings, the broadcast of the planted queries) and the exchange step itself --
`dist.all_gather_into_tensor` on RCCL followed by `knnx_merge_topk_device` -- forced at world size 1
(`ShardedIndex(force_gather=True)` / CLIPX_FORCE_GATHER=1).  Reference anchor of the call this serves: clip_back.py:362."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(args, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_under_the_launcher_at_world_size_1():
    """`bench.py --gpus 1` exactly as the driver launches N > 1 (torch.distributed.run, one rank per GPU): the process group is
    "nccl", the planted queries are broadcast, the timings go through the MAX all-reduce, and -- CLIPX_FORCE_GATHER=1 -- every kNN
    batch goes through all_gather_into_tensor + the merge kernel.  The line must parse, carry the contract's keys and pass its own
    parity gates (a failed gate prints no line)."""
    r = _launch(["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--knn-rows", "2200000", "--knn-scans", "2", "--knn-batches", "1,64,256",
                 "--no-ivf", "--no-knn-extra", "--no-pipeline", "--parity-rows", "8", "--cpu-seconds", "0", "--no-host-path"],
                {"CLIPX_FORCE_GATHER": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "headline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0
    assert line["config"]["process_group"] == "nccl (RCCL), world 1"
    assert "all_gather_into_tensor" in line["config"]["knn_exchange"] and "knnx_merge_topk_device" in line["config"]["knn_exchange"]
    assert line["parity"]["ok"] and line["parity"]["checked"] == 256
    knn = line["knn"]
    assert knn["planted_neighbour_top1"] and knn["rows_per_gpu"] == 2_200_000
    assert knn["checks"]["full_index_vs_torch_matmul_topk"]["id_sets_equal_up_to_1e-5_ties"]
    # the headline is readable from the top-level scalars alone and from the last stderr lines
    assert line["knn_qps_b256"] == [b for b in knn["by_batch"] if b["B"] == 256][0]["qps"] and line["encode_samples_per_s"] == line["value"]
    tail = [ln for ln in r.stderr.splitlines() if ln.startswith("HEADLINE")]
    assert 3 <= len(tail) <= 12 and sum(len(ln) + 1 for ln in tail) < 2000, tail
    assert tail[0].startswith("HEADLINE encode") and any("kNN checks" in ln for ln in tail)


_GATHER_SCRIPT = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.getcwd())
    from clip_retrieval_amd.distributed import ShardedIndex
    from clip_retrieval_amd.knn import Mi355xIndex
    from oracle.knn_oracle import FlatIPOracle

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    for d, n in ((768, 60_000), (640, 20_000)):     # 640: RN50x4's width, padded to 768 inside the index (ADVICE r5)
        rng = np.random.default_rng(d)
        x = rng.standard_normal((n, d)).astype(np.float32)
        x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float16)
        x[n - 1] = x[0]                               # an exact tie: ids ascending
        q = x[[0, 5, 77, n - 2]].astype(np.float32) + 0.01 * rng.standard_normal((4, d)).astype(np.float32)
        q[0] = x[0].astype(np.float32)
        ix, o = Mi355xIndex(d, id_base=1000), FlatIPOracle(d)
        ix.add(x)
        o.add(x)
        Do, Io = o.search(q, 40)
        Io = np.where(Io >= 0, Io + 1000, -1)
        plain = ShardedIndex(ix)
        forced = ShardedIndex(ix, force_gather=True)
        assert not plain.force_gather and forced.force_gather and forced.world == 1
        calls = {"n": 0}
        real = dist.all_gather_into_tensor
        def counted(*a, **k):
            calls["n"] += 1
            return real(*a, **k)
        dist.all_gather_into_tensor = counted
        D0, I0 = plain.search(q, 40)                  # one shard, no exchange
        assert calls["n"] == 0
        D1, I1 = forced.search(q, 40)                 # host queries in: upload (padded), scan, all-gather on RCCL, merge kernel
        assert calls["n"] == 1
        qd = torch.from_numpy(forced._pad_queries(q)).cuda()
        D2, I2 = forced.search_device(qd, 40)         # device tensors end to end
        assert calls["n"] == 2 and D2.is_cuda and I2.is_cuda
        D3, I3 = forced.search(q, 100)                # k > 64: local.search on the host + gather of the host results + merge kernel
        assert calls["n"] == 3
        dist.all_gather_into_tensor = real
        for D, I in ((D0, I0), (D1, I1), (D2.cpu().numpy(), I2.cpu().numpy())):
            assert np.array_equal(I, Io), (d, I[:, :5], Io[:, :5])
            assert np.abs(D - Do).max() <= 1e-5
        Do3, Io3 = o.search(q, 100)
        assert np.array_equal(I3, np.where(Io3 >= 0, Io3 + 1000, -1)) and np.abs(D3 - Do3).max() <= 1e-5
        assert forced.ntotal == n
        ix.close()
    t = torch.tensor([3.0], device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    print("GATHER_OK", flush=True)
""")


def test_sharded_index_exchange_on_rccl_at_world_size_1(tmp_path):
    """`ShardedIndex.search` / `search_device` with the exchange forced: `dist.all_gather_into_tensor` on the nccl backend (RCCL) and
    `knnx_merge_topk_device` run on the GPU and return the flat oracle's lists (ids offset by id_base, ties in id order), for a
    768- and a 640-wide index (the latter through the padded query path)."""
    script = tmp_path / "gather_world1.py"
    script.write_text(_GATHER_SCRIPT)
    r = _launch([str(script)], timeout=600)
    assert r.returncode == 0 and "GATHER_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])

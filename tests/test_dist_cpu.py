"""World-size-2 gloo test of the multi-GPU glue (runs on CPU): the record sharding rule equals
torch's DistributedSampler, shards cover the batch, and the bookkeeping reductions give the
single-process result.  The per-record compute in this test is the CPU oracle (the HIP path
needs a GPU); what is under test is ecg-byte_amd/parallel.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import load_tokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_records, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ecg_byte_amd import parallel, synth
        from oracle import oracle as O
        _, merges, pc = load_tokenizer("c1")
        trie = O.Trie(merges)
        idx = parallel.shard_indices(n_records, rank, world)
        x = synth.synth_ecg(n_records, 1000, seed=5)
        counts = torch.tensor([trie.quantize_encode(x[i], pc["percentile_1"], pc["percentile_99"]).size for i in idx],
                              dtype=torch.int32)
        wall, total = parallel.reduce_step_stats(0.5 + rank, int(counts.sum()), torch.device("cpu"))
        allc = parallel.gather_counts(counts)
        ret[rank] = (idx, wall, total, allc.tolist())
    finally:
        dist.destroy_process_group()


def test_shard_indices_equals_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    from ecg_byte_amd import parallel
    for n in (0, 1, 5, 8, 33):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                want = list(DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=False)) if n else []
                assert parallel.shard_indices(n, rank, world) == want, (n, world, rank)
                if n:   # the training sampler of main.py:239-243: shuffle=True, seed, set_epoch
                    ds = DistributedSampler(range(n), num_replicas=world, rank=rank, shuffle=True, seed=7)
                    ds.set_epoch(3)
                    assert parallel.shard_indices(n, rank, world, shuffle=True, seed=7, epoch=3) == list(ds), (n, world, rank)


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    world, n_records = 2, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_records, ret), nprocs=world, join=True)
    from ecg_byte_amd import synth
    from oracle import oracle as O
    _, merges, pc = load_tokenizer("c1")
    trie = O.Trie(merges)
    x = synth.synth_ecg(n_records, 1000, seed=5)
    single = [trie.quantize_encode(x[i], pc["percentile_1"], pc["percentile_99"]).size for i in range(n_records)]
    covered = sorted(set(ret[0][0]) | set(ret[1][0]))
    assert covered == list(range(n_records))
    for rank in range(world):
        idx, wall, total, allc = ret[rank]
        assert wall == 1.5                                            # MAX over ranks
        assert total == sum(single[i] for i in ret[0][0]) + sum(single[i] for i in ret[1][0])
        assert allc == [single[i] for i in ret[0][0]] + [single[i] for i in ret[1][0]]


def _grad_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ecg_byte_amd import parallel
        torch.manual_seed(0)
        layers = [torch.nn.Parameter(torch.randn(4, 3)) for _ in range(3)]
        x = torch.randn(6, 3)
        shard = x[rank::world]                       # this rank's records
        sync = parallel.GradAllReduce()
        for i in reversed(range(3)):                 # "backward": layer grads become final one by one
            layers[i].grad = (shard @ layers[i].detach().T).sum(0)[:, None].expand(4, 3).clone() / shard.shape[0]
            sync.on_grads_ready([layers[i]])
        sync.finish()
        ret[rank] = [p.grad.clone() for p in layers]
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_all_reduce_world_size_2_gloo():
    """GradAllReduce (the DDP replacement of HipCausalLM) averages per-layer gradients across ranks."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grad_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    torch.manual_seed(0)
    layers = [torch.randn(4, 3) for _ in range(3)]
    x = torch.randn(6, 3)
    for i in range(3):
        per_rank = [(x[r::world] @ layers[i].T).sum(0)[:, None].expand(4, 3) / x[r::world].shape[0] for r in range(world)]
        want = sum(per_rank) / world
        for r in range(world):
            assert torch.allclose(ret[r][i], want, atol=1e-6)


def _flat_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ecg_byte_amd import parallel
        g = torch.Generator().manual_seed(100 + rank)
        sizes = [1000, 300, 5000, 64, 2000]                      # "layers", in the order backward finishes them
        flat = torch.randn(sum(sizes), generator=g)
        mine = flat.clone()
        sync = parallel.GradAllReduce(bucket_bytes=4 * 1200)     # fp32 here: a bucket closes at >= 1200 elements
        lo = 0
        for n in sizes:
            sync.on_flat_ready(flat, lo, lo + n)
            lo += n
        sent_before_finish = sync.collectives
        sync.finish()
        ret[rank] = (mine, flat.clone(), sent_before_finish, sync.collectives)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_gradient_buckets_world_size_2_gloo():
    """The bucketed exchange of HipCausalLM's flat gradient buffer: adjacent ready ranges merge until a bucket is full
    (DDP's 25 MB rule, ecg_byte/main.py:165, scaled down here), every element ends as the mean over the ranks, and the number of
    collectives is the number of buckets, not the number of tensors."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_flat_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    want = (ret[0][0] + ret[1][0]) / 2
    for r in range(world):
        assert torch.allclose(ret[r][1], want, atol=1e-6)
        # 1000 + 300 >= 1200 -> bucket 1; 5000 -> bucket 2; 64 + 2000 -> bucket 3 (closed by finish... it reaches 1200 with 2000)
        assert ret[r][2] == 3 and ret[r][3] == 3


def _shard_worker(rank, world, port, cases, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ecg_byte_amd.trainer import bpe_train_sharded
        from oracle.sharded_trainer import CpuShard
        out = []
        for text, cuts, nm in cases:
            lo, hi = ([0] + cuts + [len(text)])[rank: rank + 2]
            out.append(bpe_train_sharded(CpuShard(text[lo:hi], nm), nm))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_bpe_trainer_protocol_equals_the_single_process_trainer(world):
    """the corpus split into contiguous slices over ranks, one all-gather of 8-word summaries and one
    all-reduce of the 6 x V delta slab per merge (ecg_byte_amd/trainer.py::bpe_train_sharded, the loop a multi-GPU run executes over
    RCCL) -- here over gloo with a CPU model of one rank's kernels (oracle/sharded_trainer.py).  Merges and the concatenation of the
    ranks' ids must equal the oracle trainer's on the whole text (lib.rs:58-125): random texts over small alphabets (many ties, many
    self-merges), cuts inside runs, empty and one-symbol slices, runs longer than a slice."""
    rng = np.random.default_rng(77 + world)
    cases = []
    for trial in range(24):
        k = int(rng.choice([1, 2, 3, 5]))
        n = int(rng.integers(0, 260))
        text = bytes(rng.integers(97, 97 + k, size=n).astype(np.uint8))
        if trial % 5 == 0:                                     # long runs that span the cuts
            text = b"a" * int(rng.integers(0, 90)) + text[: n // 3] + b"b" * int(rng.integers(0, 70)) + b"a" * int(rng.integers(0, 50))
        cuts = sorted(int(c) for c in rng.integers(0, len(text) + 1, size=world - 1))
        if trial % 7 == 0:
            cuts = [0] * (world - 1)                            # empty leading slices
        cases.append((text, cuts, int(rng.integers(0, 50))))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, _free_port(), cases, ret), nprocs=world, join=True)
    from oracle import oracle as O
    for c, (text, cuts, nm) in enumerate(cases):
        want_ids, _, want_merges = O.byte_pair_encoding(text, nm, fast=False)
        got_ids = sum((ret[r][c][0] for r in range(world)), [])
        for r in range(world):
            assert O.pairs_to_vocab_merges([tuple(p) for p in ret[r][c][1]])[1] == want_merges, (c, r, text, cuts, nm)
        assert got_ids == want_ids, (c, text, cuts, nm)



@pytest.mark.timeout(300)
def test_main_dis_without_rank_spawns_one_child_per_listed_gpu(tmp_path):
    """The reference's own launch (ecg_byte/main.py:356-360, scripts/train_model.sh:6-18): `main --dis --gpus 0,1,2 --ports P` with no RANK
    in the environment must mp.spawn one child per listed GPU -- rank r on GPU gpus[r], rendezvous on port P -- from a parent that has
    made no GPU call.  The children run the gloo hook (ECGB_MAIN_BACKEND / ECGB_MAIN_PROBE): rendezvous + one all-reduce, no model."""
    import json
    import subprocess
    import sys
    port = _free_port()
    probe = str(tmp_path / "probe")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ECGB_MAIN_BACKEND="gloo", ECGB_MAIN_PROBE=probe, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "ecg_byte_amd.main", "--dis", "--gpus", "2,0,3", "--ports", str(port), "--model", "none",
           "--tokenizer_check", "none", "--peft", "--batch_size", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    recs = [json.load(open(f"{probe}_{k}.json")) for k in range(3)]
    assert [x["rank"] for x in recs] == [0, 1, 2] and all(x["world"] == 3 for x in recs)
    assert [x["local_rank"] for x in recs] == [2, 0, 3]                    # local_rank = gpu_ids[rank], main.py:68-69
    assert all(x["master_port"] == str(port) for x in recs)
    assert all(x["sum"] == 6.0 for x in recs)                             # 1 + 2 + 3: the three children really met
    assert len({x["pid"] for x in recs}) == 3 and len({x["ppid"] for x in recs}) == 1


@pytest.mark.timeout(300)
def test_main_dis_under_torchrun_environment_still_initialises_from_it(tmp_path):
    """The torchrun path: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* already in the environment -> no spawn, the process IS the rank."""
    import json
    import subprocess
    import sys
    port = _free_port()
    probe = str(tmp_path / "probe")
    procs = []
    for rank in range(2):
        env = dict(os.environ, ECGB_MAIN_BACKEND="gloo", ECGB_MAIN_PROBE=probe, PYTHONPATH=ROOT, RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-m", "ecg_byte_amd.main", "--dis", "--model", "none", "--tokenizer_check", "none"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(tmp_path)))
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
    recs = [json.load(open(f"{probe}_{k}.json")) for k in range(2)]
    assert [x["local_rank"] for x in recs] == [0, 1] and all(x["sum"] == 3.0 and x["world"] == 2 for x in recs)
    assert len({x["ppid"] for x in recs}) == 1 and recs[0]["ppid"] == os.getpid()       # no intermediate spawner


def _lora_bucket_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ecg_byte_amd import parallel
        # LoRA r16 on Llama-3.2-1B widths, the stacked layout of decoder.LoraSite: per layer A [64, in] + B [out, 64] for the four sites
        # (qkv 2048 -> 3072, o 2048 -> 2048, gate|up 2048 -> 16384, down 8192 -> 2048), every slice padded to 128 elements; 16 layers, last first
        per_layer = sum((64 * i + 127) // 128 * 128 + (o * 64 + 127) // 128 * 128 for i, o in ((2048, 3072), (2048, 2048), (2048, 16384), (8192, 2048)))
        layers = 16
        g = torch.Generator().manual_seed(200 + rank)
        flat = torch.randn(layers * per_layer, generator=g).to(torch.bfloat16)
        mine = flat.clone()
        sync = parallel.GradAllReduce()                                  # DDP's 25 MB buckets
        for i in range(layers):
            sync.on_flat_ready(flat, i * per_layer, (i + 1) * per_layer)
        before = sync.collectives
        sync.finish()
        # the same exchange as reduce-scatter + all-gather for buckets of >= 1 MB (the gloo stand-in exercises cut, remainder and order)
        flat2 = mine.clone()[: layers * per_layer - 3]                   # a length the world size does not divide
        sync2 = parallel.GradAllReduce(bucket_bytes=8 << 20, rsag_mb=1)
        step = 3 * per_layer
        for lo in range(0, flat2.numel(), step):
            sync2.on_flat_ready(flat2, lo, min(lo + step, flat2.numel()))
        sync2.finish()
        # the sparse rows of an embedding gradient
        ids = torch.arange(rank, 40 + 3 * rank, 2 + rank)
        rows = torch.full((ids.numel(), 8), float(rank + 1))
        parts = parallel.GradAllReduce(sparse_embedding=True).exchange_rows(ids, rows)
        ret[rank] = (mine, flat.clone(), before, sync.collectives, per_layer, flat2.clone(), sync2.collectives, sync2.rsag_buckets,
                     [(a.tolist(), b[:, 0].tolist()) for a, b in parts])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_lora_buckets_rsag_and_sparse_rows_world_size_4_gloo():
    """World size 4 (the reference's own launch, scripts/train_model.sh: four GPUs) over the LoRA adapters' flat gradient buffer: layers of
    4.6 MiB merge into 25 MB buckets in backward order (4 collectives for 16 layers, not 128 tensors), every element ends as the mean; the
    reduce-scatter + all-gather form gives the same means; exchange_rows hands every rank every rank's (ids, rows) in rank order."""
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_lora_bucket_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    per_layer = ret[0][4]
    assert 4.0 < per_layer * 2 / 2 ** 20 < 6.0                          # MiB of bf16 per layer: several layers per 25 MB bucket
    want = sum(ret[r][0].float() for r in range(world)) / world
    per_bucket = -(-(25 << 20) // (per_layer * 2))                        # layers until a bucket holds 25 MB
    n_buckets = -(-16 // per_bucket)
    for r in range(world):
        assert torch.allclose(ret[r][1].float(), want, atol=2e-2, rtol=2e-2)
        assert ret[r][3] == n_buckets and ret[r][2] in (n_buckets - 1, n_buckets)      # the last bucket may only close in finish()
        want2 = want[: ret[r][5].numel()]
        assert torch.allclose(ret[r][5].float(), want2, atol=2e-2, rtol=2e-2)
        assert ret[r][7] >= 1 and ret[r][6] > ret[r][7]
        assert ret[r][8] == ret[0][8]
        for k, (ids, vals) in enumerate(ret[r][8]):
            assert ids == list(range(k, 40 + 3 * k, 2 + k)) and all(v == k + 1 for v in vals)

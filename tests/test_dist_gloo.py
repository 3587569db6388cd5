"""CPU, world_size 2 and 3 (gloo): batch sharding + the path's single all_gather give world-size independent results."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(local_shape, kw, seed, first):
    # a stand-in "denoising loop" on the CPU: per-sample arithmetic only (samples never interact), noise from
    # the counter-based generator exactly as the HIP sampler keys it: (seed, first + row, timestep | x_T stream, element)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import philox_ref as P
    b, per = local_shape[0], int(torch.tensor(local_shape[1:]).prod())
    x = torch.from_numpy(P.normal(per, b, first, seed, P.STREAM_XT)).view(local_shape)
    for t in (2, 1, 0):
        n = torch.from_numpy(P.normal(per, b, first, seed, t)).view(local_shape)
        x = 0.9 * x + 0.1 * n * kw["length"].view(-1, 1, 1).float() + kw["xf_proj"].sum(-1).view(-1, 1, 1)
    return x


def _worker(rank, world, port, B, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dmod = pkg("dist")
    kw = {"length": torch.arange(B) + 3, "xf_proj": torch.arange(B * 2, dtype=torch.float32).view(B, 2),
          "text": [f"t{i}" for i in range(B)], "scalar": 5}
    y = dmod.sample_sharded(_fake_sampler, (B, 4, 3), kw, seed=7)
    if rank == 0:
        torch.save(y, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,world", [(4, 2), (5, 2), (5, 3), (2, 3)])
def test_sharded_sampling_is_world_size_invariant(tmp_path, B, world):
    """(5, 3): ragged shards 2 | 2 | 1; (2, 3): the last rank holds no sample and still takes part in the gather."""
    dmod = pkg("dist")
    kw = {"length": torch.arange(B) + 3, "xf_proj": torch.arange(B * 2, dtype=torch.float32).view(B, 2),
          "text": [f"t{i}" for i in range(B)], "scalar": 5}
    single = dmod.sample_sharded(_fake_sampler, (B, 4, 3), kw, seed=7)
    out = str(tmp_path / "y.pt")
    mp.spawn(_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert torch.equal(torch.load(out), single)


def test_shard_ranges_cover_batch():
    dmod = pkg("dist")
    for B in (1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            r = [dmod.shard_range(B, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == B and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    sk = dmod.shard_kwargs({"a": torch.arange(6), "t": list("abcdef"), "s": 3}, 2, 5)
    assert sk["a"].tolist() == [2, 3, 4] and sk["t"] == ["c", "d", "e"] and sk["s"] == 3


def _fake_bucket(k, idx, T):
    # per-sample values that depend only on the sample index and the batch's frame count
    return (idx.float().view(-1, 1, 1) + 1) * torch.ones(len(idx), T, 3) + 0.001 * T


def _plan_worker(rank, world, port, lens, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dmod = pkg("dist")
    plan = dmod.plan_buckets(lens, 3, 16, 4)
    ran = []
    y = dmod.run_plan(plan, lambda k, idx, T: (ran.append(k), _fake_bucket(k, idx, T))[1], len(lens), 16, 3, "cpu")
    assert ran == [k for k in range(len(plan)) if k % world == rank]
    if rank == 1:
        torch.save(y, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_plan_and_two_rank_exchange(tmp_path):
    dmod = pkg("dist")
    lens = [16, 5, 9, 40, 8, 12, 4, 7, 11, 3]
    plan = dmod.plan_buckets(lens, 3, 16, 4)
    seen = torch.cat([idx for idx, _ in plan]).tolist()
    assert sorted(seen) == list(range(len(lens))) and all(len(idx) <= 3 for idx, _ in plan)
    clamped = [min(l, 16) for l in lens]
    for idx, T in plan:
        assert T % 4 == 0 and T <= 16 and T >= max(clamped[i] for i in idx.tolist()) > T - 4
    assert [T for _, T in plan] == sorted((T for _, T in plan), reverse=True)
    serial = [(torch.arange(lo, min(lo + 3, len(lens))), max(clamped[lo:lo + 3])) for lo in range(0, len(lens), 3)]
    assert dmod.padded_frames(plan) <= dmod.padded_frames(serial)
    single = dmod.run_plan(plan, _fake_bucket, len(lens), 16, 3, "cpu")
    for i, y in enumerate(single):
        T = next(T for idx, T in plan if i in idx.tolist())
        assert y.shape == (T, 3) and torch.all(y == (i + 1) + 0.001 * T)
    out = str(tmp_path / "plan.pt")
    mp.spawn(_plan_worker, args=(2, _free_port(), lens, out), nprocs=2, join=True)
    two = torch.load(out)
    assert len(two) == len(single) and all(torch.equal(a, b) for a, b in zip(two, single))

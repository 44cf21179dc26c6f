"""Data-parallel semantics on CPU with gloo, world_size 2 (the N > 1 path of SURVEY.md section 8e):
sharded mean-CE gradients, summed by the product's GradSync and scaled by 1/world, must equal the
single-process gradient on the global batch.  Gradients come from the CPU oracle (tests may use it);
the sharding / all-reduce / scaling logic under test is rpo_amd.dist."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd import synth
    from rpo_amd.config import vit_b16
    from rpo_amd.dist import GradSync
    sync = GradSync(backend="gloo")
    assert sync.world_size == world and sync.rank == rank and sync.enabled
    cfg = vit_b16(layers_v=1, layers_t=1, K=4)
    toks = synth.oxford_pets_base_tokens()
    # one weight generation per node (bench.py's N > 1 path): local rank 0 writes, the other rank memory-maps
    rows = np.unique(toks).tolist() + [49407]
    sd = synth.clip_state_dict_shared(cfg, 0, rows, os.path.join(out_dir, "weights.npy"), writer=sync.local_writer,
                                      barrier=sync.barrier)
    assert sync.local_writer == (rank == 0)
    assert synth.state_dict_checksum(sd) == synth.state_dict_checksum(synth.clip_state_dict(cfg, seed=0, token_rows=rows))
    tp, ip = synth.prompts(cfg, sd, seed=7)
    G = 4
    image, label = synth.images(cfg, G), synth.labels(cfg, G)
    first, count = sync.shard(G)
    m = OracleRPO(sd, toks, cfg.K, cfg.patch)
    # rank 1 starts from perturbed prompts: broadcast must make them identical
    flat = torch.cat([torch.from_numpy(tp).reshape(-1), torch.from_numpy(ip).reshape(-1)])
    if rank == 1:
        flat = flat + 1.0
    sync.broadcast(flat, src=0)
    m.set_prompts(flat[:tp.size].reshape(tp.shape).numpy(), flat[tp.size:].reshape(ip.shape).numpy())
    out, gt, gi = m.loss_and_grads(image[first:first + count], label[first:first + count])
    grads = torch.cat([gt.reshape(-1), gi.reshape(-1)])
    sync.all_reduce_sum(grads)
    grads *= sync.grad_scale
    mx = sync.max_over_ranks(float(rank), torch.device("cpu"))
    assert mx == world - 1
    assert sync.gather_floats(10.0 + rank, torch.device("cpu")) == [10.0 + r for r in range(world)]
    assert sync.broadcast_object({"dir": "x"} if rank == 0 else None) == {"dir": "x"}
    # host pinning: without a GPU the even-share rule applies -- disjoint CPU sets per local rank, threads capped
    before = sorted(os.sched_getaffinity(0))
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    info = sync.pin_host()
    if len(before) >= world:
        mine = sorted(os.sched_getaffinity(0))
        assert info["pinned"] and len(mine) == len(before) // world and mine[0] == before[rank * (len(before) // world)]
        assert torch.get_num_threads() <= len(mine)
    os.sched_setaffinity(0, before)
    sync.barrier()
    np.save(os.path.join(out_dir, f"g{rank}.npy"), grads.numpy())
    sync.close()


def test_two_rank_gradient_equals_global_batch(tmp_path):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert np.array_equal(g0, g1), "all ranks must hold identical reduced gradients"
    from oracle.rpo_oracle import OracleRPO
    from rpo_amd import synth
    from rpo_amd.config import vit_b16
    cfg = vit_b16(layers_v=1, layers_t=1, K=4)
    toks = synth.oxford_pets_base_tokens()
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    tp, ip = synth.prompts(cfg, sd, seed=7)
    m = OracleRPO(sd, toks, cfg.K, cfg.patch)
    m.set_prompts(tp, ip)
    _, gt, gi = m.loss_and_grads(synth.images(cfg, 4), synth.labels(cfg, 4))
    ref = np.concatenate([gt.numpy().reshape(-1), gi.numpy().reshape(-1)])
    assert np.abs(g0 - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_pin_host_can_be_switched_off(monkeypatch):
    from rpo_amd.dist import GradSync
    monkeypatch.setenv("RPO_NO_AFFINITY", "1")
    before = os.sched_getaffinity(0)
    assert GradSync(init=False).pin_host() == {"pinned": False} and os.sched_getaffinity(0) == before


def test_shard_rejects_uneven_batches(monkeypatch):
    from rpo_amd.dist import GradSync
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "2")
    s = GradSync(init=False)
    assert s.shard(256) == (128, 64) and s.grad_scale == 0.25
    with pytest.raises(ValueError):
        s.shard(30)

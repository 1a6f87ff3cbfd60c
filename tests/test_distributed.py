"""Data-parallel path on CPU: 2 processes over gloo (world_size 2) drive the same engine code
the GPU ranks run over RCCL -- kernels under the SIMT interpreter, all-reduce through
torch.distributed.  Mirrors the reference's no-cluster distributed tests
(tests/test_runtime/test_no_sync_equiv.py:132-172: exactly acc-1 no_sync entries and equal
weights; tests/test_runtime/test_parallel_topology.py:47-77: shard indices == DistributedSampler).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(golden_dir, emu_path):
    from specforge_amd import _lib

    _lib._inject_library_for_tests(emu_path)
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
    from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend

    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"])
    model = LlamaForCausalLMEagle3(cfg)
    sd = dict(blob["params"])
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = blob["embed"], blob["t2d"], blob["d2t"]
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=2).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(blob["head_w"]))
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-2, total_steps=100, warmup_ratio=0.0))
    backend.prepare_model(eagle)

    def batch(seed):
        from oracle import eagle3_oracle as O  # only its synthetic-data helper

        oc = O.DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"],
                           num_key_value_heads=c["nkv"], vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"],
                           target_hidden_size=c["Ht"])
        b = O.make_batch(oc, 1, 16, seed=seed, dtype=torch.bfloat16)
        b["loss_mask"][:, :3 + seed % 4] = 0      # a prompt prefix without loss, a different one per batch: every rank runs the loss-row
        #                                           compaction of the lm_head part with its OWN row counts (engine.compact_loss_rows)
        return TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                               hidden_state=b["hidden_state"], target=b["target"]), {"target_repr": "hidden_state"})

    return eagle, strat, backend, batch


def _worker(rank, world, port, golden_dir, emu_path, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    os.environ["SFEMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eagle, strat, backend, batch = _setup(golden_dir, emu_path)
    # optimizer window 1: one micro-step; window 2: two micro-steps (first one without the collective)
    backend.backward(strat.forward_loss(batch(10 + rank)).loss, is_boundary=True)
    backend.synchronize_gradients()  # the bucket all-reduces are asynchronous
    g_after_first = eagle.engine.flat.grad.float().clone()
    backend.step()
    for i, boundary in enumerate((False, True)):
        backend.backward(strat.forward_loss(batch(20 + 2 * rank + i)).loss / 2, is_boundary=boundary)
    backend.step()
    torch.save(dict(params=eagle.engine.flat.data.clone(), g1=g_after_first, no_sync=backend.no_sync_backwards),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gloo_dp_matches_single_process_accumulation(golden_dir, emu_lib_path, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), golden_dir, emu_lib_path, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    # replicas stay bit-identical (same reduced gradient, same deterministic optimizer kernel)
    assert torch.equal(r0["params"], r1["params"])
    assert torch.equal(r0["g1"], r1["g1"])
    assert r0["no_sync"] == 1 and r1["no_sync"] == 1   # exactly acc-1 skipped collectives
    # single process, same data: DP mean over ranks == accumulation of loss/world over the ranks' batches
    eagle, strat, backend, batch = _setup(golden_dir, emu_lib_path)
    for r in range(world):
        backend.backward(strat.forward_loss(batch(10 + r)).loss / world, is_boundary=(r == world - 1))
    g_single = eagle.engine.flat.grad.float().clone()
    torch.testing.assert_close(r0["g1"] / world, g_single, rtol=2e-2, atol=2e-2 * float(g_single.abs().max()))
    backend.step()
    for r in range(world):
        for i in range(2):
            backend.backward(strat.forward_loss(batch(20 + 2 * r + i)).loss / (2 * world), is_boundary=True)
    backend.step()
    from specforge_amd import _lib

    _lib._inject_library_for_tests(None)
    # after two optimizer steps with lr 1e-2 (Adam: |update| <= lr per step) the weights agree to a few bf16 ulps
    diff = (r0["params"].float() - eagle.engine.flat.data.float()).abs().max()
    assert float(diff) <= 2.5e-2, float(diff)


def test_shard_indices_match_reference_and_distributed_sampler(golden_dir):
    """bit-exact vs vectors from the reference's _distributed_sampler_indices (launch.py:219-239)
    and vs torch's DistributedSampler (reference test_parallel_topology.py:47-77)."""
    from torch.utils.data import DistributedSampler

    from specforge_amd.training import distributed_sampler_indices

    for case in torch.load(os.path.join(golden_dir, "sampler_indices.pt"), weights_only=False):
        got = distributed_sampler_indices(case["size"], dp_rank=case["dp_rank"], dp_size=case["dp_size"], seed=case["seed"],
                                          epoch=case["epoch"])
        assert got == case["idx"]
        ds = DistributedSampler(range(case["size"]), num_replicas=case["dp_size"], rank=case["dp_rank"], shuffle=True,
                                seed=case["seed"])
        ds.set_epoch(case["epoch"])
        assert got == list(iter(ds))
    assert distributed_sampler_indices(0, dp_rank=0, dp_size=2, seed=0, epoch=0) == []

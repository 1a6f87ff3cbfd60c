"""The RCCL code path of the data-parallel backend on ONE GPU (`-m gpu`): `backend="nccl"` (= RCCL on ROCm) at world size
1 with `force_collectives=True`, so that the per-bucket async all-reduces launched from inside the weight-gradient phase,
the `no_sync` skip on non-boundary micro-steps and the single-collective variant all run on the product library under the
driver's green bar.  At world size 1 a SUM all-reduce is the identity, so every variant must reproduce the plain
single-process run bit for bit (reference semantics: DDP bucketed all-reduce + no_sync,
specforge/training/backend.py:233-253,310-320).  The 2-rank protocol itself is covered on CPU by tests/test_distributed.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(golden_dir, dev, **backend_kw):
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
    from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend, TrainerCore

    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=c["fc_norm"],
                      rope_scaling=c["rope_scaling"])
    model = LlamaForCausalLMEagle3(cfg, device=dev)
    sd = dict(blob["params"])
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = blob["embed"], blob["t2d"], blob["d2t"]
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=c["ttt"]).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(blob["head_w"].to(dev)))
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-2, total_steps=100, warmup_ratio=0.0),
                                   **backend_kw)
    backend.prepare_model(eagle)
    b = blob["batch"]
    batch = TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                            hidden_state=b["hidden_state"].to(dev), target=b["target"].to(dev)), {"target_repr": "hidden_state"})
    return eagle, TrainerCore(strat, backend, accumulation_steps=2), backend, batch


def _run(golden_dir, dev, early_lm_head=False, record_buckets=False, **kw):
    eagle, core, backend, batch = _build(golden_dir, dev, **kw)
    eagle.engine.early_lm_head_wgrad = early_lm_head
    if record_buckets:
        backend.bucket_events = []
    norms = []
    for _ in range(4):                      # 2 optimizer steps, each = a non-boundary + a boundary micro-step
        r = core.train_step(batch)
        if r.stepped:
            norms.append(float(r.grad_norm))
    torch.cuda.synchronize()
    return eagle.engine.flat.data.clone(), norms, backend


def test_rccl_world1_paths_match_the_single_process_run(golden_dir):
    dev = torch.device("cuda", 0)
    ref_w, ref_n, ref_be = _run(golden_dir, dev)
    assert ref_be.module.engine.on_bucket_ready is None          # no process group: no collectives at all
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for single in (False, True):
            w, n, be = _run(golden_dir, dev, force_collectives=True, single_collective=single)
            assert be.module.engine.on_bucket_ready is not None   # the hook is live: collectives were issued
            assert torch.equal(w, ref_w), f"single_collective={single}: weights differ from the run without collectives"
            assert n == ref_n
        # the lm_head weight gradient + its bucket BEFORE the data-gradient sweep (bench.py --dp-early-lm-head), with the per-bucket event
        # record on: same weights bit for bit; every boundary backward left one (gradient complete -> all-reduce finished) pair per bucket
        w, n, be = _run(golden_dir, dev, early_lm_head=True, record_buckets=True, force_collectives=True)
        assert torch.equal(w, ref_w) and n == ref_n
        tl = be.bucket_timeline()
        nb = len(be.module.engine.bucket_bounds())
        assert len(tl) == 2 * nb and all(mb > 0 and ms >= 0 for mb, ms in tl)
        lo, hi = be.module.engine.bucket_bounds()[0]
        assert abs(tl[0][0] - (hi - lo) * 2 / 1e6) < 1e-9      # the first recorded bucket is lm_head's (the largest one at real dims)
    finally:
        dist.destroy_process_group()

"""Time the REFERENCE TRAINER ITSELF on this container's host cores.  TEST / BASELINE
INFRASTRUCTURE: build container only (imports /root/reference through oracle/ref_harness.py); nothing here is shipped.

    python oracle/time_reference_trainer.py cfg1 fp32 [steps]      # Qwen2.5-0.5B draft dims, B=1, S=256
    python oracle/time_reference_trainer.py cfg2 bf16 [steps]      # Llama-3-8B draft dims,   B=1, S=2048 (x8 = the bs 8 step)

What runs: the reference's unmodified ``build_offline_runtime -> Trainer.fit()`` (specforge/launch.py:539-661,
training/trainer.py:511), its loader, collator, ``TrainerCore``, ``BF16Optimizer``; sdpa attention backend, the loss
file's own eager ``_compute_loss`` (the Triton kernel has no CPU driver), DDP over gloo at world size 1,
``torch.set_num_threads(ncores)``.  Inputs: the synthetic files of SURVEY 8d (``tests/test_runtime/_fixtures.py:95-149``
at the config's dims).  Reported: the trainer's own ``perf/optimizer_step_time_s`` per logged step (log_interval 1; the first
step is warm-up) and tokens/s = B*S / step time.  Appends one JSON line to profiles/r5_reference_cpu_trainer.jsonl."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH  # noqa: E402

RH.setup()
import torch  # noqa: E402

CFG = {
    "cfg1": dict(json="qwen2.5-0.5b-eagle3.json", Ht=896, S=256, B=1, name="Qwen2.5-0.5B EAGLE3 draft, B=1 x S=256"),
    "cfg2": dict(json="llama3-8B-eagle3.json", Ht=4096, S=2048, B=1, name="Llama-3-8B EAGLE3 draft, B=1 x S=2048"),
}


def main():
    which, dtype_s = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    c = CFG[which]
    dtype = dict(fp32=torch.float32, bf16=torch.bfloat16)[dtype_s]
    ncores = os.cpu_count()
    torch.set_num_threads(ncores)
    from safetensors.torch import save_file

    from specforge.algorithms.builtin import builtin_algorithm_registry
    from specforge.algorithms.eagle3.model import OnlineEagle3Model
    from specforge.launch import build_offline_runtime
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig
    from specforge.modeling.target.target_head import TargetHead
    from specforge.optimizer import BF16Optimizer

    RH.init_single_rank(29591)
    dcfg = json.load(open(os.path.join(RH.REFERENCE_ROOT, "configs", c["json"])))
    Ht, S, B = c["Ht"], c["S"], c["B"]
    Vt, Vd = dcfg["vocab_size"], dcfg["draft_vocab_size"]
    work = tempfile.mkdtemp(prefix="reftime_")
    dj = os.path.join(work, "draft.json")
    json.dump(dcfg, open(dj, "w"))
    g = torch.Generator().manual_seed(0)
    n_files = B * (steps + 1)
    fd = os.path.join(work, "features")
    os.makedirs(fd)
    for i in range(n_files):
        torch.save({"input_ids": torch.randint(0, Vt, (S,), generator=g), "loss_mask": torch.ones(S, dtype=torch.long),
                    "hidden_state": torch.randn(1, S, Ht, generator=g).to(torch.bfloat16),   # TargetHead is a bf16 module

                    "aux_hidden_state": torch.randn(1, S, 3 * Ht, generator=g).to(dtype)}, os.path.join(fd, f"{i:04d}.ckpt"))
    td = os.path.join(work, "target")
    os.makedirs(td)
    json.dump({"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_size": Ht, "vocab_size": Vt,
               "num_hidden_layers": 1, "num_attention_heads": 4, "intermediate_size": 128}, open(os.path.join(td, "config.json"), "w"))
    save_file({"lm_head.weight": torch.randn(Vt, Ht, generator=g).to(torch.bfloat16)}, os.path.join(td, "model.safetensors"))
    json.dump({"metadata": {}, "weight_map": {"lm_head.weight": "model.safetensors"}},
              open(os.path.join(td, "model.safetensors.index.json"), "w"))
    ids = torch.randperm(Vt, generator=g)[:Vd].sort().values
    t2d = torch.zeros(Vt, dtype=torch.bool)
    t2d[ids] = True
    vp = os.path.join(work, "vm.pt")
    torch.save({"t2d": t2d, "d2t": (ids - torch.arange(Vd)).to(torch.int64)}, vp)

    torch.manual_seed(0)
    draft = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj), attention_backend="sdpa", torch_dtype=dtype)
    draft.load_vocab_mapping(vp)
    draft.freeze_embedding()
    head = TargetHead.from_pretrained(td, lm_head_key="lm_head.weight")
    model = OnlineEagle3Model(draft_model=draft, length=7, attention_backend="sdpa")
    logged = []
    wall = []
    alg = builtin_algorithm_registry().resolve("eagle3")

    def opt_factory(module):
        return BF16Optimizer(module, lr=1e-4, max_grad_norm=0.5, warmup_ratio=0.015, total_steps=1000)

    def logger(m, s):
        logged.append({k: v for k, v in m.items() if k.startswith("perf/") or k in ("loss", "grad_norm")})
        wall.append(time.perf_counter())
        print("step", s, {k: round(float(v), 4) for k, v in logged[-1].items()}, flush=True)

    trainer = build_offline_runtime(
        algorithm=alg, hidden_states_path=fd, draft_model=model, target_head=head, optimizer_factory=opt_factory,
        run_id="reftime", output_dir=os.path.join(work, "out"), ttt_length=7, max_len=S, batch_size=B, max_steps=steps,
        num_epochs=2, seed=0, logger=logger, log_interval=1)
    t0 = time.perf_counter()
    assert trainer.fit() == steps
    times = [m["perf/optimizer_step_time_s"] for m in logged][1:]      # first step = warm-up
    mean = sum(times) / len(times)
    rec = dict(config=c["name"], dtype=dtype_s, cores=ncores, threads=torch.get_num_threads(), steps_timed=len(times),
               step_time_s=round(mean, 3), step_times_s=[round(t, 3) for t in times], tokens_per_s=round(B * S / mean, 2),
               data_wait_s=round(sum(m["perf/data_wait_time_s"] for m in logged[1:]) / len(times), 4),
               what="reference Trainer.fit() (sdpa, eager loss, BF16Optimizer, DDP/gloo world 1), unmodified source",
               total_wall_s=round(time.perf_counter() - t0, 1))
    print(json.dumps(rec))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r5_reference_cpu_trainer.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()

"""Loss curve of the REFERENCE TRAINER at BASELINE.json configs[0]'s REAL dimensions (build container only; oracle/ref_harness.py).

    python oracle/gen_curve_realdims.py          # writes tests/golden/loss_curve_cfg1_realdims.pt  (~100 KB)

configs[0] = "Qwen2.5-0.5B EAGLE3 offline, pre-captured hidden states, bs=1 seq=256" -- the one configuration the reference runs on a CPU.
Here its own trainer (``build_offline_runtime -> Trainer.fit()``: its reader, normaliser, collator, ``_shard_offline_refs``, ``TrainerCore``,
``_reduce_eagle3_metrics``, ``BF16Optimizer``; sdpa backend, eager-loss shim, DDP over gloo at world size 1) trains the draft of
``configs/qwen2.5-0.5b-eagle3.json`` -- the draft JSON unmodified -- for 12 optimizer steps over 4 ragged feature files (3 epochs, reshuffled per epoch: every file is seen three times, so the
loss falls as the draft memorises them; some files longer than ``max_length`` 256, a prompt prefix without loss), bf16 parameters, ttt 7.  The fixture keeps what it LOGGED per step (``loss, ploss_i, acc_i,
acceptance_rate_i, grad_norm, lr``), the order it consumed the files in, and a summary of the final weights.  The inputs (1.1 GB of
embedding + head, the draft's initial weights, the feature files) are regenerated from the seed by ``oracle/seeded_case.py`` (integer
draws only: bit-identical on any host; checksums in the fixture).  tests/test_loss_curve_realdims.py (``-m gpu``) runs the HIP
ingest + strategy + backend + fused optimizer on the same files and compares step by step.
"""
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH  # noqa: E402

RH.setup()
import torch  # noqa: E402

from oracle.curve_case import CFG, input_checksums, make_inputs, weight_summary, write_files  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def draft_json():
    """the reference's own draft config for this target, unmodified"""
    return json.load(open(os.path.join(RH.REFERENCE_ROOT, "configs", "qwen2.5-0.5b-eagle3.json")))


def main():
    from safetensors.torch import save_file

    from specforge.algorithms.builtin import builtin_algorithm_registry
    from specforge.algorithms.eagle3.model import OnlineEagle3Model
    from specforge.launch import build_offline_runtime
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig
    from specforge.modeling.target.target_head import TargetHead
    from specforge.optimizer import BF16Optimizer

    RH.init_single_rank(29591)
    c = CFG
    params, embed, head_w, t2d, d2t, raws, lengths = make_inputs(c)
    sums = input_checksums(params, embed, head_w, t2d, d2t, raws)
    work = tempfile.mkdtemp(prefix="curve_real_")
    try:
        dj = os.path.join(work, "draft.json")
        json.dump(draft_json(), open(dj, "w"))
        feat = os.path.join(work, "features")
        write_files(feat, raws)
        td = os.path.join(work, "target")
        os.makedirs(td)
        json.dump({"architectures": ["Qwen2ForCausalLM"], "model_type": "qwen2", "hidden_size": c["Ht"], "vocab_size": c["Vt"],
                   "num_hidden_layers": 1, "num_attention_heads": 14, "num_key_value_heads": 2, "intermediate_size": 4864},
                  open(os.path.join(td, "config.json"), "w"))
        w = {"lm_head.weight": head_w.contiguous(), "model.embed_tokens.weight": embed.contiguous()}
        save_file(w, os.path.join(td, "model.safetensors"))
        json.dump({"metadata": {}, "weight_map": {k: "model.safetensors" for k in w}}, open(os.path.join(td, "model.safetensors.index.json"), "w"))
        vp = os.path.join(work, "vm.pt")
        torch.save({"t2d": t2d, "d2t": d2t}, vp)

        draft = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj), attention_backend="sdpa", torch_dtype=torch.bfloat16)
        missing, unexpected = draft.load_state_dict(params, strict=False)
        assert not unexpected and all(k in ("embed_tokens.weight", "t2d", "d2t") for k in missing), (missing, unexpected)
        draft.load_vocab_mapping(vp)
        draft.load_embedding(td, embedding_key="model.embed_tokens.weight")
        draft.freeze_embedding()
        head = TargetHead.from_pretrained(td, lm_head_key="lm_head.weight")
        model = OnlineEagle3Model(draft_model=draft, length=c["ttt"], attention_backend="sdpa")
        logged, order = [], []
        alg = builtin_algorithm_registry().resolve("eagle3")
        trainer = build_offline_runtime(
            algorithm=alg, hidden_states_path=feat, draft_model=model, target_head=head,
            optimizer_factory=lambda m: BF16Optimizer(m, lr=c["lr"], max_grad_norm=c["max_grad_norm"], warmup_ratio=c["warmup_ratio"],
                                                      total_steps=c["steps"]),
            run_id="curve-real", output_dir=os.path.join(work, "out"), ttt_length=c["ttt"], max_len=c["max_len"], batch_size=c["batch_size"],
            max_steps=c["steps"], num_epochs=c["num_epochs"], seed=c["seed"],
            logger=lambda m, s: logged.append((s, {k: v for k, v in m.items() if not k.startswith("perf/")})), log_interval=1)
        strat = trainer.core.strategy
        orig = strat.forward_loss

        def spy(batch, ctx=None):
            order.append([os.path.basename(str(s)) for s in batch.sample_ids])
            return orig(batch, ctx)

        strat.forward_loss = spy
        assert trainer.fit() == c["steps"]
        assert [s for s, _ in logged] == list(range(1, c["steps"] + 1))
        final = weight_summary(draft.state_dict(), c["seed"])
        blob = dict(cfg=dict(c), draft_config=draft_json(), checksums=sums, lengths=lengths, order=order, logged=[m for _, m in logged],
                    final=final, made_by="oracle/gen_curve_realdims.py", torch=torch.__version__)
        path = os.path.join(OUT, "loss_curve_cfg1_realdims.pt")
        torch.save(blob, path)
        print("loss", [round(m["loss"], 4) for _, m in logged])
        print("grad_norm", [round(m["grad_norm"], 3) for _, m in logged])
        print("acc_0", [round(m["acc_0"], 4) for _, m in logged])
        print("order", order[:4], "...", os.path.getsize(path) // 1024, "KiB")
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()

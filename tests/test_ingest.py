"""Hidden-state ingest: files in the reference's offline-feature format
(tests/test_runtime/_fixtures.py:131-149) -> normalised, right-padded, sharded batches; compared tensor for
tensor with a straight restatement of the reference's normaliser + collator
(algorithms/eagle3/data.py:10-27, data/utils.py:106-196) and with its shard indices (launch.py:174-239)."""
import os

import pytest
import torch

from specforge_amd.ingest import HiddenStateIngest, normalize_offline_sample
from specforge_amd.training import distributed_sampler_indices

HT, V = 64, 500


def _write(d, lengths, seed=0):
    g = torch.Generator().manual_seed(seed)
    files = []
    for i, L in enumerate(lengths):
        p = os.path.join(d, f"{i:04d}.ckpt")
        torch.save({"input_ids": torch.randint(0, V, (L,), generator=g), "loss_mask": torch.ones(L, dtype=torch.long),
                    "hidden_state": torch.randn(1, L, HT, generator=g).to(torch.bfloat16),
                    "aux_hidden_state": torch.randn(1, L, 3 * HT, generator=g).to(torch.bfloat16)}, p)
        files.append(p)
    return files


def _reference_collate(files, idxs, max_len):
    """normalize_offline_sample + DataCollatorWithPadding semantics, written independently (loops, no slicing tricks)"""
    samples = [torch.load(files[i]) for i in idxs]
    Ls = [min(int(s["input_ids"].shape[0]), max_len) for s in samples]
    L = max(Ls)          # the longest sample of the batch, nothing more (data/utils.py:122-142 at sp_degree 1)
    B = len(samples)
    out = dict(input_ids=torch.zeros(B, L, dtype=torch.int64), attention_mask=torch.zeros(B, L, dtype=torch.int64),
               loss_mask=torch.zeros(B, L, dtype=torch.int64), hidden_state=torch.zeros(B, L, 3 * HT, dtype=torch.bfloat16),
               target=torch.zeros(B, L, HT, dtype=torch.bfloat16))
    for b, (s, n) in enumerate(zip(samples, Ls)):
        out["input_ids"][b, :n] = s["input_ids"][:n]
        out["attention_mask"][b, :n] = 1
        out["loss_mask"][b, :n] = s["loss_mask"][:n]
        out["loss_mask"][b, n - 1] = 0
        out["hidden_state"][b, :n] = s["aux_hidden_state"][0, :n]
        out["target"][b, :n] = s["hidden_state"][0, :n]
    return out


@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_ingest_matches_reference_normaliser_collator_and_sharding(tmp_path, dev):
    lengths = [16, 9, 40, 33, 7, 24, 31, 12, 40, 5, 18]
    files = _write(str(tmp_path), lengths)
    max_len, B = 32, 2
    seen = []
    for rank in range(2):
        ing = HiddenStateIngest(files, batch_size=B, max_len=max_len, target_hidden_size=HT, device=dev, dp_rank=rank,
                                dp_size=2, seed=3)
        for epoch in (0, 1):
            want_idx = distributed_sampler_indices(len(files), dp_rank=rank, dp_size=2, seed=3, epoch=epoch)
            groups = [want_idx[i:i + B] for i in range(0, len(want_idx) - B + 1, B)]
            assert len(groups) == ing.batches_per_epoch()
            n_got = 0
            for batch, g in zip(ing.epoch(epoch), groups):   # a batch is valid until the next one is requested
                n_got += 1
                assert batch.metadata["sample_indices"] == g
                ref = _reference_collate(files, g, max_len)
                # the host-side loss-row counts of the engine's compaction: rows with loss_mask[b, s + k] != 0, per TTT step k
                lm = ref["loss_mask"]
                assert batch.metadata["loss_mask_suffix_counts"][:4] == [int((lm[:, k:] != 0).sum()) for k in range(4)]
                for k, v in ref.items():
                    t = batch.tensors[k]
                    assert t.device.type == dev and t.is_contiguous() and t.shape == v.shape, (k, t.shape, v.shape)
                    assert torch.equal(t.cpu(), v), k
                if epoch == 0:
                    seen.extend(g)
            assert n_got == len(groups)
    assert sorted(set(seen)) == sorted(set(range(len(files))) & set(seen))


def test_normalizer_matches_reference_function():
    g = torch.Generator().manual_seed(1)
    raw = {"input_ids": torch.randint(0, V, (12,), generator=g), "loss_mask": torch.ones(12, dtype=torch.long),
           "hidden_state": torch.randn(1, 12, HT, generator=g), "aux_hidden_state": torch.randn(1, 12, 3 * HT, generator=g)}
    out = normalize_offline_sample(raw, 10)
    assert out["hidden_state"].shape == (1, 10, 3 * HT) and out["target"].shape == (1, 10, HT)
    assert out["loss_mask"][0, -1] == 0 and int(out["loss_mask"].sum()) == 9 and int(raw["loss_mask"].sum()) == 12
    assert torch.equal(out["attention_mask"], torch.ones(1, 10, dtype=torch.long))


def test_ingest_matches_reference_collator_golden(tmp_path, golden_dir):
    """tests/golden/ingest_collate.pt was produced by the reference's OWN ``normalize_offline_sample`` +
    ``DataCollatorWithPadding`` (oracle/gen_fixtures_r2.py): ragged lengths, batches whose longest sample is not a
    multiple of 8, truncation at max_len -- tensor for tensor, shapes included."""
    blob = torch.load(os.path.join(golden_dir, "ingest_collate.pt"), weights_only=False)
    files = []
    for i, raw in enumerate(blob["raws"]):
        p = os.path.join(str(tmp_path), f"{i:04d}.ckpt")
        torch.save(raw, p)
        files.append(p)
    ing = HiddenStateIngest(files, batch_size=4, max_len=blob["max_len"], target_hidden_size=blob["hidden"], device="cpu")
    for g, want in zip(blob["groups"], blob["batches"]):
        got = ing.collate_indices(g)
        assert set(got) == set(want)
        for k, v in want.items():
            assert got[k].shape == v.shape and got[k].dtype == v.dtype, (k, got[k].shape, v.shape)
            assert torch.equal(got[k], v), k
    assert any(b["input_ids"].shape[1] % 8 for b in blob["batches"])   # the fixture does exercise unaligned lengths


def test_direct_reader_equals_the_torch_load_path(tmp_path):
    """the preadv fast path (bytes of the torch.save zip -> staging slot) against torch.load + normalise + copy: ragged
    lengths, truncation at max_len, a file in a format the fast path must refuse (float32 hidden states -> generic path)"""
    from specforge_amd import ingest as I

    g = torch.Generator().manual_seed(3)
    Ht, files = 16, []
    for i, L in enumerate([9, 33, 40, 17, 40, 5]):
        dt = torch.float32 if i == 3 else torch.bfloat16
        p = tmp_path / f"{i}.ckpt"
        torch.save({"input_ids": torch.randint(0, 500, (L,), generator=g), "loss_mask": (torch.rand(L, generator=g) > 0.3).long(),
                    "hidden_state": torch.randn(1, L, Ht, generator=g).to(dt), "aux_hidden_state": torch.randn(1, L, 3 * Ht, generator=g).to(dt)}, p)
        files.append(str(p))
    lay = I._sample_layout(files[0])
    assert lay is not None and set(lay) == {"input_ids", "loss_mask", "hidden_state", "aux_hidden_state"}
    raw = torch.load(files[0])
    with open(files[0], "rb") as f:                      # the layout points at the tensor's bytes
        f.seek(lay["aux_hidden_state"][0])
        got = torch.frombuffer(bytearray(f.read(raw["aux_hidden_state"].numel() * 2)), dtype=torch.bfloat16)
    assert torch.equal(got, raw["aux_hidden_state"].reshape(-1))
    kw = dict(batch_size=3, max_len=35, target_hidden_size=Ht, device="cpu", shuffle=False)
    fast, slow = I.HiddenStateIngest(files, direct=True, **kw), I.HiddenStateIngest(files, direct=False, **kw)
    calls = []
    orig = fast._read_generic
    fast._read_generic = lambda slot, b, path: (calls.append(path), orig(slot, b, path))[1]
    for grp in ([0, 1, 2], [3, 4, 5], [5, 0, 3]):
        a, b = fast.collate_indices(grp), slow.collate_indices(grp)
        assert set(a) == set(b)
        for k in a:
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), (grp, k)
    assert set(calls) == {files[3]}                       # only the float32 file fell back

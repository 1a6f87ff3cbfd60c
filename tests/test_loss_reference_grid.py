"""The reference's own Triton-CE test grid (tests/test_utils/test_loss.py:13-86) run against the HIP soft-target
CE in its drop-in mode (sf_ce_fused with tsum/pod/ids = NULL == LogSoftmaxLoss.apply): forward value and the
in-place gradient vs the file's eager ``_compute_loss`` (specforge/core/loss.py:15-21) restated in plain torch
fp32, rtol = atol = 1e-4 like the reference; plus the 7-step TTT accumulation with 0.8^i weights."""
import pytest
import torch

from specforge_amd import ops


def _eager(logits, target, mask):
    lp = torch.log_softmax(logits.float(), dim=2)
    return -torch.sum(mask * (target * lp), 2).mean()


def _hip_loss_and_grad(logits, target, mask, scale=1.0):
    B, T, V = logits.shape
    x = logits.clone().reshape(B * T, V)
    pm = mask.reshape(B, T).to(torch.int32).contiguous()
    rows = torch.empty(3, B * T, device=x.device)
    ops.ce_fused(x, target.contiguous(), S=T, Spad=T, off=0, pos_mask_pad=pm, loss_mask_pad=pm, grad_scale=scale / (B * T),
                 row_loss=rows[0], row_correct=rows[1], row_accept=rows[2])
    out = torch.empty(1, device=x.device)
    ops.reduce_sum(rows[0], B * T, 1, out, 1.0 / (B * T))
    return out[0], x.view(B, T, V)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 4])
@pytest.mark.parametrize("T", [1024, 2048, 4096, 6000])
@pytest.mark.parametrize("V", [4096, 8192, 10000])
def test_ce_matches_eager_reference_grid(B, T, V):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(B * 7 + T + V)
    logits = torch.randn(B, T, V, device=dev, generator=g)
    target = torch.softmax(torch.randn(B, T, V, device=dev, generator=g), dim=2)
    mask = (torch.rand(B, T, 1, device=dev, generator=g) > 0.5).float()
    ref_in = logits.clone().requires_grad_(True)
    ref = _eager(ref_in, target, mask)
    ref.backward()
    got, grad = _hip_loss_and_grad(logits, target, mask)
    torch.testing.assert_close(got, ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(grad, ref_in.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_ttt_accumulated_loss_with_decay_weights():
    """reference test_loss.py:41-86: sum_i 0.8^i * loss_i over 7 TTT positions, gradients per position"""
    dev, B, T, V, steps = "cuda", 2, 1024, 8192, 7
    g = torch.Generator(device=dev).manual_seed(0)
    total_ref, total = 0.0, 0.0
    for i in range(steps):
        logits = torch.randn(B, T, V, device=dev, generator=g)
        target = torch.softmax(torch.randn(B, T, V, device=dev, generator=g), dim=2)
        mask = (torch.rand(B, T, 1, device=dev, generator=g) > 0.3).float()
        ref_in = logits.clone().requires_grad_(True)
        li = _eager(ref_in, target, mask)
        (0.8 ** i * li).backward()
        total_ref = total_ref + 0.8 ** i * li.detach()
        got, grad = _hip_loss_and_grad(logits, target, mask, scale=0.8 ** i)
        total = total + 0.8 ** i * got
        torch.testing.assert_close(grad, ref_in.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(total, total_ref, rtol=1e-4, atol=1e-4)

"""Race / hazard screen of the hand-scheduled GEMM kernels on the GPU: random shapes and strides vs a torch fp32 reference,
and bitwise run-to-run determinism on chip-filling shapes (the asm-MFMA / LDS-DMA kernels carry their own hazard and
ordering management, so this is the screen the CDNA guide asks for after any sync-structure edit)."""
import sys, random, torch
sys.path.insert(0, ".")
from specforge_amd import ops
torch.manual_seed(0); random.seed(0)
dev = "cuda"
bad = 0
def check(name, out, ref, K):
    global bad
    err = float((out.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
    if err > 2e-2: bad += 1; print("MISMATCH", name, err)
for it in range(40):
    M = random.choice([8, 64, 136, 256, 264, 520, 1000, 2048, 4104]); N = random.choice([8, 48, 200, 256, 520, 1024, 4096, 6216])
    K = random.choice([64, 128, 320, 640, 4096, 8192])
    pad = random.choice([0, 8, 64])
    # TN
    a = torch.randn(K, M + pad, device=dev).to(torch.bfloat16)[:, :M]; b = torch.randn(K, N + pad, device=dev).to(torch.bfloat16)[:, :N]
    out = torch.zeros(M, N + 8, device=dev, dtype=random.choice([torch.bfloat16, torch.float32]))[:, :N]
    ws = torch.empty(2 * M * N, device=dev) if random.random() < 0.5 else None
    ops.gemm_tn(a, b, out, workspace=ws)
    check(f"tn {M}x{N}x{K}", out, a.float().t() @ b.float(), K)
    # NT (+ rowadd when M is a multiple of S)
    a2 = torch.randn(M, K + pad, device=dev).to(torch.bfloat16)[:, :K]; b2 = torch.randn(N, K + pad, device=dev).to(torch.bfloat16)[:, :K]
    out2 = torch.zeros(M, N + 8, device=dev, dtype=torch.bfloat16)[:, :N]
    ops.gemm_nt(a2, b2, out2)
    ref2 = a2.float() @ b2.float().t()
    check(f"nt {M}x{N}x{K}", out2, ref2, K)
    S = 8 if M % 8 == 0 else None
    if S and N % 4 == 0:
        T = 3; Spad = S + T; B = M // S
        add = torch.randn(B * Spad, N, device=dev)
        off = random.randint(0, T)
        rows = (torch.arange(M, device=dev) // S) * Spad + torch.arange(M, device=dev) % S + off
        out3 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm_nt_rowadd(a2, b2, out3, add, S=S, Spad=Spad, off=off)
        check(f"rowadd {M}x{N}x{K}", out3, ref2 + add[rows], K)
print("random shapes done, mismatches:", bad)
# determinism / race screen on chip-filling shapes
for (M, N, K, form) in [(8192, 4096, 8192, "nt"), (4096, 6144, 16384, "tn"), (4096, 14336, 8192, "tn")]:
    if form == "nt":
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        f = lambda o: ops.gemm_nt(a, b, o)
    else:
        a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
        ws = torch.empty(2 * M * N, device=dev)
        f = lambda o: ops.gemm_tn(a, b, o, workspace=ws)
    first = torch.empty(M, N, device=dev, dtype=torch.bfloat16); f(first)
    same = True
    for _ in range(25):
        o = torch.empty_like(first); f(o)
        same &= bool(torch.equal(o, first))
    ref = (a.float() @ b.float().t()) if form == "nt" else (a.float().t() @ b.float())
    err = float((first.float() - ref).abs().max() / ref.abs().max())
    print(form, M, N, K, "deterministic over 26 runs:", same, "relerr", f"{err:.2e}")
    if not same or err > 2e-2: bad += 1
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")

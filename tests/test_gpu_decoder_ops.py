"""GPU numerics tests of the decoder kernels against plain PyTorch fp32 references of the same op
(bf16 inputs upcast).  Tolerances: outputs are bf16, so elementwise |err| <= 2^-7 * scale (one bf16
ulp of the result magnitude) unless noted; reductions accumulate in fp32."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    from ecg_byte_amd import decoder_ops
    return decoder_ops


def _bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(torch.bfloat16)


def _close(got, ref, atol, rtol=2 ** -7):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g} (tol {tol.flatten()[err.argmax()].item():.4g})"


@pytest.mark.parametrize("tile", [128, 256, 257, 258])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 136, 128), (1024, 3072, 2048), (64, 512, 8192), (520, 300, 192),
                                   (1, 4099, 2048), (2, 640, 64), (5, 1000, 8192), (8, 132, 4096), (1, 2048, 16384), (2, 36, 8256), (7, 8203, 256), (8, 16384, 2048)])   # the last eight: the few-row (decode step) kernel
def test_gemm_nt(ops, M, N, K, tile):
    ops.set_gemm_tile(tile)
    try:
        _gemm_checks(ops, M, N, K)
    finally:
        ops.set_gemm_tile(0)


@pytest.mark.parametrize("M,N,K", [(1, 2560, 2048), (2, 2048, 16384), (5, 1000, 2048), (8, 16384, 2048), (8, 2048, 8192), (3, 36, 128)])
def test_gemm_nt_few_rows_with_second_operand_pair(ops, M, N, K):
    """A decode step with LoRA adapters: y = x W^T + t B^T for 1..8 rows is ONE launch of the column-per-wave kernels (the second pair is
    contracted behind the first), every dispatch variant (one / four waves per column, four columns per wave), plain and accumulating."""
    a, b = _bf(M, K, seed=121), _bf(N, K, scale=K ** -0.5, seed=122)
    a2, b2 = _bf(M, 64, seed=123), _bf(N, 64, scale=0.1, seed=124)
    want = a.float() @ b.float().T + a2.float() @ b2.float().T
    got = ops.gemm_nt(a, b, a2=a2, b2=b2)
    _close(got, want, atol=2e-2)
    assert torch.equal(ops.gemm_nt(a, b, a2=a2, b2=b2), got)
    base = _bf(M, N, seed=125)
    acc = ops.gemm_nt(a, b, a2=a2, b2=b2, out=base.clone(), alpha=0.5, accumulate=True)
    _close(acc, base.float() + 0.5 * want, atol=3e-2)
    f32 = torch.ones((M, N), device="cuda")
    ops.gemm_nt(a, b, a2=a2, b2=b2, out=f32, accumulate_f32=True)
    _close(f32, 1.0 + want, atol=1e-3 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 128), (300, 520, 192), (4096, 2048, 2048), (2048, 4096, 8192), (8192, 3072, 320)])
def test_gemm_phased_schedule_is_bitwise_the_unphased_kernel(ops, M, N, K):
    """The four-phase staggered 256x256 kernel (tile 256) issues the same MFMAs in the same order per accumulator as the
    one-barrier-pair kernel (tile 258): results must be IDENTICAL, launch after launch -- an LDS race (a read ahead of
    its LDS-DMA, a restage over a pending read) shows up as a rare differing tile, so the launch is repeated while other
    kernels keep the memory system busy.  K-tile counts 1, 2, 3, odd and large; ragged M / N."""
    a, b = _bf(M, K, seed=11), _bf(N, K, seed=12)
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    ops.set_gemm_tile(258)
    try:
        want = ops.gemm_nt(a, b)
        ops.set_gemm_tile(256)
        for rep in range(25):
            noise.random_()                      # concurrent traffic from the same stream's neighbours
            got = ops.gemm_nt(a, b)
            assert torch.equal(got, want), rep
        acc_w = torch.ones((M, N), device="cuda")
        acc_g = torch.ones((M, N), device="cuda")
        ops.set_gemm_tile(258); ops.gemm_nt(a, b, out=acc_w, alpha=0.25, accumulate_f32=True)
        ops.set_gemm_tile(256); ops.gemm_nt(a, b, out=acc_g, alpha=0.25, accumulate_f32=True)
        assert torch.equal(acc_g, acc_w)
    finally:
        ops.set_gemm_tile(0)


@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (128, 512, 768), (192, 264, 520), (8192, 2048, 8192), (4096, 16384, 2048), (32768, 512, 512)])
def test_gemm_tn_transposing_reads_are_bitwise_the_register_staged_kernel(ops, M, N, K):
    """dW = dY^T X by LDS-DMA + ds_read_b64_tr_b16 on the four-phase schedule (default) against the register-staged kernel
    (tile 258): identical bits (splits = 1: one workgroup sums a tile in one order), repeated under memory traffic to
    screen the schedule for LDS races; contraction lengths of 1, 2, 3 and many K-tiles, ragged output columns."""
    dy, x = _bf(M, N, seed=21), _bf(M, K, seed=22)
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    ref = dy.float().T @ x.float()
    ops.set_gemm_tile(258)
    try:
        want = ops.gemm_tn(dy, x, splits=1)
        ops.set_gemm_tile(0)
        for rep in range(15):
            noise.random_()
            got = ops.gemm_tn(dy, x, splits=1)
            assert torch.equal(got, want), rep
        _close(got, ref, atol=1e-2 * math.sqrt(M) / 8)
        _close(ops.gemm_tn(dy, x), ref, atol=1e-2 * math.sqrt(M) / 8)      # automatic split over the contraction
    finally:
        ops.set_gemm_tile(0)


def _gemm_checks(ops, M, N, K):
    a, b = _bf(M, K, seed=1), _bf(N, K, seed=2)
    ref = a.float() @ b.float().T
    _close(ops.gemm_nt(a, b), ref, atol=1e-2 * math.sqrt(K) / 8)
    # asymmetric identity check (catches transposed output maps)
    eye = torch.eye(K, device="cuda", dtype=torch.bfloat16)[:M] if M <= K else None
    if eye is not None:
        _close(ops.gemm_nt(eye, b), b.float().T[:M], atol=1e-6)
    # fp32 accumulate mode and alpha
    acc = torch.ones((M, N), device="cuda")
    ops.gemm_nt(a, b, out=acc, alpha=0.5, accumulate_f32=True)
    assert torch.allclose(acc, 1 + 0.5 * ref, atol=2e-2 * math.sqrt(K) / 8, rtol=1e-3)
    # strided operand views (column slices of a wider buffer)
    wide = _bf(M, K + 64, seed=3)
    _close(ops.gemm_nt(wide[:, 64:], b), wide[:, 64:].float() @ b.float().T, atol=1e-2 * math.sqrt(K) / 8)
    # bf16 accumulate mode (the LoRA branch adds into the base projection's output)
    base = _bf(M, N, seed=4)
    out = base.clone()
    ops.gemm_nt(a, b, out=out, alpha=0.25, accumulate=True)
    _close(out, base.float() + 0.25 * ref, atol=2e-2 * math.sqrt(K) / 8 + 2e-2 * base.float().abs().max().item())


def test_transpose_and_backward_products(ops):
    M, N, K = 384, 256, 192
    x, w, dy = _bf(M, K, seed=4), _bf(N, K, seed=5), _bf(M, N, seed=6)
    assert torch.equal(ops.transpose(x), x.T.contiguous())
    _close(ops.gemm_nt(dy, ops.transpose(w)), dy.float() @ w.float(), atol=0.2)            # dX = dY W
    _close(ops.gemm_nt(ops.transpose(dy), ops.transpose(x)), dy.float().T @ x.float(), atol=0.3)   # dW = dY^T X


@pytest.mark.parametrize("rows,H", [(300, 2048), (4100, 4096), (77, 640)], ids=["h2048", "h4096", "generic-h"])
@pytest.mark.parametrize("gemma", [False, True])
def test_rmsnorm(ops, gemma, rows, H):
    eps = 1e-5
    x, r, w = _bf(rows, H, seed=7), _bf(rows, H, seed=8), _bf(H, scale=0.5, seed=9)
    y, rstd, xs = ops.rmsnorm_fwd(x, w, eps, residual=r, gemma=gemma)
    xsum = (x.float() + r.float()).to(torch.bfloat16)
    assert torch.equal(xs, xsum)
    xf = xsum.float()
    rs = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    ref = (xf * rs * (1 + w.float())) if gemma else ((xf * rs).to(torch.bfloat16).float() * w.float())
    _close(y, ref, atol=1e-3)
    assert torch.allclose(rstd, rs.squeeze(-1), rtol=1e-5)
    # backward vs autograd of the fp32 formula
    dy, dres = _bf(rows, H, seed=10), _bf(rows, H, seed=11)
    xa = xf.clone().requires_grad_(True)
    wa = w.float().clone().requires_grad_(True)
    ya = xa * torch.rsqrt(xa.pow(2).mean(-1, keepdim=True) + eps) * ((1 + wa) if gemma else wa)
    ya.backward(dy.float())
    dw = torch.zeros(H, device="cuda")
    dx = ops.rmsnorm_bwd(xsum, w, rstd, dy, dw, dres=dres, gemma=gemma)
    _close(dx, xa.grad + dres.float(), atol=2e-2)
    assert torch.allclose(dw, wa.grad, atol=0.15 * math.sqrt(rows / 300), rtol=2e-2)
    assert torch.equal(ops.rmsnorm_bwd(xsum, w, rstd, dy, None, dres=dres, gemma=gemma), dx)     # frozen norm weights (LoRA): the same input gradient, no dw


def test_rope_forward_inverse(ops):
    T, Hq, D = 257, 6, 64
    x = _bf(T, Hq * D, seed=12)
    pos = torch.arange(T, device="cuda").float()
    inv = 1.0 / (500000.0 ** (torch.arange(0, D, 2, device="cuda").float() / D))
    fr = pos[:, None] * inv[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    y = x.clone()
    ops.rope_(y, cos, sin, Hq, D, Hq * D)
    xf = x.float().view(T, Hq, D)
    c = cos.to(torch.bfloat16).float()[:, None, :]
    s = sin.to(torch.bfloat16).float()[:, None, :]
    x1, x2 = xf[..., : D // 2], xf[..., D // 2:]
    ref = torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1).view(T, Hq * D)   # q*cos + rotate_half(q)*sin
    _close(y, ref, atol=1e-3)
    z = y.clone()
    ops.rope_(z, cos, sin, Hq, D, Hq * D, inverse=True)                       # transpose rotation = backward
    _close(z, x.float() * (c.repeat(1, Hq, 2).view(T, -1) ** 2 + s.repeat(1, Hq, 2).view(T, -1) ** 2), atol=3e-2)


@pytest.mark.parametrize("gelu", [False, True])
def test_glu(ops, gelu):
    T, I = 130, 1024
    gu, dh = _bf(T, 2 * I, seed=13), _bf(T, I, seed=14)
    g = gu[:, :I].float().clone().requires_grad_(True)
    u = gu[:, I:].float().clone().requires_grad_(True)
    act = torch.nn.functional.gelu(g, approximate="tanh") if gelu else torch.nn.functional.silu(g)
    h = act * u
    h.backward(dh.float())
    _close(ops.glu_fwd(gu, gelu_tanh=gelu), h.detach(), atol=2e-3, rtol=2 ** -6)
    d = ops.glu_bwd(gu, dh, gelu_tanh=gelu)
    _close(d[:, :I], g.grad, atol=2e-2)
    _close(d[:, I:], u.grad, atol=2e-2)


def test_embedding(ops):
    V, H, T = 500, 256, 1000
    table = _bf(V, H, seed=15)
    ids = torch.randint(0, V, (T,), device="cuda")
    assert torch.equal(ops.embed_fwd(ids, table), table[ids])
    dout = _bf(T, H, seed=16)
    g = torch.zeros(V, H, device="cuda")
    ops.embed_bwd(ids, dout, g)
    ref = torch.zeros(V, H, device="cuda").index_add_(0, ids, dout.float())
    assert torch.allclose(g, ref, atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("rows,V,ld", [(300, 1003, 1008), (41, 132015, 132096), (9, 4096, 4096), (5, 163833, 163840)],
                         ids=["small", "llama-vocab", "whole-rounds", "widest-row-in-registers"])
@pytest.mark.parametrize("in_registers", [True, False], ids=["registers", "three-sweeps"])
def test_cross_entropy_forward_backward(ops, rows, V, ld, in_registers):
    """ForCausalLMLoss (loss_utils.py:24-47) on a chunk of rows: both kernels -- the row held in registers (one read, one write: the default for
    ld <= 163 840) and the three-sweep kernel -- against torch's fp32 cross entropy; vocabularies that end inside a chunk of 8, inside a round of
    512 chunks, and exactly on one; ignored rows; padded columns zeroed."""
    ops.set_ce_in_registers(in_registers)
    try:
        _cross_entropy_case(ops, rows, V, ld)
    finally:
        ops.set_ce_in_registers(True)


def test_cross_entropy_kernels_agree_and_repeat(ops):
    rows, V, ld = 64, 132015, 132608
    logits = _bf(rows, ld, scale=4.0, seed=27)
    labels = torch.randint(0, V, (rows,), device="cuda")
    labels[::5] = -100
    inv = ops.count_labels(labels, V)
    res = []
    for mode in (True, True, False):
        ops.set_ce_in_registers(mode)
        w, t = logits.clone(), torch.zeros(1, device="cuda")
        rl = ops.ce_fwd_bwd_(w, labels, inv, t, V)
        res.append((w, rl, t))
    ops.set_ce_in_registers(True)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])          # the same bits every launch
    assert torch.allclose(res[0][1], res[2][1], rtol=2e-6, atol=2e-6)                      # row losses: the order of the fp32 sums differs
    d = (res[0][0].float() - res[2][0].float()).abs()
    assert d.max().item() <= 2 ** -8 * res[2][0].float().abs().max().item() and (d != 0).float().mean().item() < 0.02


def _cross_entropy_case(ops, rows, V, ld):
    logits = _bf(rows, ld, scale=3.0, seed=17)
    labels = torch.randint(0, V, (rows,), device="cuda")
    labels[::7] = -100
    lf = logits[:, :V].float().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, labels, ignore_index=-100, reduction="mean")
    ref.backward()
    inv = ops.count_labels(labels, V)
    assert abs(inv.item() * (labels >= 0).sum().item() - 1.0) < 1e-6
    total = torch.zeros(1, device="cuda")
    work = logits.clone()
    row_loss = ops.ce_fwd_bwd_(work, labels, inv, total, V)
    assert abs(total.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))
    assert bool((row_loss[::7] == 0).all())
    _close(work[:, :V], lf.grad, atol=2e-5, rtol=2 ** -7)
    assert bool((work[:, V:] == 0).all())


def test_softmax_causal_mask_and_backward(ops):
    B, Hh, S = 2, 3, 136
    scores = _bf(B * Hh, S, S, seed=18)
    mask = torch.ones(B, S, device="cuda")
    mask[0, :9] = 0                                   # left padding on batch entry 0
    scale = 0.125
    p = ops.softmax_causal_fwd_(scores.clone(), mask, Hh, scale)
    sf = scores.float() * scale
    vis = torch.tril(torch.ones(S, S, device="cuda", dtype=torch.bool))[None] & (mask.repeat_interleave(Hh, 0)[:, None, :] != 0)
    ref = torch.softmax(sf.masked_fill(~vis, float("-inf")), -1)
    ref = torch.nan_to_num(ref, nan=0.0)              # fully masked (pad) rows -> zeros
    _close(p, ref, atol=2e-3)
    assert bool((p[0, :9] == 0).all())
    dp = _bf(B * Hh, S, S, seed=19)
    pf = p.float()
    want = scale * pf * (dp.float() - (pf * dp.float()).sum(-1, keepdim=True))
    _close(ops.softmax_bwd_(p, dp.clone(), scale), want, atol=2e-3)


def test_adam_with_clipping_matches_torch(ops):
    n = 10007
    p0 = _bf(n, seed=20)
    g = _bf(n, scale=0.01, seed=21)
    ref_p = p0.float().clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    p = p0.clone()
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gs = (g.float() * step * 50).to(torch.bfloat16)          # norm > 1 -> clipping active
        ref_p.grad = gs.float().clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 1.0)
        opt.step()
        acc = torch.zeros(1, device="cuda")
        ops.sumsq(gs, acc)
        assert abs(acc.item() - gs.float().pow(2).sum().item()) < 1e-3 * acc.item()
        ops.adam_step_(p, gs, m, v, acc, 1.0, 1e-3, 0.9, 0.99, 1e-8, 1e-2, step)
        # the bf16 parameter follows the fp32 trajectory to within bf16 rounding of the parameter
        _close(p, ref_p.detach(), atol=1e-3 * step, rtol=2 ** -7 * step)


def test_multi_tensor_adam_is_the_per_tensor_step_bit_for_bit(ops):
    """ecgb_adam_multi_bf16 (one launch over a list of parameters, 16-byte accesses) = ecgb_adam_step per tensor, bit for bit: parameters and both moments, over three
    steps; sizes off the 8-element and the 2^20-element chunk grid, and a tensor whose storage is only 2-byte aligned (the element-wise path)."""
    sizes = [10007, 8, 1, (1 << 20) + 24, 2048 * 64, 333]
    base = [_bf(n + 1, seed=40 + i) for i, n in enumerate(sizes)]
    ps = [b[1:] if i == 1 else b[:-1] for i, b in enumerate(base)]                # tensor 1 starts 2 bytes off a 16-byte boundary
    gs = [(_bf(n + 1, scale=0.02, seed=60 + i))[1:] if i == 1 else _bf(n, scale=0.02, seed=60 + i) for i, n in enumerate(sizes)]
    ps = [p.contiguous() if i != 1 else p for i, p in enumerate(ps)]
    one = [p.clone() for p in ps]; many = [p.clone() for p in ps]
    if True:                                                                      # keep the odd alignment in both copies
        buf1, buf2 = torch.empty(sizes[1] + 1, dtype=torch.bfloat16, device="cuda"), torch.empty(sizes[1] + 1, dtype=torch.bfloat16, device="cuda")
        buf1[1:].copy_(ps[1]); buf2[1:].copy_(ps[1]); one[1], many[1] = buf1[1:], buf2[1:]
    m1 = [torch.zeros(n, device="cuda") for n in sizes]; v1 = [torch.zeros(n, device="cuda") for n in sizes]
    m2 = [torch.zeros(n, device="cuda") for n in sizes]; v2 = [torch.zeros(n, device="cuda") for n in sizes]
    chunks = ops.SumsqPlan(sizes, torch.device("cuda"))
    tables = ops.AdamMultiPlan(many, gs, m2, v2, torch.device("cuda"))
    assert tables.matches(many, gs, m2, v2) and not tables.matches(one, gs, m2, v2)
    for step in range(1, 4):
        acc = torch.zeros(1, device="cuda")
        for g in gs:
            ops.sumsq(g.contiguous(), acc)
        for p, g, m, v in zip(one, gs, m1, v1):
            ops.adam_step_(p, g, m, v, acc, 1.0, 1e-3, 0.9, 0.99, 1e-8, 1e-2, step)
        ops.adam_multi_(tables, chunks, acc, 1.0, 1e-3, 0.9, 0.99, 1e-8, 1e-2, step)
        for i in range(len(sizes)):
            assert torch.equal(one[i], many[i]) and torch.equal(m1[i], m2[i]) and torch.equal(v1[i], v2[i]), (step, i)
    assert not torch.equal(one[0], ps[0])


@pytest.mark.parametrize("B,S,Hq,Hkv", [(2, 128, 4, 1), (1, 192, 2, 2), (3, 64, 2, 1), (2, 320, 8, 2)])
def test_fused_attention_forward_backward(ops, B, S, Hq, Hkv):
    """Fused attention (no S x S tensor) vs an fp32 PyTorch reference with the reference's mask semantics
    (causal AND key not padded; modeling_llama.py:1047-1100), incl. left padding and partial 128-row blocks."""
    D = 64
    QKV = Hq * D + 2 * Hkv * D
    qkv = _bf(B * S, QKV, seed=30)
    mask = torch.ones(B, S, device="cuda")
    mask[0, : S // 3] = 0
    scale = 1.0 / math.sqrt(D)
    o, lse = ops.attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale)
    x = qkv.float().view(B, S, QKV).clone().requires_grad_(True)
    q = x[..., : Hq * D].reshape(B, S, Hq, D).transpose(1, 2)
    k = x[..., Hq * D: Hq * D + Hkv * D].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    v = x[..., Hq * D + Hkv * D:].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1)
    vis = torch.tril(torch.ones(S, S, device="cuda", dtype=torch.bool))[None, None] & (mask[:, None, None, :] != 0)
    sc = (q @ k.transpose(-1, -2)) * scale
    p = torch.nan_to_num(torch.softmax(sc.masked_fill(~vis, float("-inf")), -1), nan=0.0)
    ref = (p.to(torch.bfloat16).float() @ v).transpose(1, 2).reshape(B * S, Hq * D)
    _close(o, ref, atol=2e-2)
    assert bool((o.view(B, S, -1)[0, : S // 3] == 0).all())                 # pad rows -> zeros
    do = _bf(B * S, Hq * D, seed=31)
    ref.backward(do.float())
    d_qkv = ops.attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale)
    want = x.grad.view(B * S, QKV)
    for name, lo, hi in (("dq", 0, Hq * D), ("dk", Hq * D, Hq * D + Hkv * D), ("dv", Hq * D + Hkv * D, QKV)):
        g, w = d_qkv[:, lo:hi].float(), want[:, lo:hi]
        rel = (g - w).norm() / w.norm()
        assert rel.item() < 2e-2, (name, rel.item())


@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (512, 512, 256), (4096, 3072, 2048), (192, 264, 520), (128, 8, 40)])
def test_gemm_tn_weight_gradient(ops, M, N, K):
    """dW = dY^T X straight from the row-major operands (no transposed copies), incl. ragged N/K edges."""
    dy, x = _bf(M, N, seed=40), _bf(M, K, seed=41)
    ref = dy.float().T @ x.float()
    _close(ops.gemm_tn(dy, x), ref, atol=1e-2 * math.sqrt(M) / 4)
    _close(ops.gemm_tn(dy, x, alpha=0.25), 0.25 * ref, atol=1e-2 * math.sqrt(M) / 8)


@pytest.mark.parametrize("M,N,K,splits", [(128, 256, 512, 8), (1024, 520, 264, 3), (4096, 2048, 2048, 4), (2048, 64, 2048, 32)])
def test_gemm_tn_k_slices_meet_in_slabs(ops, M, N, K, splits):
    """The contraction cut into slices (more slices than K-tiles in the first case: the empty ones contribute zeros): each slice's
    partial product is stored to its own fp32 slab and the slabs are summed in order, so the result is the same bits on every run;
    accumulate adds to the bf16 already in the destination."""
    dy, x = _bf(M, N, seed=42), _bf(M, K, seed=43)
    ref = dy.float().T @ x.float()
    out = torch.full((N, K), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm_tn(dy, x, splits=splits, out=out)
    _close(out, ref, atol=1e-2 * math.sqrt(M) / 4)
    again = ops.gemm_tn(dy, x, splits=splits)
    assert torch.equal(out, again)
    base = _bf(N, K, seed=44)
    acc = base.clone()
    ops.gemm_tn(dy, x, splits=splits, out=acc, accumulate=True)
    want = base.float() + ref
    tol = 1e-2 * math.sqrt(M) / 4 + 2 ** -7 * (base.float().abs() + ref.abs())      # one bf16 rounding of the sum (two with splits=1)
    assert bool(((acc.float() - want).abs() <= tol).all())
    acc1 = base.clone()
    ops.gemm_tn(dy, x, splits=1, out=acc1, accumulate=True)
    assert bool(((acc1.float() - want).abs() <= tol).all())


@pytest.mark.parametrize("gelu_tanh", [False, True])
@pytest.mark.parametrize("M,inter,K,K2", [(4096, 8192, 2048, 0), (4000, 6144, 512, 64), (8192, 3072, 64, 0)])
def test_gemm_glu_epilogue_is_bitwise_the_separate_kernels(ops, M, inter, K, K2, gelu_tanh):
    """gate|up projection with the GLU in its epilogue (gate and up rows of the weight interleaved per tile) against the plain GEMM
    followed by glu_fwd: the same bits in gate|up and in act(gate) * up; ragged row count, the LoRA K-concatenation, and the
    inference form that never writes gate|up."""
    assert ops.glu_fusable(M, inter)
    a, b = _bf(M, K, seed=51), _bf(2 * inter, K, scale=K ** -0.5, seed=52)
    a2 = _bf(M, K2, seed=53) if K2 else None
    b2 = _bf(2 * inter, K2, scale=0.1, seed=54) if K2 else None
    gu_ref = ops.gemm_nt(a, b, a2=a2, b2=b2)
    h_ref = ops.glu_fwd(gu_ref, gelu_tanh=gelu_tanh)
    gu, h = ops.gemm_nt_glu(a, b, gelu_tanh=gelu_tanh, a2=a2, b2=b2)
    assert torch.equal(gu, gu_ref)
    assert torch.equal(h, h_ref)
    none, h2 = ops.gemm_nt_glu(a, b, gelu_tanh=gelu_tanh, keep_gu=False, a2=a2, b2=b2)
    assert none is None and torch.equal(h2, h_ref)
    g, u = gu_ref[:, :inter].float(), gu_ref[:, inter:].float()
    act = torch.nn.functional.gelu(g, approximate="tanh") if gelu_tanh else torch.nn.functional.silu(g)
    _close(h, act.to(torch.bfloat16).float() * u, atol=1e-3)


@pytest.mark.parametrize("gelu_tanh", [False, True])
@pytest.mark.parametrize("p", [0.0, 0.05])
def test_lora_dx_with_glu_backward_is_bitwise_the_two_kernels(ops, p, gelu_tanh):
    """The down-projection site's adapter contribution to d(act(gate) * up) and the GLU backward in one pass: the same bits as
    ecgb_lora_dx followed by ecgb_glu_bwd, and the input gradient itself is left untouched."""
    T, inter = 1000, 2048
    dx, gu = _bf(T, inter, seed=71), _bf(T, 2 * inter, seed=72)
    dt = _bf(T, 64, scale=0.5, seed=73)
    dt[:, 16:] = 0
    At = _bf(inter, 64, scale=0.05, seed=74)
    want_dx = ops.lora_dx_(dx.clone(), dt, At, 1, 1, 2.0, p, 4321)
    want = ops.glu_bwd(gu, want_dx, gelu_tanh=gelu_tanh)
    keep = dx.clone()
    got = ops.lora_dx_glu(dx, dt, At, gu, 2.0, p, 4321, gelu_tanh=gelu_tanh)
    assert torch.equal(got, want)
    assert torch.equal(dx, keep)


@pytest.mark.parametrize("T,K,n_sub,n_fields,p", [(1024, 512, 3, 3, 0.25), (1000, 2048, 1, 1, 0.05), (4096, 2048, 2, 2, 0.05), (700, 128, 4, 2, 0.1),
                                                  (2048, 8192, 1, 1, 0.05), (32768, 2048, 3, 3, 0.05), (333, 320, 4, 4, 0.5), (640, 2048, 2, 1, 0.0)])
def test_lora_da_replays_the_forward_masks(ops, T, K, n_sub, n_fields, p):
    """The adapters' weight gradient from x itself (ecgb_lora_da: the dropout masks evaluated again from the seed, one pass over x for all
    modules of the site) against dt_f^T . xd_f on the masked copies ecgb_lora_down stores when asked: every mask bit must be the forward's.
    Ragged rows (a last tile of 8 / 40 / 13 rows), columns that are not a multiple of the 256-column tile, 1..4 sub-blocks over 1..4
    modules, no dropout, accumulation into an existing gradient; the same bits on every launch (ordered slabs, no atomics)."""
    x, A = _bf(T, K, seed=91), _bf(64, K, scale=0.05, seed=92)
    dt = _bf(T, 64, scale=0.5, seed=93)
    dt[:, 16 * n_sub:] = 0
    _, xd = ops.lora_down(x, A, n_sub, n_fields, 2.0, p, 777, keep_masked=True)
    inv = 2.0 / (1.0 - int(p * 65536) / 65536)
    w = 16 * n_sub // n_fields
    want = torch.cat([dt[:, w * f: w * f + w].float().T @ (xd[f] if xd is not None else x).float() for f in range(n_fields)], 0) * inv
    got = torch.zeros(16 * n_sub, K, dtype=torch.bfloat16, device="cuda")
    ops.lora_da(x, dt, got, n_sub, n_fields, 2.0, p, 777)
    _close(got, want, atol=2e-2 * float(want.abs().max()), rtol=1e-2)
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    for rep in range(4):
        noise.random_()
        again = torch.zeros_like(got)
        assert torch.equal(ops.lora_da(x, dt, again, n_sub, n_fields, 2.0, p, 777), got), rep
    other = ops.lora_da(x, dt, torch.zeros_like(got), n_sub, n_fields, 2.0, p, 778)
    assert p == 0.0 or not torch.equal(other, got)                          # another seed, other masks
    base = _bf(16 * n_sub, K, seed=94)
    acc = ops.lora_da(x, dt, base.clone(), n_sub, n_fields, 2.0, p, 777, accumulate=True)
    _close(acc, base.float() + want, atol=2e-2 * float(want.abs().max()) + 2e-2, rtol=1e-2)


@pytest.mark.parametrize("M,N,K", [(4096, 2048, 3072), (4000, 2048, 64), (32768, 2048, 128), (8192, 8192, 2048), (3000, 2056, 192)])
def test_gemm_nn_is_bitwise_the_nt_kernel_on_a_transposed_copy(ops, M, N, K):
    """dX = dY . W against the weight as stored ([out, in] row-major): the NN kernel (A side of the NT kernel, B side by transposing LDS
    reads) against gemm_nt on W^T -- identical bits, repeated under memory traffic (LDS race screen); 1, 2, 3 and many K-tiles, ragged
    rows and columns; the accumulate modes."""
    assert ops.nn_eligible(32768, 2048, 3072) and not ops.nn_eligible(8, 2048, 3072) and not ops.nn_eligible(4096, 2048, 40)
    a, w = _bf(M, K, seed=81), _bf(K, N, scale=K ** -0.5, seed=82)
    want = ops.gemm_nt(a, ops.transpose(w))
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    for rep in range(8):
        noise.random_()
        assert torch.equal(ops.gemm_nn(a, w), want), rep
    _close(want, a.float() @ w.float(), atol=1e-2 * math.sqrt(K) / 8)
    base = _bf(M, N, seed=83)
    o1, o2 = base.clone(), base.clone()
    ops.gemm_nt(a, ops.transpose(w), out=o1, alpha=0.5, accumulate=True)
    ops.gemm_nn(a, w, out=o2, alpha=0.5, accumulate=True)
    assert torch.equal(o1, o2)
    f1, f2 = torch.ones((M, N), device="cuda"), torch.ones((M, N), device="cuda")
    ops.gemm_nt(a, ops.transpose(w), out=f1, accumulate_f32=True)
    ops.gemm_nn(a, w, out=f2, accumulate_f32=True)
    assert torch.equal(f1, f2)


@pytest.mark.parametrize("M,N,K,cat", [(32768, 3072, 2048, False), (8192, 16384, 128, True), (16384, 8192, 192, False), (32768, 2048, 2112, True),
                                       (4096, 132608, 256, False)])
def test_gemm_nt_persistent_tile_loop_is_bitwise_the_one_tile_kernel(ops, M, N, K, cat):
    """Big problems of whole 256x256 tiles run the persistent form of the NT (and NN) kernel (a workgroup per CU walks the tiles, the next tile's first
    K-tiles land under the epilogue): the same MFMA sequence per tile, so the same bits as the one-tile-per-workgroup kernel (tile code 259) --
    even and odd K-tile counts (the ring position of a tile's first K-tile alternates), 2 and many K-tiles, the K-concatenated second operand
    pair, the GLU epilogue; repeated under memory traffic (a tile reading LDS that the next tile's DMA has already overwritten would show)."""
    a, b = _bf(M, K, seed=111), _bf(N, K, scale=K ** -0.5, seed=112)
    kw = dict(a2=_bf(M, 64, seed=113), b2=_bf(N, 64, scale=0.1, seed=114)) if cat else {}
    try:
        bn = _bf(K, N, scale=K ** -0.5, seed=115)                            # the NN kernel's operand ([K, N] as stored)
        gu_in = _bf(M, 2 * N, seed=116) if ops.nn_glu_bwd_eligible(M, N, K) and M * N <= 1 << 28 else None
        ops.set_gemm_tile(259)
        want = ops.gemm_nt(a, b, **kw)
        want_glu = ops.gemm_nt_glu(a, b, **kw) if N % 256 == 0 and ops.glu_fusable(M, N // 2) else None
        want_nn = ops.gemm_nn(a, bn)
        want_gb = ops.gemm_nn_glu_bwd(a, bn, gu_in) if gu_in is not None else None
        ops.set_gemm_tile(0)
        noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
        for rep in range(4):
            noise.random_()
            assert torch.equal(ops.gemm_nt(a, b, **kw), want), rep
            if want_glu is not None:
                gu, h = ops.gemm_nt_glu(a, b, **kw)
                assert torch.equal(gu, want_glu[0]) and torch.equal(h, want_glu[1]), rep
            assert torch.equal(ops.gemm_nn(a, bn), want_nn), rep
            if want_gb is not None:
                assert torch.equal(ops.gemm_nn_glu_bwd(a, bn, gu_in), want_gb), rep
        ops.set_gemm_backward_persistent(False)                 # what a data-parallel run selects: the same bits from the one-tile kernels
        assert torch.equal(ops.gemm_nn(a, bn), want_nn)
        if want_gb is not None:
            assert torch.equal(ops.gemm_nn_glu_bwd(a, bn, gu_in), want_gb)
    finally:
        ops.set_gemm_tile(0)
        ops.set_gemm_backward_persistent(True)


@pytest.mark.parametrize("gelu_tanh", [False, True])
@pytest.mark.parametrize("M,I,K", [(4096, 8192, 2048), (16384, 3072, 512), (8192, 2048, 192), (8192, 8192, 2048)])
def test_gemm_nn_with_glu_backward_is_bitwise_the_two_kernels(ops, M, I, K, gelu_tanh):
    """The down projection's input gradient with the GLU backward in the GEMM's epilogue (full fine-tune): d(gate|up) must be the bits of
    gemm_nn followed by glu_bwd, on every launch; shapes outside whole 256x256 tiles are refused (the model takes the two kernels there).  The last shape is one
    the entry point sends to the four-wave kernel (128 K-tiles per workgroup), the others stay on the eight-wave kernels: both against the two-kernel bits."""
    assert ops.nn_glu_bwd_eligible(32768, 8192, 2048) and not ops.nn_glu_bwd_eligible(1000, 8192, 2048) and not ops.nn_glu_bwd_eligible(4096, 8320, 2048)
    dy, w = _bf(M, K, seed=101), _bf(K, I, scale=K ** -0.5, seed=102)
    gu = _bf(M, 2 * I, seed=103)
    want = ops.glu_bwd(gu, ops.gemm_nn(dy, w), gelu_tanh=gelu_tanh)
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    for rep in range(4):
        noise.random_()
        assert torch.equal(ops.gemm_nn_glu_bwd(dy, w, gu, gelu_tanh=gelu_tanh), want), rep
    with pytest.raises(Exception):
        ops.gemm_nn_glu_bwd(dy[:1000], w, gu[:1000], gelu_tanh=gelu_tanh)


@pytest.mark.parametrize("gelu_tanh", [False, True])
@pytest.mark.parametrize("p,seed", [(0.05, 4321), (0.0, 0), (0.3, (7 << 32) + 99)])
@pytest.mark.parametrize("M,I,K", [(8192, 8192, 2048), (16384, 4096, 4096)])
def test_down_site_backward_of_a_lora_fine_tune_in_one_launch_is_bitwise_the_two_kernels(ops, M, I, K, p, seed, gelu_tanh):
    """The frozen down projection's input gradient, the adapter's share scale / (1 - p) * mask . (dt A) and the GLU backward in ONE launch of the four-wave
    kernel (ecgb_gemm_nn_glu_bwd_lora_bf16, round 4): d(gate|up) must be the bits of gemm_nn followed by lora_dx_glu -- the forward's dropout mask regenerated
    in the epilogue from the same (seed, element) rule, with and without dropout, a 64-bit seed, both activations, on every launch.  Shapes the four-wave
    kernel does not take return None (the model then runs the two kernels), and so does the switch."""
    dy, w = _bf(M, K, seed=141), _bf(K, I, scale=K ** -0.5, seed=142)
    gu = _bf(M, 2 * I, seed=143)
    dt, At = _bf(M, 64, scale=0.5, seed=144), _bf(I, 64, scale=0.05, seed=145)
    dt[:, 16:] = 0
    At[:, 16:] = 0
    want = ops.lora_dx_glu(ops.gemm_nn(dy, w), dt, At, gu, 2.0, p, seed, gelu_tanh=gelu_tanh)
    if p > 0:                                                        # the adapter's share and its mask are really in there
        assert not torch.equal(want, ops.glu_bwd(gu, ops.gemm_nn(dy, w), gelu_tanh=gelu_tanh))
        assert not torch.equal(want, ops.lora_dx_glu(ops.gemm_nn(dy, w), dt, At, gu, 2.0, 0.0, seed, gelu_tanh=gelu_tanh))
    noise = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    for rep in range(3):
        noise.random_()
        got = ops.gemm_nn_glu_bwd_lora(dy, w, gu, dt, At, 2.0, p, seed, gelu_tanh=gelu_tanh)
        assert got is not None and torch.equal(got, want), rep
    assert ops.gemm_nn_glu_bwd_lora(dy[:1024], w, gu[:1024], dt[:1024], At, 2.0, p, seed, gelu_tanh=gelu_tanh) is None      # too few K-tiles per workgroup
    ops.set_fuse_lora_dx_glu(False)
    try:
        assert ops.gemm_nn_glu_bwd_lora(dy, w, gu, dt, At, 2.0, p, seed, gelu_tanh=gelu_tanh) is None
    finally:
        ops.set_fuse_lora_dx_glu(True)


@pytest.mark.parametrize("p,seed", [(0.05, 4321), (0.0, 0)])
@pytest.mark.parametrize("M,N,K", [(32768, 2048, 2048), (16384, 4096, 4096)])
def test_input_gradient_of_a_single_module_site_in_one_launch_is_bitwise_the_two_kernels(ops, M, N, K, p, seed):
    """The o projection's input gradient of a LoRA fine-tune: product and adapter share in one launch (ecgb_gemm_nn_lora_bf16) against gemm_nn followed by
    lora_dx_ (one block): the same bits, every launch; None where the four-wave kernel does not take the shape."""
    dy, w = _bf(M, K, seed=161), _bf(K, N, scale=K ** -0.5, seed=162)
    dt, At = _bf(M, 64, scale=0.5, seed=163), _bf(N, 64, scale=0.05, seed=164)
    dt[:, 16:] = 0
    At[:, 16:] = 0
    want = ops.lora_dx_(ops.gemm_nn(dy, w), dt, At, 1, 1, 2.0, p, seed)
    assert not torch.equal(want, ops.gemm_nn(dy, w))
    for rep in range(3):
        got = ops.gemm_nn_lora(dy, w, dt, At, 2.0, p, seed)
        assert got is not None and torch.equal(got, want), rep
    assert ops.gemm_nn_lora(dy[:1024], w, dt[:1024], At, 2.0, p, seed) is None


@pytest.mark.parametrize("gemma", [False, True])
@pytest.mark.parametrize("rows,with_residual", [(4096, True), (1000, False), (7, True)])
def test_rmsnorm_forward_with_the_row_in_registers_is_the_generic_kernel_bit_for_bit(ops, rows, with_residual, gemma):
    """hidden 2048 (both model families) takes a kernel that keeps the row in registers between the sum of squares and the scaling: y, rstd and the
    residual sum must be the generic kernel's bits."""
    x, w = _bf(rows, 2048, seed=171), _bf(2048, scale=0.2, seed=172)
    res = _bf(rows, 2048, seed=173) if with_residual else None
    try:
        ops.set_rmsnorm_fwd_rows(False)
        want = ops.rmsnorm_fwd(x, w, 1e-6, residual=res, gemma=gemma)
        ops.set_rmsnorm_fwd_rows(True)
        got = ops.rmsnorm_fwd(x, w, 1e-6, residual=res, gemma=gemma)
    finally:
        ops.set_rmsnorm_fwd_rows(True)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_single_module_sites_draw_one_hash_per_element_pair_and_the_two_fields_are_independent(ops):
    """A site with one module (o, down) spends one 32-bit hash on elements 2k, 2k + 1 (low and high 16-bit field): the drop rate of even and odd columns is p,
    neighbours are dropped together p^2 of the time (independent fields), another seed gives another mask, and lora_da / lora_dx replay exactly this mask
    (test_lora_da_replays_the_forward_masks, the test above)."""
    T, K, p = 4096, 2048, 0.25
    x = _bf(T, K, seed=151).abs() + 1                               # no zeros: a zero in the masked copy is a dropped element
    A = _bf(64, K, scale=0.02, seed=152)
    _, xd = ops.lora_down(x, A, 1, 1, 2.0, p, 99, keep_masked=True)
    drop = (xd[0] == 0)
    assert torch.equal(xd[0], torch.where(drop, torch.zeros_like(x), x))
    even, odd = drop[:, 0::2].float(), drop[:, 1::2].float()
    assert abs(even.mean().item() - p) < 3e-3 and abs(odd.mean().item() - p) < 3e-3
    assert abs((even * odd).mean().item() - p * p) < 3e-3                                   # the pair's two fields
    assert abs((odd[:, :-1] * even[:, 1:]).mean().item() - p * p) < 3e-3                    # neighbours of different pairs
    assert abs((drop[:-1] & drop[1:]).float().mean().item() - p * p) < 3e-3                 # rows
    _, xd2 = ops.lora_down(x, A, 1, 1, 2.0, p, 100, keep_masked=True)
    assert abs(((xd2[0] == 0) & drop).float().mean().item() - p * p) < 3e-3


@pytest.mark.parametrize("M,N,K", [(1280, 2048, 132608), (320, 768, 51584), (1000, 2048, 49152), (4096, 2048, 132608)])
def test_gemm_nn_k_slices(ops, M, N, K):
    """The input gradient of the loss head (few labelled rows x hidden, contraction over the vocabulary): K-slices into fp32 slabs, summed in slice
    order -- against the fp32 product, the same bits every launch, ragged rows and a last slice shorter than the others."""
    splits = ops.nn_splitk_plan(M, N, K)
    assert splits >= 2 and ops.nn_splitk_plan(32768, 2048, 2048) == 0 and ops.nn_splitk_plan(8, 2048, 132608) == 0
    a, b = _bf(M, K, seed=131), _bf(K, N, scale=K ** -0.5, seed=132)
    got = ops.gemm_nn_splitk(a, b, splits)
    _close(got, a.float() @ b.float(), atol=2e-2)
    assert splits == ops.nn_splitk_plan(256, N, K)                     # the slicing depends on the contraction alone
    for s in (2, 3, splits):
        again = ops.gemm_nn_splitk(a, b, s)
        _close(again, a.float() @ b.float(), atol=2e-2)
        assert torch.equal(ops.gemm_nn_splitk(a, b, s), again)


def test_embedding_scatter_sorted_is_exact_and_repeatable(ops):
    """Rows scattered into a bf16 table in sorted order (no atomics): against an fp64 index_add on top of the table's previous contents,
    heavy repeats of a few ids, a skipped padding id, and the same bits on every call."""
    T, H, V = 5000, 1024, 700
    g = torch.Generator(device="cuda").manual_seed(91)
    ids = torch.randint(0, V, (T,), device="cuda", generator=g)
    ids[:1500] = 7
    ids[1500:1900] = 699
    ids[1900:2400] = 3                                       # the padding id: skipped
    dout = _bf(T, H, seed=92)
    base = _bf(V, H, seed=93)
    want = base.double()
    keep = ids != 3
    want.index_add_(0, ids[keep], dout[keep].double() * 0.5)
    got = ops.embed_bwd_sorted_(ids, dout, base.clone(), scale=0.5, skip_id=3)
    assert torch.equal(got[3], base[3])
    err = (got.double() - want).abs()
    assert bool((err <= 2 ** -8 * want.abs() + 1e-6).all()), err.max().item()
    for _ in range(3):
        assert torch.equal(ops.embed_bwd_sorted_(ids, dout, base.clone(), scale=0.5, skip_id=3), got)


def test_sumsq_multi_equals_per_tensor_sum(ops):
    """One launch over a list of gradient tensors (sizes from 8 elements to a few chunks of 2^20, an unaligned view among
    them) against the fp32 sum of squares."""
    sizes = [8, 2048, 1 << 20, (1 << 20) + 24, 3 * (1 << 20) + 5000, 77]
    ts = [_bf(n, seed=30 + i) for i, n in enumerate(sizes)]
    ts.append(_bf(4104, seed=40)[3:4099])                    # not 16-byte aligned: scalar path
    plan = ops.SumsqPlan([t.numel() for t in ts], "cuda")
    acc = torch.zeros(1, device="cuda")
    ops.sumsq_multi(ts, acc, plan)
    want = sum(float((t.float() ** 2).sum()) for t in ts)
    assert abs(acc.item() - want) <= 1e-4 * want


@pytest.mark.parametrize("D,Hq,Hkv", [(64, 4, 2), (128, 4, 1), (256, 8, 1)])
@pytest.mark.parametrize("B,n", [(1, 700), (3, 1500), (2, 37)])
def test_decode_attention_single_and_split_kernels(ops, D, Hq, Hkv, B, n):
    """One query per head against a KV cache with padded positions: the one-workgroup kernel and the kernel that splits a
    head's keys over workgroups, against softmax(q.K^T * scale + mask) . V in fp32 with P cast to bf16 (as SDPA does)."""
    torch.manual_seed(D + n)
    cap = n + 11
    cache = (torch.randn(B, cap, 2 * Hkv * D, device="cuda") * 0.7).to(torch.bfloat16)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device="cuda").to(torch.bfloat16)
    mask = torch.ones(B, cap, device="cuda")
    mask[:, :5] = 0                                             # left padding
    mask[B - 1, 5:9] = 0
    scale = D ** -0.5
    q = qkv[:, :Hq * D].float().view(B, Hq, D)
    k = cache[:, :n, :Hkv * D].float().view(B, n, Hkv, D).repeat_interleave(Hq // Hkv, dim=2)
    v = cache[:, :n, Hkv * D:].float().view(B, n, Hkv, D).repeat_interleave(Hq // Hkv, dim=2)
    s = torch.einsum("bhd,bnhd->bhn", q, k) * scale
    s = s.masked_fill(mask[:, None, :n] == 0, float("-inf"))
    p = torch.softmax(s, dim=-1).to(torch.bfloat16).float()
    want = torch.einsum("bhn,bnhd->bhd", p, v).reshape(B, Hq * D)
    one = ops.attn_decode(qkv, cache, mask, n, Hq, Hkv, D, scale).float()
    assert (one - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-3
    for ns in (2, 5, 16):
        got = ops.attn_decode_split(qkv, cache, mask, n, Hq, Hkv, D, scale, ns).float()
        assert (got - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-3, ns
        assert (got - one).abs().max().item() <= 1e-2 * want.abs().max().item() + 1e-3, ns   # same arithmetic per key, other summation order
        # the length in device memory (a replayed graph's step), a caller-owned scratch: the same bits
        n_dev = torch.full((1,), n, dtype=torch.int32, device="cuda")
        own = ops.decode_split_scratch(cap, B, Hq, D, ns, "cuda")
        assert torch.equal(ops.attn_decode_split(qkv, cache, mask, n_dev, Hq, Hkv, D, scale, ns, scratch=own).float(), got), ns
    assert ops.decode_splits(100, 1, 8) == 1 and ops.decode_splits(2048, 1, 8) == 32 and ops.decode_splits(700, 8, 8) == 10


def test_decode_advance_is_the_generate_loops_bookkeeping(ops):
    """ecgb_decode_advance against the element-wise torch lines it replaces in the replayed decode step (generation/utils.py:3208-3232): finished sequences take the pad id,
    the eos test on the padded token, the token into `out` / `tok`, the mask's new column, the counters; with and without eos ids."""
    B, cap, pad = 5, 40, 7
    g = torch.Generator(device="cuda").manual_seed(3)
    for eos in (None, torch.tensor([11, 13], device="cuda"), torch.tensor([pad], device="cuda")):
        col = torch.randint(5, 30, (B, 1), device="cuda", generator=g)
        st = dict(tok=torch.zeros(B, dtype=torch.long, device="cuda"), pos=torch.randint(0, 50, (B,), device="cuda", generator=g), col=col.clone(),
                  n_dev=torch.full((1,), 9, dtype=torch.int32, device="cuda"), out=torch.randint(0, 99, (B, cap), device="cuda", generator=g),
                  mask=torch.zeros((B, cap), device="cuda"), unfinished=torch.tensor([1, 0, 1, 1, 0], device="cuda"))
        want = {k: v.clone() for k, v in st.items()}
        for step in range(3):
            nx = torch.tensor([11, 5, 20 + step, 13 if step == 1 else 2, 9], device="cuda")
            # the torch lines
            t = nx
            if eos is not None:
                t = nx * want["unfinished"] + pad * (1 - want["unfinished"])
                want["unfinished"] = want["unfinished"] * (t[:, None] != eos[None, :]).all(1).long()
            want["out"].scatter_(1, want["col"], t[:, None])
            want["mask"].scatter_(1, want["col"], torch.ones((B, 1), device="cuda"))
            want["tok"] = t.clone()
            want["pos"] = want["pos"] + 1
            want["n_dev"] = want["n_dev"] + 1
            want["col"] = want["col"] + 1
            ops.decode_advance_(nx, st["tok"], st["pos"], st["col"], st["n_dev"], st["out"], st["mask"], st["unfinished"], pad, eos)
            for k in st:
                assert torch.equal(st[k], want[k]), (k, step, eos)


def test_rope_table_is_the_torch_expression_bit_for_bit(ops):
    """ecgb_rope_table = (pos.float()[:, None] * inv_freq[None, :]).cos() / .sin() (modeling_llama.py:119-139 in fp32): the decode step's one launch for four."""
    for D, theta in ((64, 500000.0), (256, 10000.0), (128, 10000.0)):
        inv_freq = (1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))).cuda()
        pos = torch.cat([torch.arange(0, 4200), torch.tensor([8191, 32767, 131071, 1 << 20])]).cuda()
        fr = pos.float()[:, None] * inv_freq[None, :]
        cos, sin = ops.rope_table(pos, inv_freq)
        assert torch.equal(cos, fr.cos()) and torch.equal(sin, fr.sin()), D


@pytest.mark.parametrize("rows,n,ld", [(1, 259759, 259776), (3, 1000, 1000), (8, 4099, 4103), (2, 7, 9), (5, 300, 304), (4, 70001, 70005), (16, 33000, 33008), (17, 33000, 33008)])
def test_argmax_rows_is_torch_argmax(ops, rows, n, ld):
    """ecgb_argmax_bf16 (the greedy token choice of generate()) = torch.argmax over the first n columns: the FIRST index of the maximum with ties
    (bf16 logits tie often), -0 == +0, a NaN counts as the maximum; rows that start off a 16-byte boundary (odd widths)."""
    g = torch.Generator(device="cuda").manual_seed(rows * 1000 + n)
    x = (torch.randn(rows, ld, device="cuda", generator=g) * 3).to(torch.bfloat16)
    x[:, n:] = 100.0                                               # past the end: never chosen
    assert torch.equal(ops.argmax_rows(x, n), x[:, :n].float().argmax(-1))
    assert torch.equal(ops.argmax_rows(x, n), x[:, :n].float().argmax(-1))      # (a wide row of few goes over several workgroups that meet in a slot: left clean for the next launch)
    # ties: few distinct values
    y = torch.randint(-2, 3, (rows, ld), device="cuda", generator=g).to(torch.bfloat16)
    assert torch.equal(ops.argmax_rows(y, n), y[:, :n].float().argmax(-1))
    # all negative, zeros of both signs
    z = -x.abs()
    z[:, n // 2] = -0.0
    z[:, n - 1] = 0.0
    assert torch.equal(ops.argmax_rows(z, n), torch.full((rows,), n // 2, device="cuda"))
    z[0, n // 3] = float("nan")
    assert int(ops.argmax_rows(z, n)[0]) == n // 3
    w = torch.full((rows, ld), float("-inf"), device="cuda").to(torch.bfloat16)
    assert torch.equal(ops.argmax_rows(w, n), torch.zeros(rows, dtype=torch.long, device="cuda"))


@pytest.mark.parametrize("M,N,K,pad", [(512, 512, 64, 0), (768, 256, 192, 8), (4096, 8192, 256, 0), (8192, 8192, 1024, 64)])
def test_gemm_nt_four_wave_kernel_same_bits(ops, M, N, K, pad):
    """csrc/gemm_w4.hip (four waves per workgroup, 128x128 wave tiles, persistent) = the eight-wave kernels bit for bit (same MFMA, same order over K), for
    every tile order, with padded row strides and a scale; ecgb_gemm_nt_bf16 dispatches to it from 256 K-tiles per CU on and stays on the eight-wave kernels
    when the switch is off; shapes off whole tiles are refused by name."""
    a = _bf(M, K + pad, seed=51)[:, :K]
    b = _bf(N, K + pad, seed=52)[:, :K]
    ops.set_gemm_w4(False)
    try:
        ref = ops.gemm_nt(a, b)
        ref_s = ops.gemm_nt(a, b, alpha=0.37)
    finally:
        ops.set_gemm_w4(True)
    want = a.float() @ b.float().T
    assert (ref.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    for g in (0, 3, 16, 1000):
        ops.set_gemm_w4_group_m(g)
        try:
            assert torch.equal(ops.gemm_nt_w4(a, b), ref), g
        finally:
            ops.set_gemm_w4_group_m(8)
    assert torch.equal(ops.gemm_nt_w4(a, b, alpha=0.37), ref_s)
    out = torch.full((M, N + 24), 7.0, device="cuda", dtype=torch.bfloat16)       # into a wider buffer: the columns past N stay
    ops.gemm_nt_w4(a, b, out=out[:, :N])
    assert torch.equal(out[:, :N], ref) and bool((out[:, N:] == 7.0).all())
    assert torch.equal(ops.gemm_nt(a, b), ref)                                   # the dispatch, whatever it picks (the last shape: four-wave)
    with pytest.raises(Exception):
        ops.gemm_nt_w4(a[:-16], b)


@pytest.mark.parametrize("gelu", [False, True])
def test_gemm_nt_glu_four_wave_kernel_same_bits(ops, gelu):
    """The gate|up projection with the GLU in the epilogue on the four-wave kernel (a shape ecgb_gemm_nt_glu_bf16 dispatches there: 256 K-tiles per CU) = the
    eight-wave kernel bit for bit: gate|up, act(gate) * up, and the inference form that never writes gate|up."""
    M, I, K = 8192, 8192, 2048
    x, w = _bf(M, K, seed=61), _bf(2 * I, K, seed=62) * 0.05
    ops.set_gemm_w4(False)
    try:
        gu8, h8 = ops.gemm_nt_glu(x, w, gelu_tanh=gelu, keep_gu=True)
    finally:
        ops.set_gemm_w4(True)
    gu4, h4 = ops.gemm_nt_glu(x, w, gelu_tanh=gelu, keep_gu=True)
    assert torch.equal(gu4, gu8) and torch.equal(h4, h8)
    none, h4n = ops.gemm_nt_glu(x, w, gelu_tanh=gelu, keep_gu=False)
    assert none is None and torch.equal(h4n, h8)
    assert torch.equal(ops.glu_fwd(gu4, gelu_tanh=gelu), h4)                      # and the unfused kernel on the stored projection
    want = x[:64].float() @ w.float().T
    assert (gu4[:64].float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()


def test_gemm_nt_cat_four_wave_kernel_same_bits(ops):
    """A second operand pair behind the first (the LoRA branch inside the projection's launch) on the four-wave kernel, plain and with the GLU epilogue, at a
    shape the entry points dispatch there: the eight-wave kernels bit for bit; K2 = 64 and 128."""
    M, N, K = 8192, 16384, 2048
    x, w = _bf(M, K, seed=71), _bf(N, K, seed=72) * 0.05
    for K2 in (64, 128):
        t, bl = _bf(M, K2, seed=73), _bf(N, K2, seed=74) * 0.1
        ops.set_gemm_w4(False)
        try:
            y8 = ops.gemm_nt(x, w, a2=t, b2=bl)
            gu8, h8 = ops.gemm_nt_glu(x, w, a2=t, b2=bl)
        finally:
            ops.set_gemm_w4(True)
        assert torch.equal(ops.gemm_nt(x, w, a2=t, b2=bl), y8), K2
        ops.set_gemm_w4(2)                                                       # (the GLU form with a pair stays on eight waves by default)
        try:
            gu4, h4 = ops.gemm_nt_glu(x, w, a2=t, b2=bl)
        finally:
            ops.set_gemm_w4(True)
        assert torch.equal(gu4, gu8) and torch.equal(h4, h8), K2
        want = x[:32].float() @ w.float().T + t[:32].float() @ bl.float().T
        assert (y8[:32].float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    assert not torch.equal(y8, ops.gemm_nt(x, w))                                 # (the branch is there)


@pytest.mark.parametrize("with_pair", [False, True])
def test_gemm_nt_rope_epilogue_is_gemm_then_rope_bit_for_bit(ops, with_pair):
    """ecgb_gemm_nt_bf16_rope (the q|k|v projection with RoPE's forward in the four-wave kernel's epilogue) = ecgb_gemm_nt_bf16[_cat] followed by ecgb_rope on the
    q and k heads: the same bits, the v columns untouched; a shape the kernel does not take goes through the two calls by itself."""
    M, K, Hq, Hkv, D = 8192, 512, 32, 8, 64
    N, cols = (Hq + 2 * Hkv) * D, (Hq + Hkv) * D
    x, w = _bf(M, K, seed=91), _bf(N, K, seed=92) * 0.1
    kw = dict(a2=_bf(M, 64, seed=93), b2=_bf(N, 64, seed=94) * 0.1) if with_pair else {}
    pos = (torch.arange(M, device="cuda") % 1024).float()
    fr = pos[:, None] * (1.0 / (500000.0 ** (torch.arange(0, D, 2, device="cuda").float() / D)))[None, :]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    plain = ops.gemm_nt(x, w, **kw)
    apart = ops.rope_(plain.clone(), cos, sin, Hq + Hkv, D, N)
    assert torch.equal(ops.gemm_nt_rope(x, w, cos, sin, cols, **kw), apart)      # the default: the two calls (faster at the step's shape)
    ops.set_gemm_rope_fusion(True)
    try:
        fused = ops.gemm_nt_rope(x, w, cos, sin, cols, **kw)
        small = ops.gemm_nt_rope(x[:300], w, cos[:300].contiguous(), sin[:300].contiguous(), cols, **({k: v[:300] if k == "a2" else v for k, v in kw.items()}))
    finally:
        ops.set_gemm_rope_fusion(False)
    assert torch.equal(fused, apart)
    assert torch.equal(fused[:, cols:], plain[:, cols:]) and not torch.equal(fused[:, :cols], plain[:, :cols])
    assert torch.equal(small, apart[:300])                                        # 300 rows: not whole tiles, the fallback


@pytest.mark.parametrize("M,N,K", [(512, 512, 128), (768, 256, 192), (8192, 2048, 4096)])
def test_gemm_nn_four_wave_kernel_same_bits(ops, M, N, K):
    """The four-wave kernel on the NN layout (B [K, N] row-major: dX = dY . W, the fragments of B by transposing LDS reads) = the eight-wave NN kernels bit for bit;
    the last shape is one ecgb_gemm_nn_bf16 dispatches there (256 K-tiles per CU); padded strides; refused off whole tiles."""
    a = _bf(M, K + 8, seed=95)[:, :K]
    b = (_bf(K, N + 16, seed=96) * 0.1)[:, :N]
    ops.set_gemm_w4(False)
    try:
        ref = ops.gemm_nn(a, b)
    finally:
        ops.set_gemm_w4(True)
    want = a.float() @ b.float()
    assert (ref.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    for g in (0, 3, 8):
        ops.set_gemm_w4_group_m(g)
        try:
            assert torch.equal(ops.gemm_nn_w4(a, b), ref), g
        finally:
            ops.set_gemm_w4_group_m(8)
    assert torch.equal(ops.gemm_nn(a, b), ref)                                   # the dispatch, whatever it picks
    with pytest.raises(Exception):
        ops.gemm_nn_w4(a, b[:, :-16])


@pytest.mark.parametrize("Kc,M,N", [(128, 512, 512), (192, 768, 256), (16384, 2048, 2048)])
def test_gemm_tn_four_wave_kernel_same_bits(ops, Kc, M, N):
    """The four-wave kernel on the TN layout (dW = dY^T . X: both operands staged as contraction rows, fragments by transposing LDS reads) = the eight-wave TN kernel with
    one K-slice bit for bit; the last shape is one ecgb_gemm_tn_bf16 dispatches there; padded strides; refused off whole tiles."""
    a = _bf(Kc, M + 8, seed=97)[:, :M]
    b = (_bf(Kc, N + 16, seed=98) * 0.1)[:, :N]
    ops.set_gemm_w4(False)
    try:
        ref = ops.gemm_tn(a, b, splits=1)
    finally:
        ops.set_gemm_w4(True)
    want = a.float().T @ b.float()
    assert (ref.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()
    for g in (0, 8):
        ops.set_gemm_w4_group_m(g)
        try:
            assert torch.equal(ops.gemm_tn_w4(a, b), ref), g
        finally:
            ops.set_gemm_w4_group_m(8)
    assert torch.equal(ops.gemm_tn(a, b, splits=1), ref)                          # the dispatch, whatever it picks
    with pytest.raises(Exception):
        ops.gemm_tn_w4(a[:, :-16], b)


@pytest.mark.parametrize("M", [1, 2, 5, 8])
@pytest.mark.parametrize("gelu", [False, True])
def test_few_row_glu_projection_is_gemm_then_glu_bit_for_bit(ops, M, gelu):
    """A decode step's gate|up projection with the GLU folded into the few-row kernel (gemm_nt_skinny_glu_kernel) = the few-row GEMM followed by ecgb_glu_fwd: the same
    bits for act(gate) * up and for gate|up when it is kept, with and without the LoRA pair."""
    I, K = 1024, 2048
    x, w = _bf(M, K, seed=101), _bf(2 * I, K, seed=102) * 0.05
    t, bl = _bf(M, 64, seed=103), _bf(2 * I, 64, seed=104) * 0.05
    for kw in ({}, dict(a2=t, b2=bl)):
        gu = ops.gemm_nt(x, w, **kw)
        h = ops.glu_fwd(gu, gelu_tanh=gelu)
        gu2, h2 = ops.gemm_nt_glu(x, w, gelu_tanh=gelu, keep_gu=True, **kw)
        assert torch.equal(gu2, gu) and torch.equal(h2, h)
        none, h3 = ops.gemm_nt_glu(x, w, gelu_tanh=gelu, keep_gu=False, **kw)
        assert none is None and torch.equal(h3, h)


@pytest.mark.parametrize("B,Hq,Hkv,D", [(1, 8, 1, 256), (3, 32, 8, 64), (2, 4, 4, 128)])
def test_rope_append_is_rope_then_cache_copy(ops, B, Hq, Hkv, D):
    """ecgb_rope_append (a decode step's RoPE on q and k and the KV-cache append in one launch) = ecgb_rope followed by the copy of k | v into the cache row, with the
    length as an argument and in device memory; the other cache rows stay."""
    QKV, cap, n = (Hq + 2 * Hkv) * D, 37, 20
    qkv = _bf(B, QKV, seed=111)
    pos = torch.arange(B, device="cuda").float() * 7 + 3
    fr = pos[:, None] * (1.0 / (10000.0 ** (torch.arange(0, D, 2, device="cuda").float() / D)))[None, :]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    cache0 = _bf(B * cap, 2 * Hkv * D, seed=112).view(B, cap, 2 * Hkv * D).contiguous()
    want_q = ops.rope_(qkv.clone(), cos, sin, Hq + Hkv, D, QKV)
    want_c = cache0.clone()
    want_c[:, n - 1] = want_q[:, Hq * D:]
    for dyn in (False, True):
        q2, c2 = qkv.clone(), cache0.clone()
        ops.rope_append_(q2, cos, sin, Hq, Hkv, D, c2, torch.full((1,), n, dtype=torch.int32, device="cuda") if dyn else n)
        assert torch.equal(q2, want_q) and torch.equal(c2, want_c), dyn


@pytest.mark.parametrize("rows", [1, 2, 5, 8])
@pytest.mark.parametrize("gemma", [False, True])
def test_rmsnorm_with_lora_down_projection_is_the_two_launches(ops, rows, gemma):
    """ecgb_rmsnorm_lora_fwd (a decode step's RMSNorm with the adapter site's t = scale * y A^T formed in the same launch) = ecgb_rmsnorm_fwd followed by the few-row
    GEMM: y, rstd, the residual sum and t, bit for bit."""
    H = 2048
    x, res, w = _bf(rows, H, seed=121), _bf(rows, H, seed=122), _bf(H, seed=123) * 0.1 + 1.0
    A = _bf(64, H, seed=124) * 0.05
    A[40:] = 0
    for residual in (None, res):
        y0, r0, s0 = ops.rmsnorm_fwd(x, w, 1e-5, residual=residual, gemma=gemma)
        t0 = ops.gemm_nt(y0, A, alpha=2.0)
        y1, r1, s1, t1 = ops.rmsnorm_fwd(x, w, 1e-5, residual=residual, gemma=gemma, lora=(A, 2.0))
        assert torch.equal(y1, y0) and torch.equal(r1, r0) and torch.equal(s1, s0) and torch.equal(t1, t0)

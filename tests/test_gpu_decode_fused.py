"""The fused decode-step kernels of csrc/decode.hip (round 5) against the separate kernels they replace, bit for bit: every sum is formed in the same order
(rmsnorm_lora_fwd_kernel, gemm_nt_skinny_kernel<2, 1 / 4>, gemm_nt_skinny_glu_kernel, rope_append_kernel, attn_decode_scores / values / combine).
Reference of the ops themselves: modeling_llama.py:67-72,193-224,238-258,526-614; the separate kernels are pinned against fp32 torch in tests/test_gpu_decoder_ops.py."""
import math

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


def _site(in_dim, out_dim, n_sub, seed):
    """A LoRA site as decoder.LoraSite lays it out: A [64, in] with 16 * n_sub used rows, B [out, 64]."""
    A = torch.zeros(64, in_dim, dtype=torch.bfloat16, device="cuda")
    A[: 16 * n_sub] = _bf(16 * n_sub, in_dim, scale=in_dim ** -0.5, seed=seed)
    B = _bf(out_dim, 64, scale=0.05, seed=seed + 1)
    B[:, 16 * n_sub:] = 0
    return A, 16 * n_sub, 2.0, B


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("gemma", [False, True])
@pytest.mark.parametrize("lora", [False, True])
@pytest.mark.parametrize("with_delta", [True, False])
def test_norm_gemv_equals_rmsnorm_then_few_row_gemm(ops, M, gemma, lora, with_delta):
    H, N = 2048, 2560
    x, delta, w = _bf(M, H, seed=1), (_bf(M, H, scale=0.3, seed=2) if with_delta else None), _bf(H, scale=0.2, seed=3)
    W = _bf(N, H, scale=H ** -0.5, seed=4)
    site = _site(H, N, 3, 10) if lora else None
    y, xs = ops.decode_norm_gemv(x, delta, w, 1e-5, gemma, W, lora=site)
    if lora:
        h, _, xs_ref, t = ops.rmsnorm_fwd(x, w, 1e-5, residual=delta, gemma=gemma, lora=(site[0], site[2]))
        ref = ops.gemm_nt(h, W, a2=t, b2=site[3])
    else:
        h, _, xs_ref = ops.rmsnorm_fwd(x, w, 1e-5, residual=delta, gemma=gemma)
        ref = ops.gemm_nt(h, W)
    assert torch.equal(y, ref)
    assert torch.equal(xs, xs_ref)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("gemma", [False, True])
def test_norm_gemv_glu_equals_rmsnorm_then_glu_gemv(ops, M, gemma):
    H, I = 2048, 8192
    x, delta, w = _bf(M, H, seed=5), _bf(M, H, scale=0.3, seed=6), _bf(H, scale=0.2, seed=7)
    W = _bf(2 * I, H, scale=H ** -0.5, seed=8)
    y, xs = ops.decode_norm_gemv(x, delta, w, 1e-6, gemma, W, glu=2 if gemma else 1)
    h, _, xs_ref = ops.rmsnorm_fwd(x, w, 1e-6, residual=delta, gemma=gemma)
    _, ref = ops.gemm_nt_glu(h, W, gelu_tanh=gemma, keep_gu=False)
    assert torch.equal(y, ref) and torch.equal(xs, xs_ref)


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("K,N", [(2048, 2048), (16384, 2048), (8192, 2048)])
@pytest.mark.parametrize("mode", ["plain", "lora-inside", "t-given"])
def test_gemv_equals_few_row_gemm(ops, M, K, N, mode):
    if mode == "lora-inside" and K > 4096:
        pytest.skip("t is formed inside only for K <= 4096")
    a, W = _bf(M, K, seed=11), _bf(N, K, scale=K ** -0.5, seed=12)
    site = _site(K, N, 1, 20)
    if mode == "plain":
        assert torch.equal(ops.decode_gemv(a, W), ops.gemm_nt(a, W))
        return
    t_ref = ops.gemm_nt(a, site[0], alpha=site[2])
    ref = ops.gemm_nt(a, W, a2=t_ref, b2=site[3])
    if mode == "lora-inside":
        got = ops.decode_gemv(a, W, lora=site)
    else:
        t = ops.decode_lora_t(a, site[0], site[1], site[2]) if K % 2048 == 0 else t_ref
        assert torch.equal(t, t_ref)
        got = ops.decode_gemv(a, W, lora=site, t=t)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,Hq,Hkv,D", [(1, 8, 1, 256), (2, 32, 8, 64), (1, 8, 2, 128), (2, 8, 8, 64)])
@pytest.mark.parametrize("dyn", [False, True])
def test_decode_attn_equals_rope_append_then_split_attention(ops, B, Hq, Hkv, D, dyn):
    """RoPE + append + attention in two launches = ecgb_rope_append + ecgb_attn_decode_split with the same number of splits: the same output, the same cache row.
    Left-padded rows (mask 0) and a length that is not a multiple of anything."""
    cap, n = 768, 613
    QKV = (Hq + 2 * Hkv) * D
    qkv = _bf(B, QKV, seed=31)
    cache = _bf(B, cap, 2 * Hkv * D, seed=32)
    mask = torch.ones(B, cap, device="cuda")
    mask[:, n:] = 0
    mask[0, :37] = 0
    pos = torch.tensor([n - 1 - 37, n - 1][:B], device="cuda").float()
    fr = pos[:, None] * torch.rand(D // 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    scale = 1.0 / math.sqrt(D)
    ns = ops.decode_attn_splits(cap)
    kv = torch.full((1,), n, dtype=torch.int32, device="cuda") if dyn else n
    # reference: the separate kernels
    q2, c2 = qkv.clone(), cache.clone()
    ops.rope_append_(q2, cos, sin, Hq, Hkv, D, c2, kv)
    ref = ops.attn_decode_split(q2, c2, mask, kv, Hq, Hkv, D, scale, ns)
    c1 = cache.clone()
    scratch = ops.decode_attn_scratch(cap, B, Hq, Hkv, D, ns, "cuda")
    got = ops.decode_attn(qkv.clone(), cos, sin, c1, mask, kv, Hq, Hkv, D, scale, ns, scratch)
    assert torch.equal(c1, c2)
    assert torch.equal(got, ref)
    # a second call on the same scratch (the tickets went back to zero): the same result
    c3 = cache.clone()
    assert torch.equal(ops.decode_attn(qkv.clone(), cos, sin, c3, mask, kv, Hq, Hkv, D, scale, ns, scratch), ref)


@pytest.mark.parametrize("family,lora", [("llama", False), ("llama", True), ("gemma", True)])
def test_fused_decode_step_equals_the_separate_kernels(family, lora):
    """A whole generate() on a small model: the fused decode step against the twelve-launch step of round 4, token for token and (no graph) logit for logit."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    kw = dict(vocab_size=1000, hidden_size=512, intermediate_size=2048, num_hidden_layers=3, num_attention_heads=8, num_key_value_heads=2, head_dim=64, pad_token_id=999)
    if family == "gemma":
        kw.update(model_type="gemma", num_key_value_heads=1, head_dim=128, num_attention_heads=4, rms_norm_eps=1e-6)
    m = HipCausalLM(DecoderConfig(**kw), seed=3)
    if lora:
        m.enable_lora(r=16, alpha=32, dropout=0.05)
        g = torch.Generator(device="cuda").manual_seed(9)
        for sites in m.lora:
            for s in sites.values():
                s.B.data.copy_((torch.randn(s.B.shape, device="cuda", generator=g) * 0.05).to(torch.bfloat16) * s.bmask)
    m.eval()
    ids = torch.randint(0, 990, (2, 40), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    mask = torch.ones(2, 40, device="cuda")
    mask[0, :7] = 0
    outs = {}
    for mode in ("separate", "fused", "fused+attn"):
        m.decode_fused, m.decode_fused_attn = mode != "separate", mode == "fused+attn"
        m.__dict__.pop("_gen_graphs", None)
        seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=12, pad_token_id=999, return_logits=True, use_graph=False)
        seq_g = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=12, pad_token_id=999)
        assert torch.equal(seq, seq_g), mode                         # the replayed graph = the eager loop
        outs[mode] = (seq, logits)
    assert torch.equal(outs["fused"][0], outs["separate"][0]) and torch.equal(outs["fused"][1], outs["separate"][1])        # the same attention kernels: the same bits
    # the two-launch attention splits the keys differently from the short-cache kernel: the same tokens, logits equal up to the order of fp32 sums
    assert torch.equal(outs["fused+attn"][0], outs["separate"][0])
    assert torch.allclose(outs["fused+attn"][1], outs["separate"][1], atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("dyn", [False, True], ids=["host-length", "device-length"])
@pytest.mark.parametrize("B,Hq,Hkv,D,cap,n,ns", [(1, 8, 1, 256, 768, 729, 12), (1, 8, 1, 256, 768, 600, 12), (2, 8, 2, 64, 1024, 1024, 16), (1, 32, 8, 64, 640, 577, 8), (1, 32, 8, 64, 768, 700, 12),
                                                (2, 4, 1, 128, 2048, 1301, 32), (1, 8, 1, 256, 768, 14, 12)],
                         ids=["c5-step", "c5-first-step", "two-sequences", "llama1b", "llama1b-384-workgroups", "d128-long", "more-splits-than-keys"])
def test_one_launch_attention_equals_rope_append_and_split_attention(ops, B, Hq, Hkv, D, cap, n, ns, dyn):
    """ecgb_attn_decode_one (round 6: RoPE + append + scores + values + combine, the splits' workgroups meeting through counters inside the launch) against
    ecgb_rope_append followed by ecgb_attn_decode_split with the same split count: the attention output AND the cache bit for bit; repeated launches on one scratch
    (the counters return to zero) repeat the bits; q|k|v itself is left unrotated."""
    qkv = _bf(B, (Hq + 2 * Hkv) * D, seed=41)
    cache = _bf(B, cap, 2 * Hkv * D, seed=42)
    mask = torch.ones(B, cap, device="cuda")
    mask[:, n:] = 0
    mask[0, : min(37, n - 1)] = 0
    pos = torch.tensor([n - 1 - min(37, n - 1), n - 1][:B], device="cuda").float()
    fr = pos[:, None] * torch.rand(D // 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(6))[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    scale = 1.0 / math.sqrt(D)
    assert ops.decode_one_ok(B, Hq, D, ns, cap)
    kv = torch.full((1,), n, dtype=torch.int32, device="cuda") if dyn else n
    q2, c2 = qkv.clone(), cache.clone()
    ops.rope_append_(q2, cos, sin, Hq, Hkv, D, c2, kv)
    ref = ops.attn_decode_split(q2, c2, mask, kv, Hq, Hkv, D, scale, ns)
    scratch = ops.decode_one_scratch(B, Hq, D, ns, "cuda")
    q1 = qkv.clone()
    for rep in range(3):
        c1 = cache.clone()
        got = ops.attn_decode_one(q1, cos, sin, c1, mask, kv, Hq, Hkv, D, scale, ns, scratch=scratch)
        assert torch.equal(c1, c2), rep
        assert torch.equal(got, ref), rep
    assert torch.equal(q1, qkv)
    assert float(scratch[-2 * B * Hq:].abs().max()) == 0.0 and bool((scratch.view(torch.int32)[: B * Hq * ns * 2] == 0x7FC0DEAD).all())   # counters back to zero, statistics back to "not stored yet"


def test_one_launch_attention_refuses_launches_that_could_not_be_resident(ops):
    assert not ops.decode_one_ok(4, 32, 64, 16, 1024) and not ops.decode_one_ok(1, 8, 256, 1, 256) and not ops.decode_one_ok(1, 8, 96, 8, 1024)
    assert ops.decode_one_ok(1, 32, 64, 12, 768)                               # Llama-3.2-1B, one sequence, a 768-row cache: 384 workgroups, under two a CU
    from ecg_byte_amd import _lib
    B, Hq, Hkv, D, cap, ns = 4, 32, 8, 64, 1024, 16
    qkv, cache, mask = _bf(B, (Hq + 2 * Hkv) * D), _bf(B, cap, 2 * Hkv * D), torch.ones(B, cap, device="cuda")
    cos = sin = torch.zeros(B, D // 2, device="cuda")
    with pytest.raises(_lib.EcgbError):
        ops.attn_decode_one(qkv, cos, sin, cache, mask, 1000, Hq, Hkv, D, 0.125, ns)


@pytest.mark.parametrize("family,lora", [("llama", False), ("gemma", True)])
def test_generate_with_one_launch_attention_equals_four_launches(family, lora):
    """generate() with caches long enough for the split attention (prompt 600): the one-launch attention step (default) against round 4's four launches, token for token,
    logit for logit, eager loop and replayed graph."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    kw = dict(vocab_size=1000, hidden_size=512, intermediate_size=2048, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, head_dim=64, pad_token_id=999)
    if family == "gemma":
        kw.update(model_type="gemma", num_key_value_heads=1, head_dim=256, num_attention_heads=4, rms_norm_eps=1e-6)
    m = HipCausalLM(DecoderConfig(**kw), seed=5)
    if lora:
        m.enable_lora(r=16, alpha=32, dropout=0.05)
        g = torch.Generator(device="cuda").manual_seed(9)
        with torch.no_grad():
            for sites in m.lora:
                for s in sites.values():
                    s.B.copy_((torch.randn(s.B.shape, device="cuda", generator=g) * 0.05).to(torch.bfloat16) * s.bmask)
    m.eval()
    ids = torch.randint(0, 990, (1, 600), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    mask = torch.ones(1, 600, device="cuda")
    mask[0, :11] = 0
    outs = {}
    for one in (False, True):
        m.decode_attn_one = one
        seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=10, pad_token_id=999, return_logits=True, use_graph=False)
        seq_g = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=10, pad_token_id=999)
        seq_g2 = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=10, pad_token_id=999)     # the captured step replayed by a second call
        assert torch.equal(seq, seq_g) and torch.equal(seq, seq_g2), one
        outs[one] = (seq, logits)
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])


def test_one_launch_attention_with_nan_inputs_returns(ops):
    """Garbage in must come out as garbage, not as a wait that never ends: the splits wait for each other's statistics on a marker no arithmetic produces, so a
    NaN sum of exponentials (NaN keys or queries) ends the wait like any value."""
    B, Hq, Hkv, D, cap, n, ns = 1, 8, 1, 256, 768, 700, 12
    qkv = _bf(B, (Hq + 2 * Hkv) * D, seed=51)
    cache = _bf(B, cap, 2 * Hkv * D, seed=52)
    cache[0, 100, :D] = float("nan")
    qkv[0, 5] = float("nan")
    mask = torch.ones(B, cap, device="cuda"); mask[:, n:] = 0
    cos = sin = torch.zeros(B, D // 2, device="cuda")
    scratch = ops.decode_one_scratch(B, Hq, D, ns, "cuda")
    out = ops.attn_decode_one(qkv, cos, sin, cache, mask, n, Hq, Hkv, D, 1.0 / 16, ns, scratch=scratch)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out.float()).any())
    assert bool((scratch.view(torch.int32)[: B * Hq * ns * 2] == 0x7FC0DEAD).all())


@pytest.mark.parametrize("M,K,N", [(1, 2048, 2048), (2, 2048, 2048), (1, 16384, 2048), (2, 16384, 2048), (1, 512, 1536), (1, 8192, 512), (1, 2048, 8200)])
def test_one_launch_adapter_site_equals_the_two_launches(ops, M, K, N):
    """Round 6: a decode step's o / down adapter site -- t = scale * x A^T by the launch's first workgroups, x W^T + t B^T by the others, t handed over in 64-bit words that carry a step
    counter (ecgb_gemm_nt_bf16_lora_decode) -- against the launch pair it replaces, bit for bit, over several steps on ONE word buffer (every step a new counter value, as
    ecgb_decode_advance_e gives it: the words of the step before are stale, not wrong)."""
    w, a, b = _bf(N, K, scale=0.05, seed=1), _bf(64, K, scale=0.05, seed=2), _bf(N, 64, scale=0.1, seed=3)
    a[48:] = 0                                                                  # (a site of three modules: rows past 16 * n_sub are zero)
    t64 = torch.zeros((M, 64), dtype=torch.int64, device="cuda")
    epoch = torch.ones(1, dtype=torch.int32, device="cuda")
    assert ops.lora_decode_ok(M, N, K)
    for step in range(6):
        x = _bf(M, K, seed=10 + step)
        t_ref = ops.gemm_nt(x, a, alpha=0.5)
        y_ref = ops.gemm_nt(x, w, a2=t_ref, b2=b)
        y = ops.gemm_nt_lora_decode(x, w, a, 0.5, b, t64, epoch)
        assert torch.equal(y, y_ref), step
        assert torch.equal((t64 & 0xFFFF).to(torch.int16).view(torch.bfloat16), t_ref) and bool(((t64 >> 16) == epoch.long()).all())
        epoch += 1


def test_one_launch_adapter_site_refuses_what_the_two_kernels_would_split_differently(ops):
    from ecg_byte_amd import _lib
    assert not ops.lora_decode_ok(1, 16384, 8192) and not ops.lora_decode_ok(3, 2048, 2048)
    x, w, a, b = _bf(1, 8192), _bf(16384, 8192, scale=0.05), _bf(64, 8192, scale=0.05), _bf(16384, 64)
    with pytest.raises(_lib.EcgbError):
        ops.gemm_nt_lora_decode(x, w, a, 1.0, b, torch.zeros((1, 64), dtype=torch.int64, device="cuda"), torch.ones(1, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("family", ["llama", "gemma"])
def test_generate_with_one_launch_adapter_sites_equals_the_separate_launches(family):
    """generate() with adapters: the o / down sites as one launch each (opt-in: decode_lora_one) against the launch pairs, token for token and logit for logit, the eager loop and the replayed
    graph (whose warm-up step, capture and first replay all see the same cache length: the step counter is what tells their launches apart), twice on one captured graph."""
    from ecg_byte_amd.decoder import DecoderConfig, HipCausalLM
    kw = dict(vocab_size=1000, hidden_size=512, intermediate_size=2048, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, head_dim=64, pad_token_id=999)
    if family == "gemma":
        kw.update(model_type="gemma", num_key_value_heads=1, head_dim=256, num_attention_heads=4, rms_norm_eps=1e-6, intermediate_size=8192)    # (8 192: the down site's four-wave columns)
    m = HipCausalLM(DecoderConfig(**kw), seed=5)
    m.enable_lora(r=16, alpha=32, dropout=0.05)
    g = torch.Generator(device="cuda").manual_seed(9)
    with torch.no_grad():
        for sites in m.lora:
            for s in sites.values():
                s.B.copy_((torch.randn(s.B.shape, device="cuda", generator=g) * 0.05).to(torch.bfloat16) * s.bmask)
    m.eval()
    ids = torch.randint(0, 990, (1, 300), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    mask = torch.ones(1, 300, device="cuda")
    mask[0, :7] = 0
    outs = {}
    for one in (False, True):
        m.decode_lora_one = one
        seq, logits = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=12, pad_token_id=999, return_logits=True, use_graph=False)
        seq_g = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=12, pad_token_id=999)
        seq_g2 = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=12, pad_token_id=999)     # the captured step replayed by a second call
        assert torch.equal(seq, seq_g) and torch.equal(seq, seq_g2), one
        outs[one] = (seq, logits)
    m.decode_lora_one = False
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])

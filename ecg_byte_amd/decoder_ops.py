"""Thin wrappers over the decoder C ABI (include/ecgbyte_decoder.h): torch tensors in, torch
tensors out, all compute in libecgbyte_hip.so.  bf16 everywhere unless noted."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _L():
    return _lib.lib()


require_current = _lib.require_current


def _bf(t):
    assert t.dtype == torch.bfloat16 and t.is_cuda and t.is_contiguous(), "expects contiguous CUDA bf16"
    return t


def embed_fwd(ids, table, scale=1.0):
    out = torch.empty(ids.shape + (table.shape[1],), dtype=torch.bfloat16, device=table.device)
    _lib.check(_L().ecgb_embed_fwd(_p(ids.contiguous()), _p(_bf(table)), _p(out), ids.numel(), table.shape[1], float(scale), _st()))
    return out


_fuse_rope_fwd = False        # measured at the C3 shape (scripts/dev_rope_fusion.py): 0.382 ms fused against 0.362 ms apart (0.396 / 0.376 with a LoRA pair) -- the
                              # four-wave kernel is slower than the eight-wave one on this short contraction and its epilogue is exposed: off


def set_gemm_rope_fusion(on=True):
    """A/B switch: RoPE's forward in the q|k|v projection's epilogue (ecgb_gemm_nt_bf16_rope, where it can) or as its own pass (default: faster, see above)."""
    global _fuse_rope_fwd
    _fuse_rope_fwd = bool(on)


def gemm_nt_rope(a, b, cos, sin, rope_cols, a2=None, b2=None, alpha=1.0):
    """C = alpha * (A B^T [+ A2 B2^T]) with every head of 64 columns below rope_cols rotated by RoPE (row t with row t of cos / sin [M, 32] fp32): the q|k|v projection
    and ecgb_rope in one launch (ecgb_gemm_nt_bf16_rope), the same bits; where that kernel does not take the shape, the two calls."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    if _fuse_rope_fwd and cos.shape[1] == 32:
        K2 = a2.shape[1] if a2 is not None else 0
        rc = _L().ecgb_gemm_nt_bf16_rope(_p(a), a.stride(0), _p(b), b.stride(0), _p(a2), a2.stride(0) if K2 else 0, _p(b2), b2.stride(0) if K2 else 0, K2,
                                         _p(out), out.stride(0), M, N, K, float(alpha), _p(cos), _p(sin), int(rope_cols), _st())
        if rc == 0:
            return out
        if rc != -3:                                                     # ECGB_ERR_UNSUPPORTED: the two steps apart, below
            _lib.check(rc)
    gemm_nt(a, b, out=out, alpha=alpha, a2=a2, b2=b2)
    return rope_(out, cos, sin, rope_cols // (2 * cos.shape[1]), 2 * cos.shape[1], N)


def set_gemm_w4(on=True):
    """A/B switch: plain NT products of whole 256x256 tiles on the four-wave kernel (default) or on the eight-wave kernels; on=2 also sends the forms there
    that measured faster on eight waves (gate|up + GLU with a LoRA pair)."""
    _lib.check(_L().ecgb_set_gemm_w4(int(on)))


def set_gemm_w4_group_m(g=8):
    """Tile order of the four-wave kernel: blocks of g tile rows (0: row by row)."""
    _lib.check(_L().ecgb_set_gemm_w4_group_m(int(g)))


def set_gemm_w4_min_ktiles(n=128):
    """K-tiles per workgroup from which the GEMM entry points pick the four-wave kernel."""
    _lib.check(_L().ecgb_set_gemm_w4_min_ktiles(int(n)))


def set_gemm_w4_sched(s=1):
    """K-tile schedule of the four-wave kernel: 1 four barriers behind counted waits (default), 0 one rendezvous per K-tile (round 3).  Same bits."""
    _lib.check(_L().ecgb_set_gemm_w4_sched(int(s)))


def gemm_tn_w4(a, b, alpha=1.0, out=None):
    """a [K, M]^T . b [K, N] (both row-major: the weight gradient dY^T . X) on the four-wave kernel (ecgb_gemm_tn_w4_bf16): M, N multiples of 256, K of 64."""
    K, M = a.shape
    N = b.shape[1]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if out is None else out
    _lib.check(_L().ecgb_gemm_tn_w4_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha), _st()))
    return out


def gemm_nn_w4(a, b, alpha=1.0, out=None):
    """a [M, K] . b [K, N] (b row-major) on the four-wave kernel (ecgb_gemm_nn_w4_bf16): M, N multiples of 256, K of 64."""
    M, K = a.shape
    N = b.shape[1]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if out is None else out
    _lib.check(_L().ecgb_gemm_nn_w4_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha), _st()))
    return out


def gemm_nt_w4(a, b, alpha=1.0, out=None):
    """a [M, K] . b [N, K]^T on the four-wave kernel (ecgb_gemm_nt_w4_bf16): M, N multiples of 256, K of 64."""
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device) if out is None else out
    _lib.check(_L().ecgb_gemm_nt_w4_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha), _st()))
    return out


def rope_table(pos, inv_freq):
    """(cos, sin) [n, half] fp32 of float(pos) * inv_freq: the bits of the four torch kernels of `(pos.float()[:, None] * inv_freq[None, :]).cos() / .sin()`, one launch."""
    pos = pos.reshape(-1)
    assert pos.dtype == torch.int64 and inv_freq.dtype == torch.float32 and inv_freq.is_contiguous()
    n, half = pos.numel(), inv_freq.numel()
    cos = torch.empty((n, half), dtype=torch.float32, device=pos.device)
    sin = torch.empty_like(cos)
    _lib.check(_L().ecgb_rope_table(_p(pos.contiguous()), n, _p(inv_freq), half, _p(cos), _p(sin), _st()))
    return cos, sin


def decode_advance_(nxt, tok, pos, col, n_dev, out, mask, unfinished=None, pad_id=0, eos=None, epoch=None):
    """The bookkeeping of one generated token per sequence in one launch (ecgb_decode_advance): finished sequences take pad_id, the token goes to out[:, col] and into
    `tok`, mask[:, col] = 1, pos / col / n_dev advance, `unfinished` drops sequences that produced an eos id.  Everything in place.  epoch (int32[1]): += 1 too
    (ecgb_decode_advance_e: the step counter of the one-launch adapter sites)."""
    B = nxt.shape[0]
    assert nxt.dtype == torch.int64 and out.dtype == torch.int64 and mask.dtype == torch.float32 and n_dev.dtype == torch.int32
    assert out.stride(1) == 1 and mask.stride(1) == 1 and col.numel() == B and tok.numel() == B and pos.numel() == B
    n_eos = 0 if eos is None else int(eos.numel())
    if epoch is not None:
        assert epoch.dtype == torch.int32 and epoch.numel() == 1
        _lib.check(_L().ecgb_decode_advance_e(_p(nxt), B, _p(tok), _p(pos), _p(col), _p(n_dev), _p(out), out.stride(0), _p(mask), mask.stride(0),
                                              _p(unfinished) if n_eos else None, int(pad_id), _p(eos) if n_eos else None, n_eos, _p(epoch), _st()))
        return
    _lib.check(_L().ecgb_decode_advance(_p(nxt), B, _p(tok), _p(pos), _p(col), _p(n_dev), _p(out), out.stride(0), _p(mask), mask.stride(0),
                                        _p(unfinished) if n_eos else None, int(pad_id), _p(eos) if n_eos else None, n_eos, _st()))


def lora_decode_ok(M, N, K, K2=64):
    """Does ecgb_gemm_nt_bf16_lora_decode take the site (one or two rows; both products in the same column-per-wave kernel)?"""
    return M <= 2 and K2 <= 64 and K % 64 == 0 and not (K >= 8192 and K % 32 == 0 and N > 8192)


def gemm_nt_lora_decode(x, w, lora_a, t_scale, lora_b, t64, epoch):
    """A decode step's adapter site in one launch (ecgb_gemm_nt_bf16_lora_decode): y = x W^T + (t_scale * x A^T) B^T, the bits of gemm_nt(x, A, alpha=t_scale) followed by
    gemm_nt(x, w, a2=t, b2=B).  t64: int64[M, K2] owned by the site (t's bf16 bits under the epoch, zeros at first); epoch: int32[1], a value no earlier call on the site had."""
    M, K = x.shape
    N, K2 = w.shape[0], lora_a.shape[0]
    assert lora_a.shape[1] == K and lora_b.shape == (N, K2) and t64.dtype == torch.int64 and t64.numel() >= M * K2 and epoch.dtype == torch.int32
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    _lib.check(_L().ecgb_gemm_nt_bf16_lora_decode(_p(x), x.stride(0), _p(w), w.stride(0), _p(lora_a), lora_a.stride(0), float(t_scale), _p(lora_b), lora_b.stride(0), K2,
                                                   _p(t64), _p(y), y.stride(0), M, N, K, _p(epoch), _st()))
    return y


def argmax_rows(x, n=None):
    """torch.argmax(x[:, :n], -1) of a bf16 matrix (first index of the maximum), one launch without workspace (ecgb_argmax_bf16)."""
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    n = x.shape[1] if n is None else int(n)
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    _lib.check(_L().ecgb_argmax_bf16(_p(x), x.stride(0), x.shape[0], n, _p(out), _st()))
    return out


def embed_bwd_sorted_(ids, dout, grad_table_bf16, scale=1.0, skip_id=-1):
    """grad_table_bf16[id] += scale * sum of dout rows with that id, in place, no atomics (ecgb_embed_bwd_sorted): the same bits every call."""
    flat = ids.reshape(-1)
    ids_sorted, order = torch.sort(flat, stable=True)
    assert grad_table_bf16.dtype == torch.bfloat16 and grad_table_bf16.is_contiguous()
    _lib.check(_L().ecgb_embed_bwd_sorted(_p(ids_sorted), _p(order), _p(_bf(dout)), _p(grad_table_bf16), flat.numel(), dout.shape[-1], float(scale),
                                          int(skip_id), _st()))
    return grad_table_bf16


def embed_bwd(ids, dout, grad_table_f32, scale=1.0):
    _lib.check(_L().ecgb_embed_bwd(_p(ids.contiguous()), _p(_bf(dout)), _p(grad_table_f32), ids.numel(), dout.shape[-1], float(scale), _st()))


def rmsnorm_fwd(x, w, eps, residual=None, gemma=False, lora=None):
    """Returns (y, rstd, x_sum): x_sum = x + residual when residual is given (else x itself).
    lora = (A [n, H] bf16, scale), up to eight rows (a decode step with adapters): also returns t = gemm_nt(y, A, alpha=scale) as a fourth value, computed in the
    same launch (ecgb_rmsnorm_lora_fwd; the same bits as the few-row GEMM)."""
    H = x.shape[-1]
    rows = x.numel() // H
    y = torch.empty_like(x)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    s = torch.empty_like(x) if residual is not None else None
    if lora is not None:
        A, scale = lora
        if rows <= 8 and H % 512 == 0 and A.stride(1) == 1:
            t = torch.empty((rows, A.shape[0]), dtype=torch.bfloat16, device=x.device)
            _lib.check(_L().ecgb_rmsnorm_lora_fwd(_p(_bf(x)), _p(residual), _p(_bf(w)), _p(y), _p(s), _p(rstd), rows, H, float(eps), int(gemma),
                                                  _p(A), A.stride(0), A.shape[0], float(scale), _p(t), t.stride(0), _st()))
            return y, rstd, (s if residual is not None else x), t
        _lib.check(_L().ecgb_rmsnorm_fwd(_p(_bf(x)), _p(residual), _p(_bf(w)), _p(y), _p(s), _p(rstd), rows, H, float(eps), int(gemma), _st()))
        return y, rstd, (s if residual is not None else x), gemm_nt(y.view(rows, H), A, alpha=scale)
    _lib.check(_L().ecgb_rmsnorm_fwd(_p(_bf(x)), _p(residual), _p(_bf(w)), _p(y), _p(s), _p(rstd), rows, H, float(eps), int(gemma), _st()))
    return y, rstd, (s if residual is not None else x)


def set_rmsnorm_fwd_rows(on=True):
    """A/B switch: rmsnorm_fwd at hidden 2048 with the row kept in registers (default) or the generic kernel.  Same bits."""
    _lib.check(_L().ecgb_set_rmsnorm_fwd_rows(int(bool(on))))


def rmsnorm_bwd(x, w, rstd, dy, dw_f32, dres=None, gemma=False):
    """dx (+ dres) of RMSNorm; dw_f32 [H] receives += the weight gradient, or None: frozen norm weights (LoRA), nothing is computed for them."""
    H = x.shape[-1]
    dx = torch.empty_like(x)
    rows = x.numel() // H
    nf = _L().ecgb_rmsnorm_bwd_scratch_floats(rows, H) if dw_f32 is not None else 0     # per-workgroup partial rows of dw, added in order (the same bits every launch)
    scratch = torch.empty(nf, dtype=torch.float32, device=x.device) if nf else None
    _lib.check(_L().ecgb_rmsnorm_bwd(_p(_bf(x)), _p(_bf(w)), _p(rstd), _p(_bf(dy)), _p(dres), _p(dx), _p(dw_f32), rows, H, int(gemma), _p(scratch), _st()))
    return dx


def layernorm_fwd(x, w, b, eps, residual=None):
    """nn.LayerNorm.  Returns (y, mean, rstd, x_sum): x_sum = x + residual when residual is given (else x itself)."""
    H = x.shape[-1]
    rows = x.numel() // H
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    s = torch.empty_like(x) if residual is not None else None
    _lib.check(_L().ecgb_layernorm_fwd(_p(_bf(x)), _p(residual), _p(_bf(w)), _p(_bf(b)), _p(y), _p(s), _p(mean), _p(rstd), rows, H, float(eps), _st()))
    return y, mean, rstd, (s if residual is not None else x)


def layernorm_bwd(x, w, mean, rstd, dy, dw_f32, db_f32, dres=None):
    H = x.shape[-1]
    dx = torch.empty_like(x)
    rows = x.numel() // H
    scratch = torch.empty(_L().ecgb_layernorm_bwd_scratch_floats(rows, H), dtype=torch.float32, device=x.device)    # ordered dw / db sums
    _lib.check(_L().ecgb_layernorm_bwd(_p(_bf(x)), _p(_bf(w)), _p(mean), _p(rstd), _p(_bf(dy)), _p(dres), _p(dx), _p(dw_f32), _p(db_f32),
                                       rows, H, _p(scratch), _st()))
    return dx


def bias_(u, bias):
    """u += bias (broadcast over rows), in place."""
    N = u.shape[-1]
    _lib.check(_L().ecgb_bias_act(_p(_bf(u)), _p(_bf(bias)), None, u.numel() // N, N, 0, _st()))
    return u


def bias_gelu_new_(u, bias):
    """u += bias in place (the pre-activation the backward needs); returns gelu_new(u)."""
    N = u.shape[-1]
    h = torch.empty_like(u)
    _lib.check(_L().ecgb_bias_act(_p(_bf(u)), _p(_bf(bias)), _p(h), u.numel() // N, N, 1, _st()))
    return h


def gelu_new_bwd(pre, dh):
    d = torch.empty_like(pre)
    _lib.check(_L().ecgb_gelu_new_bwd(_p(_bf(pre)), _p(_bf(dh)), _p(d), pre.numel(), _st()))
    return d


def colsum(dy):
    """fp32 column sums of a 2-D bf16 tensor (a bias gradient)."""
    out = torch.zeros(dy.shape[-1], dtype=torch.float32, device=dy.device)
    rows, n = dy.numel() // dy.shape[-1], dy.shape[-1]
    scratch = torch.empty(_L().ecgb_colsum_scratch_floats(rows, n), dtype=torch.float32, device=dy.device)          # ordered partial rows
    _lib.check(_L().ecgb_colsum(_p(_bf(dy)), _p(out), rows, n, _p(scratch), _st()))
    return out


def rope_(x, cos, sin, n_heads, head_dim, row_stride, inverse=False):
    """In place on a [tokens, ...] view whose heads start at x.data_ptr(): x may be a slice of a fused qkv buffer."""
    tokens = cos.shape[0]
    _lib.check(_L().ecgb_rope(_p(x), _p(cos), _p(sin), tokens, n_heads, head_dim, row_stride, int(inverse), _st()))
    return x


def rope_append_(qkv, cos, sin, Hq, Hkv, D, cache, kv_len):
    """A decode step's rope_ on the q and k heads of qkv [B, (Hq + 2 Hkv) * D] (in place) and the append of the rotated k and of v at row kv_len - 1 of cache
    [B, cap, 2 * Hkv * D], one launch (ecgb_rope_append).  kv_len: an int, or an int32[1] device tensor (a replayed graph)."""
    B, cap, W = cache.shape
    assert W == 2 * Hkv * D and cache.is_contiguous()
    dyn = torch.is_tensor(kv_len)
    _lib.check(_L().ecgb_rope_append(_p(qkv), _p(cos), _p(sin), B, Hq, Hkv, D, qkv.stride(0), _p(cache), cap, 0 if dyn else int(kv_len),
                                     _p(kv_len) if dyn else None, _st()))
    return qkv


def glu_fwd(gate_up, gelu_tanh=False):
    inter = gate_up.shape[-1] // 2
    h = torch.empty(gate_up.shape[:-1] + (inter,), dtype=torch.bfloat16, device=gate_up.device)
    _lib.check(_L().ecgb_glu_fwd(_p(_bf(gate_up)), _p(h), gate_up.numel() // (2 * inter), inter, int(gelu_tanh), _st()))
    return h


def glu_bwd(gate_up, dh, gelu_tanh=False):
    inter = gate_up.shape[-1] // 2
    d = torch.empty_like(gate_up)
    _lib.check(_L().ecgb_glu_bwd(_p(_bf(gate_up)), _p(_bf(dh)), _p(d), gate_up.numel() // (2 * inter), inter, int(gelu_tanh), _st()))
    return d


def add(a, b, out=None):
    """out = a + b (out may be a or b: elementwise)."""
    o = torch.empty_like(a) if out is None else out
    _lib.check(_L().ecgb_add_bf16(_p(_bf(a)), _p(_bf(b)), _p(o), a.numel(), _st()))
    return o


def transpose(x2d):
    R, Cc = x2d.shape
    o = torch.empty((Cc, R), dtype=torch.bfloat16, device=x2d.device)
    _lib.check(_L().ecgb_transpose_bf16(_p(_bf(x2d)), _p(o), R, Cc, _st()))
    return o


class TransposePlan:
    """Pointer / shape tables of `transpose_multi` for a fixed list of 2-D bf16 tensors (built once: the tensors and their transposed
    destinations are persistent buffers updated in place)."""

    def __init__(self, tensors):
        dev = tensors[0].device
        self.src = list(tensors)
        self.dst = [torch.empty((t.shape[1], t.shape[0]), dtype=torch.bfloat16, device=dev) for t in tensors]
        tiles = [((t.shape[0] + 63) // 64) * ((t.shape[1] + 63) // 64) for t in tensors]
        off = [0]
        for n in tiles:
            off.append(off[-1] + n)
        self.total = off[-1]
        self.n = len(tensors)
        self.src_ptrs = torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=dev)
        self.dst_ptrs = torch.tensor([t.data_ptr() for t in self.dst], dtype=torch.int64, device=dev)
        self.rows = torch.tensor([t.shape[0] for t in tensors], dtype=torch.int32, device=dev)
        self.cols = torch.tensor([t.shape[1] for t in tensors], dtype=torch.int32, device=dev)
        self.tile_off = torch.tensor(off[:-1], dtype=torch.int32, device=dev)

    def valid_for(self, tensors):
        return len(tensors) == self.n and all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(tensors, self.src))


def transpose_multi(plan):
    """plan.dst[t] = plan.src[t]^T for every tensor of the plan, one launch."""
    _lib.check(_L().ecgb_transpose_multi_bf16(_p(plan.src_ptrs), _p(plan.dst_ptrs), _p(plan.rows), _p(plan.cols), _p(plan.tile_off), plan.n,
                                              plan.total, _st()))
    return plan.dst


def set_gemm_tile(tile: int):
    """0 = automatic, 128 / 256 = force that output tile (tests, tuning)."""
    _lib.check(_L().ecgb_set_gemm_tile(int(tile)))


def set_gemm_group_m(group_m: int = 0):
    """Tile order of the big GEMM kernels inside an XCD's range: 0 = row by row (default), g = blocks of g tile rows (A/B; measured slower).  Same results."""
    _lib.check(_L().ecgb_set_gemm_group_m(int(group_m)))


def set_gemm_backward_persistent(on: bool):
    """Input-gradient GEMMs with the persistent tile loop (default) or one tile per workgroup: the latter beside a gradient exchange that
    occupies CUs (parallel.GradAllReduce turns it off for world size > 1; ecgb_set_gemm_backward_persistent)."""
    _lib.check(_L().ecgb_set_gemm_backward_persistent(int(bool(on))))


def get_gemm_backward_persistent() -> bool:
    """The switch as it stands (ecgb_get_gemm_backward_persistent)."""
    return bool(_L().ecgb_get_gemm_backward_persistent())


def gemm_nt(a, b, out=None, alpha=1.0, accumulate_f32=False, accumulate=False, a2=None, b2=None):
    """C[M,N] = alpha * A[M,K] @ B[N,K]^T.  a, b: 2-D bf16 (row stride = shape[1] or a column slice view).
    accumulate_f32: `out` is fp32 and receives +=;  accumulate: `out` is bf16 and receives +=.
    a2 [M,K2], b2 [N,K2]: a second operand pair contracted in the same launch, C = alpha * (A B^T + A2 B2^T)."""
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if accumulate_f32 else torch.bfloat16, device=a.device)
    mode = 1 if accumulate_f32 else (2 if accumulate else 0)
    if a2 is not None:
        K2 = a2.shape[1]
        assert a2.shape[0] == M and b2.shape == (N, K2) and a2.stride(1) == 1 and b2.stride(1) == 1
        _lib.check(_L().ecgb_gemm_nt_bf16_cat(_p(a), a.stride(0), _p(b), b.stride(0), _p(a2), a2.stride(0), _p(b2), b2.stride(0), K2,
                                              _p(out), out.stride(0), M, N, K, float(alpha), mode, _st()))      # (few rows: the decode step's kernels, one launch too)
        return out
    _lib.check(_L().ecgb_gemm_nt_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha),
                                      mode, 1, 0, 0, 0, _st()))
    return out


def nn_eligible(M, N, K):
    """Shapes the NN kernel (256x256 tiles only) takes; smaller or ragged-K problems go through gemm_nt on a transposed copy."""
    return M >= 256 and N >= 256 and N % 8 == 0 and K % 64 == 0 and ((M + 255) // 256) * ((N + 255) // 256) >= 192


def gemm_nn(a, b, out=None, alpha=1.0, accumulate_f32=False, accumulate=False):
    """C[M,N] = alpha * A[M,K] @ B[K,N], B row-major (dX = dY W against the weight as stored, [out, in]): no transposed copy of B
    (ecgb_gemm_nn_bf16; the same bits as gemm_nt(a, b.T.contiguous()))."""
    M, K = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if accumulate_f32 else torch.bfloat16, device=a.device)
    mode = 1 if accumulate_f32 else (2 if accumulate else 0)
    _lib.check(_L().ecgb_gemm_nn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, float(alpha), mode, _st()))
    return out


def nn_splitk_plan(M, N, K, n_cu=256):
    """K-slices for gemm_nn_splitk, or 0: few 256x256 output tiles and a long contraction (the loss head's input gradient: ~1 300 labelled rows x hidden
    over the vocabulary)."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if M < 256 or N < 256 or N % 8 or K % 64 or tiles >= 192:
        return 0
    # the slice count depends on the contraction length ALONE (about 384 K-tiles a slice: 6 slices over a 132 608-entry vocabulary): a row's sum is
    # then formed in the same order whatever the number of rows -- the labelled-rows loss head and the full-logits one give the same bits
    splits = min(-(-(K // 64) // 384), 64)
    return splits if splits >= 2 else 0


def gemm_nn_splitk(a, b, splits, alpha=1.0):
    """C[M,N] = alpha * A[M,K] @ B[K,N] (B row-major) with the contraction in `splits` K-slices: fp32 slabs summed in slice order (ecgb_gemm_nn_splitk_bf16;
    deterministic)."""
    M, K = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and a.stride(1) == 1 and b.stride(1) == 1
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    slabs = torch.empty((splits, M, N), dtype=torch.float32, device=a.device)
    _lib.check(_L().ecgb_gemm_nn_splitk_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), _p(slabs), M, N, K, int(splits), float(alpha), _st()))
    return out


def nn_glu_bwd_eligible(M, inter, K):
    """Shapes the NN kernel with the GLU backward in its epilogue takes: whole 256x256 tiles (the epilogue pairs every element of the product with
    its gate / up), enough of them to fill the chip."""
    return M % 256 == 0 and inter % 256 == 0 and K % 64 == 0 and (M // 256) * (inter // 256) >= 192


def gemm_nn_glu_bwd(dy, w, gate_up, gelu_tanh=False):
    """d(gate|up) [M, 2I] = glu_bwd(gate|up, dy @ w) with w = [K, I] (the down projection as stored): the input gradient of the down projection and the
    GLU backward in one launch (ecgb_gemm_nn_glu_bwd_bf16; the same bits as gemm_nn followed by glu_bwd, d(act(gate) * up) is never written)."""
    M, K = dy.shape
    inter = w.shape[1]
    assert w.shape[0] == K and gate_up.shape == (M, 2 * inter) and dy.stride(1) == 1 and w.stride(1) == 1 and gate_up.stride(1) == 1
    d = torch.empty_like(gate_up)
    _lib.check(_L().ecgb_gemm_nn_glu_bwd_bf16(_p(_bf(dy)), dy.stride(0), _p(_bf(w)), w.stride(0), _p(_bf(gate_up)), gate_up.stride(0), _p(d), d.stride(0),
                                              M, inter, K, int(gelu_tanh), _st()))
    return d


def glu_fusable(M, inter):
    """Shapes the GLU-epilogue GEMM takes: the 256x256 tile needs whole tiles of 128 gate + 128 up columns and enough of them to fill
    the chip; up to eight rows (a decode step) go through the few-row kernel with the GLU folded in."""
    if M <= 8:
        return inter % 128 == 0                              # a decode step: the few-row kernel with the GLU folded in (gemm_nt_skinny_glu_kernel)
    return M >= 256 and inter % 128 == 0 and ((M + 255) // 256) * (2 * inter // 256) >= 192


def gemm_nt_glu(a, b, gelu_tanh=False, keep_gu=True, a2=None, b2=None, alpha=1.0):
    """(gate|up, h):  gate|up [M, 2I] = alpha * (A B^T [+ A2 B2^T]) with B = [2I, K] (gate rows, then up rows) and h [M, I] =
    act(gate) * up computed in the GEMM's epilogue (ecgb_gemm_nt_glu_bf16); keep_gu=False: gate|up is not written (returns None)."""
    M, K = a.shape
    inter = b.shape[0] // 2
    assert b.shape == (2 * inter, K) and a.stride(1) == 1 and b.stride(1) == 1
    gu = torch.empty((M, 2 * inter), dtype=torch.bfloat16, device=a.device) if keep_gu else None
    h = torch.empty((M, inter), dtype=torch.bfloat16, device=a.device)
    K2 = 0
    if a2 is not None:
        K2 = a2.shape[1]
        assert a2.shape[0] == M and b2.shape == (2 * inter, K2) and a2.stride(1) == 1 and b2.stride(1) == 1
    _lib.check(_L().ecgb_gemm_nt_glu_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(a2), a2.stride(0) if K2 else 0, _p(b2),
                                          b2.stride(0) if K2 else 0, K2, _p(gu), 2 * inter, _p(h), inter, M, inter, K, float(alpha),
                                          int(gelu_tanh), _st()))
    return gu, h


def dropout(x, p, seed, out=None):
    """Inverted dropout; the same (p, seed) reproduces the mask (call it on the gradient for the backward)."""
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_L().ecgb_dropout_bf16(_p(_bf(x)), _p(out), x.numel(), float(p), int(seed), _st()))
    return out


def lora_down(x, A, n_sub, n_fields, scale, p=0.0, seed=0, keep_masked=False):
    """t = scale / (1 - p) * (mask_f . x) A_f^T for the stacked adapters A [64, in] (n_sub 16-row sub-blocks) of n_fields modules, each
    with its own dropout mask (ecgb_lora_down).  Returns (t [T, 64], masked copies of x [n_fields, T, in] or None)."""
    T, K = x.shape
    t = torch.empty((T, 64), dtype=torch.bfloat16, device=x.device)
    xd = torch.empty((n_fields, T, K), dtype=torch.bfloat16, device=x.device) if keep_masked and p > 0 else None
    _lib.check(_L().ecgb_lora_down(_p(_bf(x)), _p(_bf(A)), _p(t), _p(xd), T, K, n_sub, n_fields, float(scale), float(p), int(seed), _st()))
    return t, xd


def lora_dx_(dx, dt, At, n_sub, n_fields, scale, p=0.0, seed=0):
    """dx += scale / (1 - p) * sum_f mask_f . (dt_f A_f), in place; dt [T, 64], At = A^T [in, 64] (ecgb_lora_dx)."""
    T, K = dx.shape
    assert dt.shape == (T, 64) and At.shape == (K, 64)
    _lib.check(_L().ecgb_lora_dx(_p(_bf(dt)), _p(_bf(At)), _p(_bf(dx)), T, K, n_sub, n_fields, float(scale), float(p), int(seed), _st()))
    return dx


def lora_da(x, dt, out, n_sub, n_fields, scale, p=0.0, seed=0, accumulate=False):
    """out [16 * n_sub, in] (bf16, contiguous: the first rows of the stacked A's gradient) (+)= scale / (1 - p) * dt[:, :16 n_sub]^T . (mask_f . x),
    the dropout masks of ecgb_lora_down evaluated again from (seed, element index) (ecgb_lora_da): one pass over x for all modules of the site,
    no masked copies of x kept by the forward; row chunks meet in fp32 slabs added in order."""
    T, K = x.shape
    assert dt.shape == (T, 64) and out.shape == (16 * n_sub, K) and out.dtype == torch.bfloat16 and out.is_contiguous()
    nbytes = _L().ecgb_lora_da_scratch_bytes(T, K, n_sub)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    _lib.check(_L().ecgb_lora_da(_p(_bf(x)), _p(_bf(dt)), _p(out), T, K, n_sub, n_fields, float(scale), float(p), int(seed), int(accumulate),
                                 _p(scratch), nbytes, _st()))
    return out


def lora_dx_glu(dx, dt, At, gate_up, scale, p=0.0, seed=0, gelu_tanh=False):
    """d(gate|up) = glu_bwd(gate|up, dx + scale / (1 - p) * mask . (dt A)) in one pass (ecgb_lora_dx_glu: the down-projection site, one
    adapter block); dx [T, I] is only read."""
    T, inter = dx.shape
    assert dt.shape == (T, 64) and At.shape == (inter, 64) and gate_up.shape == (T, 2 * inter)
    d = torch.empty_like(gate_up)
    _lib.check(_L().ecgb_lora_dx_glu(_p(_bf(dt)), _p(_bf(At)), _p(_bf(dx)), _p(_bf(gate_up)), _p(d), T, inter, 1, 1, float(scale), float(p), int(seed),
                                     int(gelu_tanh), _st()))
    return d


_fuse_lora_dx_glu = True


def set_fuse_lora_dx_glu(on=True):
    """A/B switch: a single-module site's input gradient with the adapter share in the GEMM's epilogue (down: gemm_nn_glu_bwd_lora, with the GLU backward; o:
    gemm_nn_lora) or as gemm_nn + lora_dx_glu / lora_dx_."""
    global _fuse_lora_dx_glu
    _fuse_lora_dx_glu = bool(on)


def gemm_nn_glu_bwd_lora(dy, w, gate_up, dt, At, scale, p=0.0, seed=0, gelu_tanh=False):
    """d(gate|up) [M, 2I] = glu_bwd(gate|up, bf16(dy @ w) + scale / (1 - p) * mask . (dt A)) in ONE launch (ecgb_gemm_nn_glu_bwd_lora_bf16: the down-projection
    site of a LoRA fine-tune -- gemm_nn followed by lora_dx_glu, the same bits, d(act(gate) * up) never written), or None where the four-wave kernel does not
    take the shape (the caller runs the two)."""
    M, K = dy.shape
    inter = w.shape[1]
    assert w.shape[0] == K and gate_up.shape == (M, 2 * inter) and dt.shape == (M, 64) and At.shape == (inter, 64)
    assert dy.stride(1) == 1 and w.stride(1) == 1 and gate_up.stride(1) == 1 and dt.is_contiguous() and At.is_contiguous()
    if not _fuse_lora_dx_glu:
        return None
    d = torch.empty_like(gate_up)
    rc = _L().ecgb_gemm_nn_glu_bwd_lora_bf16(_p(_bf(dy)), dy.stride(0), _p(_bf(w)), w.stride(0), _p(_bf(gate_up)), gate_up.stride(0), _p(_bf(dt)), _p(_bf(At)),
                                             _p(d), d.stride(0), M, inter, K, int(gelu_tanh), float(scale), float(p), int(seed), _st())
    if rc == -3:                                                         # ECGB_ERR_UNSUPPORTED
        return None
    _lib.check(rc)
    return d


def gemm_nn_lora(dy, w, dt, At, scale, p=0.0, seed=0):
    """dx [M, in] = bf16(dy @ w) + scale / (1 - p) * mask . (dt A) in ONE launch (ecgb_gemm_nn_lora_bf16: the input gradient of a frozen projection with one LoRA
    module on it -- gemm_nn followed by lora_dx_, the same bits), or None where the four-wave kernel does not take the shape / the switch is off."""
    M, K = dy.shape
    n_in = w.shape[1]
    assert w.shape[0] == K and dt.shape == (M, 64) and At.shape == (n_in, 64) and dy.stride(1) == 1 and w.stride(1) == 1 and dt.is_contiguous() and At.is_contiguous()
    if not _fuse_lora_dx_glu:
        return None
    dx = torch.empty((M, n_in), dtype=torch.bfloat16, device=dy.device)
    rc = _L().ecgb_gemm_nn_lora_bf16(_p(_bf(dy)), dy.stride(0), _p(_bf(w)), w.stride(0), _p(_bf(dt)), _p(_bf(At)), _p(dx), dx.stride(0), M, n_in, K,
                                     float(scale), float(p), int(seed), _st())
    if rc == -3:                                                         # ECGB_ERR_UNSUPPORTED
        return None
    _lib.check(rc)
    return dx


def tn_splits(N, K, M, n_cu=256):
    """K-slices per 256x256 output tile of a weight-gradient product.  Skinny outputs (LoRA adapter gradients) are a pass over the long
    operand: enough workgroups to pull it at HBM speed.  Otherwise the slice count that minimises the makespan in contraction steps when
    tiles * s workgroups are dealt over the compute units one each (measured on dW of qkv [3072 x 2048] and o [2048 x 2048] at 32 768
    rows: 5 and 4 slices, 0.49 and 0.31 ms against 0.99 and 0.96 unsplit), plus ~tiles / 20 steps per slice for writing and re-reading
    its fp32 slab."""
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    kt = max(1, M // 64)
    if min(N, K) <= 64:
        return max(1, min(32, n_cu // tiles, kt))
    best = None
    for s in range(1, min(8, kt) + 1):
        cost = -(-tiles * s // n_cu) * -(-kt // s) + s * tiles / 20.0
        if best is None or cost < best[0] - 1e-9:
            best = (cost, s)
    return best[1]


def sum_slabs(slabs, out, accumulate=False):
    """out (bf16, contiguous) = [out +] sum over the leading axis of slabs (fp32 [n_slabs, out.numel()]), added in slab order."""
    n_slabs = slabs.shape[0]
    _lib.check(_L().ecgb_sum_slabs_bf16(_p(slabs), slabs.stride(0), n_slabs, _p(out), out.numel(), int(accumulate), _st()))
    return out


def gemm_tn(a, b, alpha=1.0, splits=None, out=None, accumulate=False):
    """C[N,K] = alpha * A[M,N]^T @ B[M,K]  (weight gradient dW = dY^T X) without transposed copies.
    splits: workgroups sharing one output tile's contraction (None: enough to fill the chip); each writes its fp32 partial product to
    its own slab and one pass sums them in slice order (deterministic).
    out: contiguous bf16 [N, K] destination (e.g. a parameter's slice of the flat gradient buffer); accumulate: out += result."""
    M, N = a.shape
    K = b.shape[1]
    assert b.shape[0] == M and a.stride(1) == 1 and b.stride(1) == 1
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    if splits is None:
        splits = tn_splits(N, K, M)
    if out is None:
        assert not accumulate
        out = torch.empty((N, K), dtype=torch.bfloat16, device=a.device)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape == (N, K)
    if splits == 1 and not accumulate:
        _lib.check(_L().ecgb_gemm_tn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), K, M, N, K, float(alpha), 1, _st()))
        return out
    if splits == 1:
        tmp = torch.empty((N, K), dtype=torch.bfloat16, device=a.device)
        _lib.check(_L().ecgb_gemm_tn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(tmp), K, M, N, K, float(alpha), 1, _st()))
        return add(out, tmp, out=out)
    slabs = torch.empty((splits, N * K), dtype=torch.float32, device=a.device)
    _lib.check(_L().ecgb_gemm_tn_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(slabs), K, M, N, K, float(alpha), splits, _st()))
    return sum_slabs(slabs, out, accumulate)


def gemm_nt_heads(a, lda, b, ldb, c, ldc, M, N, K, alpha, batch, inner, outer_a, inner_a, div_a, outer_b, inner_b, div_b,
                  outer_c, inner_c, accumulate_f32=False):
    """a, b, c: tensors or (tensor, element_offset) pairs."""
    def ptr(x):
        if isinstance(x, tuple):
            return C.c_void_p(x[0].data_ptr() + x[1] * x[0].element_size())
        return _p(x)
    _lib.check(_L().ecgb_gemm_nt_bf16_heads(ptr(a), lda, ptr(b), ldb, ptr(c), ldc, M, N, K, float(alpha), int(accumulate_f32), batch,
                                            inner, outer_a, inner_a, div_a, outer_b, inner_b, div_b, outer_c, inner_c, _st()))


def transpose_strided(src, src_off, dst, dst_off, rows, cols, ld_in, ld_out, batch, inner, outer_in, inner_in, outer_out, inner_out):
    _lib.check(_L().ecgb_transpose_bf16_strided(C.c_void_p(src.data_ptr() + 2 * src_off), C.c_void_p(dst.data_ptr() + 2 * dst_off),
                                                rows, cols, ld_in, ld_out, batch, inner, outer_in, inner_in, outer_out, inner_out, _st()))


def softmax_causal_fwd_(scores, attn_mask, n_heads, scale):
    BH, S, _ = scores.shape
    _lib.check(_L().ecgb_softmax_causal_fwd(_p(_bf(scores)), _p(attn_mask), BH, n_heads, S, float(scale), _st()))
    return scores


def softmax_bwd_(p, dp, scale):
    BH, S, _ = p.shape
    _lib.check(_L().ecgb_softmax_bwd(_p(_bf(p)), _p(_bf(dp)), BH, S, float(scale), _st()))
    return dp


def count_labels(labels, vocab):
    inv = torch.empty(1, dtype=torch.float32, device=labels.device)
    _lib.check(_L().ecgb_count_labels(_p(labels), labels.numel(), vocab, _p(inv), _st()))
    return inv


def ce_fwd_bwd_(logits, labels, inv_count, sum_loss, vocab):
    """logits [rows, ld] bf16 overwritten with dlogits; returns per-row losses (fp32)."""
    rows, ld = logits.shape
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    _lib.check(_L().ecgb_ce_fwd_bwd(_p(_bf(logits)), _p(labels), _p(row_loss), None, _p(inv_count), rows, vocab, ld, _st()))
    sum_loss.add_(row_loss.sum() * inv_count)               # in a fixed order (the kernel's own float atomics add in order of arrival)
    return row_loss


def sumsq(g, acc):
    _lib.check(_L().ecgb_sumsq(_p(g), g.numel(), int(g.dtype == torch.float32), _p(acc), _st()))


class SumsqPlan:
    """Chunk table of `sumsq_multi` for a fixed list of tensor sizes (built once, reused every step)."""

    def __init__(self, counts, device):
        ct, co = [], []
        for t, n in enumerate(counts):
            for off in range(0, n, 1 << 20):
                ct.append(t)
                co.append(off)
        self.counts_host = list(counts)
        self.counts = torch.tensor(counts, dtype=torch.int64, device=device)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=device)
        self.chunk_off = torch.tensor(co, dtype=torch.int64, device=device)
        self.n_chunks = len(ct)
        self.partials = torch.empty(len(ct), dtype=torch.float32, device=device)       # per-chunk sums, added in chunk order


def sumsq_multi(tensors, acc, plan):
    """acc += sum over all (bf16, contiguous) tensors of their squared elements, one launch."""
    ptrs = torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64).to(acc.device, non_blocking=True)
    _lib.check(_L().ecgb_sumsq_multi_bf16(_p(ptrs), _p(plan.counts), _p(plan.chunk_tensor), _p(plan.chunk_off), plan.n_chunks, _p(acc),
                                          _p(plan.partials), _st()))


def adam_step_(p, g, m, v, sumsq_acc, max_norm, lr, beta1, beta2, eps, weight_decay, step):
    _lib.check(_L().ecgb_adam_step(_p(p), _p(g), int(g.dtype == torch.float32), _p(m), _p(v), p.numel(), _p(sumsq_acc),
                                   float(max_norm), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                   int(step), _st()))


class AdamMultiPlan:
    """Pointer tables (parameters, gradients, moments) of `adam_multi_` for a fixed list of tensors; rebuilt by the caller when an address changes."""

    def __init__(self, params, grads, ms, vs, device):
        self.key = tuple(t.data_ptr() for ts in (params, grads, ms, vs) for t in ts)
        tab = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64).to(device)
        self.p, self.g, self.m, self.v = tab(params), tab(grads), tab(ms), tab(vs)

    def matches(self, params, grads, ms, vs):
        return self.key == tuple(t.data_ptr() for ts in (params, grads, ms, vs) for t in ts)


def adam_multi_(tables, chunks, sumsq_acc, max_norm, lr, beta1, beta2, eps, weight_decay, step):
    """One launch of the Adam step over every tensor of `tables` (AdamMultiPlan; bf16 parameters and gradients, fp32 moments, all contiguous); `chunks` is the
    SumsqPlan of the same tensor sizes.  The same bits as adam_step_ per tensor."""
    _lib.check(_L().ecgb_adam_multi_bf16(_p(tables.p), _p(tables.g), _p(tables.m), _p(tables.v), _p(chunks.counts), _p(chunks.chunk_tensor), _p(chunks.chunk_off),
                                         chunks.n_chunks, _p(sumsq_acc), float(max_norm), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                         int(step), _st()))


FUSED_HEAD_DIMS = (64, 128, 256)     # head dims ecgb_attn_fwd / ecgb_attn_bwd take


def _off(t, off):
    return C.c_void_p(t.data_ptr() + off * t.element_size())


def attn_fwd(qkv, mask, B, S, Hq, Hkv, D, scale):
    """Fused attention on a fused qkv buffer [B*S, Hq*D + 2*Hkv*D] (q | k | v).  Returns (o [B*S, Hq*D], lse)."""
    QKV = qkv.shape[1]
    o = torch.empty((B * S, Hq * D), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, Hq, S), dtype=torch.float32, device=qkv.device)
    _lib.check(_L().ecgb_attn_fwd(_off(qkv, 0), QKV, _off(qkv, Hq * D), QKV, _off(qkv, Hq * D + Hkv * D), QKV, _p(mask),
                                  _p(o), Hq * D, _p(lse), B, S, Hq, Hkv, D, float(scale), _st()))
    return o, lse


def set_stream_grid_cap(n=1 << 20):
    """A/B: the most workgroups glu_fwd / glu_bwd launch (rounds 2-5: 4096)."""
    _lib.check(_L().ecgb_set_stream_grid_cap(int(n)))


def set_ce_in_registers(on: bool = True):
    """ecgb_ce_fwd_bwd: hold each row of logits in registers (one read, one write; default) or run the three-sweep kernel (A/B, tests)."""
    _lib.check(_L().ecgb_set_ce_in_registers(int(bool(on))))


def set_attn_lean_waves(waves: int = 4):
    """Waves per workgroup of the lean head_dim-64 attention kernels (4: default, 8: A/B)."""
    _lib.check(_L().ecgb_set_attn_lean_waves(int(waves)))


def set_attn_d256_pass_p(on: bool = False):
    """head_dim 256 backward (tests, A/B): both passes of the pair kernel form the scores (default), or the dK pass hands its probabilities to a dV kernel through the scratch
    (attn_bwd then allocates B * Hq * S^2 bf16 more a call: 3 % faster at the C5 shape, the same bits)."""
    _lib.check(_L().ecgb_set_attn_d256_pass_p(int(bool(on))))


def set_attn_fwd_staging(mode: int = 2):
    """head_dim 64 forward / backward kernels: 2 = lean LDS-DMA kernels (default: softmax constants in the MFMA accumulators, deferred running
    maximum), 1 = the round-2 LDS-DMA kernels, 0 = the register-staged kernels (tests, A/B).  With 2: | 0x100 / 0x200 / 0x400 keeps the forward /
    dQ / dK-dV kernel alone on mode 1."""
    _lib.check(_L().ecgb_set_attn_fwd_staging(int(mode)))


_fuse_rope_bwd = True


def set_attn_bwd_rope_fusion(on=True):
    """A/B switch: RoPE's backward inside the attention backward kernels' stores (default, where they can: ecgb_attn_bwd_rope) or as its own pass."""
    global _fuse_rope_bwd
    _fuse_rope_bwd = bool(on)


def attn_bwd(qkv, mask, o, do, lse, B, S, Hq, Hkv, D, scale, rope=None):
    """Returns d_qkv with the same fused layout as qkv.  rope = (cos, sin) [B * S, D / 2] fp32: qkv holds the ROTATED q and k and d_qkv comes back as the
    gradient of the unrotated projection -- RoPE's backward in the attention kernels' own stores where they can (ecgb_attn_bwd_rope: head_dim 64, lean kernels),
    otherwise as the separate pass it replaces (ecgb_rope, inverse); the same bits either way."""
    QKV = qkv.shape[1]
    d_qkv = torch.empty_like(qkv)
    delta = torch.empty((B, Hq, S), dtype=torch.float32, device=qkv.device)
    nbytes = _L().ecgb_attn_bwd_scratch_bytes(B, S, Hq, Hkv, D)          # head_dim 256: partial dK / dV slabs (few key blocks) + the probabilities from the dK pass to the dV kernel; else 0
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=qkv.device) if nbytes else None
    if rope is not None and _fuse_rope_bwd and D == 64:
        cos, sin = rope
        rc = _L().ecgb_attn_bwd_rope(_off(qkv, 0), QKV, _off(qkv, Hq * D), QKV, _off(qkv, Hq * D + Hkv * D), QKV, _p(mask),
                                     _p(o), _p(_bf(do)), Hq * D, _p(lse), _p(delta), _off(d_qkv, 0), QKV, _off(d_qkv, Hq * D), QKV,
                                     _off(d_qkv, Hq * D + Hkv * D), QKV, _p(cos), _p(sin), B, S, Hq, Hkv, D, float(scale), _p(scratch), nbytes, _st())
        if rc == 0:
            return d_qkv
        if rc != -3:                                                     # ECGB_ERR_UNSUPPORTED: the two steps apart, below
            _lib.check(rc)
    _lib.check(_L().ecgb_attn_bwd(_off(qkv, 0), QKV, _off(qkv, Hq * D), QKV, _off(qkv, Hq * D + Hkv * D), QKV, _p(mask),
                                  _p(o), _p(_bf(do)), Hq * D, _p(lse), _p(delta), _off(d_qkv, 0), QKV, _off(d_qkv, Hq * D), QKV,
                                  _off(d_qkv, Hq * D + Hkv * D), QKV, B, S, Hq, Hkv, D, float(scale), _p(scratch), nbytes, _st()))
    if rope is not None:
        rope_(d_qkv, rope[0], rope[1], Hq + Hkv, D, QKV, inverse=True)
    return d_qkv


def attn_decode(qkv_new, cache, mask, kv_len, Hq, Hkv, D, scale):
    """One decode step.  qkv_new [B, (Hq+2Hkv)*D] holds the new token's roped q|k|v; cache [B, cap, 2*Hkv*D] (k | v) already
    contains the new token's k, v in row kv_len-1; mask [B, cap] fp32.  Returns o [B, Hq*D]."""
    B, cap, W = cache.shape
    q = qkv_new[:, :Hq * D].contiguous()
    o = torch.empty((B, Hq * D), dtype=torch.bfloat16, device=q.device)
    _lib.check(_L().ecgb_attn_decode(_p(q), _off(cache, 0), _off(cache, Hkv * D), W, cap, _p(mask), mask.stride(0), _p(o),
                                     B, int(kv_len), Hq, Hkv, D, float(scale), _st()))
    return o


_split_scratch = {}


def decode_splits(kv_len, B, Hq):
    """Workgroups a head's keys are split over in a decode step: about 64 keys each (a wave then needs one batch of key
    rows), at most 32 per head and 1024 in all; 1 = the single-workgroup kernel (short caches: three launches cost more
    than they save).  generate() passes the CAPACITY of its caches (prompt + new tokens, rounded up to 128 rows), not the keys valid
    at a step: one split count for the whole call, the same in the eager loop and in the replayed graph -- the same bits."""
    if kv_len < 512:
        return 1
    return int(max(1, min(kv_len // 64, 32, max(1, 1024 // (B * Hq)))))


def decode_split_scratch(cap, B, Hq, D, n_splits, device):
    """The fp32 scratch of attn_decode_split for a cache of `cap` rows (scores, per-split statistics, partial outputs)."""
    return torch.empty(_L().ecgb_attn_decode_split_scratch_bytes(cap, B, Hq, D, n_splits), dtype=torch.uint8, device=device)


def attn_decode_split(qkv_new, cache, mask, kv_len, Hq, Hkv, D, scale, n_splits, scratch=None):
    """attn_decode with the keys of every head split over n_splits workgroups (ecgb_attn_decode_split).  kv_len: an int, or an int32[1] device
    tensor (ecgb_attn_decode_split_dyn: replayable from a captured graph; the same bits for the same n_splits).  scratch: a caller-owned buffer
    (decode_split_scratch) -- a captured graph must own the one it was captured with; otherwise one cached buffer per shape."""
    B, cap, W = cache.shape
    q = qkv_new[:, :Hq * D].contiguous()
    o = torch.empty((B, Hq * D), dtype=torch.bfloat16, device=q.device)
    need = _L().ecgb_attn_decode_split_scratch_bytes(cap, B, Hq, D, n_splits)
    buf = scratch
    if buf is None:
        key = (q.device, need)
        buf = _split_scratch.get(key)
        if buf is None:
            _split_scratch.clear()                       # one live buffer: the shapes of a generate() call do not change
            buf = _split_scratch[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
    assert buf.numel() >= need
    if torch.is_tensor(kv_len):
        _lib.check(_L().ecgb_attn_decode_split_dyn(_p(q), _off(cache, 0), _off(cache, Hkv * D), W, cap, _p(mask), mask.stride(0), _p(o),
                                                   B, _p(kv_len), Hq, Hkv, D, float(scale), int(n_splits), _p(buf), buf.numel(), _st()))
    else:
        _lib.check(_L().ecgb_attn_decode_split(_p(q), _off(cache, 0), _off(cache, Hkv * D), W, cap, _p(mask), mask.stride(0), _p(o),
                                               B, int(kv_len), Hq, Hkv, D, float(scale), int(n_splits), _p(buf), buf.numel(), _st()))
    return o


def decode_one_ok(B, Hq, D, n_splits, cap):
    """Does the one-launch decode attention (ecgb_attn_decode_one) take this step?  Its workgroups wait for each other inside the launch: all resident at once."""
    return n_splits > 1 and n_splits <= 64 and n_splits * Hq * B <= 2 * _n_cus() and D in (64, 128, 256) and -(-cap // n_splits) <= 2048


_n_cus_cached = {}


def _n_cus():
    dev = torch.cuda.current_device()
    if dev not in _n_cus_cached:
        _n_cus_cached[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
    return _n_cus_cached[dev]


def decode_one_scratch(B, Hq, D, n_splits, device):
    """Scratch of attn_decode_one: the splits' statistics (the bit pattern 0x7FC0DEAD = not stored yet), partial outputs, the counters (zero); every launch leaves them as it found them."""
    buf = torch.zeros(_L().ecgb_attn_decode_one_scratch_floats(B, Hq, D, n_splits), dtype=torch.float32, device=device)
    buf.view(torch.int32)[: B * Hq * n_splits * 2] = 0x7FC0DEAD
    return buf


_one_scratch = {}


def attn_decode_one(qkv_new, cos, sin, cache, mask, kv_len, Hq, Hkv, D, scale, n_splits, scratch=None):
    """rope_append_ + attn_decode_split in ONE launch (ecgb_attn_decode_one): RoPE of the new token's q and k, its rotated k and its v into cache row kv_len - 1,
    softmax(q K^T scale + mask) V over the kv_len cached keys; qkv_new is left as it is.  The same bits as the separate launches with the same n_splits.
    kv_len: an int or an int32[1] device tensor; scratch: a caller-owned buffer (decode_one_scratch) -- a captured graph must own the one it was captured with."""
    B, cap, W = cache.shape
    o = torch.empty((B, Hq * D), dtype=torch.bfloat16, device=qkv_new.device)
    buf = scratch
    if buf is None:
        key = (qkv_new.device, B, Hq, D, n_splits)
        buf = _one_scratch.get(key)
        if buf is None:
            _one_scratch.clear()
            buf = _one_scratch[key] = decode_one_scratch(B, Hq, D, n_splits, qkv_new.device)
    dyn = torch.is_tensor(kv_len)
    _lib.check(_L().ecgb_attn_decode_one(_p(_bf(qkv_new)), qkv_new.stride(0), _p(cos), _p(sin), _p(cache), W, cap, _p(mask), mask.stride(0), _p(o), B,
                                         0 if dyn else int(kv_len), _p(kv_len) if dyn else None, Hq, Hkv, D, float(scale), int(n_splits), _p(buf), buf.numel(), _st()))
    return o


def attn_decode_dyn(qkv_new, cache, mask, kv_len_dev, Hq, Hkv, D, scale):
    """attn_decode with the number of valid cache rows in device memory (int32[1]): replayable from a captured graph."""
    B, cap, W = cache.shape
    q = qkv_new[:, :Hq * D].contiguous()
    o = torch.empty((B, Hq * D), dtype=torch.bfloat16, device=q.device)
    _lib.check(_L().ecgb_attn_decode_dyn(_p(q), _off(cache, 0), _off(cache, Hkv * D), W, cap, _p(mask), mask.stride(0), _p(o),
                                         B, _p(kv_len_dev), Hq, Hkv, D, float(scale), _st()))
    return o


def kv_append(qkv_new, col_off, cache, kv_len_dev):
    """cache[:, *kv_len_dev - 1] = qkv_new[:, col_off : col_off + cache.shape[2]]"""
    B, cap, W = cache.shape
    _lib.check(_L().ecgb_kv_append(_p(qkv_new), qkv_new.stride(0), col_off, W, _p(cache), cap, B, _p(kv_len_dev), _st()))


# ---- the fused decode step (csrc/decode.hip): one or two sequences -------------------------------------------------------------------------------------
def decode_fusable(B, H, Hq, Hkv, D):
    """Shapes the fused decode kernels take: at most two sequences, hidden a multiple of 512 (<= 8192), head_dim 64 / 128 / 256, at most eight query heads a KV head."""
    return B <= 2 and H % 512 == 0 and H <= 8192 and D in (64, 128, 256) and Hq % Hkv == 0 and Hq // Hkv <= 8


def decode_norm_gemv(x, delta, norm_w, eps, gemma, w, lora=None, glu=0):
    """(y, x_sum): y = rmsnorm(x + delta) W^T (+ the adapter branch), one launch (ecgb_decode_norm_gemv).  lora = (A [64, H], n_a used rows, scale, B [N or 2N, 64]).
    glu 1 / 2: W = gate rows then up rows, y = act(gate) * up [M, N / 2 rows of W each]."""
    M, H = x.shape
    N = w.shape[0] // 2 if glu else w.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    xs = torch.empty_like(x) if delta is not None else x
    A, n_a, scale, Bm = lora if lora is not None else (None, 0, 0.0, None)
    _lib.check(_L().ecgb_decode_norm_gemv(_p(x), _p(delta), _p(norm_w), float(eps), int(gemma), M, H, _p(xs) if delta is not None else None, _p(w), w.stride(0), N,
                                          _p(A), A.stride(0) if A is not None else 0, int(n_a), float(scale), _p(Bm), Bm.stride(0) if Bm is not None else 0,
                                          _p(y), y.stride(0), int(glu), _st()))
    return y, xs


def decode_gemv(a, w, lora=None, t=None):
    """y = a W^T (+ t B^T) for one or two rows.  lora = (A, n_a, scale, B): t formed in the kernel (K <= 4096) unless `t` [M, 64] is given (decode_lora_t)."""
    M, K = a.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    A, n_a, scale, Bm = lora if lora is not None else (None, 0, 0.0, None)
    _lib.check(_L().ecgb_decode_gemv(_p(a), a.stride(0), M, K, _p(w), w.stride(0), N, _p(A) if t is None else None, A.stride(0) if A is not None else 0, int(n_a), float(scale),
                                     _p(t), _p(Bm), Bm.stride(0) if Bm is not None else 0, _p(y), y.stride(0), _st()))
    return y


def decode_lora_t(a, A, n_a, scale):
    """t [M, 64] = bf16(scale * a A^T) over the n_a used rows of A (the down-projection site of a decode step: K a multiple of 2048)."""
    M, K = a.shape
    t = torch.empty((M, 64), dtype=torch.bfloat16, device=a.device)
    _lib.check(_L().ecgb_decode_lora_t(_p(a), a.stride(0), M, K, _p(A), A.stride(0), int(n_a), float(scale), _p(t), _st()))
    return t


def decode_attn_splits(cap):
    """Workgroups the keys of a (KV head, sequence) are split over: about 32 keys each, at most 32, at least what keeps a split within 1 024 keys."""
    return int(max(1, min(32, cap // 32), -(-cap // 1024)))


def decode_attn_scratch(cap, B, Hq, Hkv, D, n_splits, device):
    """The zeroed fp32 scratch of decode_attn (scores, statistics, partial outputs, tickets: the tickets must start at zero and are left at zero)."""
    return torch.zeros(_L().ecgb_decode_attn_scratch_floats(cap, B, Hq, Hkv, D, n_splits), dtype=torch.float32, device=device)


def decode_attn(qkv, cos, sin, cache, mask, kv_len, Hq, Hkv, D, scale, n_splits, scratch):
    """One decode step's attention from the RAW q|k|v projection: RoPE, cache append at row kv_len - 1, split softmax attention (ecgb_decode_attn, two launches).
    kv_len: an int or an int32[1] device tensor."""
    B, cap, W = cache.shape
    assert W == 2 * Hkv * D and cache.is_contiguous()
    o = torch.empty((B, Hq * D), dtype=torch.bfloat16, device=qkv.device)
    dyn = torch.is_tensor(kv_len)
    _lib.check(_L().ecgb_decode_attn(_p(qkv), qkv.stride(0), _p(cos), _p(sin), _p(cache), cap, _p(mask), mask.stride(0), _p(o), B, 0 if dyn else int(kv_len),
                                     _p(kv_len) if dyn else None, Hq, Hkv, D, float(scale), int(n_splits), _p(scratch), scratch.numel(), _st()))
    return o

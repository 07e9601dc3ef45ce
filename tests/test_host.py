"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/ecgbyte.h declares, the packed trie layout reproduces the oracle's greedy parse, the
quantiser's threshold staircase is exact, and device entry points fail loudly without a GPU.
No compute kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN
from helpers import load_tokenizer, random_merges
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ecg_byte_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    return _lib.lib()


def _header_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b(ecgb_[a-z_0-9]+)\s*\(", src)))


@pytest.mark.parametrize("header,at_least", [("ecgbyte.h", 30), ("ecgbyte_decoder.h", 70)])
def test_library_exports_every_declared_symbol(lib, header, at_least):
    names = _header_functions(header)
    assert len(names) >= at_least, (header, len(names))
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/{header} but not exported"
    assert lib.ecgb_version() >> 16 == 1


def _make_tok(lib, merges):
    from ecg_byte_amd.tokenizer import flatten_merges
    flat, off, ids = flatten_merges(merges)
    h = C.c_void_p()
    u32p = C.POINTER(C.c_uint32)
    rc = lib.ecgb_tokenizer_create(flat.ctypes.data_as(u32p), off.ctypes.data_as(u32p),
                                   ids.ctypes.data_as(u32p), len(merges), C.byref(h))
    return rc, h


def _nodes(lib, h):
    n = lib.ecgb_tokenizer_copy_nodes(h, None, 0)
    out = np.empty(n, dtype=np.uint64)
    lib.ecgb_tokenizer_copy_nodes(h, out.ctypes.data_as(C.POINTER(C.c_uint64)), n)
    return out


CONT, HEAD, BRANCH = 1 << 30, 1 << 31, (1 << 30) - 1
OTHER = 29


def _runbits(lib, h):
    n = lib.ecgb_tokenizer_copy_runbits(h, None, 0)
    out = np.empty(n, dtype=np.uint32)
    lib.ecgb_tokenizer_copy_runbits(h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out


def _ctz32(x):
    return 32 if x == 0 else (x & -x).bit_length() - 1


def _walk_packed(nodes, byte_to_class, text, lens, runbits=None):
    """Greedy longest match over the packed device trie -- mirrors the step of encode_flow_kernel in encode.hip: "the
    symbol repeats" selects the continuation flag instead of the class bit, the child comes from bitmap + popcount either
    way, inside a same-class chain (with `runbits`) the step takes up to 32 equal symbols at once, and when the step fails
    the token is the stopped-at node's BEST token (its length from the id -> length table `lens`)."""
    out, i, n = [], 0, len(text)
    nodes = [int(v) for v in nodes]
    cls = [byte_to_class.get(b, OTHER) for b in text] + [OTHER] * 40
    diff = [1] + [int(cls[k] != cls[k - 1]) for k in range(1, len(cls))]     # the kernel's change map
    rb = [int(v) for v in runbits] if runbits is not None else None
    while i < n:
        node, j = 0, i
        while True:
            rec = nodes[node]
            bm, fc, tok = rec & 0xFFFFFFFF, (rec >> 32) & 0xFFFF, rec >> 48
            norep = node == 0 or diff[j] == 1
            bit = (1 << cls[j]) if norep else CONT
            if not bm & bit:
                break
            inchain = not norep and not bm & HEAD
            if not inchain:
                node = fc + bin(bm & (bit - 1)).count("1")
                j += 1
            elif rb is None:
                node += 1
                j += 1
            else:
                k, sh = node >> 5, node & 31
                cw = ((rb[2 * k] | (rb[2 * k + 2] << 32)) >> sh) & 0xFFFFFFFF
                dw = sum(diff[j + t] << t for t in range(32))
                m = min(_ctz32((dw | ~cw) & 0xFFFFFFFF), 32)
                assert m >= 1
                node += m
                j += m
        if tok == 0xFFFF:                                       # only the root has no best token: unmatched byte
            assert node == 0
            out.append(text[i])
            i += 1
        else:
            out.append(tok)
            i += lens[tok]
    return out


def _lens(merges):
    lens = {b: 1 for b in range(256)}
    lens.update({tid: len(seq) for seq, tid in merges})
    return lens


def _classes(merges):
    b2c = {ord("a") + c: c for c in range(26)}
    extra = sorted({b for seq, _ in merges for b in seq} - set(b2c))
    for k, b in enumerate(extra):
        b2c[b] = 26 + k
    return b2c


@pytest.mark.parametrize("seed", range(4))
def test_packed_trie_reproduces_oracle(lib, seed):
    rng = np.random.default_rng(seed)
    alphabet = [b"abcdef", b"ab", b"abcxyz.,!", b"mnop\x80\xff"][seed]
    merges = random_merges(rng, int(rng.integers(5, 300)), alphabet=alphabet)
    rc, h = _make_tok(lib, merges)
    assert rc == 0
    try:
        nodes = _nodes(lib, h)
        rb = _runbits(lib, h)
        own = [(int(rb[2 * (u >> 5) + 1]) >> (u & 31)) & 1 for u in range(nodes.size)]
        for u in range(nodes.size):       # the first bit table restates the continuation flag
            assert (int(rb[2 * (u >> 5)]) >> (u & 31)) & 1 == (int(nodes[u]) >> 30) & 1
        # a node's token field is its own token if it carries one (second bit table), else its parent's field
        assert int(nodes[0]) >> 48 == 0xFFFF and not own[0]
        seen = 1
        for u in range(nodes.size):
            bm, fc = int(nodes[u]) & 0xFFFFFFFF, (int(nodes[u]) >> 32) & 0xFFFF
            kids = [fc + k for k in range(bin(bm & BRANCH).count("1"))]
            if bm & CONT:
                kids.append(fc + len(kids) if (bm & HEAD) else u + 1)
            for v in kids:
                seen += 1
                if not own[v]:
                    assert int(nodes[v]) >> 48 == int(nodes[u]) >> 48
        assert seen == nodes.size
        ids_in_trie = {int(nodes[u]) >> 48 for u in range(nodes.size) if own[u]}
        assert ids_in_trie <= set(_lens(merges))
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        assert lib.ecgb_tokenizer_info(h, C.byref(a), C.byref(b), C.byref(c)) == 0
        assert a.value == nodes.size and b.value == max(len(m[0]) for m in merges)
        b2c = _classes(merges)
        lens = _lens(merges)
        assert c.value == len(b2c)
        for _ in range(10):
            pool = np.frombuffer(alphabet + b"qz~", dtype=np.uint8)
            text = bytes(rng.choice(pool, size=int(rng.integers(0, 500))))
            want = O.encode_text(text, merges)
            assert _walk_packed(nodes, b2c, text, lens) == want
            assert _walk_packed(nodes, b2c, text, lens, rb) == want
            runs = b"".join(bytes([rng.choice(pool)]) * int(rng.integers(1, 70)) for _ in range(40))   # long runs: the run step
            assert _walk_packed(nodes, b2c, runs, lens, rb) == O.encode_text(runs, merges)
    finally:
        lib.ecgb_tokenizer_destroy(h)


def test_packed_trie_fixture_tokenizer(lib):
    _, merges, pc = load_tokenizer("c1")
    from ecg_byte_amd import synth
    rc, h = _make_tok(lib, merges)
    assert rc == 0
    nodes = _nodes(lib, h)
    rb = _runbits(lib, h)
    lib.ecgb_tokenizer_destroy(h)
    x = synth.synth_ecg(1, 1000, seed=0)
    text = O.symbols_to_text(O.quantize(x[0], pc["percentile_1"], pc["percentile_99"]))[:3000]
    assert _walk_packed(nodes, _classes(merges), text, _lens(merges)) == O.encode_text(text, merges)
    assert _walk_packed(nodes, _classes(merges), text, _lens(merges), rb) == O.encode_text(text, merges)


def test_merges_beyond_the_packed_trie_take_the_general_form(lib):
    """More than 29 distinct byte values, token ids >= 65 535, 65 535 or more trie nodes: round 4 returned ECGB_ERR_UNSUPPORTED, `encode_text` (lib.rs:149-193)
    refuses no merges list.  The handle is now built in the general form (edge table + 32-bit ids, walked by encode_general_kernel); the GPU parity of that walk
    is tests/test_gpu_parity.py::test_general_form_*."""
    many = [([i, i + 1], 256 + i) for i in range(0, 80, 2)]          # > 29 symbol classes
    rc, h = _make_tok(lib, many)
    assert rc == 0 and h
    n = C.c_uint32()
    assert lib.ecgb_tokenizer_info(h, C.byref(n), None, None) == 0 and n.value == 1 + 256 + 40      # root, the 256 bytes, one node per merge
    assert lib.ecgb_tokenizer_copy_nodes(h, None, 0) == 0                                            # no packed nodes
    lib.ecgb_tokenizer_destroy(h)
    rc, h = _make_tok(lib, [([97, 98], 70000)])                        # token id does not fit 16 bits
    assert rc == 0 and h
    lib.ecgb_tokenizer_destroy(h)
    rng = np.random.default_rng(0)
    big = [(rng.integers(97, 123, size=6).tolist(), 256 + i) for i in range(24000)]                 # > 65 535 nodes
    rc, h = _make_tok(lib, big)
    assert rc == 0 and h
    assert lib.ecgb_tokenizer_info(h, C.byref(n), None, None) == 0 and n.value >= 65535
    lib.ecgb_tokenizer_destroy(h)


def test_expansion_elements_above_255_are_unreachable_nodes_like_the_reference(lib):
    """lib.rs:140-146 keys trie children by u32: an expansion element > 255 is an edge no input byte follows, so the entry's
    token can never be emitted, but the prefix in front of it still adds (token-less) interior nodes.  The handle accepts such a
    list and parses exactly like the oracle, whose edge key is (node << 32) | element (no aliasing with byte edges: with the old
    (node << 8) | byte key, element 300 = 0x12C under node n aliased byte 0x2C under node n + 1)."""
    merges = [([97, 98], 256), ([97, 300], 257), ([97, 98, 99, 400, 100], 258), ([98, 99], 259), ([97, 98, 99], 260),
              ([0x12C], 261), ([97, 0x161], 262)]
    rc, h = _make_tok(lib, merges)
    assert rc == 0, lib.ecgb_last_error()
    nodes = _nodes(lib, h)
    lib.ecgb_tokenizer_destroy(h)
    lens = _lens([m for m in merges if max(m[0]) <= 255])
    b2c = _classes([m for m in merges if max(m[0]) <= 255])
    from oracle import lib_rs_literal as R
    for text in ("abcabd", "aabbcc", "abc,abc", "a,bca", "ab" * 9 + "c", "abca"):
        ref = O.encode_text(text, merges)
        assert ref == R.encode_text(text, merges)
        assert 257 not in ref and 258 not in ref and 261 not in ref and 262 not in ref
        assert _walk_packed(nodes, b2c, text.encode(), lens) == ref, text


def test_rust_bpe_cache_honours_in_place_edits_of_the_merges_list():
    """The reference rebuilds its trie from `merges` on every call (lib.rs:153-161): an edit of a middle entry -- even of an inner
    list, in place -- must not be served from a stale handle."""
    from ecg_byte_amd import rust_bpe
    merges = [([97, 97], 256), ([97, 98], 257), ([98, 98], 258), ([97, 97, 98], 259), ([99, 99], 260), ([98, 99], 261), ([100, 100], 262)]
    t0 = rust_bpe.tokenizer_for(merges)
    assert rust_bpe.tokenizer_for(merges) is t0                       # unchanged content: cached
    assert rust_bpe.tokenizer_for([(list(s), i) for s, i in merges]) is t0   # an equal copy too
    n0 = t0.nodes().copy()
    merges[3][0][2] = 97                                              # inner list of a middle entry, in place: aab -> aaa
    t1 = rust_bpe.tokenizer_for(merges)
    assert t1 is not t0 and not np.array_equal(t1.nodes(), n0)
    merges[2] = ([98, 98, 98], 258)                                   # a replaced middle entry
    t2 = rust_bpe.tokenizer_for(merges)
    assert t2 is not t1 and t2.n_nodes == t1.n_nodes + 1
    npm = [(np.array(s, dtype=np.uint32), np.int64(i)) for s, i in merges]   # numpy element types take the pickle path
    assert rust_bpe.tokenizer_for(npm).n_nodes == t2.n_nodes


def test_rust_bpe_cache_key_is_a_function_of_the_content_not_of_object_identity():
    """marshal formats 3+ write back-references for objects it has seen before (by identity, and only when their reference count says they may be shared): two
    equal lists -- one built from shared int objects, one from distinct ones -- serialised differently and missed the cache.  The key uses format 2."""
    import marshal
    from ecg_byte_amd import rust_bpe
    big = 70000                                                        # not a cached small int: `int(str(big))` is a distinct object every time
    shared = [([big, big, big], 256 + 1), ([big, 257], 256 + 2)]
    shared = [(seq, tid) for seq, tid in shared]
    shared.append(shared[0])                                           # the same tuple object twice
    distinct = [([int(str(e)) for e in seq], int(str(tid))) for seq, tid in shared]
    assert shared == distinct
    assert marshal.dumps(shared) != marshal.dumps(distinct)            # what the default format did
    assert rust_bpe._content_key(shared) == rust_bpe._content_key(distinct)
    assert rust_bpe._content_key(shared) != rust_bpe._content_key(distinct[:-1])


def test_device_entry_points_fail_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, h = _make_tok(lib, [([97, 98], 256)])
    assert rc == 0                                                     # host-only handle
    dummy = (C.c_uint8 * 4096)()
    p = C.cast(dummy, C.c_void_p)
    rc = lib.ecgb_quantize_encode_hip(h, p, 1, 16, 0.0, 1.0, p, 16, p, p, 4096, None)
    assert rc == -5 and b"no device copy" in lib.ecgb_last_error()
    rc = lib.ecgb_encode_hip(h, p, 1, 16, p, 16, p, p, 4096, None)
    assert rc == -5
    lib.ecgb_tokenizer_destroy(h)
    from ecg_byte_amd.tokenizer import HipTokenizer
    with pytest.raises(TypeError):
        HipTokenizer([([97, 98], 256)]).quantize_encode(torch.zeros(1, 12, 10, dtype=torch.float64),
                                                        {"percentile_1": 0.0, "percentile_99": 1.0})


def test_quantiser_staircase_is_exact(lib):
    """Classifying by the host-computed thresholds == the reference arithmetic, for the
    reference-generated golden inputs (incl. +-4 ulp around every bin edge)."""
    z = np.load(os.path.join(GOLDEN, "quantize_ref.npz"))
    for i in range(int(z["n_cases"])):
        p1, p99 = map(float, z[f"p_{i}"])
        thr = np.empty(25, dtype=np.float64)
        assert lib.ecgb_quantizer_thresholds(p1, p99, thr.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.all(np.diff(thr) >= 0)
        x = z[f"x_{i}"].reshape(-1)
        level = np.searchsorted(thr, x, side="right").astype(np.uint8)   # #{k : x >= thr[k]}
        assert np.array_equal(level, z[f"sym_{i}"].reshape(-1)), f"case {i}"
        # each threshold is the first float64 of its level
        below = np.nextafter(thr, -np.inf)
        assert np.array_equal(O.quantize(thr, p1, p99), np.arange(1, 26))
        assert np.array_equal(O.quantize(below, p1, p99), np.arange(0, 25))
    thr = np.empty(25)
    assert lib.ecgb_quantizer_thresholds(5.0, -10.0, thr.ctypes.data_as(C.POINTER(C.c_double))) == -3


def test_rust_bpe_vocab_bookkeeping():
    from ecg_byte_amd import rust_bpe
    vocab, merges = rust_bpe.vocab_merges_from_pairs([(97, 97), (256, 98), (200, 257)])
    assert merges == [([97, 97], 256), ([97, 97, 98], 257), ([200, 97, 97, 98], 258)]
    assert vocab[258] == "<200>aab" and vocab[65] == "A" and vocab[128] == "<128>"   # lib.rs:50-56
    assert (vocab, merges) == O.pairs_to_vocab_merges([(97, 97), (256, 98), (200, 257)])


def test_tokenizer_pickle_format_roundtrip(tmp_path):
    from ecg_byte_amd import tokenizer_utils as tu
    vocab, merges, _ = load_tokenizer("c1")
    p = tmp_path / "tok.pkl"
    tu.save_vocab_and_merges(vocab, merges, str(p))
    v2, m2 = tu.load_vocab_and_merges(str(p))
    assert v2 == vocab and m2 == merges
    sym = np.array([["a", "z"], ["m", "b"]])
    back = tu.reverse_normalize_all(sym, {"percentile_1": -1.0, "percentile_99": 1.0})
    assert np.array_equal(back, O.dequantize(np.array([[0, 25], [12, 1]], dtype=np.uint8), -1.0, 1.0))


def test_inline_asm_lds_reads_are_not_used_before_their_wait():
    """gemm.hip's transposing LDS reads are inline asm whose wait is a later statement (the phase schedule puts the LDS-DMA issue between them):
    hipcc may use or copy a destination register before that wait -- it did so in an attention kernel, stale data only on a busy chip.
    scripts/check_asm_lds_reads.py screens the compiled assembly for that; here: the screen itself on two hand-written snippets, then on
    the real file (hipcc cross-compiles gfx950 without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_lds_reads", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "check_asm_lds_reads.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    good = """_Zkernel:
	;;#ASMSTART
	ds_read_b64_tr_b16 v[10:11], v3 offset:0
	;;#ASMEND
	v_add_u32_e32 v5, v6, v7
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mfma_f32_16x16x32_bf16 v[20:23], v[8:11], v[12:15], v[20:23]
"""
    bad = good.replace("v_add_u32_e32 v5, v6, v7", "v_mov_b64_e32 v[40:41], v[10:11]")
    assert chk.check(good) == []
    hits = chk.check(bad)
    assert len(hits) == 1 and hits[0][3] == [10, 11]
    assert chk.main() == 0


def test_bench_refuses_a_world_size_other_than_gpus():
    """bench.py --gpus N started with another WORLD_SIZE (a plain process asked for 2 GPUs, or a 2-rank launcher with --gpus 1) exits 2 before it generates or
    initialises anything and prints no JSON line: a scaling record cannot be mislabelled.  (The process group's size and, under RCCL, one device per rank are
    checked again after initialisation: tests/test_gpu_pipeline.py runs the two-rank path.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "WORLD_SIZE" in r.stderr
    env["WORLD_SIZE"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_four_wave_gemm_refuses_operands_its_32_bit_dma_offsets_cannot_span(lib):
    """The four-wave GEMM sets its DMA descriptor once per output tile and walks the contraction with 32-bit scalar offsets: an operand stored as contraction rows
    reaches (K + 15) * ld * 2 bytes, a row operand 63 * ld * 2 + K * 2.  Operands of 4 GiB and more must go to the eight-wave kernels (round 4's advisor finding:
    dW of gate|up with T >= 131 072 tokens at ld 16 384)."""
    ok = lib.ecgb_gemm_w4_span_ok
    assert ok(0, 2048, 2048, 2048) == 1 and ok(1, 8192, 2048, 8192) == 1 and ok(2, 16384, 2048, 32768) == 1      # the C3 step's shapes
    assert ok(2, 16384, 2048, 131072) == 0          # dW of gate|up, T = 131 072: (K + 15) * 16 384 * 2 > 2^32
    assert ok(2, 16384, 2048, 131072 - 64) == 1 and ok(2, 16384, 2048, 65536) == 1          # (131 023 rows of 32 KiB: 1.6 MB short of 2^32)
    assert ok(1, 8192, 16384, 131072) == 0 and ok(1, 8192, 16384, 65536) == 1      # NN: B is the contraction-row operand
    assert ok(1, 1 << 26, 2048, 4096) == 0          # NN's A is a row operand: 63 rows of 2^27 bytes
    assert ok(0, 2048, 2048, 1 << 31) == 0          # NT: the contraction itself runs along the row
    assert ok(0, 0, 2048, 64) == 0

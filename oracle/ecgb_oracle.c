/*
 * ecgb_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the reference's tokenizer hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (ecg-byte_amd/) never links, imports or calls it.
 *
 * Restated (all paths relative to /root/reference):
 *   ecg_byte/utils/tokenizer_utils.py:14-19   normalize_all          -> ecgb_oracle_quantize
 *   ecg_byte/utils/tokenizer_utils.py:22-28   reverse_normalize_all  -> ecgb_oracle_dequantize
 *   ecg_byte/rust_bpe/src/lib.rs:10-26        merge                  -> ecgb_oracle_merge
 *   ecg_byte/rust_bpe/src/lib.rs:28-48        get_stats              -> ecgb_oracle_get_stats
 *   ecg_byte/rust_bpe/src/lib.rs:58-125       byte_pair_encoding     -> ecgb_oracle_bpe_train
 *   ecg_byte/rust_bpe/src/lib.rs:127-147      TrieNode               -> struct trie
 *   ecg_byte/rust_bpe/src/lib.rs:149-193      encode_text            -> ecgb_oracle_encode
 *
 * Parity pinning: the Rust crate cannot be built here (no cargo/rustc) and the
 * reference ships no tests or golden vectors for this path.  The quantiser is pinned
 * bit-for-bit against the imported Python reference (tests/golden/make_golden.py);
 * encode/merge/get_stats are pinned by the hand-derived known-answer vectors of
 * SURVEY.md §8c and the reference's own round-trip check (train_tokenizer.py:58-60).
 * TRAINER TIE-BREAKS: parity unpinned -- the reference picks among equal-count pairs
 * by rayon/FxHashMap iteration order, which is schedule dependent; this restatement
 * defines the tie-break as the numerically smallest (left, right) pair.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: no FMA contraction, IEEE double).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ALPHABET_LEN 26 /* tokenizer_utils.py:12 */

/* ------------------------------------------------------------------------------------
 * normalize_all  (tokenizer_utils.py:14-19).  Operation order is the reference's:
 *   normalized = (signal - (p1 - 0.5)) / ((p99 + 0.5) - (p1 - 0.5) + 1e-6)
 *   clipped    = clip(normalized, 0, 1)
 *   scaled     = minimum(floor(clipped * 26), 25).astype(uint8)
 * `sym` receives the alphabet index 0..25 (the reference's character is 'a'+sym).
 * NaN input: numpy's NaN->uint8 cast is unspecified; this restatement yields 0.
 * ---------------------------------------------------------------------------------- */
void ecgb_oracle_quantize(const double *x, size_t n, double p1, double p99,
                          double *clipped_out, uint8_t *sym)
{
    const double a = p1 - 0.5;
    const double d = ((p99 + 0.5) - (p1 - 0.5)) + 1e-6;
    for (size_t i = 0; i < n; ++i) {
        double nrm = (x[i] - a) / d;
        double c = nrm;
        if (c < 0.0) c = 0.0; /* np.clip = minimum(maximum(x, 0), 1); NaN propagates */
        if (c > 1.0) c = 1.0;
        double s = floor(c * (double)ALPHABET_LEN);
        if (s > (double)(ALPHABET_LEN - 1)) s = (double)(ALPHABET_LEN - 1);
        if (clipped_out) clipped_out[i] = c;
        sym[i] = (s == s) ? (uint8_t)s : 0;
    }
}

/* reverse_normalize_all (tokenizer_utils.py:22-28): idx/25 * (max-min) + min. */
void ecgb_oracle_dequantize(const uint8_t *sym, size_t n, double p1, double p99, double *out)
{
    const double mn = p1 - 0.5, mx = p99 + 0.5;
    for (size_t i = 0; i < n; ++i) {
        double c = (double)sym[i] / (double)(ALPHABET_LEN - 1);
        out[i] = c * (mx - mn) + mn;
    }
}

/* ------------------------------------------------------------------------------------
 * merge (lib.rs:10-26): in-place left-to-right non-overlapping replacement.
 * Returns the new length.
 * ---------------------------------------------------------------------------------- */
size_t ecgb_oracle_merge(uint32_t *ids, size_t n, uint32_t left, uint32_t right, uint32_t new_id)
{
    size_t i = 0, w = 0;
    while (i < n) {
        if (i + 1 < n && ids[i] == left && ids[i + 1] == right) {
            ids[w++] = new_id;
            i += 2;
        } else {
            ids[w++] = ids[i];
            i += 1;
        }
    }
    return w;
}

/* ------------------------------------------------------------------------------------
 * get_stats (lib.rs:28-48): histogram of ALL adjacent windows (overlapping ones too:
 * "aaa" -> (a,a):2).  Open-addressing map keyed by (left<<32 | right).
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint64_t *keys;   /* UINT64_MAX = empty */
    uint32_t *vals;
    size_t cap, len;
} pairmap;

static uint64_t mix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

static int pairmap_init(pairmap *m, size_t cap)
{
    m->cap = cap; m->len = 0;
    m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->vals = (uint32_t *)calloc(cap, sizeof(uint32_t));
    if (!m->keys || !m->vals) return -1;
    memset(m->keys, 0xff, cap * sizeof(uint64_t));
    return 0;
}

static void pairmap_free(pairmap *m) { free(m->keys); free(m->vals); }

static int pairmap_add(pairmap *m, uint64_t key, uint32_t inc);

static int pairmap_grow(pairmap *m)
{
    pairmap n;
    if (pairmap_init(&n, m->cap * 2)) return -1;
    for (size_t i = 0; i < m->cap; ++i)
        if (m->keys[i] != UINT64_MAX) pairmap_add(&n, m->keys[i], m->vals[i]);
    pairmap_free(m);
    *m = n;
    return 0;
}

static int pairmap_add(pairmap *m, uint64_t key, uint32_t inc)
{
    if ((m->len + 1) * 2 > m->cap && pairmap_grow(m)) return -1;
    size_t h = (size_t)mix64(key) & (m->cap - 1);
    while (m->keys[h] != UINT64_MAX && m->keys[h] != key) h = (h + 1) & (m->cap - 1);
    if (m->keys[h] == UINT64_MAX) { m->keys[h] = key; m->len++; }
    m->vals[h] += inc;
    return 0;
}

/* Exposed for tests: writes up to `cap` (left,right,count) triples sorted by key; returns
 * the number of distinct pairs. */
size_t ecgb_oracle_get_stats(const uint32_t *ids, size_t n, uint32_t *out_lrc, size_t cap)
{
    pairmap m;
    if (pairmap_init(&m, 1024)) return 0;
    for (size_t i = 0; i + 1 < n; ++i) /* ids.windows(2) */
        pairmap_add(&m, ((uint64_t)ids[i] << 32) | ids[i + 1], 1);
    size_t k = 0, total = m.len;
    /* deterministic order for the caller: selection sort by key over the few pairs asked */
    uint64_t last = 0; int first = 1;
    while (k < cap && k < total) {
        uint64_t best = UINT64_MAX; uint32_t bv = 0;
        for (size_t i = 0; i < m.cap; ++i) {
            uint64_t key = m.keys[i];
            if (key == UINT64_MAX) continue;
            if (!first && key <= last) continue;
            if (key < best) { best = key; bv = m.vals[i]; }
        }
        out_lrc[3 * k + 0] = (uint32_t)(best >> 32);
        out_lrc[3 * k + 1] = (uint32_t)best;
        out_lrc[3 * k + 2] = bv;
        last = best; first = 0; ++k;
    }
    pairmap_free(&m);
    return total;
}

/* ------------------------------------------------------------------------------------
 * byte_pair_encoding (lib.rs:58-125), the literal O(N * num_merges) loop:
 *   for i in 0..num_merges: pairs = get_stats(ids); if empty break;
 *     best = argmax count  [tie-break DEFINED here: smallest (left,right)];
 *     new_id = 256 + i; merge(ids, best, new_id); record merge.
 * ids_io: in = bytes widened to u32 (lib.rs:72), out = final ids; *n_io its length.
 * out_pairs receives (left,right) per performed merge; returns the number performed.
 * The caller derives vocab / merges byte expansions (lib.rs:101-110) from the pairs.
 * ---------------------------------------------------------------------------------- */
uint32_t ecgb_oracle_bpe_train(uint32_t *ids_io, size_t *n_io, uint32_t num_merges,
                               uint32_t *out_pairs)
{
    size_t n = *n_io;
    uint32_t done = 0;
    for (uint32_t i = 0; i < num_merges; ++i) {
        pairmap m;
        if (pairmap_init(&m, 1024)) break;
        for (size_t k = 0; k + 1 < n; ++k)
            pairmap_add(&m, ((uint64_t)ids_io[k] << 32) | ids_io[k + 1], 1);
        if (m.len == 0) { pairmap_free(&m); break; } /* lib.rs:88-90 */
        uint64_t best = UINT64_MAX; uint32_t best_c = 0;
        for (size_t k = 0; k < m.cap; ++k) {
            if (m.keys[k] == UINT64_MAX) continue;
            if (m.vals[k] > best_c || (m.vals[k] == best_c && m.keys[k] < best)) {
                best_c = m.vals[k]; best = m.keys[k];
            }
        }
        pairmap_free(&m);
        uint32_t l = (uint32_t)(best >> 32), r = (uint32_t)best;
        n = ecgb_oracle_merge(ids_io, n, l, r, 256u + i); /* lib.rs:97-99 */
        out_pairs[2 * done] = l; out_pairs[2 * done + 1] = r;
        ++done;
    }
    *n_io = n;
    return done;
}

/* ------------------------------------------------------------------------------------
 * TrieNode (lib.rs:127-147) + encode_text (lib.rs:149-193).
 * The reference holds a HashMap<u32,TrieNode> per node; here one open-addressing edge
 * map keyed (node<<32 | element) serves every node (u32 elements, lib.rs:128) -- same mapping, same semantics:
 *   insert(): walk/create children, then node.token_id = Some(id)  (last insert wins)
 *   encode(): from i walk while a child exists, remember the deepest node carrying a
 *             token_id, emit it and advance by its length; no match -> emit the raw id.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint64_t *ekeys;    /* (node<<32)|element: the reference keys children by u32 (lib.rs:128), so an expansion element
                           > 255 is its own edge -- a node no input byte reaches -- and never aliases a byte edge; UINT64_MAX empty */
    uint32_t *echild;
    size_t ecap, elen;
    int64_t *token;     /* per node, -1 = None */
    size_t n_nodes, ncap;
} trie;

static int trie_init(trie *t, size_t ecap, size_t ncap)
{
    t->ecap = ecap; t->elen = 0; t->ncap = ncap; t->n_nodes = 1; /* node 0 = root */
    t->ekeys = (uint64_t *)malloc(ecap * sizeof(uint64_t));
    t->echild = (uint32_t *)malloc(ecap * sizeof(uint32_t));
    t->token = (int64_t *)malloc(ncap * sizeof(int64_t));
    if (!t->ekeys || !t->echild || !t->token) return -1;
    memset(t->ekeys, 0xff, ecap * sizeof(uint64_t));
    t->token[0] = -1;
    return 0;
}

static void trie_free(trie *t) { free(t->ekeys); free(t->echild); free(t->token); }

static int64_t trie_child(const trie *t, uint32_t node, uint32_t byte)
{
    uint64_t key = ((uint64_t)node << 32) | byte;
    size_t h = (size_t)mix64(key) & (t->ecap - 1);
    while (t->ekeys[h] != UINT64_MAX) {
        if (t->ekeys[h] == key) return t->echild[h];
        h = (h + 1) & (t->ecap - 1);
    }
    return -1;
}

static int trie_grow_edges(trie *t)
{
    size_t ncap = t->ecap * 2;
    uint64_t *k = (uint64_t *)malloc(ncap * sizeof(uint64_t));
    uint32_t *c = (uint32_t *)malloc(ncap * sizeof(uint32_t));
    if (!k || !c) return -1;
    memset(k, 0xff, ncap * sizeof(uint64_t));
    for (size_t i = 0; i < t->ecap; ++i) {
        if (t->ekeys[i] == UINT64_MAX) continue;
        size_t h = (size_t)mix64(t->ekeys[i]) & (ncap - 1);
        while (k[h] != UINT64_MAX) h = (h + 1) & (ncap - 1);
        k[h] = t->ekeys[i]; c[h] = t->echild[i];
    }
    free(t->ekeys); free(t->echild);
    t->ekeys = k; t->echild = c; t->ecap = ncap;
    return 0;
}

static int trie_insert(trie *t, const uint32_t *seq, size_t len, uint32_t token_id)
{
    uint32_t node = 0;
    for (size_t i = 0; i < len; ++i) {
        int64_t ch = trie_child(t, node, seq[i]);
        if (ch < 0) { /* children.entry(id).or_insert_with(TrieNode::new) */
            if ((t->elen + 1) * 2 > t->ecap && trie_grow_edges(t)) return -1;
            if (t->n_nodes == t->ncap) {
                size_t nc = t->ncap * 2;
                int64_t *nt = (int64_t *)realloc(t->token, nc * sizeof(int64_t));
                if (!nt) return -1;
                t->token = nt; t->ncap = nc;
            }
            uint64_t key = ((uint64_t)node << 32) | seq[i];
            size_t h = (size_t)mix64(key) & (t->ecap - 1);
            while (t->ekeys[h] != UINT64_MAX) h = (h + 1) & (t->ecap - 1);
            t->ekeys[h] = key; t->echild[h] = (uint32_t)t->n_nodes; t->elen++;
            t->token[t->n_nodes] = -1;
            ch = (int64_t)t->n_nodes++;
        }
        node = (uint32_t)ch;
    }
    t->token[node] = token_id; /* node.token_id = Some(token_id): later duplicates overwrite */
    return 0;
}

/* merges are passed flattened: expansion i = flat[offsets[i] .. offsets[i+1]), id = ids[i]. */
static int trie_build(trie *t, const uint32_t *flat, const uint32_t *offsets,
                      const uint32_t *ids, size_t n_merges)
{
    if (trie_init(t, 4096, 1024)) return -1;
    for (uint32_t b = 0; b <= 255u; ++b) /* lib.rs:155-157 */
        if (trie_insert(t, &b, 1, b)) return -1;
    for (size_t i = 0; i < n_merges; ++i) /* lib.rs:159-161, list order */
        if (trie_insert(t, flat + offsets[i], offsets[i + 1] - offsets[i], ids[i])) return -1;
    return 0;
}

static size_t trie_encode(const trie *t, const uint8_t *text, size_t n, uint32_t *out)
{
    size_t i = 0, w = 0;
    while (i < n) { /* lib.rs:165 */
        uint32_t node = 0;
        size_t match_len = 0; int64_t match_id = -1;
        for (size_t j = i; j < n; ++j) { /* lib.rs:170-181 */
            int64_t ch = trie_child(t, node, text[j]);
            if (ch < 0) break;
            node = (uint32_t)ch;
            if (t->token[node] >= 0) { match_len = j - i + 1; match_id = t->token[node]; }
        }
        if (match_id >= 0) { out[w++] = (uint32_t)match_id; i += match_len; }
        else { out[w++] = text[i]; i += 1; } /* lib.rs:186-189 */
    }
    return w;
}

/* Reference-faithful call: the trie is rebuilt on every call (lib.rs:153-161).
 * `out` must hold n entries.  Returns the token count, or (size_t)-1 on allocation failure. */
size_t ecgb_oracle_encode(const uint8_t *text, size_t n, const uint32_t *flat,
                          const uint32_t *offsets, const uint32_t *ids, size_t n_merges,
                          uint32_t *out)
{
    trie t;
    if (trie_build(&t, flat, offsets, ids, n_merges)) return (size_t)-1;
    size_t w = trie_encode(&t, text, n, out);
    trie_free(&t);
    return w;
}

/* Build-once handle for the "trie built once" CPU-baseline variant. */
void *ecgb_oracle_trie_create(const uint32_t *flat, const uint32_t *offsets,
                              const uint32_t *ids, size_t n_merges)
{
    trie *t = (trie *)malloc(sizeof(trie));
    if (!t) return NULL;
    if (trie_build(t, flat, offsets, ids, n_merges)) { free(t); return NULL; }
    return t;
}

size_t ecgb_oracle_trie_nodes(const void *h) { return ((const trie *)h)->n_nodes; }

size_t ecgb_oracle_trie_encode(const void *h, const uint8_t *text, size_t n, uint32_t *out)
{
    return trie_encode((const trie *)h, text, n, out);
}

void ecgb_oracle_trie_destroy(void *h)
{
    if (h) { trie_free((trie *)h); free(h); }
}

/* Whole reference per-sample front end (data_loader.py:74-76): quantise a (12,L) float64
 * record, flatten lead-major, 'a'+sym, encode.  scratch must hold n bytes. */
size_t ecgb_oracle_quantize_encode(const void *h, const double *x, size_t n, double p1,
                                   double p99, uint8_t *scratch, uint32_t *out)
{
    ecgb_oracle_quantize(x, n, p1, p99, NULL, scratch);
    for (size_t i = 0; i < n; ++i) scratch[i] = (uint8_t)('a' + scratch[i]);
    return trie_encode((const trie *)h, scratch, n, out);
}

/* Diagnostic: the greedy step of lib.rs:170-181 evaluated at EVERY start position
 * (match length and id the scan would take if it stood at i).  Used by tests and by the
 * chunk-synchronisation statistics in DESIGN.md; not a reference entry point. */
void ecgb_oracle_trie_match_all(const void *h, const uint8_t *text, size_t n,
                                uint32_t *len_out, uint32_t *id_out, uint32_t *walk_out)
{
    const trie *t = (const trie *)h;
    for (size_t i = 0; i < n; ++i) {
        uint32_t node = 0; size_t match_len = 0; int64_t match_id = -1; size_t j;
        for (j = i; j < n; ++j) {
            int64_t ch = trie_child(t, node, text[j]);
            if (ch < 0) break;
            node = (uint32_t)ch;
            if (t->token[node] >= 0) { match_len = j - i + 1; match_id = t->token[node]; }
        }
        if (match_id < 0) { match_len = 1; match_id = text[i]; }
        len_out[i] = (uint32_t)match_len; id_out[i] = (uint32_t)match_id;
        if (walk_out) walk_out[i] = (uint32_t)(j - i); /* symbols consumed by the walk */
    }
}

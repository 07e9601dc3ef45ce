// tokenizer.cpp -- host side of the tokenizer handle: builds the byte trie that
// rust_bpe.encode_text builds on every call (reference: ecg_byte/rust_bpe/src/lib.rs:127-161),
// once, and lays it out for the device.
//
// Layout choice (MI355X): the encode kernel walks the trie with one dependent lookup per
// step, so a node must be ONE aligned 8-byte LDS/L2 read:
//   branch-child bitmap (one bit per symbol class) + 2 flags | first-branch-child id | best token id
// Branch children (classes other than the one the node was entered by) are numbered
// consecutively in class order:
//   child(node, cls) = first + popcount(bitmap & ((1 << cls) - 1)).
// Quantised ECG is mostly runs of equal symbols and ~3/4 of a trained trie's nodes lie on
// same-class chains (m, mm, mmm, ...), so those chains are numbered consecutively: the child
// by the entering class is node + 1 (for the head of a chain: right after its branch
// children), and a run of k equal symbols is ONE step of k nodes.  Numbering is otherwise
// breadth-first, so shallow (hot) nodes get low ids; the kernel keeps nodes [0, n_lds) in
// LDS and reads any deeper remainder through L2.
#include "tokenizer.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstring>
#include <new>
#include <queue>

namespace ecgb {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

}  // namespace ecgb

extern "C" const char *ecgb_last_error(void) { return ecgb::g_last_error.c_str(); }

extern "C" uint32_t ecgb_version(void) { return (1u << 16) | 0u; }

namespace {

struct BuildNode {
    std::array<int32_t, 32> child;
    int64_t token = -1;
    BuildNode() { child.fill(-1); }
};

inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// The general form: lib.rs:127-161 literally -- insert the 256 single bytes with id = byte, then every (expansion, id) in list order (a later duplicate
// overwrites the token, lib.rs:145); an element above 255 is an edge no input byte follows, so only the prefix in front of it adds (token-less) nodes.
void build_general(ecgb_tokenizer *tok, const uint32_t *flat_bytes, const uint32_t *offsets, const uint32_t *ids, size_t n_merges)
{
    size_t total = 256;
    for (size_t i = 0; i < n_merges; ++i) total += offsets[i + 1] - offsets[i];
    size_t cap = 1024;
    while (cap < 2 * total + 2) cap <<= 1;
    tok->g_keys.assign(cap, 0);
    tok->g_child.assign(cap, 0);
    tok->g_token.assign(1, ecgb::kNoToken32);      // node 0 = root
    auto child = [&](uint32_t node, uint32_t byte, bool create) -> int64_t {
        const uint64_t key = (((uint64_t)node << 8) | byte) + 1;
        size_t h = (size_t)mix64(key) & (cap - 1);
        while (tok->g_keys[h]) {
            if (tok->g_keys[h] == key) return tok->g_child[h];
            h = (h + 1) & (cap - 1);
        }
        if (!create) return -1;
        tok->g_keys[h] = key;
        tok->g_child[h] = (uint32_t)tok->g_token.size();
        tok->g_token.push_back(ecgb::kNoToken32);
        return tok->g_child[h];
    };
    uint32_t depth_max = 1;
    auto insert = [&](const uint32_t *seq, size_t len, uint32_t id, bool with_token) {
        uint32_t node = 0;
        for (size_t k = 0; k < len; ++k) node = (uint32_t)child(node, seq[k], true);
        if (with_token) tok->g_token[node] = id;
        depth_max = std::max<uint32_t>(depth_max, (uint32_t)len);
    };
    for (uint32_t b = 0; b < 256; ++b) insert(&b, 1, b, true);
    for (size_t i = 0; i < n_merges; ++i) {
        const uint32_t *seq = flat_bytes + offsets[i];
        const size_t len = offsets[i + 1] - offsets[i];
        size_t reach = 0;
        while (reach < len && seq[reach] <= 255u) ++reach;
        insert(seq, reach, ids[i], reach == len);
    }
    tok->general = true;
    tok->g_nodes = (uint32_t)tok->g_token.size();
    tok->max_depth = depth_max;
    tok->nodes.clear(); tok->runbits.clear(); tok->tok_len.clear();
}

// uploads the general form; the packed form's device pointers stay null except `lut_dev`/`nodes_dev` markers the entry points test
int upload_general(ecgb_tokenizer *tok)
{
    int dev = -1, n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return ECGB_OK; }
    tok->device = dev;
    hipDeviceProp_t prop;
    tok->n_cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    const size_t kb = tok->g_keys.size() * 8, cb = tok->g_child.size() * 4, tb = tok->g_token.size() * 4;
    if (hipMalloc((void **)&tok->g_keys_dev, kb) != hipSuccess || hipMalloc((void **)&tok->g_child_dev, cb) != hipSuccess || hipMalloc((void **)&tok->g_token_dev, tb) != hipSuccess) {
        ecgb::set_error("ecgb_tokenizer_create: hipMalloc failed (general trie)");
        return ECGB_ERR_NOMEM;
    }
    if (hipMemcpy(tok->g_keys_dev, tok->g_keys.data(), kb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(tok->g_child_dev, tok->g_child.data(), cb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(tok->g_token_dev, tok->g_token.data(), tb, hipMemcpyHostToDevice) != hipSuccess) {
        ecgb::set_error("ecgb_tokenizer_create: hipMemcpy failed (general trie)");
        return ECGB_ERR_HIP;
    }
    return ECGB_OK;
}

}  // namespace

extern "C" int ecgb_tokenizer_create(const uint32_t *flat_bytes, const uint32_t *offsets,
                                     const uint32_t *ids, size_t n_merges, ecgb_tokenizer **out)
{
    using namespace ecgb;
    if (!out || (n_merges && (!flat_bytes || !offsets || !ids))) {
        set_error("ecgb_tokenizer_create: NULL argument");
        return ECGB_ERR_INVALID;
    }
    *out = nullptr;
    ecgb_tokenizer *tok = new (std::nothrow) ecgb_tokenizer();
    if (!tok) { set_error("ecgb_tokenizer_create: out of host memory"); return ECGB_ERR_NOMEM; }
    try {
        // ---- symbol classes: 0..25 are 'a'..'z' (the quantiser's alphabet index IS the
        // class), further byte values that occur in an expansion get 26..28.
        std::memset(tok->byte_to_class, kOtherClass, sizeof(tok->byte_to_class));
        for (uint32_t c = 0; c < ECGB_ALPHABET; ++c) {
            tok->byte_to_class['a' + c] = (uint8_t)c;
            tok->class_to_byte[c] = (uint8_t)('a' + c);
        }
        uint32_t n_classes = ECGB_ALPHABET;
        bool used[256] = { false };
        for (size_t i = 0; i < n_merges; ++i) {
            if (offsets[i + 1] < offsets[i]) { delete tok; set_error("ecgb_tokenizer_create: offsets not monotone"); return ECGB_ERR_INVALID; }
            // An element > 255 is accepted as the reference accepts it (lib.rs:140-146 keys children by u32): no input byte
            // ever follows that edge, so everything from it on -- the entry's token included -- is unreachable.  Only the
            // prefix in front of it adds (token-less) nodes.
            for (uint32_t k = offsets[i]; k < offsets[i + 1] && flat_bytes[k] <= 255u; ++k) used[flat_bytes[k]] = true;
        }
        for (uint32_t b = 0; b < 256; ++b) {
            if (!used[b] || tok->byte_to_class[b] != kOtherClass) continue;
            if (n_classes == kMaxClasses) {      // more than 29 distinct byte values (a..z plus 3): the general form
                build_general(tok, flat_bytes, offsets, ids, n_merges);
                const int rc = upload_general(tok);
                if (rc) { ecgb_tokenizer_destroy(tok); return rc; }
                *out = tok;
                return ECGB_OK;
            }
            tok->byte_to_class[b] = (uint8_t)n_classes;
            tok->class_to_byte[n_classes] = (uint8_t)b;
            ++n_classes;
        }
        tok->n_classes = n_classes;
        tok->n_merges = (uint32_t)n_merges;

        // ---- pointer trie, insertion order of lib.rs:155-161
        std::vector<BuildNode> bn(1);
        auto insert = [&](const uint32_t *seq, size_t len, uint32_t token_id) {
            int32_t node = 0;
            for (size_t k = 0; k < len; ++k) {
                const uint8_t cls = tok->byte_to_class[seq[k]];   // < kMaxClasses: every merge byte has a class
                int32_t ch = bn[node].child[cls];
                if (ch < 0) {
                    ch = (int32_t)bn.size();
                    bn[node].child[cls] = ch;
                    bn.emplace_back();
                }
                node = ch;
            }
            bn[node].token = token_id;  // later duplicates overwrite (lib.rs:145)
        };
        for (uint32_t c = 0; c < n_classes; ++c) {  // the single-byte tokens that have a class
            uint32_t b = tok->class_to_byte[c];
            insert(&b, 1, b);
        }
        auto walk_only = [&](const uint32_t *seq, size_t len) {   // the reachable prefix of an entry with an element > 255
            int32_t node = 0;
            for (size_t k = 0; k < len; ++k) {
                const uint8_t cls = tok->byte_to_class[seq[k]];
                int32_t ch = bn[node].child[cls];
                if (ch < 0) { ch = (int32_t)bn.size(); bn[node].child[cls] = ch; bn.emplace_back(); }
                node = ch;
            }
        };
        for (size_t i = 0; i < n_merges; ++i) {
            const uint32_t *seq = flat_bytes + offsets[i];
            const size_t len = offsets[i + 1] - offsets[i];
            size_t reach = 0;
            while (reach < len && seq[reach] <= 255u) ++reach;
            if (reach == len) insert(seq, len, ids[i]);
            else walk_only(seq, reach);
        }

        bool wide_id = false;
        for (size_t i = 0; i < n_merges; ++i) wide_id = wide_id || ids[i] >= kNoToken;
        if (bn.size() >= 65535 || wide_id) {    // node numbers or token ids that do not fit the packed node's 16-bit fields: the general form
            build_general(tok, flat_bytes, offsets, ids, n_merges);
            const int rc = upload_general(tok);
            if (rc) { ecgb_tokenizer_destroy(tok); return rc; }
            *out = tok;
            return ECGB_OK;
        }
        // ---- renumbering: breadth-first over blocks; a block = the branch children of a node in
        // class order, followed (for a chain head) by the whole same-class chain below it
        const size_t nb = bn.size();
        std::vector<int32_t> cin(nb, -1), parent(nb, -1);      // entering class, parent (old ids)
        std::vector<uint32_t> depth_old(nb, 0);
        {
            std::vector<int32_t> stack{0};
            while (!stack.empty()) {
                const int32_t v = stack.back();
                stack.pop_back();
                for (uint32_t c = 0; c < kMaxClasses; ++c) {
                    const int32_t ch = bn[v].child[c];
                    if (ch < 0) continue;
                    cin[ch] = (int32_t)c;
                    parent[ch] = v;
                    depth_old[ch] = depth_old[v] + 1;
                    stack.push_back(ch);
                }
            }
        }
        auto is_head = [&](int32_t v) { return v != 0 && (parent[v] == 0 || cin[parent[v]] != cin[v]); };
        auto cont_of = [&](int32_t v) { return (v != 0) ? bn[v].child[cin[v]] : -1; };
        std::vector<int32_t> order;  // new id -> old id
        order.reserve(nb);
        order.push_back(0);
        std::vector<uint32_t> first_child(nb, 0);   // by new id
        for (size_t head = 0; head < order.size(); ++head) {
            const int32_t v = order[head];
            first_child[head] = (uint32_t)order.size();
            for (uint32_t c = 0; c < kMaxClasses; ++c)
                if (bn[v].child[c] >= 0 && (v == 0 || (int32_t)c != cin[v])) order.push_back(bn[v].child[c]);
            if (is_head(v))
                for (int32_t w = cont_of(v); w >= 0; w = cont_of(w)) order.push_back(w);
        }
        tok->nodes.resize(order.size());
        tok->runbits.assign(2 * (order.size() / 32 + 3), 0u);
        std::vector<uint32_t> new_id(nb, 0), best(order.size(), kNoToken);   // best token on the path root..node, by new id
        for (size_t i = 0; i < order.size(); ++i) new_id[order[i]] = (uint32_t)i;
        uint32_t max_depth = 0;
        for (size_t i = 0; i < order.size(); ++i) {
            const int32_t v = order[i];
            const BuildNode &n = bn[v];
            uint32_t bitmap = 0;
            for (uint32_t c = 0; c < kMaxClasses; ++c)
                if (n.child[c] >= 0 && (v == 0 || (int32_t)c != cin[v])) bitmap |= 1u << c;
            if (is_head(v)) bitmap |= kHeadFlag;
            if (cont_of(v) >= 0) {
                bitmap |= kContFlag;
                tok->runbits[2 * (i / 32)] |= 1u << (i % 32);
            }
            uint32_t token = kNoToken;
            if (i != 0 && n.token >= 0) {  // the root's own token is never consulted (lib.rs:170-181)
                if (n.token >= (int64_t)kNoToken) {      // (unreachable: wide ids took the general form above)
                    delete tok;
                    set_error("ecgb_tokenizer_create: token id >= 65535");
                    return ECGB_ERR_UNSUPPORTED;
                }
                token = (uint32_t)n.token;
                tok->runbits[2 * (i / 32) + 1] |= 1u << (i % 32);
            }
            // a parent is numbered before its children, so its best token is known
            best[i] = (token != kNoToken || i == 0) ? token : best[new_id[parent[v]]];
            tok->nodes[i] = pack_node(bitmap, first_child[i], best[i]);
            max_depth = std::max(max_depth, depth_old[v]);
        }
        tok->max_depth = max_depth;
        // token id -> length, for the encoder's pointer-following pass (only when every length fits a byte)
        if (max_depth <= 255) {
            uint32_t max_id = 255;
            for (size_t i = 0; i < n_merges; ++i) if (ids[i] < kNoToken) max_id = std::max(max_id, ids[i]);
            tok->tok_len.assign(((size_t)max_id + 1 + 7) & ~(size_t)7, 0);
            for (uint32_t b = 0; b < 256; ++b) tok->tok_len[b] = 1;
            std::vector<uint8_t> seen(tok->tok_len.size(), 0);
            bool ambiguous = false;   // one id on expansions of different lengths: the id does not tell the length
            for (size_t i = 0; i < order.size(); ++i) {
                const int64_t t = bn[order[i]].token;
                if (i == 0 || t < 0) continue;
                const uint8_t d = (uint8_t)depth_old[order[i]];
                if (seen[(size_t)t] && tok->tok_len[(size_t)t] != d) ambiguous = true;
                seen[(size_t)t] = 1;
                tok->tok_len[(size_t)t] = d;
            }
            if (ambiguous) tok->tok_len.clear();   // the encoder then takes the kernel that does not use the table
        }
        // root children are nodes 1..n_classes in class order
        for (uint32_t c = 0; c < 32; ++c) { tok->single_id[c] = 0; if (c >= n_classes) tok->class_to_byte[c] = 0; }
        for (uint32_t c = 0; c < n_classes; ++c) tok->single_id[c] = (uint16_t)(tok->nodes[1 + c] >> 48);
    } catch (const std::bad_alloc &) {
        delete tok;
        set_error("ecgb_tokenizer_create: out of host memory");
        return ECGB_ERR_NOMEM;
    }

    // ---- upload
    // Without a device the handle stays host-only (introspection works, every device entry
    // point then returns ECGB_ERR_NODEVICE): lets the trie layout be tested on CPU-only boxes.
    int dev = -1, n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        *out = tok;
        return ECGB_OK;
    }
    tok->device = dev;
    hipDeviceProp_t prop;
    tok->n_cus = (hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    const size_t nbytes = tok->nodes.size() * sizeof(uint64_t);
    uint8_t lut[256 + 64 + 32];
    std::memcpy(lut, tok->byte_to_class, 256);
    std::memcpy(lut + 256, tok->single_id, 64);
    std::memcpy(lut + 320, tok->class_to_byte, 32);
    const size_t rbytes = tok->runbits.size() * sizeof(uint32_t);
    if (hipMalloc((void **)&tok->nodes_dev, nbytes) != hipSuccess ||
        hipMalloc((void **)&tok->runbits_dev, rbytes) != hipSuccess ||
        hipMalloc((void **)&tok->toklen_dev, std::max<size_t>(8, tok->tok_len.size())) != hipSuccess ||
        hipMalloc((void **)&tok->lut_dev, sizeof(lut)) != hipSuccess) {
        ecgb_tokenizer_destroy(tok);
        ecgb::set_error("ecgb_tokenizer_create: hipMalloc failed");
        return ECGB_ERR_NOMEM;
    }
    if (hipMemcpy(tok->nodes_dev, tok->nodes.data(), nbytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(tok->runbits_dev, tok->runbits.data(), rbytes, hipMemcpyHostToDevice) != hipSuccess ||
        (!tok->tok_len.empty() &&
         hipMemcpy(tok->toklen_dev, tok->tok_len.data(), tok->tok_len.size(), hipMemcpyHostToDevice) != hipSuccess) ||
        hipMemcpy(tok->lut_dev, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess) {
        ecgb_tokenizer_destroy(tok);
        ecgb::set_error("ecgb_tokenizer_create: hipMemcpy failed");
        return ECGB_ERR_HIP;
    }
    *out = tok;
    return ECGB_OK;
}

extern "C" void ecgb_tokenizer_destroy(ecgb_tokenizer *tok)
{
    if (!tok) return;
    if (tok->nodes_dev) (void)hipFree(tok->nodes_dev);
    if (tok->runbits_dev) (void)hipFree(tok->runbits_dev);
    if (tok->toklen_dev) (void)hipFree(tok->toklen_dev);
    if (tok->lut_dev) (void)hipFree(tok->lut_dev);
    if (tok->g_keys_dev) (void)hipFree(tok->g_keys_dev);
    if (tok->g_child_dev) (void)hipFree(tok->g_child_dev);
    if (tok->g_token_dev) (void)hipFree(tok->g_token_dev);
    delete tok;
}

extern "C" size_t ecgb_tokenizer_copy_nodes(const ecgb_tokenizer *tok, uint64_t *out, size_t cap)
{
    if (!tok) return 0;
    const size_t n = tok->nodes.size();
    if (out) std::memcpy(out, tok->nodes.data(), std::min(n, cap) * sizeof(uint64_t));
    return n;
}

extern "C" size_t ecgb_tokenizer_copy_runbits(const ecgb_tokenizer *tok, uint32_t *out, size_t cap)
{
    if (!tok) return 0;
    const size_t n = tok->runbits.size();
    if (out) std::memcpy(out, tok->runbits.data(), std::min(n, cap) * sizeof(uint32_t));
    return n;
}

extern "C" int ecgb_tokenizer_info(const ecgb_tokenizer *tok, uint32_t *n_nodes, uint32_t *max_depth,
                                   uint32_t *n_classes)
{
    if (!tok) { ecgb::set_error("ecgb_tokenizer_info: NULL handle"); return ECGB_ERR_INVALID; }
    if (n_nodes) *n_nodes = tok->general ? tok->g_nodes : (uint32_t)tok->nodes.size();
    if (max_depth) *max_depth = tok->max_depth;
    if (n_classes) *n_classes = tok->n_classes;
    return ECGB_OK;
}

// tokenizer.hpp -- host-side tokenizer handle shared by the C-ABI translation units.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ecgbyte.h"

namespace ecgb {

constexpr uint32_t kMaxClasses = 29;      // usable symbol classes (bits 0..28 of a node's child bitmap)
constexpr uint32_t kOtherClass = 29;      // byte that occurs in no expansion and is not a..z; also the
                                          // end-of-stream sentinel: bit 29 is never set in any bitmap
constexpr uint32_t kNoToken = 0xFFFFu;    // node carries no token id
constexpr uint32_t kNoToken32 = 0xFFFFFFFFu;   // the same in the general form's 32-bit token array
constexpr uint32_t kContFlag = 1u << 30;  // node has a child of the class it was entered by (its "continuation")
constexpr uint32_t kHeadFlag = 1u << 31;  // node was entered by a different class than its parent was (or from the root)
constexpr uint32_t kBranchMask = (1u << 30) - 1u;

// Device trie node, 8 bytes:
//   [28:0]  bitmap of the BRANCH children: classes other than the one the node was entered by
//   [30]    kContFlag, [31] kHeadFlag   (kContFlag - 1 == kBranchMask: the continuation child of a head is "the child of
//           class 30", after every branch child, by the same bitmap + popcount formula)
//   [47:32] id of the first branch child (branch children are consecutive, in class order)
//   [63:48] BEST token of the node: the id carried by the deepest token-carrying node on the path root..node (the node
//           itself if it carries one), kNoToken if there is none (the root).  This is what lib.rs:170-189 emits when its
//           walk stops at this node, so a walk does not have to remember the last token it passed; which nodes carry a
//           token THEMSELVES is the second bit table of `runbits`.
// The continuation child of a head node sits right after its branch children
// (first + popcount(bitmap)), followed by the rest of the same-class chain: for every other
// node the continuation child is node + 1.  A run of equal symbols therefore walks consecutive
// node ids and can be taken several symbols at a time (encode.hip, walk_chunk).  The root has
// no entering class: all its children are branch children and both flags are clear.
inline uint64_t pack_node(uint32_t bitmap, uint32_t first_child, uint32_t token)
{
    return (uint64_t)bitmap | ((uint64_t)(first_child & 0xFFFFu) << 32) | ((uint64_t)(token & 0xFFFFu) << 48);
}

void set_error(const std::string &msg);

}  // namespace ecgb

struct ecgb_tokenizer {
    std::vector<uint64_t> nodes;       // packed trie, node 0 = root, its children 1..n_classes in class order
    std::vector<uint32_t> runbits;     // word pairs k: [2k] bit u%32 = node 32k+u has kContFlag, [2k+1] = carries a token;
                                       // two zero pairs of padding (the kernel reads a 64-node window past any node)
    std::vector<uint8_t> tok_len;      // token id -> expansion length (1 for the byte tokens; 0 = id not in the vocabulary),
                                       // padded to a multiple of 8 entries; empty if some expansion is longer than 255
                                       // or one id stands for expansions of different lengths
    uint8_t byte_to_class[256];        // raw byte -> symbol class (kOtherClass if none)
    uint16_t single_id[32];            // token id of the length-1 match of each class
    uint8_t class_to_byte[32];
    uint32_t n_classes = 0;
    uint32_t max_depth = 0;
    uint32_t n_merges = 0;
    // GENERAL form (merges the packed layout cannot hold: more than 29 byte values, 65 535 or more nodes, token ids >= 65 535): the trie of lib.rs:127-147 as it
    // is -- one open-addressing edge table keyed (node << 8 | byte) and a 32-bit token per node (kNoToken32 = none) -- walked by encode_general_kernel, one lane
    // per stream: slow, and total.  `nodes` stays empty.
    bool general = false;
    std::vector<uint64_t> g_keys;      // (node << 8 | byte) + 1, 0 = empty slot; capacity a power of two
    std::vector<uint32_t> g_child;
    std::vector<uint32_t> g_token;     // per node
    uint32_t g_nodes = 0;
    uint64_t *g_keys_dev = nullptr;
    uint32_t *g_child_dev = nullptr, *g_token_dev = nullptr;
    // device copies
    uint64_t *nodes_dev = nullptr;
    uint32_t *runbits_dev = nullptr;
    uint8_t *toklen_dev = nullptr;
    uint8_t *lut_dev = nullptr;        // 256 B byte_to_class | 64 B single_id | 32 B class_to_byte
    int device = -1;
    int n_cus = 0;                     // compute units of `device` (persistent grid size)
};

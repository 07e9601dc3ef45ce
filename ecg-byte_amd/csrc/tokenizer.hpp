// tokenizer.hpp -- host-side tokenizer handle shared by the C-ABI translation units.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ecgbyte.h"

namespace ecgb {

constexpr uint32_t kMaxClasses = 31;      // usable symbol classes (bits 0..30 of a node's child bitmap)
constexpr uint32_t kOtherClass = 31;      // byte that occurs in no expansion and is not a..z; also the
                                          // end-of-stream sentinel: bit 31 is never set in any bitmap
constexpr uint32_t kNoToken = 0xFFFFu;    // node carries no token id

// Device trie node, 8 bytes:  [31:0] child bitmap over symbol classes,
// [47:32] id of the first child (children of a node are consecutive, in class order),
// [63:48] token id carried by the node (kNoToken if none).
inline uint64_t pack_node(uint32_t bitmap, uint32_t first_child, uint32_t token)
{
    return (uint64_t)bitmap | ((uint64_t)(first_child & 0xFFFFu) << 32) | ((uint64_t)(token & 0xFFFFu) << 48);
}

void set_error(const std::string &msg);

}  // namespace ecgb

struct ecgb_tokenizer {
    std::vector<uint64_t> nodes;       // breadth-first packed trie, node 0 = root
    uint8_t byte_to_class[256];        // raw byte -> symbol class (kOtherClass if none)
    uint16_t single_id[32];            // token id of the length-1 match of each class
    uint8_t class_to_byte[32];
    uint32_t n_classes = 0;
    uint32_t max_depth = 0;
    uint32_t n_merges = 0;
    // device copies
    uint64_t *nodes_dev = nullptr;
    uint8_t *lut_dev = nullptr;        // 256 B byte_to_class | 64 B single_id | 32 B class_to_byte
    int device = -1;
    int n_cus = 0;                     // compute units of `device` (persistent grid size)
};

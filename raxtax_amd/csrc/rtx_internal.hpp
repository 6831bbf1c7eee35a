// Internal declarations shared by the host-side C++ and the HIP translation unit of
// libraxtax_hip.so.  Not part of the ABI (include/raxtax_hip.h is).
#pragma once

#include <cstdint>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "raxtax_hip.h"

namespace rtx {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

constexpr uint32_t kNoNode = 0xFFFFFFFFu;

// Host thread budget (host_threads.cpp): CPUs of the affinity mask capped by the cgroup CPU quota -- what
// std::thread::available_parallelism gives the reference's rayon pool (main.rs:40-57) -- and a worker count for a pool that wants
// up to `want` threads while `sharers` handles of this process (rtx_raxtax_multi) and host_share() ranks on this host
// (rtx_set_host_share / LOCAL_WORLD_SIZE) work side by side.  Never std::thread::hardware_concurrency(): it ignores the quota.
unsigned available_parallelism();
unsigned host_share();
unsigned host_threads(unsigned want, unsigned sharers = 1);

// (rtx_api_index.hip) the device of a handle; a handle that shares its device with another one driven beside it runs on one stream
int index_device(const rtx_index *index);
void index_set_shared_device(rtx_index *index, bool shared);
uint32_t index_swap_min_subs(rtx_index *index, uint32_t v);  // returns the previous value
uint32_t index_swap_run_ahead(rtx_index *index, uint32_t v);  // RTX_OPT_RUN_AHEAD, returns the previous value
bool hw_queues_for_run_ahead();  // (host_threads.cpp) GPU_MAX_HW_QUEUES reads six or more: transfers do not share a hardware queue with kernels

#ifndef RTX_NODE_TYPES_DEFINED
#define RTX_NODE_TYPES_DEFINED
enum NodeType : uint8_t { kInner = 0, kTaxon = 1, kSequence = 2 };  // src/tree.rs:181-186
#endif

struct Node {  // src/tree.rs:188-194, children as arena indices
    std::string label;
    uint64_t lo = 0, hi = 0;  // confidence_range
    NodeType type = kInner;
    std::vector<uint32_t> children;
};

struct BytesHash {
    size_t operator()(std::string_view s) const noexcept;
};

// Breadth-first flattening handed to the device (see rtx_nodes_view).
struct FlatNodes {
    std::vector<uint32_t> begin, end, first_child, n_children, parent, depth;
    std::vector<uint8_t> type;
    uint32_t size() const { return (uint32_t)begin.size(); }
    uint32_t max_depth = 0;
};

// Builds parent/depth from first_child/n_children; validates the arrays.  Returns false
// (with set_error) if they do not describe a BFS-ordered tree rooted at node 0.
bool derive_flat_nodes(uint64_t n_refs, uint32_t n_nodes, const uint32_t *begin, const uint32_t *end,
                       const uint32_t *first_child, const uint32_t *n_children, const uint8_t *type,
                       FlatNodes &out);

}  // namespace rtx

// Host mirror of `Tree` (src/tree.rs:36-43).
struct rtx_tree {
    uint64_t n = 0;         // input sequences
    uint64_t num_tips = 0;  // Tree.num_tips
    std::vector<std::string> lineages;  // sorted (tree.rs:53-54,128-129)
    std::vector<uint64_t> orig_idx;     // sorted index -> input index
    std::vector<uint8_t> seq_bytes;     // sequences in sorted order
    std::vector<uint64_t> seq_off;
    std::vector<uint64_t> csr_off;      // 65537, Tree.k_mer_map
    std::vector<uint32_t> postings;
    std::unordered_map<std::string_view, std::vector<uint32_t>, rtx::BytesHash> sequences;  // Tree.sequences
    std::vector<rtx::Node> nodes;       // arena, root = 0 (Tree.root)
    rtx::FlatNodes flat;
};

struct rtx_queries {
    std::vector<std::string> labels;
    std::vector<uint8_t> bases;
    std::vector<uint64_t> base_off{0};
};

// C++ face of the host mirror of raxtax() (src/raxtax.rs:14-97): what the C++ CLI links
// against.  Same argument order and meaning as the reference function.
#pragma once

#include <functional>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "raxtax_hip.h"

namespace raxtax {

// (label, out_lines, tsv_lines?) -- the message of the reference's crossbeam channel
// (raxtax.rs:19,85-87).  Returns false when the sink is closed.
using Sender = std::function<bool(const std::string &, std::string &&, std::optional<std::string> &&)>;

// Returns RTX_OK or a negative RTX_ERR_* (RTX_ERR_SENDER = the reference's Err on a closed channel).
int raxtax(const std::vector<std::pair<std::string, std::vector<uint8_t>>> &queries, const rtx_tree *tree,
           rtx_index *index, bool skip_exact_matches, bool raw_confidence, size_t chunk_size, const Sender &sender,
           bool tsv);

}  // namespace raxtax

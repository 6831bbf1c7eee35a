"""raxtax_amd -- MI355X (gfx950) implementation of raxtax's per-query k-mer classification
hot path behind a C ABI (include/raxtax_hip.h).  See DESIGN.md."""
import os as _os

# hardware queues of the HIP runtime (read when it initialises; csrc/host_threads.cpp says why the library wants more than the default four):
# asked for here too, in case the process initialises HIP (torch.cuda ...) between this import and the loading of the library
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .api import (EvaluationResult, Index, Result, Tree, parse_query_fasta_str, parse_reference_fasta_str,  # noqa: F401
                  raxtax, raxtax_last_timing)
from ._lib import (RTX_RAW_CONFIDENCE, RTX_SKIP_EXACT_MATCHES, RtxError)  # noqa: F401

__all__ = ["Tree", "Index", "Result", "EvaluationResult", "raxtax", "raxtax_last_timing", "parse_reference_fasta_str",
           "parse_query_fasta_str", "RtxError", "RTX_SKIP_EXACT_MATCHES", "RTX_RAW_CONFIDENCE"]

"""raxtax_amd -- MI355X (gfx950) implementation of raxtax's per-query k-mer classification
hot path behind a C ABI (include/raxtax_hip.h).  See DESIGN.md."""
from .api import (EvaluationResult, Index, Result, Tree, parse_query_fasta_str, parse_reference_fasta_str,  # noqa: F401
                  raxtax, raxtax_last_timing)
from ._lib import (RTX_RAW_CONFIDENCE, RTX_SKIP_EXACT_MATCHES, RtxError)  # noqa: F401

__all__ = ["Tree", "Index", "Result", "EvaluationResult", "raxtax", "raxtax_last_timing", "parse_reference_fasta_str",
           "parse_query_fasta_str", "RtxError", "RTX_SKIP_EXACT_MATCHES", "RTX_RAW_CONFIDENCE"]

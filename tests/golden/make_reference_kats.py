#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json: the known-answer vectors (inputs and
expected outputs only) held by the reference's own unit tests for the hot path,
transcribed by hand from noahares/raxtax v1.5.0:

  F1  src/utils.rs:237-243   test_map
  F2  src/utils.rs:246-263   test_sequence_to_kmers
  F3  src/utils.rs:266-273   test_decompress_sequence
  F4  src/utils.rs:209-234   test_euclidean_norm / _distance / test_cosine_similarity
  F5  src/parser.rs:167-217  test_str_parser
  F6  src/parser.rs:220-233  test_query_parser
  F7  src/parser.rs:236-299  test_kmers
  F8  src/lineage.rs:192-239 test_tree_construction
  F9  src/lineage.rs:242-302 test_variable_lineage_length
  F10 src/lineage.rs:305-334 test_likelihood_edge_case
  P1  src/prob.rs:209-227    test_pmf            (property, tolerance 1e-7)
  P2  src/prob.rs:230-235    test_hit_prob       (property, tolerance 1e-7)

These are data, not code; nothing here is executed from /root/reference.
"""
import json
import math
from pathlib import Path

BADA = ">Badabing|Badabum;tax="

kats = {
    "source": "noahares/raxtax v1.5.0 inline unit tests (see make_reference_kats.py header)",
    "F1_map": {"in": [1, 2, 4, 8, 10], "out": [0, 1, 2, 3, None]},
    "F2_sequence_to_kmers": {
        "sequence": [1, 2, 1, 4, 8, 2, 8, 4, 1, 4, 8, 2, 8, 4, 1, 4],
        "kmers": [
            0b0001_0010_1101_1110,
            0b0010_1101_1110_0010,
            0b0100_1011_0111_1000,
            0b0111_1000_1011_0111,
            0b1000_1011_0111_1000,
            0b1011_0111_1000_1011,
            0b1101_1110_0010_1101,
            0b1110_0010_1101_1110,
        ],
    },
    "F3_decompress": {
        "sequence": [1, 2, 1, 4, 8, 2, 8, 4, 1, 4, 8, 2, 8, 4, 1, 4],
        "text": "ACAGTCTGAGTCTGAG",
    },
    "F4_norms": {
        "tol": 1e-7,
        "euclidean_norm": [
            {"v": [1.0, 2.0, 3.0, 4.0], "out": math.sqrt(30.0)},
            {"v": [0.5, 0.5, 0.25, 0.2], "out": math.sqrt(0.6025)},
        ],
        "euclidean_distance_l1": [
            {"a": [1.0, 0.0, 0.0], "b": [0.0, 1.0, 0.0], "out": math.sqrt(2.0)},
            {"a": [0.5, 0.1, 0.1], "b": [1.0, 1.0, 0.5], "out": 0.4100771455544949},
        ],
        "cosine_similarity": [
            {"a": [1.0, 0.0, 0.0], "b": [0.0, 1.0, 0.0], "out": 0.0},
            {"a": [0.5, 0.5], "b": [0.5, 0.5], "out": 1.0},
        ],
    },
    "F5_str_parser": {
        "fasta": "\n".join([
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species1;",
            "AAACCCTTTGGGA",
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species2;",
            "ATACGCTTTGGGA",
            BADA + "p:Phylum1,c:Class1,o:Order4,f:Family5,g:Genus2,s:Species3;",
            "ATCCGCTATGGGA",
            BADA + "p:Phylum1,c:Class2,o:Order2,f:Family3,g:Genus3,s:Species6;",
            "ATACGCTTTGCGT",
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species2;",
            "GTGCGCTATGCGA",
            BADA + "p:Phylum2,c:Class3,o:Order3,f:Family4,g:Genus4,s:Species5;",
            "ATACGCTTTGCGT",
        ]),
        "k_mer_map": {
            str(0b1_0101_1111_1110): [0],
            str(0b11_0001_1001_1111): [1, 4, 5],
            str(0b110_0111_0011_1010): [3],
        },
        "num_tips": 6,
        "lineages": [
            "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species1",
            "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species2",
            "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species2",
            "p:Phylum1,c:Class1,o:Order4,f:Family5,g:Genus2,s:Species3",
            "p:Phylum1,c:Class2,o:Order2,f:Family3,g:Genus3,s:Species6",
            "p:Phylum2,c:Class3,o:Order3,f:Family4,g:Genus4,s:Species5",
        ],
    },
    "F6_query_parser": [
        {"fasta": ">label1\nAAACCCTTTGGGA", "sequence": [1, 1, 1, 2, 2, 2, 8, 8, 8, 4, 4, 4, 1]},
        {"fasta": ">label1\nACGTWSMKRYBDHVN",
         "sequence": [1, 2, 4, 8, 9, 6, 3, 12, 5, 10, 14, 13, 11, 7, 15]},
    ],
    "F7_kmers": {
        "fasta": "\n".join([
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species1;",
            "AAACCCCGT",
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus1,s:Species1;",
            "TAACCCCGG",
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus2,s:Species3;",
            "TTTAAAACC",
            BADA + "p:Phylum1,c:Class1,o:Order1,f:Family1,g:Genus2,s:Species3;",
            "TTTAAAACA",
            BADA + "p:Phylum1,c:Class2,o:Order2,f:Family2,g:Genus3,s:Species4;",
            "AAACCCCGG",
        ]),
        "k_mer_map": {
            str(0b1_0101_0110): [0, 4],
            str(0b101_0101_1010): [1, 4],
            str(0b101_0101_1011): [0],
            str(0b1100_0001_0101_0110): [1],
            str(0b1111_0000_0000_0101): [2],
            str(0b1111_1100_0000_0001): [2, 3],
        },
    },
    "F8_tree_construction": {
        "lineages": [
            "Animalia,Chordata,Mammalia,Primates,Hominidae,Homo",
            "Animalia,Chordata,Mammalia,Primates,Hominidae,Pan",
            "Animalia,Chordata,Mammalia,Carnivora,Canidae,Canis",
            "Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis",
            "Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis",
        ],
        "sequence_code": 0, "sequence_len": 9,
        "confidence_values": [0.1, 0.3, 0.4, 0.004, 0.004],
        "expected": [
            ["Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis", [0.81, 0.81, 0.81, 0.8, 0.7, 0.7]],
            ["Animalia,Chordata,Mammalia,Carnivora,Canidae,Canis", [0.81, 0.81, 0.81, 0.8, 0.1, 0.1]],
            ["Animalia,Chordata,Mammalia,Primates,Hominidae,Pan", [0.81, 0.81, 0.81, 0.01, 0.01, 0.01]],
        ],
    },
    "F9_variable_lineage_length": {
        "lineages": [
            "Animalia,Chordata,Mammalia,Primates,Hominidae,Homo,Homo_sapiens",
            "Animalia,Chordata,Mammalia,Primates,Hominidae,Pan",
            "Animalia,Chordata,Mammalia,Carnivora,Canidae,Canis",
            "Animalia,Chordata,Mammalia,Carnivora,Doggo",
            "Animalia,Chordata,Mammalia,Mouse",
            "Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis",
            "Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis",
        ],
        "sequence_code": 0, "sequence_len": 9,
        "confidence_values": [0.05, 0.1, 0.3, 0.4, 0.1, 0.004, 0.004],
        "expected": [
            ["Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis", [0.96, 0.96, 0.96, 0.85, 0.7, 0.7]],
            ["Animalia,Chordata,Mammalia,Carnivora,Doggo", [0.96, 0.96, 0.96, 0.85, 0.1]],
            ["Animalia,Chordata,Mammalia,Carnivora,Canidae,Canis", [0.96, 0.96, 0.96, 0.85, 0.05, 0.05]],
            ["Animalia,Chordata,Mammalia,Mouse", [0.96, 0.96, 0.96, 0.1]],
            ["Animalia,Chordata,Mammalia,Primates,Hominidae,Pan", [0.96, 0.96, 0.96, 0.01, 0.01, 0.01]],
        ],
    },
    "F10_likelihood_edge_case": {
        "lineages": [
            "Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis",
            "Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis_ferrocius",
            "Animalia,Chordata,Mammalia,Carnivora,Canidae,Canis",
        ],
        "sequence_code": 0, "sequence_len": 9,
        "confidence_values": [0.004, 0.004, 0.004],
        "expected": [
            ["Animalia,Chordata,Mammalia,Carnivora,Felidae,Felis_ferrocius",
             [0.01, 0.01, 0.01, 0.01, 0.01, 0.01]],
        ],
    },
    "P1_pmf": {"t": 200, "n": 32, "m": 50, "tol": 1e-7},
    "P2_hit_prob": {"t": 400, "n": 200, "sizes_range": [0, 400], "tol": 1e-7},
}

out = Path(__file__).with_name("reference_kats.json")
out.write_text(json.dumps(kats, indent=1) + "\n")
print("wrote", out)

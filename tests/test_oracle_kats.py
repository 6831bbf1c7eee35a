"""Pins the CPU oracle against every known-answer vector the reference's own unit
tests hold for the hot path (tests/golden/reference_kats.json)."""
import math

import numpy as np
import pytest


def test_f1_map(oracle, kats):
    k = kats["F1_map"]
    assert [oracle.map_four_to_two_bit_repr(c) for c in k["in"]] == k["out"]


def test_f2_sequence_to_kmers(oracle, kats):
    k = kats["F2_sequence_to_kmers"]
    got = oracle.sequence_to_kmers(k["sequence"])
    assert list(got) == k["kmers"]
    assert all(a <= b for a, b in zip(got, got[1:]))


def test_f3_decompress(oracle, kats):
    k = kats["F3_decompress"]
    assert oracle.decompress_sequence(k["sequence"]) == k["text"]


def test_f4_norms(oracle, kats):
    k = kats["F4_norms"]
    for c in k["euclidean_norm"]:
        assert abs(oracle.euclidean_norm(c["v"]) - c["out"]) < k["tol"]
    for c in k["euclidean_distance_l1"]:
        assert abs(oracle.euclidean_distance_l1(c["a"], c["b"]) - c["out"]) < k["tol"]
    for c in k["cosine_similarity"]:
        assert abs(oracle.cosine_similarity(c["a"], c["b"]) - c["out"]) < k["tol"]


def test_f5_str_parser(oracle, kats):
    k = kats["F5_str_parser"]
    tree = oracle.parse_reference_fasta_str(k["fasta"])
    for kmer, ids in k["k_mer_map"].items():
        assert sorted(tree.kmer_list(int(kmer))) == ids
    assert tree.num_tips == k["num_tips"]
    assert tree.lineages == k["lineages"]


def test_f6_query_parser(oracle, kats):
    for c in kats["F6_query_parser"]:
        (label, seq), = oracle.parse_query_fasta_str(c["fasta"])
        assert label == "label1"
        assert list(seq) == c["sequence"]


def test_f7_kmers(oracle, kats):
    k = kats["F7_kmers"]
    tree = oracle.parse_reference_fasta_str(k["fasta"])
    for kmer, ids in k["k_mer_map"].items():
        assert sorted(tree.kmer_list(int(kmer))) == ids


@pytest.mark.parametrize("name", ["F8_tree_construction", "F9_variable_lineage_length",
                                  "F10_likelihood_edge_case"])
def test_f8_f10_lineage(oracle, kats, name):
    k = kats[name]
    seqs = [np.full(k["sequence_len"], k["sequence_code"], np.uint8) for _ in k["lineages"]]
    tree = oracle.tree_new(k["lineages"], seqs)
    rows = tree.lineage_evaluate(k["confidence_values"])
    got = [[tree.lineage(r["idx"]), r["conf"]] for r in rows]
    assert got == k["expected"]


def test_p1_pmf(oracle, kats):
    k = kats["P1_pmf"]
    t, n, m, tol = k["t"], k["n"], k["m"], k["tol"]
    ln_total = oracle.ln_binomial(t + n - 1, n)
    p = oracle.iterative_pmf_ln(t, n, m, ln_total)

    def pmf(i):  # prob.rs:178-206 closed form
        return math.exp(oracle.ln_binomial(m + i - 1, i) + oracle.ln_binomial((t - m) + (n - i) - 1, n - i)
                        - ln_total)

    p2 = [pmf(i) for i in range(n + 1)]
    assert abs(sum(math.exp(x) for x in p) - 1.0) < tol
    assert abs(sum(p2) - 1.0) < tol
    for a, b in zip(p, p2):
        assert abs(math.exp(a) - b) < tol


def test_p2_hit_prob(oracle, kats):
    k = kats["P2_hit_prob"]
    lo, hi = k["sizes_range"]
    probs = oracle.highest_hit_prob_per_reference(k["t"], k["n"], np.arange(lo, hi + 1))
    assert abs(probs.sum() - 1.0) < k["tol"]
    assert all(a <= b for a, b in zip(probs, probs[1:]))
    # scratch-model values of the last three entries
    assert abs(probs[-1] - 0.335) < 1e-6
    assert abs(probs[-2] - 0.223146911) < 1e-6
    assert abs(probs[-3] - 0.148515837) < 1e-6


def test_statrs_ln_gamma_restatement(oracle):
    # the Lanczos restatement of statrs::function::gamma::ln_gamma agrees with libm lgamma
    for x in [0.5, 1.0, 1.5, 2.0, 10.0, 171.0, 172.5, 500.0, 1000.0, 1302.0, 65536.0, 131070.0]:
        a = oracle.lib.orc_ln_gamma(x)
        b = math.lgamma(x)
        assert abs(a - b) <= 1e-13 * max(1.0, abs(b)), (x, a, b)
    for n in [0, 1, 5, 170, 171, 400, 1301]:
        assert abs(oracle.lib.orc_ln_factorial(n) - math.lgamma(n + 1.0)) <= 1e-12 * max(1.0, math.lgamma(n + 1.0))

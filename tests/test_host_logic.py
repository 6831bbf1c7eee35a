"""Host-side mirror (libraxtax_hip.so, no GPU needed): Tree::new / FASTA parsing / formatting
against the reference's known-answer vectors and against the oracle on seeded inputs."""
import numpy as np
import pytest

import raxtax_amd as rx
from raxtax_amd import synth


def test_f5_str_parser_host(kats):
    k = kats["F5_str_parser"]
    tree = rx.parse_reference_fasta_str(k["fasta"])
    for kmer, ids in k["k_mer_map"].items():
        assert sorted(tree.k_mer_map(int(kmer))) == ids
    assert tree.num_tips == k["num_tips"]
    assert tree.lineages == k["lineages"]


def test_f6_query_parser_host(kats):
    for c in kats["F6_query_parser"]:
        (label, seq), = rx.parse_query_fasta_str(c["fasta"])
        assert label == "label1"
        assert list(seq) == c["sequence"]


def test_f7_kmers_host(kats):
    k = kats["F7_kmers"]
    tree = rx.parse_reference_fasta_str(k["fasta"])
    for kmer, ids in k["k_mer_map"].items():
        assert sorted(tree.k_mer_map(int(kmer))) == ids


def test_parser_error_paths():
    with pytest.raises(rx.RtxError):
        rx.parse_reference_fasta_str("")                       # "File is empty", parser.rs:47-49
    with pytest.raises(rx.RtxError):
        rx.parse_reference_fasta_str("ACGT\n>x;tax=a;\nACGT")  # "Not a valid FASTA file"
    with pytest.raises(rx.RtxError):
        rx.parse_reference_fasta_str(">x;nolineage\nACGT")     # missing tax=...;
    with pytest.raises(rx.RtxError):
        rx.parse_reference_fasta_str(">x;tax=a;\nACGU")        # unexpected character
    with pytest.raises(rx.RtxError):
        rx.parse_reference_fasta_str(">x;tax=a;\n>y;tax=b;\nACGT")  # label/sequence count mismatch
    # comment and blank lines are skipped, lines are trimmed, case-insensitive
    t = rx.parse_reference_fasta_str(";c\n\n  >x;tax=a,b;  \n acgtacgta \n")
    assert t.num_tips == 1 and t.lineages == ["a,b"]


def test_query_parser_skip_and_multiline():
    qs = rx.parse_query_fasta_str(">a\nACGT\nACGT\n>b\nTTTT\n>c\nGG", ["b"])
    assert [q[0] for q in qs] == ["a", "c"]
    assert list(qs[0][1]) == [1, 2, 4, 8, 1, 2, 4, 8]


@pytest.mark.parametrize("n_refs", [300, 5184])
def test_tree_matches_oracle(oracle, n_refs):
    db = synth.make_db(n_refs)
    ot = oracle.tree_new_flat(db.lineages, db.seq_bytes, db.seq_off)
    ht = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    assert ht.num_tips == ot.num_tips == n_refs
    assert ht.lineages == ot.lineages
    assert np.array_equal(ht.original_index(), ot.original_index())
    oo, op = ot.csr()
    ho, hp = ht.csr()
    assert np.array_equal(oo, ho) and np.array_equal(op, hp)
    # exact-match map
    qs = synth.make_queries(db, 64, exact_frac=0.5)
    for i in range(qs.n):
        assert np.array_equal(ht.exact_matches(qs.seq(i)), ot.exact_matches(qs.seq(i)))
    # flattened taxonomy == the oracle's tree without childless Sequence nodes
    on = ot.nodes()
    hn = ht.nodes()
    keep = [i for i in range(len(on["type"])) if not (on["type"][i] == 2 and on["n_children"][i] == 0)]
    assert len(keep) == len(hn["type"])
    o_ranges = sorted((int(on["lo"][i]), int(on["hi"][i]), int(on["type"][i])) for i in keep)
    h_ranges = sorted((int(hn["begin"][i]), int(hn["end"][i]), int(hn["type"][i])) for i in range(len(hn["type"])))
    assert o_ranges == h_ranges
    # BFS invariants the device relies on
    nxt = 1
    for v in range(len(hn["type"])):
        if hn["n_children"][v]:
            assert hn["first_child"][v] == nxt
            for c in range(nxt, nxt + hn["n_children"][v]):
                assert hn["parent"][c] == v
            nxt += int(hn["n_children"][v])
    assert nxt == len(hn["type"])


def test_tree_variable_depth_and_duplicates(oracle, kats):
    k = kats["F9_variable_lineage_length"]
    seqs = [np.full(9, 0, np.uint8) for _ in k["lineages"]]
    ht = rx.Tree.new(k["lineages"], seqs)
    ot = oracle.tree_new(k["lineages"], seqs)
    assert ht.lineages == ot.lineages
    hn = ht.nodes()
    # all 7 identical sequences are exact matches of one another (ids in sorted order)
    assert list(ht.exact_matches(seqs[0])) == list(range(7))
    assert int(hn["end"][0]) == 7 and int(hn["type"][0]) == 0


def test_native_result_pack_equals_python_pack():
    """rtx_result_pack (the record a rank ships in the multi-GPU gather) == dist_util.pack_records."""
    import ctypes as C

    from raxtax_amd import _lib, dist_util

    lib = _lib.load()
    rng = np.random.default_rng(5)
    nq = 1000
    count = rng.integers(0, 5, nq).astype(np.uint32)
    order = rng.permutation(nq)                         # rows stored in an order of their own
    begin = np.zeros(nq, np.uint64)
    run = 0
    for q in order:
        begin[q] = run
        run += int(count[q])
    nr = run
    lin = rng.integers(0, 50000, nr).astype(np.uint32)
    node = np.zeros(nr, np.uint32)
    depth = rng.integers(1, 14, nr).astype(np.uint32)    # deeper than 8 levels: every level must travel
    conf = np.zeros((nr, 32))
    conf[:, :13] = rng.integers(0, 101, (nr, 13)) / 100.0
    conf[np.arange(32)[None, :] >= depth[:, None]] = 0.0
    local = rng.random(nr)
    gs = rng.random(nq)
    t = rng.integers(0, 652, nq).astype(np.uint32)
    status = (rng.random(nq) < 0.05).astype(np.uint8)
    lin[0] = 0xFFFFFFF0                                  # reference ids are u32 (tree.rs:21-22)
    v = _lib.ResultView()
    v.n_queries, v.n_rows = nq, nr
    P = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    v.t, v.status, v.global_signal = P(t, C.c_uint32), P(status, C.c_uint8), P(gs, C.c_double)
    v.row_begin, v.row_count = P(begin, C.c_uint64), P(count, C.c_uint32)
    v.row_lineage, v.row_node, v.row_depth = P(lin, C.c_uint32), P(node, C.c_uint32), P(depth, C.c_uint32)
    v.row_conf, v.row_local_signal = P(conf, C.c_double), P(local, C.c_double)
    need = lib.rtx_result_pack(C.byref(v), None, 0)
    buf = np.zeros(need, np.uint8)
    assert lib.rtx_result_pack(C.byref(v), P(buf, C.c_uint8), need) == need
    want = dist_util.pack_records(None, lin, depth, conf, local, gs, row_begin=begin, row_count=count, t=t, status=status)
    assert np.array_equal(buf, want)
    back = dist_util.unpack_records(buf)
    assert np.array_equal(np.diff(back["row_off"]), count)
    assert np.array_equal(back["t"], t) and np.array_equal(back["status"], status)
    src = np.repeat(begin.astype(np.int64) - back["row_off"][:-1], count) + np.arange(nr)
    assert np.array_equal(back["row_lineage"], lin[src]) and back["row_lineage"].dtype == np.uint32
    assert np.array_equal(back["row_conf"], conf[src][:, : back["row_conf"].shape[1]]) and back["row_conf"].shape[1] == 13


def _messy_fasta(n_records, seed, reference):
    """Several MiB of FASTA with everything the line rules of parser.rs have to cope with: wrapped sequences,
    blank and ';' comment lines, CRLF, indented headers, headers without bases (queries only), lower case,
    ambiguity codes.  Large enough to be cut into several pieces by the parallel parsers."""
    rng = np.random.default_rng(seed)
    letters = np.frombuffer(b"ACGTacgtNRYKMSWBDHV", dtype=np.uint8)
    out = []
    for i in range(n_records):
        head = f">r{i};tax=p:P{i % 7},c:C{i % 31},s:S{i % 997};" if reference else f">q{i} some description"
        if rng.random() < 0.02:
            head = "  " + head + "  "
        out.append(head + ("\r" if rng.random() < 0.1 else ""))
        if not reference and rng.random() < 0.01:
            continue                                        # header without bases: replaced by the next header
        L = int(rng.integers(200, 700))
        seq = letters[rng.integers(0, len(letters) if rng.random() < 0.05 else 4, L)].tobytes().decode()
        w = int(rng.integers(40, 120))
        for a in range(0, L, w):
            out.append(seq[a:a + w] + ("\r" if rng.random() < 0.02 else ""))
            if rng.random() < 0.01:
                out.append("")
            if rng.random() < 0.01:
                out.append("; a comment >with a bracket")
    return "\n".join(out) + "\n"


def test_parallel_fasta_parsers_equal_the_serial_oracle(oracle):
    """rtx_queries_parse_fasta / rtx_tree_parse_reference_fasta cut large inputs at header lines and parse the
    pieces on several threads; the result must be what the reference's sequential rules give (oracle)."""
    qtext = _messy_fasta(12000, 1, reference=False)
    assert len(qtext) > 4 << 20
    want = oracle.parse_query_fasta_str(qtext, ["q5", "q11999"])
    got = rx.parse_query_fasta_str(qtext, ["q5", "q11999"])
    assert len(got) == len(want) > 11000
    for (gl, gs), (wl, ws) in zip(got, want):
        assert gl == wl and np.array_equal(gs, ws)
    rtext = _messy_fasta(9000, 2, reference=True)
    assert len(rtext) > 3 << 20
    otree = oracle.parse_reference_fasta_str(rtext)
    tree = rx.parse_reference_fasta_str(rtext)
    assert tree.num_tips == otree.num_tips and tree.lineages == otree.lineages
    off, post = tree.csr()
    ooff, opost = otree.csr()
    assert np.array_equal(off, ooff) and np.array_equal(post, opost)


@pytest.mark.parametrize("block", [1500, 40_000, 1_000_000])
def test_blockwise_query_parsing_equals_whole_file(oracle, block):
    """rtx_fasta_block_end + rtx_queries_parse_fasta_block (the CLI's streamed ingest): reading a file in blocks cut in
    front of header lines gives exactly the records of the whole-file parse, including headers without bases at block
    ends, comment lines and a skip list."""
    import ctypes as C

    from raxtax_amd import _lib

    lib = _lib.load()
    text = _messy_fasta(3000, 7, reference=False).encode()
    want = oracle.parse_query_fasta_str(text.decode(), ["q17"])
    got = []
    buf = b""
    pos = 0
    first = True
    skip = (C.c_char_p * 1)(b"q17")
    while True:
        chunk = text[pos:pos + block]
        pos += len(chunk)
        buf += chunk
        eof = pos >= len(text)
        end = len(buf)
        flags = 0 if first else 2
        if not eof:
            end = lib.rtx_fasta_block_end(buf, len(buf))
            if end == 0:
                continue
            flags |= 1
        h = C.c_void_p()
        _lib.check(lib.rtx_queries_parse_fasta_block(buf, end, skip, 1, flags, C.byref(h)))
        n = lib.rtx_queries_len(h)
        bases, off = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
        lib.rtx_queries_data(h, C.byref(bases), C.byref(off))
        o = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
        b = np.ctypeslib.as_array(bases, shape=(max(int(o[-1]), 1),)).copy()
        for i in range(n):
            got.append((lib.rtx_queries_label(h, i).decode(), b[int(o[i]):int(o[i + 1])]))
        lib.rtx_queries_destroy(h)
        buf = buf[end:]
        first = False
        if eof:
            break
    assert len(got) == len(want)
    for (gl, gs), (wl, ws) in zip(got, want):
        assert gl == wl and np.array_equal(gs, ws)


def test_pack_bases_two_per_byte():
    """The packing of the upload (rtx_batch_prefetch: 4-bit codes of parser.rs:11-34, two per byte over PCIe) against numpy, at sizes
    around the 32-base SSE step and the thread cuts, odd lengths, and a batch with a byte above 15 (reported, sent unpacked)."""
    import ctypes as C

    from raxtax_amd import _lib
    from raxtax_amd._lib import ptr, u8p

    lib = _lib.load()
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 31, 32, 33, 63, 64, 65, 1000, (1 << 20) + 77, (3 << 20) + 1):
        b = rng.integers(0, 16, size=max(n, 1), dtype=np.uint8)[:n]
        out = np.full((n + 1) // 2 + 8, 0xEE, np.uint8)
        assert lib.rtx_pack_bases(ptr(b, u8p) if n else None, n, ptr(out, u8p)) == 1
        pad = np.concatenate([b, np.zeros(n & 1, np.uint8)])
        want = pad[0::2] | (pad[1::2] << 4)
        assert np.array_equal(out[: (n + 1) // 2], want), n
        assert (out[(n + 1) // 2:] == 0xEE).all(), n          # nothing written behind the packed bytes
    bad = rng.integers(0, 16, size=(1 << 20) + 5, dtype=np.uint8)
    bad[777_777] = 0x41
    out = np.zeros(len(bad) // 2 + 8, np.uint8)
    assert lib.rtx_pack_bases(ptr(bad, u8p), len(bad), ptr(out, u8p)) == 0


def test_format_query_numbers_are_printf_numbers():
    """rtx_format_query prints confidences "{:.2}" and signals "{:.5}" (lineage.rs:17-29) without snprintf (host_format.cpp: put_fixed);
    the digits must be printf's / Rust's -- the exact binary value rounded half to even -- for every value, also those that sit on or
    next to a rounding boundary (there the fast path hands over to snprintf)."""
    import ctypes as C

    from raxtax_amd import _lib

    lib = _lib.load()
    tree = rx.Tree.new(["k:A,p:B,c:C,o:D,f:E,g:F,s:G"], [np.array([1, 2, 4, 8, 1, 2, 4, 8, 1, 2], np.uint8)])
    rng = np.random.default_rng(17)
    vals = list(rng.random(3000)) + list(rng.random(500) * 1e-3) + list(rng.random(200) * 50) + [0.0, 1.0, 0.5, 0.000005, 0.0000149999999, 0.123455,
            0.123465, 2.5e-6, 7.5e-6, 0.999995, 0.9999949999, 0.99999500001, 1e-300, 123456.7890149, 999999.999994, 1e6, 3e7, float("inf")]
    vals += [k / 1e5 + 5e-6 for k in range(0, 2000, 7)] + [np.nextafter(k / 1e5 + 5e-6, 0) for k in range(0, 2000, 7)] + [np.nextafter(k / 1e5 + 5e-6, 1) for k in range(0, 2000, 7)]
    nq = len(vals)
    depth = np.full(nq, 7, np.uint32)
    conf = np.zeros((nq, 32))
    conf[:, :7] = rng.integers(0, 101, (nq, 7)) / 100.0
    local = np.array(vals, dtype=np.float64)
    gs = np.array(vals[::-1], dtype=np.float64)
    begin = np.arange(nq, dtype=np.uint64)
    count = np.ones(nq, np.uint32)
    lin = np.zeros(nq, np.uint32)
    t = np.full(nq, 3, np.uint32)
    status = np.zeros(nq, np.uint8)
    v = _lib.ResultView()
    v.n_queries, v.n_rows = nq, nq
    P = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    v.t, v.status, v.global_signal = P(t, C.c_uint32), P(status, C.c_uint8), P(gs, C.c_double)
    v.row_begin, v.row_count = P(begin, C.c_uint64), P(count, C.c_uint32)
    v.row_lineage, v.row_node, v.row_depth = P(lin, C.c_uint32), P(lin, C.c_uint32), P(depth, C.c_uint32)
    v.row_conf, v.row_local_signal = P(conf, C.c_double), P(local, C.c_double)
    buf = C.create_string_buffer(8192)
    seq = np.array([1, 2, 4, 8], np.uint8)
    for q in range(nq):
        n = lib.rtx_format_query(tree._h, C.byref(v), q, b"q", P(seq, C.c_uint8), 4, None, 0, 2, buf, 8192, None, 0, None)
        assert n > 0
        want = "q\tk:A,p:B,c:C,o:D,f:E,g:F,s:G\t" + ",".join("%.2f" % c for c in conf[q, :7]) + "\t%.5f\t%.5f" % (local[q], gs[q])
        assert buf.raw[:n].decode() == want, (q, local[q], gs[q])


def test_a_taxonomy_deeper_than_the_device_walk_is_refused_with_its_lineage():
    """RTX_MAX_DEPTH = 32 levels per result row (the reference has no limit, lineage.rs:119-179): a deeper lineage is refused where the tree
    is built, and the error names it."""
    import raxtax_amd as rx

    seq = np.array([1, 2, 4, 8] * 5, np.uint8)
    ok = ",".join(f"l{i}" for i in range(32))
    deep = ",".join(f"d{i}" for i in range(33))
    rx.Tree.new([ok, "a,b"], [seq, seq])                       # 32 levels: fine
    with pytest.raises(rx.RtxError) as e:
        rx.Tree.new([ok, deep], [seq, seq])
    assert e.value.code == -6 and "33 levels" in str(e.value) and "d0,d1,d2" in str(e.value)
    with pytest.raises(rx.RtxError) as e:
        rx.parse_reference_fasta_str(f">x;tax={deep};\nACGTACGTACGTACGTAAAA\n>y;tax=a,b;\nACGTACGTACGTACGTAAAC\n")
    assert e.value.code == -6


def test_the_package_asks_for_hardware_queues_before_hip_can_initialise():
    """raxtax_amd/__init__.py (and the library's own initialiser, csrc/host_threads.cpp): GPU_MAX_HW_QUEUES is exported -- without overwriting a value the
    process came with -- so that the transfer streams of a handle do not share a hardware queue with its compute streams (RTX_OPT_RUN_AHEAD needs that)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    ROOT = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    code = "import os, raxtax_amd; print(os.environ.get('GPU_MAX_HW_QUEUES'))"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(ROOT)).stdout.strip() == "8"
    env["GPU_MAX_HW_QUEUES"] = "5"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(ROOT)).stdout.strip() == "5"

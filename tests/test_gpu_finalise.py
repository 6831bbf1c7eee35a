"""The result rows are finalised on the device (rtx_finalise.hip: sort lineage.rs:91-93, local signal lineage.rs:95-102, final layout) and the
host only copies them: a download must not depend on the CPUs the library may use, and must equal the oracle's rows."""
from pathlib import Path

import numpy as np
import pytest

FIELDS = ("row_off", "row_lineage", "row_node", "row_depth", "row_conf", "row_local_signal", "global_signal", "t", "status")
FASTA = Path(__file__).resolve().parent / "golden" / "diptera_subset.fasta"


@pytest.mark.gpu
def test_rows_do_not_depend_on_the_host_share(oracle):
    """Real barcodes (many rows per query, ties, several sub-batches so that the download streams): the same batch with the library's host
    pools sized for 1 and for 8 ranks on this host -- every field of the view identical; the rows in the oracle's order with its local signal."""
    import raxtax_amd as rx

    lib = rx._lib.load()
    text = FASTA.read_text()
    tree = rx.parse_reference_fasta_str(text)
    otree = oracle.parse_reference_fasta_str(text)
    queries = rx.parse_query_fasta_str(text)
    reps = 40                                                     # 24 000 queries: four sub-batches
    seqs = [s for _, s in queries] * reps
    bases = np.concatenate(seqs)
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    ix = rx.Index(tree)
    got = {}
    try:
        for share in (1, 8):
            rx._lib.check(lib.rtx_set_host_share(share))
            ix.upload(bases, off)
            ix.run(rx.RTX_SKIP_EXACT_MATCHES)
            got[share] = ix.download()
    finally:
        rx._lib.check(lib.rtx_set_host_share(1))
    a, b = got[1], got[8]
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.row_conf.shape[1] == max(l.count(",") + 1 for l in tree.lineages)
    n_multi = 0
    for q in range(0, len(queries), 7):
        rows, _ = otree.classify(queries[q][1], skip_exact=True, raw_confidence=True)
        for rep in (0, reps - 1):
            mine = a.rows(q + rep * len(queries))
            assert len(mine) == len(rows)
            # (exact ties between sibling taxa may pick another lineage: compare what does not depend on them)
            assert [r.confidence_values for r in mine] == [r["conf"] for r in rows]
            assert np.allclose([r.local_signal for r in mine], [r["local_signal"] for r in rows], rtol=0, atol=1e-9) or \
                sorted(round(r.local_signal, 8) for r in mine) == sorted(round(r["local_signal"], 8) for r in rows)
        n_multi += len(rows) > 1
    assert n_multi > 10


@pytest.mark.gpu
def test_view_layout_of_abi_5():
    """row_conf is [n_rows][row_conf_stride] with the stride of the deepest lineage; the byte arrays hold what rtx_result_pack ships."""
    import ctypes as C

    import raxtax_amd as rx
    from raxtax_amd import dist_util, synth

    db = synth.make_db(3000, fanouts=(2, 2, 3, 3, 3, 2))
    qs = synth.make_queries(db, 500, exact_frac=0.2)
    tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
    ix = rx.Index(tree)
    ix.upload(qs.bases, qs.base_off)
    ix.run(0)
    view = ix.download(copy=False)
    depth_max = max(l.count(",") + 1 for l in db.lineages)
    assert view.row_conf_stride == depth_max
    nr = int(view.n_rows)
    d8 = np.ctypeslib.as_array(view.row_depth_u8, shape=(nr,))
    d32 = np.ctypeslib.as_array(view.row_depth, shape=(nr,))
    hund = np.ctypeslib.as_array(view.row_conf_hundredths, shape=(nr, depth_max))
    conf = np.ctypeslib.as_array(view.row_conf, shape=(nr, depth_max))
    assert np.array_equal(d8, d32) and np.array_equal(hund / 100.0, conf)
    lib = rx._lib.load()
    need = lib.rtx_result_pack(C.byref(view), None, 0)
    buf = np.zeros(need, np.uint8)
    assert lib.rtx_result_pack(C.byref(view), buf.ctypes.data_as(C.POINTER(C.c_uint8)), need) == need
    rec = dist_util.unpack_records(buf)
    from raxtax_amd.api import Result

    res = Result(view)
    assert np.array_equal(rec["row_lineage"], res.row_lineage) and np.array_equal(rec["row_conf"], res.row_conf[:, : rec["row_conf"].shape[1]])
    assert np.array_equal(rec["row_local_signal"], res.row_local_signal)

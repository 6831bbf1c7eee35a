"""The reference's `.bin` database (bincode 1.3.3 default options, src/tree.rs:147-164).
No upstream test pins the layout, so it is cross-checked structurally: the file written by the host mirror is
parsed with an independent struct-level reader (this file) and compared field by field with the ORACLE's tree;
then it is loaded back and must reproduce the original."""
import struct

import numpy as np
import pytest

import raxtax_amd as rx
from raxtax_amd import synth


class Bincode:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def u32(self):
        v, = struct.unpack_from("<I", self.d, self.p); self.p += 4; return v

    def u64(self):
        v, = struct.unpack_from("<Q", self.d, self.p); self.p += 8; return v

    def raw(self, n):
        b = self.d[self.p:self.p + n]; self.p += n; return b

    def string(self):
        return self.raw(self.u64()).decode()

    def node(self, out, parent):
        me = len(out)
        out.append(None)
        label = self.string()
        lo, hi = self.u64(), self.u64()
        n = self.u64()
        for _ in range(n):
            self.node(out, me)
        out[me] = (label, lo, hi, self.u32(), n, parent)


def parse_bin(data: bytes):
    b = Bincode(data)
    nodes = []
    b.node(nodes, -1)
    lineages = [b.string() for _ in range(b.u64())]
    sequences = {}
    for _ in range(b.u64()):
        key = b.raw(b.u64())
        sequences[key] = [b.u32() for _ in range(b.u64())]
    n_lists = b.u64()
    k_mer_map = [[b.u32() for _ in range(b.u64())] for _ in range(n_lists)]
    num_tips = b.u64()
    assert b.p == len(data)
    return nodes, lineages, sequences, k_mer_map, num_tips


@pytest.fixture(scope="module")
def small_db():
    db = synth.make_db(300, fanouts=(2, 2, 2, 2, 2, 2))
    # a duplicated sequence and variable-depth lineages, as in the reference's own tests
    lineages = list(db.lineages)
    lineages[7] = lineages[7].rsplit(",", 1)[0]
    seq_bytes = db.seq_bytes.copy()
    L = db.length
    seq_bytes[11 * L:12 * L] = seq_bytes[10 * L:11 * L]
    return lineages, seq_bytes, db.seq_off


def test_bin_layout_matches_oracle_tree(tmp_path, oracle, small_db):
    lineages, seq_bytes, seq_off = small_db
    tree = rx.Tree.new_flat(lineages, seq_bytes, seq_off)
    otree = oracle.tree_new_flat(lineages, seq_bytes, seq_off)
    path = tmp_path / "db.bin"
    tree.save_to_file(path)
    nodes, lins, sequences, k_mer_map, num_tips = parse_bin(path.read_bytes())
    assert num_tips == otree.num_tips == 300
    assert lins == otree.lineages
    assert len(k_mer_map) == 65536
    off, post = otree.csr()
    for k in range(0, 65536, 97):
        assert k_mer_map[k] == list(post[int(off[k]):int(off[k + 1])])
    assert sum(map(len, k_mer_map)) == len(post)
    # Tree.root in pre-order, all node types (incl. the per-reference Sequence nodes)
    on = otree.nodes()
    assert len(nodes) == len(on["type"])
    for i, (label, lo, hi, ty, nch, parent) in enumerate(nodes):
        assert (label, lo, hi, ty, nch, parent) == (on["label"][i], int(on["lo"][i]), int(on["hi"][i]), int(on["type"][i]),
                                                    int(on["n_children"][i]), int(on["parent"][i]))
    # Tree.sequences: unique encoded sequence -> ids in lineage-sorted order
    orig = otree.original_index()
    assert sum(len(v) for v in sequences.values()) == 300
    for key, ids in sequences.items():
        assert ids == sorted(ids)
        for r in ids:
            o = int(orig[r])
            assert bytes(seq_bytes[int(seq_off[o]):int(seq_off[o + 1])]) == key
    assert max(len(v) for v in sequences.values()) >= 2     # duplicated sequences share one map entry


def test_bin_round_trip(tmp_path, small_db):
    lineages, seq_bytes, seq_off = small_db
    tree = rx.Tree.new_flat(lineages, seq_bytes, seq_off)
    path = tmp_path / "db.bin"
    tree.save_to_file(path)
    back = rx.Tree.load_from_file(path)
    assert back.num_tips == tree.num_tips and back.lineages == tree.lineages
    a, b = tree.csr(), back.csr()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    na, nb = tree.nodes(), back.nodes()
    for k in na:
        assert np.array_equal(na[k], nb[k]), k
    orig = tree.original_index()
    for r in (0, 10, 11, 299):
        o = int(orig[r])
        s = seq_bytes[int(seq_off[o]):int(seq_off[o + 1])]
        assert np.array_equal(back.exact_matches(s), tree.exact_matches(s))
    # a second save of the loaded tree parses to the same content (map order may differ)
    path2 = tmp_path / "db2.bin"
    back.save_to_file(path2)
    p1, p2 = parse_bin(path.read_bytes()), parse_bin(path2.read_bytes())
    assert p1[0] == p2[0] and p1[1] == p2[1] and p1[2] == p2[2] and p1[3] == p2[3] and p1[4] == p2[4]


def test_bin_rejects_garbage(tmp_path, small_db):
    lineages, seq_bytes, seq_off = small_db
    tree = rx.Tree.new_flat(lineages, seq_bytes, seq_off)
    path = tmp_path / "db.bin"
    tree.save_to_file(path)
    data = path.read_bytes()
    (tmp_path / "trunc.bin").write_bytes(data[: len(data) // 2])
    (tmp_path / "fasta.bin").write_text(">x;tax=a;\nACGT\n")
    for name in ("trunc.bin", "fasta.bin", "missing.bin"):
        with pytest.raises(rx.RtxError):          # Tree::load_from_file errors -> the caller falls back to FASTA
            rx.Tree.load_from_file(tmp_path / name)

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


def write_bin_from_oracle(otree, lineages_sorted, seq_of_sorted, rng) -> bytes:
    """An independent bincode 1.3.3 WRITER of `Tree` (tree.rs:36-43,181-194: little-endian, u64 lengths, u32 enum tags,
    structs as the concatenation of their fields, the HashMap in arbitrary order) fed from the ORACLE's tree: what the
    reference's `Tree::save_to_file` would produce for it."""
    out = bytearray()
    u32 = lambda v: out.extend(struct.pack("<I", v))
    u64 = lambda v: out.extend(struct.pack("<Q", v))

    def string(s):
        b = s.encode()
        u64(len(b))
        out.extend(b)

    on = otree.nodes()
    children = [[] for _ in on["type"]]
    for i, p in enumerate(on["parent"]):
        if p >= 0:
            children[int(p)].append(i)

    def node(i):                      # Node { label, confidence_range, children, node_type }
        string(on["label"][i])
        u64(int(on["lo"][i]))
        u64(int(on["hi"][i]))
        u64(len(children[i]))
        for c in children[i]:
            node(c)
        u32(int(on["type"][i]))

    node(0)
    u64(len(lineages_sorted))
    for l in lineages_sorted:
        string(l)
    groups = {}
    for r, s in enumerate(seq_of_sorted):
        groups.setdefault(bytes(s), []).append(r)
    keys = list(groups)
    rng.shuffle(keys)                 # a HashMap has no order
    u64(len(keys))
    for k in keys:
        u64(len(k))
        out.extend(k)
        u64(len(groups[k]))
        for r in groups[k]:
            u32(r)
    off, post = otree.csr()
    u64(65536)
    for k in range(65536):
        a, b = int(off[k]), int(off[k + 1])
        u64(b - a)
        out.extend(post[a:b].astype("<u4").tobytes())
    u64(otree.num_tips)
    return bytes(out)


def test_bin_written_by_an_independent_encoder_loads(tmp_path, oracle, small_db):
    """The other direction: a file produced by an independent encoder of the format (from the oracle's tree) is read by
    rtx_tree_load_bin and gives the tree the host mirror builds from the same input."""
    lineages, seq_bytes, seq_off = small_db
    otree = oracle.tree_new_flat(lineages, seq_bytes, seq_off)
    orig = otree.original_index()
    seq_sorted = [seq_bytes[int(seq_off[int(o)]):int(seq_off[int(o) + 1])] for o in orig]
    data = write_bin_from_oracle(otree, otree.lineages, seq_sorted, np.random.default_rng(3))
    path = tmp_path / "independent.bin"
    path.write_bytes(data)
    back = rx.Tree.load_from_file(path)
    tree = rx.Tree.new_flat(lineages, seq_bytes, seq_off)
    assert back.num_tips == tree.num_tips and back.lineages == tree.lineages
    a, b = tree.csr(), back.csr()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    na, nb = tree.nodes(), back.nodes()
    for k in na:
        assert np.array_equal(na[k], nb[k]), k
    for r in (0, 10, 11, 150, 299):
        assert np.array_equal(back.exact_matches(seq_sorted[r]), tree.exact_matches(seq_sorted[r]))
    # and the mirror's own file parses to the same content as the independent one (map order aside)
    tree.save_to_file(tmp_path / "own.bin")
    p1, p2 = parse_bin(data), parse_bin((tmp_path / "own.bin").read_bytes())
    assert p1[0] == p2[0] and p1[1] == p2[1] and p1[2] == p2[2] and p1[3] == p2[3] and p1[4] == p2[4]


def test_bin_loader_rejects_hostile_counts(tmp_path, oracle, small_db):
    """Counts are checked against the bytes that are left BEFORE they are multiplied (a posting count of 2^62 wraps
    to 0 bytes), and posting lists must ascend strictly (the reference-sharded index cuts them with lower_bound)."""
    lineages, seq_bytes, seq_off = small_db
    otree = oracle.tree_new_flat(lineages, seq_bytes, seq_off)
    orig = otree.original_index()
    seq_sorted = [seq_bytes[int(seq_off[int(o)]):int(seq_off[int(o) + 1])] for o in orig]
    good = bytearray(write_bin_from_oracle(otree, otree.lineages, seq_sorted, np.random.default_rng(4)))
    off, post = otree.csr()
    k = int(np.argmax(np.diff(off.astype(np.int64)) >= 2))            # a k-mer with at least two postings
    tail = sum(8 + 4 * int(off[j + 1] - off[j]) for j in range(k, 65536)) + 8
    pos = len(good) - tail                                              # where the list of k-mer k starts
    assert struct.unpack_from("<Q", good, pos)[0] == int(off[k + 1] - off[k])
    bad = bytearray(good)
    struct.pack_into("<Q", bad, pos, 1 << 62)
    swapped = bytearray(good)
    a, b = struct.unpack_from("<II", good, pos + 8)
    struct.pack_into("<II", swapped, pos + 8, b, a)
    for name, data in (("huge.bin", bad), ("unsorted.bin", swapped)):
        (tmp_path / name).write_bytes(bytes(data))
        with pytest.raises(rx.RtxError):
            rx.Tree.load_from_file(tmp_path / name)
    (tmp_path / "good.bin").write_bytes(bytes(good))
    assert rx.Tree.load_from_file(tmp_path / "good.bin").num_tips == 300

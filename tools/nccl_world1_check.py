"""Exercises the RCCL code path of raxtax_amd.dist_util (device tensors, pinned staging, asynchronous gather) with a
process group of ONE rank on one GPU, overlapped with a classification running on the library's own stream -- what
bench.py does at N > 1, minus the other ranks.  Usage: python tools/nccl_world1_check.py"""
import ctypes, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import torch.distributed as dist
import raxtax_amd as rx
from raxtax_amd import dist_util, synth

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
db = synth.make_db(20000); qs = synth.make_queries(db, 40000)
tree = rx.Tree.new_flat(db.lineages, db.seq_bytes, db.seq_off)
ix = rx.Index(tree)
ix.upload(qs.bases, qs.base_off, *ix.exact_matches(qs.bases, qs.base_off))
lib = rx._lib.load()
cache = [{}, {}]; bufs = [None, None]; pending = None; prev = None; got = []
def ship(view, k):
    global pending
    need = lib.rtx_result_pack(ctypes.byref(view), None, 0)
    if bufs[k] is None or bufs[k].shape[0] < need: bufs[k] = dist_util.pinned_bytes(int(need * 1.25) + 64)
    n = lib.rtx_result_pack(ctypes.byref(view), bufs[k].ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), bufs[k].shape[0])
    if pending is not None: got.append(dist_util.gather_finish(pending))
    pending = dist_util.gather_start(dist, bufs[k][:n], 0, 1, device="cuda", cache=cache[k])
t0 = time.perf_counter()
for i in range(6):
    ix.run(0)
    if prev is not None: ship(prev, i & 1)
    prev = ix.download(copy=False)
ship(prev, 0); got.append(dist_util.gather_finish(pending))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 6
ref = dist_util.unpack_records(got[0][0])
for g in got[1:]:
    u = dist_util.unpack_records(g[0])
    assert u["n_queries"] == 40000 and np.array_equal(u["row_lineage"], ref["row_lineage"]) and np.array_equal(u["row_conf"], ref["row_conf"])
res = ix.download()
assert np.array_equal(ref["row_off"], res.row_off) and np.array_equal(ref["row_lineage"], res.row_lineage)
print(f"nccl world-1 gather ok: {len(got)} gathers, {dt * 1e3:.2f} ms per step of 40000 queries")
dist.destroy_process_group()

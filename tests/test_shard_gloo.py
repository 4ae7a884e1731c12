"""The N>1 path on CPU: world_size-2 gloo processes exercise exactly the code bench.py runs
under RCCL -- shard_range + broadcast_csr -- and check, with the oracle as the checker, that
per-rank shards of the batch reproduce the full-batch result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    oracle = ge.load_oracle()
    synth = pkg.synth
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    s = synth.shape("shard", 7, 16, 14, 14, 24, 3, pad=1, group=2, sparsity=0.8)
    mg, cg = s.M // s.group, s.C // s.group
    csr = None
    if rank == 0:                       # only rank 0 holds the dense blob (WeightAlign there)
        w = synth.pruned_weights(s, 5)
        rps, cis, vas, ngs = [], [], [], []
        for g in range(s.group):
            rp, ci, va = oracle.dense2csr(w[g * mg:(g + 1) * mg].reshape(mg, cg * s.KH * s.KW))
            rps.append(rp); cis.append(ci); vas.append(va); ngs.append(len(ci))
        csr = (np.concatenate(rps), np.concatenate(cis), np.concatenate(vas), np.array(ngs, np.int32))
    rp, ci, va, ng = pkg.shard.broadcast_csr(csr, s.group, s.group * (mg + 1), synth.nnz_of(s), src=0)
    # rebuild the dense weights from the broadcast CSR and run this rank's shard of the batch
    w_rx = np.zeros((s.M, cg * s.KH * s.KW), np.float32)
    off = 0
    for g in range(s.group):
        r = rp[g * (mg + 1):(g + 1) * (mg + 1)]
        for m in range(mg):
            w_rx[g * mg + m, ci[off + r[m]:off + r[m + 1]]] = va[off + r[m]:off + r[m + 1]]
        off += ng[g]
    b, e = pkg.shard.shard_range(s.N, rank, world)
    x = synth.activations(s, 9, b, e - b)          # seeded by GLOBAL image index
    geom = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, group=s.group)
    top = oracle.conv_forward(geom, x, w_rx.reshape(s.M, cg, s.KH, s.KW), None, gate=False)
    # the byte-blob broadcast bench.py uses for the aligned form (size known to the source only)
    blob = np.arange(100003, dtype=np.uint64).view(np.uint8)[5:] if rank == 0 else None
    got = pkg.shard.broadcast_blob(blob, src=0)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), top=top, b=b, e=e, w=w_rx, blob_sum=int(got.astype(np.uint64).sum()),
             blob_len=got.size)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_broadcast(tmp_path, pkg, oracle, synth):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s = synth.shape("shard", 7, 16, 14, 14, 24, 3, pad=1, group=2, sparsity=0.8)
    w = synth.pruned_weights(s, 5)
    x = synth.activations(s, 9)
    geom = oracle.geom(s.C, s.H, s.W, s.M, s.KH, s.KW, s.pad_h, s.pad_w, group=s.group)
    full = oracle.conv_forward(geom, x, w, None, gate=False)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2)]
    assert [(int(p["b"]), int(p["e"])) for p in parts] == [(0, 4), (4, 7)]
    for p in parts:
        assert np.array_equal(p["w"].reshape(w.shape), w)      # broadcast delivered the weights
    assert np.array_equal(np.concatenate([p["top"] for p in parts]), full)
    want = np.arange(100003, dtype=np.uint64).view(np.uint8)[5:]
    for p in parts:
        assert int(p["blob_len"]) == want.size and int(p["blob_sum"]) == int(want.astype(np.uint64).sum())


def test_shard_range_covers_batch(pkg):
    for n in (0, 1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [pkg.shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.shard.shard_range(8, 2, 2)


def test_pack_unpack_roundtrip(pkg):
    rp = np.array([0, 1, 3, 0, 2, 2], np.int32)
    ci = np.array([4, 0, 8, 1, 5], np.int32)
    va = np.array([0.5, -1.25, 3e-8, 7, -0.0], np.float32)
    ng = np.array([3, 2], np.int32)
    buf = pkg.shard.pack_csr(rp, ci, va, ng)
    assert len(buf) == pkg.shard.packed_len(2, 6, 5)
    a, b, c, d = pkg.shard.unpack_csr(buf)
    assert np.array_equal(a, rp) and np.array_equal(b, ci) and np.array_equal(d, ng)
    assert c.tobytes() == va.tobytes()

"""Batch-dimension sharding of the sconv forward over the GPUs of one node.

Images are independent in the forward pass (reference: conv_layer.cu:19 loops over n), so the
path shards with NO data-path collective: rank r owns images [r*N/P, (r+1)*N/P).  The only
exchange is a one-time broadcast of the sparse weights from rank 0 -- the counterpart of
NCCL<Dtype>::Broadcast (src/caffe/parallel.cpp:189-200), here ``torch.distributed.broadcast``
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np


def shard_range(n_images, rank, world):
    """[begin, end) of rank's contiguous slice; the first (n % world) ranks get one extra image."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    base, extra = divmod(n_images, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def pack_csr(rowptr, colidx, values, nnz_per_group):
    """One int32 buffer [n_groups, nnz_per_group..., rowptr..., colidx..., values-as-bits...]:
    a single message per layer instead of four (bigger, fewer collectives)."""
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colidx = np.ascontiguousarray(colidx, np.int32)
    values = np.ascontiguousarray(values, np.float32)
    ng = np.ascontiguousarray(nnz_per_group, np.int32)
    head = np.array([len(ng), len(rowptr), len(colidx)], np.int32)
    return np.concatenate([head, ng, rowptr, colidx, values.view(np.int32)])


def unpack_csr(buf):
    buf = np.ascontiguousarray(buf, np.int32)
    n_g, n_rp, nnz = (int(v) for v in buf[:3])
    o = 3
    ng = buf[o:o + n_g]; o += n_g
    rp = buf[o:o + n_rp]; o += n_rp
    ci = buf[o:o + nnz]; o += nnz
    va = buf[o:o + nnz].view(np.float32)
    return rp.copy(), ci.copy(), va.copy(), ng.copy()


def packed_len(n_groups, n_rowptr, nnz):
    return 3 + n_groups + n_rowptr + 2 * nnz


def broadcast_csr(csr, n_groups, n_rowptr, nnz, src=0, device="cpu", group=None):
    """Broadcast one layer's CSR from rank ``src``.  ``csr`` = (rowptr, colidx, values,
    nnz_per_group) on the source rank, ignored elsewhere; sizes are known to every rank from
    the layer geometry and the exact-count sparsity (or from a prior size broadcast)."""
    import torch
    import torch.distributed as dist
    n = packed_len(n_groups, n_rowptr, nnz)
    if dist.get_rank(group) == src:
        t = torch.from_numpy(pack_csr(*csr)).to(device)
        assert t.numel() == n
    else:
        t = torch.empty(n, dtype=torch.int32, device=device)
    dist.broadcast(t, src=src, group=group)
    return unpack_csr(t.cpu().numpy())


def broadcast_blob(blob, src=0, device="cpu", group=None, keep_on_device=False):
    """Broadcast a byte blob (numpy uint8 on the source rank, ignored elsewhere) whose size only the source
    knows: the aligned form of a layer (escoin_plan_export_aligned: CSR + channel deal + unit table + code
    object), so that receivers load the code rank 0 generated instead of generating their own.
    keep_on_device: return the tensor the collective filled (a CUDA tensor under RCCL) -- Plan.import_aligned takes
    it as it is; otherwise a numpy copy."""
    import torch
    import torch.distributed as dist
    is_src = dist.get_rank(group) == src
    n = torch.tensor([int(blob.size) if is_src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src, group=group)
    if is_src:
        t = torch.from_numpy(np.ascontiguousarray(blob, np.uint8)).to(device)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src, group=group)
    if keep_on_device and t.is_cuda:
        return t
    return t.cpu().numpy()

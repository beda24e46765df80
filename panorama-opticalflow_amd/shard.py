"""Multi-GPU host logic of the path: overlap pairs are independent units, so they
are sharded round-robin over ranks (one process per GPU) with NO collective on the data path; the only
exchange is the final gather of the blended strips to rank 0 (RCCL over xGMI on GPUs, gloo in CPU tests).
"""
import torch
import torch.distributed as dist


def pairs_for_rank(n_pairs, rank, world):
    """Static round-robin: pair i runs on rank i % world (SURVEY.md 8(e) 'Partitioning')."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_pairs, world))


def run_sharded(n_pairs, rank, world, make_pair, process_pair):
    """Each rank processes its own pairs; returns {pair_index: result_tensor} for the local shard."""
    return {i: process_pair(*make_pair(i)) for i in pairs_for_rank(n_pairs, rank, world)}


def gather_to_rank0(local, n_pairs, rank, world, like):
    """Final gather of per-pair results (all the same shape/dtype as `like`) to rank 0, in pair order.
    Uses grouped send/recv semantics of dist.gather round by round (round j moves pair j*world + r)."""
    out = [None] * n_pairs if rank == 0 else None
    rounds = (n_pairs + world - 1) // world
    for j in range(rounds):
        idx = j * world + rank
        mine = local.get(idx)
        buf = mine if mine is not None else torch.zeros_like(like)
        if world == 1:
            if idx < n_pairs:
                out[idx] = buf
            continue
        recv = [torch.empty_like(like) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, recv, dst=0)
        if rank == 0:
            for r in range(world):
                k = j * world + r
                if k < n_pairs:
                    out[k] = recv[r]
    return out


class OverlappedGather:
    """The path's only exchange, taken off the critical path: the gather of pair k's result to rank 0 runs (on the
    collective's own stream) while pair k+1 is being computed.  Results alternate between two buffers so that the one a
    gather is still reading is never the one being written; at most one gather is in flight."""

    def __init__(self, like, world, rank):
        self.world, self.rank, self.pending, self.k = world, rank, None, 0
        self.bufs = [like, torch.empty_like(like)]
        self.recv = [[torch.empty_like(like) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]

    def out_buffer(self):
        """where the next result has to be written"""
        return self.bufs[self.k % 2]

    def submit(self):
        """start gathering the buffer just written; waits for the previous gather first"""
        self.wait()
        j = self.k % 2
        self.pending = dist.gather(self.bufs[j], self.recv[j], dst=0, async_op=True)
        self.k += 1

    def wait(self):
        """Host-blocking: on return the previous gather has finished READING its buffer.  For the NCCL/RCCL backend
        Work.wait() only orders torch's current stream, and libpanoflow writes the buffers from its own HIP streams,
        so the host additionally waits for that stream before the buffer may be overwritten."""
        if self.pending is not None:
            self.pending.wait()
            if self.bufs[0].is_cuda:
                torch.cuda.current_stream(self.bufs[0].device).synchronize()
            self.pending = None

    def last(self):
        """rank 0: the most recently gathered list of results (after wait())"""
        return self.recv[(self.k - 1) % 2] if self.k else None


class RcclGather:
    """Same job as OverlappedGather, but through the product's own C ABI (pf_dist_*: grouped ncclSend/ncclRecv on a
    dedicated HIP stream inside libpanoflow.so) -- torch is not on the data path; it only carried the 128-byte
    ncclUniqueId to the ranks.  `dist_obj` is a pyabi.Dist; buffers are torch CUDA tensors used as raw HBM."""

    def __init__(self, like, dist_obj):
        self.d, self.k = dist_obj, 0
        self.bufs = [like, torch.empty_like(like)]
        self.nbytes = like.numel() * like.element_size()
        self.recv = [torch.empty((dist_obj.world,) + tuple(like.shape), dtype=like.dtype, device=like.device) for _ in range(2)] if dist_obj.rank == 0 else [None, None]

    def out_buffer(self):
        return self.bufs[self.k % 2]

    def submit(self):
        """start gathering the buffer just written (the producing pf_* call is synchronous on return); pf_dist_gather_async
        itself first waits for the previous gather, so a buffer is never overwritten while it is being sent"""
        j = self.k % 2
        self.d.gather_async(self.bufs[j].data_ptr(), self.recv[j].data_ptr() if self.recv[j] is not None else 0, self.nbytes)
        self.k += 1

    def wait(self):
        self.d.wait()

    def last(self):
        return self.recv[(self.k - 1) % 2] if self.k else None


def max_over_ranks(seconds, device="cpu"):
    """The job's time is the slowest rank's (bench.py contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

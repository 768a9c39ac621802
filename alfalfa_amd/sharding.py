"""Multi-GPU plan for the path: replicas sharded by independent stream, no data-path collective.

One process per GPU.  Rank r of W owns streams [r*S, (r+1)*S) (weak scaling: fixed work per GPU).  The only exchange is
the ExCamera-style entry-state hand-off before steady state: DecoderState blob (host, ~1.2 KB + segmentation map) and the
last reference raster (device, 3.1 MB at 1080p) broadcast from the rank that decoded the shared GOP head; RCCL over xGMI
on GPUs (`torch.distributed` backend "nccl"), gloo in the CPU tests (decoder.cc:43-52,171-175, decode-bundle.cc:56-99).
"""
import hashlib


def stream_ids(rank, world, streams_per_gpu, first=100):
    """Global stream ids (== synthetic seeds) owned by `rank`; disjoint across ranks, contiguous, weak scaling."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return [first + rank * streams_per_gpu + i for i in range(streams_per_gpu)]


def broadcast_bytes(dist, payload, src, device=None):
    """Broadcast a bytes object from `src` with torch.distributed (length first, then payload)."""
    import torch
    rank = dist.get_rank()
    n = torch.tensor([len(payload) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def digests_agree(dist, digest, device=None):
    """All ranks pass a 32-byte digest; returns True on every rank iff all are identical."""
    import torch
    world = dist.get_world_size()
    mine = torch.frombuffer(bytearray(digest), dtype=torch.uint8).to(device) if device is not None else torch.frombuffer(bytearray(digest), dtype=torch.uint8)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return all(bool((g == gathered[0]).all()) for g in gathered)


def sha256(b):
    return hashlib.sha256(b).digest()

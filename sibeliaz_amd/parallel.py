"""Multi-GPU block finding: one rank per GPU over torch.distributed (backend "nccl" = RCCL over xGMI; gloo in CPU tests).

The reference has no distributed path (OpenMP only). The exact round engine (csrc/engine.cpp) deals the
seeds of each speculative round to the ranks round-robin; each rank runs its share on its own MI355X (tables and the
`used` bitmap are replicated), then per-seed results and footprints are all-gathered — two collectives per round (sizes,
padded payload), KBs to a few MB, latency bound — and every rank runs the identical ordered commit, so all ranks hold
the same `used` state and block list without further traffic. The job launches that recompute invalidated or conflicting
seeds against predicted `used` views are deterministic, latency-bound chains; every rank repeats them on its own GPU.

This module only supplies the all-gather callback the C++ engine calls; the engine, commit and kernels are native.
"""
import ctypes as C

import numpy as np

from .api import ALLGATHER_CB, INSTANCE_DTYPE, MARK_CB, PROCESS_CB, RESET_CB, SEED_DTYPE, Hooks


def make_allgather(group=None, tensor_device=None):
    """Returns (callback, keepalive). The callback all-gathers `bytes` bytes from every rank into recv[world * bytes]."""
    import torch
    import torch.distributed as dist

    dev = tensor_device if tensor_device is not None else torch.device("cpu")
    world = dist.get_world_size(group)

    def cb(_user, send, nbytes, recv):
        try:
            n = int(nbytes)
            src = np.frombuffer((C.c_uint8 * n).from_address(send), dtype=np.uint8)
            t = torch.from_numpy(src.copy()).to(dev)
            out = torch.empty(world * n, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(out, t, group=group)
            C.memmove(recv, out.cpu().numpy().ctypes.data, world * n)
            return 0
        except Exception as e:  # noqa: BLE001 - reported through the C return code
            print("all-gather callback failed:", e, flush=True)
            return 1

    fn = ALLGATHER_CB(cb)
    return fn, (fn, cb)


def make_hooks(rank=0, world=1, group=None, tensor_device=None, processor=None, round_phases=0, **engine):
    """Builds an api.Hooks. `processor` (optional, tests) is an object with process(seeds)->(offsets, inst), mark(ranges),
    reset() that stands in for the device."""
    keep = []
    h = Hooks()
    h.rank, h.world, h.round_phases, h.progress = rank, world, round_phases, 0
    for k, v in engine.items():
        setattr(h, k, int(v))
    if world > 1:
        fn, ka = make_allgather(group, tensor_device)
        h.allgather = fn
        keep.append(ka)
    if processor is not None:
        def process(_u, seeds_ptr, n, offsets, inst_ptr, cap):
            try:
                seeds = np.frombuffer((C.c_char * (n * SEED_DTYPE.itemsize)).from_address(seeds_ptr), dtype=SEED_DTYPE, count=n)
                off, inst = processor.process(seeds)
                for i in range(n + 1):
                    offsets[i] = int(off[i])
                if int(off[n]) > cap:
                    return 1
                inst = np.ascontiguousarray(inst, dtype=INSTANCE_DTYPE)
                if len(inst):
                    C.memmove(inst_ptr, inst.ctypes.data, inst.nbytes)
                return 0
            except Exception as e:  # noqa: BLE001
                print("process callback failed:", e, flush=True)
                return -1

        def mark(_u, ranges, n):
            r = np.frombuffer((C.c_uint64 * (2 * n)).from_address(ranges), dtype=np.uint64).reshape(-1, 2)
            processor.mark(r)
            return 0

        def reset(_u):
            processor.reset()
            return 0

        h.process, h.mark, h.reset = PROCESS_CB(process), MARK_CB(mark), RESET_CB(reset)
        keep.append((h.process, h.mark, h.reset, process, mark, reset))
    return h, keep


def find_blocks_distributed(finder, min_block, max_branch, seeds, device, group=None, tensor_device=None, round_phases=0):
    """BlocksFinder::FindBlocks across the ranks of `group`; returns the pre-trim blocks (identical on every rank)."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    hooks, keep = make_hooks(rank, world, group, tensor_device, None, round_phases)
    blocks = finder.FindBlocks(min_block, max_branch, device=device, seeds=seeds, hooks=hooks)
    del keep
    return blocks, finder.stats

"""HIP stream roles of a training step.

The data-gradient chain (forward, dgrad GEMMs, attention / LayerNorm backward, optimiser) is the critical path; the
weight-gradient GEMMs run on a side stream (towers.TransformerStack.backward).  Running the chain on a HIGH-priority
stream makes the dispatcher hand free CUs to the chain first, so the side stream only takes what the chain leaves:
measured +1.1 % pairs/s on ViT-B/16 (one MI355X, interleaved runs).  ``SC_CHAIN_PRIO=0`` keeps the caller's stream.
"""
import contextlib
import os

import torch

_chain = {}


@contextlib.contextmanager
def chain_stream():
    """Context: run the enclosed work on this device's high-priority chain stream, ordered after what the caller's
    stream has enqueued so far; the caller's stream waits for it on exit."""
    if os.environ.get("SC_CHAIN_PRIO", "1") == "0" or not torch.cuda.is_available():
        yield None
        return
    dev = torch.cuda.current_device()
    s = _chain.get(dev)
    if s is None:
        s = _chain[dev] = torch.cuda.Stream(priority=-1)
    cur = torch.cuda.current_stream()
    if cur == s:
        yield s
        return
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        yield s
    cur.wait_stream(s)


def consumer_streams():
    """The caller's current stream and, when they exist, this device's chain stream and second-tower stream: the streams that
    will read a tensor handed to the training loop by another stream (data producer -> ``Tensor.record_stream``)."""
    cur = torch.cuda.current_stream()
    out = [cur]
    for s in (_chain.get(torch.cuda.current_device()), _tower.get(torch.cuda.current_device())):
        if s is not None and s != cur:
            out.append(s)
    return out


_tower = {}


def tower_stream(device=None):
    """This device's stream for the SECOND tower when the two towers of a step run side by side (net.py): the text tower /
    gene transformer and the vision tower are independent until the loss, and at small token counts neither fills the chip
    (ViT-B-32 + CLIP text tower at batch 32: 20-84 tiles per GEMM launch on 256 CUs).  Same priority as the chain."""
    key = torch.cuda.current_device() if device is None else torch.device(device).index
    s = _tower.get(key)
    if s is None:
        s = _tower[key] = torch.cuda.Stream(priority=-1)
    return s


_comm = {}


def comm_stream(device=None):
    """This device's ONE communication stream (high priority): the feature all-gathers of the step (comm.FeatureGather) and
    the weight all-gathers / copy refreshes of the sharded optimiser (comm.ShardedGradExchange) share it.  Every extra HIP
    stream is another candidate for one of the few hardware queues of the device (4 by default): with a fifth busy stream
    the low-priority weight-gradient stream ended up multiplexed with the chain and the overlapped backward ran 2.3 ms
    SLOWER (measured, one rank, profiles/r05_grad_exchange_one_rank.txt) -- streams are created per ROLE, not per object."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _comm.get(key)
    if s is None:
        s = _comm[key] = torch.cuda.Stream(device=dev, priority=-1)
    return s

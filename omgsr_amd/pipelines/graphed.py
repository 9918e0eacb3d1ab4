"""HIP-graph replay of a pipeline's forward pass (the brief's "HIP streams and graphs instead of a tracing compiler").

The reference driver is strictly batch 1 (infer/infer_omgsr_s.py:92-93): one 128 -> 512 image is ~1 000 kernel launches of a few
microseconds each, and issuing them from Python costs more host time than the GPU needs to run them (a ctypes call, argument
structs, output allocation per launch). Every launch of the hot path has fixed arguments once the input shape, the tier and the
prompt are fixed (weights, folded constants and cached K / V^T live at fixed addresses; the caching allocator replays its
allocations inside a captured region), so the whole `run()` body of `forward()` is captured ONCE per (shape, dtype, tile geometry,
prompt tensor, tier, policy) into a hipGraph and replayed afterwards: one host call per image.

What stays outside the graph, exactly where the reference has it: the synchronisation on both sides of the timed region, and the
accurate tier's fp16 range-guard read (one 4-byte read at that synchronisation; a call that overflowed is recomputed eagerly by
precision.RangeFallback and the graphs of the fp16-operand mode are dropped).

Use: `pipe.enable_graphs()` (or OMGSR_GRAPH=1); `forward()` keeps its signature and return value. The first call with a new key runs
eagerly (it also warms the weight packing, constant folding and kernel attributes, none of which may happen under capture), the second
captures, later ones replay. Inputs are copied into the graph's static input buffer (12 MB per 1024^2 image: ~3 us of HBM time);
the returned image is a fresh tensor, like the reference's.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch


class WeightsStamp:
    """Cheap identity of every parameter of the pipeline's models for the graph key: (owner, name, parameter) triples are cached and
    re-walked when an owner no longer holds that object (load_state_dict(assign=True), swap_tensors-style conversion), and the stamp
    folds each parameter's version AND storage address (`.to()` / `.data =` replace the storage without a version bump)."""

    def __init__(self, *models):
        self.models = [m for m in models if m is not None]
        self._owners = None

    def __call__(self) -> tuple:
        own = self._owners
        if own is None or not all(m._parameters.get(n) is p for m, n, p in own):
            own = self._owners = [(sub, n, p) for model in self.models for sub in model.modules()
                                  for n, p in sub._parameters.items() if p is not None]
        v = a = 0
        for _, _, p in own:
            v += p._version
            a = (a * 1000003 + p.data_ptr()) & 0xFFFFFFFFFFFFFFFF      # order-dependent: two parameters swapping storage do not cancel
        return (len(own), v, a)


class GraphCache:
    def __init__(self, max_graphs: int = 8):
        self.enabled = False
        self.max_graphs = max_graphs
        self._seen: Dict[tuple, int] = {}
        self._graphs: Dict[tuple, tuple] = {}
        self._eager_only: set = set()       # keys whose capture failed: run eagerly from then on instead of retrying the capture per call
        self._last = None                   # (key, cache epoch) when the previous call returned: a capture needs the one-slot caches WARM for its key
        self.capture_failures: Dict[str, int] = {}      # reason -> count (a capture that throws is not silent)
        self.replays = 0
        self.captures = 0
        self.stale_drops = 0

    def clear(self) -> None:
        self._graphs.clear()
        self._seen.clear()
        self._eager_only.clear()
        self._last = None

    def _eager(self, full, fn, dynamic):
        from .. import nn as onn
        out = fn(*dynamic)
        self._last = (full, onn.cache_epoch())
        return out

    @staticmethod
    def _ident(t: Optional[torch.Tensor]):
        return None if t is None else (id(t), t._version, t.data_ptr(), tuple(t.shape), t.dtype)

    def call(self, key: tuple, dynamic: list, fixed: list, fn: Callable[..., torch.Tensor]) -> torch.Tensor:
        """fn(*dynamic_inputs) -> output tensor. `dynamic`: tensors whose VALUES change per call (the LQ image): copied into static
        buffers. `fixed`: tensors read by address (prompt embeddings, ids, posterior noise override): part of the key by identity and
        version - a new tensor object, or an in-place edit, is a new graph.
        A graph also reads, by address, device values that one-slot caches OWN (cross-attention K / V^T of the current prompt, FLUX
        modulation / rope tables, packed weights): every cache rebuild bumps nn.cache_epoch(), each graph remembers the epoch it was
        captured under, and a graph whose epoch is not the current one is dropped instead of replayed (prompt A -> B -> A: graph A's
        K / V^T were freed when B's replaced them in the slot)."""
        from .. import nn as onn, ops, precision
        from .._lib import OmgsrError
        if not self.enabled:
            return fn(*dynamic)
        # everything that selects kernels or packed-weight forms without touching a parameter: tier, precision policy, batch-invariant
        # dispatch, the fp16 range guard (it adds the overflow word to every launch)
        state = (ops.mode_key(), precision.policy_epoch(), ops._BATCH_INVARIANT, ops._GUARD)
        full = (key, state, tuple((tuple(d.shape), d.dtype, str(d.device)) for d in dynamic), tuple(self._ident(t) for t in fixed))
        hit = self._graphs.get(full)
        if hit is not None and hit[4] != onn.cache_epoch():
            del self._graphs[full]              # some cache rebuilt since the capture: its old value (baked into the graph) may be freed
            self._seen[full] = 0
            self.stale_drops += 1
            hit = None
        if hit is not None:
            g, static_in, static_out, keep, _ = hit
            for s, d in zip(static_in, dynamic):
                s.copy_(d)
            g.replay()
            self.replays += 1
            self._last = (full, onn.cache_epoch())
            return static_out.clone()
        if full in self._eager_only:
            return self._eager(full, fn, dynamic)
        if len(self._eager_only) > 256:      # (bounded like _seen: a service must not grow a set per failed key forever)
            self._eager_only.clear()
        n = self._seen.get(full, 0)
        if len(self._seen) > 4096:       # keys are cheap (ints / shapes), but a service that sees ever-new prompts must not grow without bound
            self._seen.clear()
        # Capture only over WARM caches: the immediately preceding call had this key and no cache has rebuilt since it returned. A builder that
        # runs under capture is recorded, not executed - prompt A, B, A as single calls reaches the third call with _seen[A] == 1 while the
        # cross-attention K / V^T slot holds B's: that call runs eagerly (it re-warms the slot) and counts as a first sighting.
        warm = n > 0 and self._last == (full, onn.cache_epoch())
        self._seen[full] = n + 1 if warm else 1
        if not warm:                     # first sight / cold caches: eager (warms packed weights, folded constants, kernel attributes, the allocator)
            return self._eager(full, fn, dynamic)
        if len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        static_in = [d.clone() for d in dynamic]
        torch.cuda.synchronize()
        epoch0 = onn.cache_epoch()
        g = torch.cuda.CUDAGraph()
        onn.begin_rebuild_log()
        try:
            with torch.cuda.graph(g):
                static_out = fn(*static_in)
        except RuntimeError as exc:
            # a capture that throws (an op that synchronises, an allocation the graph pool refuses: both surface as RuntimeError) must not be
            # retried on every call; anything else (a kernel wrapper's OmgsrError, a TypeError) is a real error and propagates. Values built under
            # the aborted capture never ran: empty those slots before the eager run.
            for invalidate in onn.end_rebuild_log():
                invalidate()
            if isinstance(exc, OmgsrError):
                raise
            self._eager_only.add(full)
            reason = f"{type(exc).__name__}: {str(exc).splitlines()[0] if str(exc) else ''}"[:200]
            self.capture_failures[reason] = self.capture_failures.get(reason, 0) + 1
            torch.cuda.synchronize()
            return self._eager(full, fn, dynamic)
        except BaseException:
            for invalidate in onn.end_rebuild_log():
                invalidate()
            raise
        rebuilt = onn.end_rebuild_log()
        if onn.cache_epoch() != epoch0:
            # a cache rebuilt DURING the capture: its value lives in the graph's private pool, was never computed (the graph is not replayed)
            # and the slot now points at it - drop the graph, EMPTY every such slot, and run eagerly (which rebuilds them for real). The warm
            # check above exists so that this never happens; if it does, stay eager for this key.
            for invalidate in rebuilt:
                invalidate()
            self._eager_only.add(full)
            self.capture_failures["cache rebuilt during capture"] = self.capture_failures.get("cache rebuilt during capture", 0) + 1
            del g
            torch.cuda.synchronize()
            return self._eager(full, fn, dynamic)
        self._graphs[full] = (g, static_in, static_out, tuple(fixed), epoch0)      # `fixed` kept alive: their addresses are baked into the graph
        self.captures += 1
        g.replay()
        self.replays += 1
        self._last = (full, onn.cache_epoch())
        return static_out.clone()

"""Host-side watch over the persistent kernels' abort words.

A persistent launch (csrc/opnet_xcd_kernels.hip, opnet_xcd4_kernels.hip, seq_xcd_kernels.hip) bounds every spin: a workgroup
that cannot see its producers for 1.5 s raises an abort word in the launch's workspace, every poller leaves and the outputs
are NaN.  The reference's model call (baselines/inference_main.py:203-207, training_main.py:186) cannot fail that way, so the
callers here must never consume such an output: behind every persistent launch the module enqueues one 16-byte copy of the
status words into pinned host memory and an event (`watch`); at the next point where the caller synchronises anyway it calls
`verify`, which re-runs every aborted batch through the launch-per-step chain INTO THE SAME OUTPUT TENSORS (results the caller
already holds views of are healed in place) and warns once.  Nothing here synchronises on the hot path.

A watch entry keeps its launch's input alive through `redo` (NonLinearLstm's features: 4.6 MB per clip), so entries do not wait
for `verify`: every `watch` first drops the entries whose launch has already completed clean (`event.query()`, no sync) - a
driver that defers `verify` to the end of a data set holds the inputs of the launches still in flight, not of all of them.
"""
from __future__ import annotations

import warnings
from typing import Callable, List, Optional, Tuple

import torch

_SLOTS = 64
_warned = False


class LaunchMonitor:
    def __init__(self):
        self._host: Optional[torch.Tensor] = None          # pinned [SLOTS, 4] int32
        self._free: List[int] = list(range(_SLOTS))
        self._pending: List[Tuple[torch.cuda.Event, int, Optional[Callable[[], None]], str]] = []
        self.aborted = 0                                   # launches found aborted so far
        self.healed = 0                                    # ... of which re-run on the chain

    def watch(self, workspace: torch.Tensor, offset: int, redo: Optional[Callable[[], None]], what: str) -> None:
        """call right after the persistent launch was enqueued on the current stream; `workspace` is the launch's uint8
        workspace, `offset` the byte offset of its status words; `redo()` re-runs the batch on the chain into the same outputs"""
        if self._host is None:
            self._host = torch.zeros((_SLOTS, 4), dtype=torch.int32).pin_memory()
        self.reap()
        if not self._free:
            self.verify(limit=1)                           # 64 launches in flight: wait for the oldest, frees its slot
        slot = self._free.pop()
        self._host[slot].copy_(workspace[offset:offset + 16].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(workspace.device))
        self._pending.append((ev, slot, redo, what))

    def pending(self) -> int:
        return len(self._pending)

    def reap(self) -> int:
        """drop every watched launch that has completed with clean status words - without waiting for any (event.query()) -
        and with it the reference to its input that `redo` holds; an aborted one stays for `verify` to heal.  Returns how
        many entries were dropped."""
        keep = []
        for entry in self._pending:
            ev, slot, _redo, _what = entry
            if ev.query() and int(self._host[slot, 0]) == 0:
                self._free.append(slot)
            else:
                keep.append(entry)
        n = len(self._pending) - len(keep)
        self._pending = keep
        return n

    def _close(self, entry) -> int:
        """one watched launch whose event is (made) complete: free its slot; heal it if it aborted.  Returns 1 if it had."""
        global _warned
        ev, slot, redo, what = entry
        ev.synchronize()
        code, block, phase, _ = (int(v) for v in self._host[slot])
        self._free.append(slot)
        if code == 0:
            return 0
        self.aborted += 1
        if not _warned:
            _warned = True
            warnings.warn(f"objectpermanence_amd: a persistent launch ({what}) gave up (code {code}, block {block}, phase "
                          f"{phase}); the batch is re-run on the launch-per-step chain", RuntimeWarning, stacklevel=3)
        if redo is None:
            raise RuntimeError(f"persistent launch ({what}) aborted (code {code}, block {block}, phase {phase}) and "
                               "cannot be re-run here")
        redo()
        self.healed += 1
        return 1

    def verify(self, limit: Optional[int] = None) -> int:
        """wait for the watched launches (oldest first; at most `limit` of them) and heal the aborted ones; returns how many
        were aborted.  An aborted launch without a redo raises."""
        todo = self._pending if limit is None else self._pending[:limit]
        self._pending = [] if limit is None else self._pending[limit:]
        return sum(self._close(entry) for entry in todo)

    def settle(self) -> int:
        """the non-blocking form: close (and heal) every watched launch that HAS completed, leave the ones still running.  A
        caller that knows a later event of the same stream has completed (DeferredConsumer) may then use that launch's output."""
        done, keep = [], []
        for entry in self._pending:
            (done if entry[0].query() else keep).append(entry)
        self._pending = keep
        return sum(self._close(entry) for entry in done)


def _monitors(model: torch.nn.Module):
    for owner in (model, getattr(model, "_runner", None)):
        mon = getattr(owner, "_monitor", None)
        if mon is not None:
            yield mon


class HostEvent:
    """stands in for torch.cuda.Event where the forward ran on the host (CPU tests of the drivers): always complete"""

    def record(self, *_):
        pass

    def query(self) -> bool:
        return True

    def synchronize(self) -> None:
        pass


class DeferredConsumer:
    """Outputs of enqueued forwards, consumed - post-processed, reduced to what the caller keeps, and DROPPED - as soon as their
    launch is known complete and clean, instead of holding a data set's outputs until one sync at its end (the evaluation loops of
    training_main.py:59-93 / inference_main.py:191-217 hold ~30 KB per clip that way: fine for CATER's 5.5 k videos, unbounded in
    principle).  `add(event, *payload)`: `event` was recorded on the forward's stream after the forward (and its watch) was
    enqueued; `consume(*payload)` runs once the event has completed and every completed launch of the model has been settled
    (an aborted one re-run into the same tensors first).  Never blocks while fewer than `max_pending` entries wait."""

    def __init__(self, model: torch.nn.Module, consume: Callable[..., None], max_pending: int = 16):
        self.model, self.consume, self.max_pending = model, consume, int(max_pending)
        self._q: List[Tuple[torch.cuda.Event, tuple]] = []
        self.peak_pending = 0

    def add(self, event, *payload) -> None:
        self._q.append((event, payload))
        self.peak_pending = max(self.peak_pending, len(self._q))
        self.drain(block=len(self._q) > self.max_pending)

    def drain(self, block: bool = False, all_: bool = False) -> None:
        """consume what is ready; block: wait for the oldest entry first (all_: for every entry - the end of the data set)"""
        while self._q:
            ev, payload = self._q[0]
            if not ev.query():
                if not block:
                    return
                ev.synchronize()
            for mon in _monitors(self.model):
                mon.settle()
            self._q.pop(0)
            self.consume(*payload)
            block = all_


def verify_launches(model: torch.nn.Module) -> int:
    """verify every launch monitor found on `model` (and its runners); 0 when the model has none"""
    return sum(mon.verify() for mon in _monitors(model))

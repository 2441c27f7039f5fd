#!/usr/bin/env python3
"""HIP-event timeline of ONE training step, streams NOT serialised (rocprofv3's kernel trace runs the dispatches of
different streams one after the other on this pool: profiles/r03/step_r03_c2.txt shows the side-stream index build
finishing before the forward starts, which is not what an un-profiled step does).

    python3 profiles/step_timeline.py c2        # BASELINE configs[1] training step (SGD)
    python3 profiles/step_timeline.py c2adam

Every library call that enqueues work is bracketed by a pair of events on the stream it is issued to (the calls are
wrapped at the ctypes boundary, as `record_plan` does); offsets are relative to an event recorded on the main
stream just before the step.  An event pair brackets a CALL: a call of several launches shows as one line."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO]

import torch  # noqa: E402

import bench  # noqa: E402
from besskge import _native as nat  # noqa: E402
from besskge import runtime  # noqa: E402
from besskge.collectives import SingleProcessGroup  # noqa: E402


class EventLib:
    """The loaded library with an event pair around every call whose last argument is a stream."""

    def __init__(self, lib):
        self._lib, self.rec, self.on = lib, [], False

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        sig = nat.SIGNATURES.get(name)
        if not self.on or sig is None or not sig or sig[-1] is not nat._vp or not self._lib.bess_plan_knows(name.encode()):
            return fn

        def call(*args):
            st = torch.cuda.current_stream()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            rc = fn(*args)
            b.record(st)
            self.rec.append((name, st.cuda_stream, a, b))
            return rc

        return call


def main(workload: str) -> None:
    dev = torch.device("cuda", 0)
    model, sharding, k_pair = bench.build_c2(bench.N_ENTITY_C2, 1, 0, dev, SingleProcessGroup(1), False)
    batches = bench.make_batches_c2(1, 0, sharding, k_pair, pool=4, dev=dev)
    opt = runtime.Adam(lr=1e-3, weight_decay=1e-2) if workload == "c2adam" else 1e-3
    for i in range(6):
        model.train_step_replicas([batches[i % 4]], opt)
    torch.cuda.synchronize()
    proxy = EventLib(nat.load())
    nat._lib = proxy
    try:
        best = None
        for i in range(5):
            proxy.rec, proxy.on = [], True
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model.train_step_replicas([batches[i % 4]], opt)
            e1.record()
            torch.cuda.synchronize()
            proxy.on = False
            span = e0.elapsed_time(e1) * 1e3
            if best is None or span < best[0]:
                best = (span, e0, list(proxy.rec))
    finally:
        nat._lib = proxy._lib
    span, e0, rec = best
    main_stream = torch.cuda.current_stream().cuda_stream
    rows = sorted(((e0.elapsed_time(a) * 1e3, a.elapsed_time(b) * 1e3, st, name) for name, st, a, b in rec))
    print(f"# HIP-event timeline of one {workload} training step (best of 5; events on the calls' own streams, nothing serialised)")
    print(f"{'start us':>9} {'dur us':>8}  stream  call")
    busy_main = 0.0
    for t, d, st, name in rows:
        print(f"{t:9.1f} {d:8.1f}  {'main' if st == main_stream else 'side':>6}  {name}")
        if st == main_stream:
            busy_main += d
    print(f"# step span (main stream) {span:.1f} us; calls on the main stream sum to {busy_main:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "c2")

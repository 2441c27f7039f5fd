"""Time of the segment index build (bess_build_segment_index: own stable radix sort + run-length encode + scan)."""
import pathlib
import sys
import time

REPO = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "bess-kge_amd")]
import torch  # noqa: E402

from besskge import _native as nat  # noqa: E402

dev = torch.device("cuda", 0)
for n, rows in ((1 << 20, 93_773), (4096 * 256, 312_576), (8192 * 64, 62_500_000), (65_536, 93_773), (20_000, 93_773)):
    idx = torch.randint(0, rows, (n,), dtype=torch.int32, device=dev)
    for _ in range(3):
        nat.SegmentIndex(idx, rows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        nat.SegmentIndex(idx, rows)
    torch.cuda.synchronize()
    print(f"n = {n:>8} refs, {rows:>9} rows: {1e6 * (time.perf_counter() - t0) / reps:8.1f} us per index", flush=True)

"""host-side pieces of bench.py that only run with several ranks: the collective log (wraps torch.distributed's collectives, async Work handles
included) -- exercised here on CPU with a one-rank gloo group and a dummy event clock."""
import os
import sys
import time

import torch
import torch.distributed as dist

from conftest import ROOT

sys.path.insert(0, ROOT)


class _Clock:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


def test_collective_log_counts_blocking_and_async_collectives():
    import bench
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x, out = torch.ones(8), torch.empty(8)
        with bench._CollectiveLog(event_factory=_Clock) as log:
            for _ in range(2):
                dist.all_reduce(x)
                w = dist.all_reduce(x, async_op=True)
                dist.all_gather_into_tensor(out, x)
                w.wait()
                assert w.is_completed()                       # (the proxy forwards everything else)
        assert dist.all_reduce.__module__ != "bench"          # restored
        s = log.summary(2)
        assert s["per_step"] == 3
        rows = s["by_kind_and_payload"]
        assert rows["all_reduce:32B"]["per_step"] == 1 and rows["all_reduce(async):32B"]["per_step"] == 1 and rows["all_gather_into_tensor:32B"]["per_step"] == 1
        assert all(r["mean_us_stream_held"] is not None and r["mean_us_stream_held"] >= 0 for r in rows.values())
        assert s["sum_us_stream_held_per_step"] >= 0
    finally:
        dist.destroy_process_group()

"""Host cost of the stream-ordering calls of the binding (microseconds per call, idle streams)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "jpeg-encoder_amd", "python"))
import torch
from jpegenc_amd import binding as B

ctx = B.Context(0)
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
s = torch.cuda.current_stream(0)
side = torch.cuda.Stream()


def t(name, fn, n=2000):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print("%-46s %7.2f us" % (name, dt))


t("torch.cuda.current_stream(0)", lambda: torch.cuda.current_stream(0))
t("null stream .query()", lambda: s.query())
t("side stream .query()", lambda: side.query())
t("ctx.wait_for_stream(null)", lambda: ctx.wait_for_stream(0))
t("ctx.wait_for_stream(side)", lambda: ctx.wait_for_stream(side.cuda_stream))
t("ctx.stream_wait_for_ctx(null)", lambda: ctx.stream_wait_for_ctx(0))
t("ctx.stream_wait_for_ctx(side)", lambda: ctx.stream_wait_for_ctx(side.cuda_stream))
t("ctx._after_torch()", lambda: ctx._after_torch())
t("ctx._before_torch()", lambda: ctx._before_torch())
ctx.close()

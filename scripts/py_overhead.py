"""Where do the microseconds between kbench (C++ loop) and bench.py (Python loop) go?  Times the same one-call step
from Python with (a) bench.py's lambda (argument marshalling every call), (b) arguments bound once."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd as de
import bench
dev = torch.device("cuda", 0)
sc = bench.Scale(de, 12, 2, 192, 640, dev, 0, False)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); st = s.cuda_stream
p = lambda t: t.data_ptr()
args = (p(sc.K), p(sc.Kinv), p(sc.pose), p(sc.depth), p(sc.src), p(sc.tgt), 0, 12, 2, 192, 640, p(sc.warped), p(sc.out), p(sc.idt),
        p(sc.aux), p(sc.sums), p(sc.g_warped), p(sc.g_depth), p(sc.g_pose), 0, p(sc.ws_step), sc.ws_step_n, st)
fn = sc.lib.ds_loss_step_fwdbwd
def timed(body, n=300):
    for _ in range(20): body()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): body()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6
print("lambda (marshal per call): %.1f us per step (host issue %.1f us)" % timed(lambda: sc.one_call(st)))
print("bound args:                %.1f us per step (host issue %.1f us)" % timed(lambda: fn(*args)))
print("Scale.step:                %.1f us per step (host issue %.1f us)" % timed(lambda: sc.step(st)))

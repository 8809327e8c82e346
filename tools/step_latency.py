#!/usr/bin/env python3
"""Where does the ~1.45 ms per synchronized region go?  Host time of the pieces of one forward and the completion time of
k forwards launched into an idle GPU."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsgnet_amd.weights import synth_input

cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
dev = torch.device("cuda", 0)
model = model.to(dev)
x = synth_input(batch, wh, seed=0).to(dev)


def step():
    with torch.no_grad():
        return model(x)


for _ in range(30):
    step()
torch.cuda.synchronize()
for k in (1, 2, 3, 5, 10):
    rows = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    print(f"{k} forwards into an idle GPU: host enqueue ms / completion ms:", " ".join(f"{a:.3f}/{b:.3f}" for a, b in rows), flush=True)
plan = model.plan_for(x)
st = torch.cuda.current_stream(dev).cuda_stream
rows = []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.ops[0][1](st)
    t1 = time.perf_counter()
    plan.graph.replay()
    t2 = time.perf_counter()
    for _, op in plan.ops[len(plan.ops) - plan.n_tail:]:
        op(st)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append(tuple(round(v * 1e3, 3) for v in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))
print("one forward, idle GPU: host ms of [input launch, graph.replay(), tail launches, wait for completion]:", rows, flush=True)
# the graph alone, back to back, events around each replay
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
torch.cuda.synchronize()
for e0, e1 in evs:
    e0.record()
    plan.graph.replay()
    e1.record()
torch.cuda.synchronize()
print("graph.replay() alone, 12 back to back after a synchronize: device ms each:", " ".join(f"{e0.elapsed_time(e1):.3f}" for e0, e1 in evs))
print("   start-to-start ms:", " ".join(f"{evs[i][0].elapsed_time(evs[i + 1][0]):.3f}" for i in range(11)))
# eager launch list (no graph) into an idle GPU
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _, op in plan.ops:
        op(st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager launch list once: host enqueue {(t1 - t0) * 1e3:.3f} ms, completion {(t2 - t0) * 1e3:.3f} ms")
# where the host time of a forward call goes
import cProfile, pstats, io
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:170] for l in buf.getvalue().splitlines()[:60]))

"""Rehearsal of `bench.py --gpus N` (N = argv[1], default 2, at most 4: the GPU boxes allow six processes on the card) on a
ONE-GPU box: the ranks share cuda:0 and talk over gloo (NCCL refuses two ranks on one device).  Checks the control flow of the multi-rank path -- rendezvous, barriers, MAX over ranks, the gather,
rank 0 printing the one JSON line, clean shutdown -- not its speed.  Run: python tools/rehearse_bench_ranks.py"""
import json, os, subprocess, sys

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
assert 2 <= N <= 4

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, runpy
import torch, torch.distributed as dist
_init = dist.init_process_group
def init(backend=None, **kw):
    kw.pop("device_id", None)
    return _init("gloo", **kw)
dist.init_process_group = init
_ar, _ag = dist.all_reduce, dist.all_gather
def all_reduce(t, op=dist.ReduceOp.SUM, **kw):
    c = t.cpu(); _ar(c, op=op, **kw); t.copy_(c)
def all_gather(outs, t, **kw):
    co = [o.cpu() for o in outs]; _ag(co, t.cpu(), **kw)
    for o, c in zip(outs, co): o.copy_(c)
def all_gather_into_tensor(out, t, **kw):
    co = list(out.cpu().chunk(dist.get_world_size())); _ag(co, t.cpu(), **kw)
    out.copy_(torch.cat(co))
dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor = all_reduce, all_gather, all_gather_into_tensor
sys.argv = ["bench.py", "--gpus", "NRANKS", "--steps", "24", "--warmup", "4", "--no-cpu-baseline"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
'''.replace("ROOT", repr(ROOT)).replace("NRANKS", str(N))
procs = []
for rank in range(N):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(N), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29717")
    procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
outs = [p.communicate(timeout=600) for p in procs]
for rank, (p, (o, e)) in enumerate(zip(procs, outs)):
    print(f"rank {rank}: exit {p.returncode}, stdout lines {len(o.strip().splitlines())}")
    if p.returncode != 0:
        print(e[-2000:])
line = json.loads(outs[0][0].strip().splitlines()[-1])
for r in range(1, N):
    assert not any(l.startswith("{") for l in outs[r][0].splitlines()), "only rank 0 prints the JSON line: " + outs[r][0][:200]
assert line["n_gpus"] == N and line["steps"] == 24 and line["scaling"] == "weak"
img = line["image"]
assert img["n_gpus"] == N and img["scaling"] == "strong" and img["ms_per_image"] > 0 and 0.3 < img["acc_mean"] < 1.0, img
print("image line:", {k: img[k] for k in ("ms_per_image", "rays_per_s", "collective", "acc_mean")})
print("rank 0 line:", {k: line[k] for k in ("metric", "value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling")})

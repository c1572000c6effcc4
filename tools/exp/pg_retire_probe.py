"""What does the process group's flight recorder say about eager collectives the watchdog still tracks?  (trainer.py: _drain_collective_watchdog)"""
import os, pickle, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
if os.environ.get("TBS"):
    os.environ["TORCH_FR_BUFFER_SIZE"] = os.environ["TBS"]      # unset: the build's default
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
from torch._C._distributed_c10d import _dump_nccl_trace
for rep in range(3):
    w = dist.all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize()
    t0 = time.time()
    for i in range(40):
        tr = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
        ents = tr.get("entries", [])
        last = ents[-1] if ents else None
        if i == 0 and rep == 0:
            print("keys", list(tr.keys()), "entry keys", list(last.keys()) if last else None)
        if last is None: print("no entries"); break
        if last.get("retired"):
            print(f"rep {rep}: retired after {1e3 * (time.time() - t0):.1f} ms, state {last.get('state')}, n {len(ents)}"); break
        time.sleep(0.01)
    else:
        print("not retired in 400 ms", last.get("state"), last.get("retired"))
act = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=True)).get("entries", [])
print("active entries now:", len(act))
dist.destroy_process_group()

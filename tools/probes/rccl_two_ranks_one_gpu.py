"""Round 6 (VERDICT r5 "next" #7): can RCCL's own kernels be made resident on ONE GPU?  Two attempts, both recorded as they come out:
  (a) a process group of TWO ranks on cuda:0 with backend nccl (= RCCL) - the only way to make RCCL launch a real ring kernel on a one-GPU box;
  (b) a group of ONE rank looping a 405-MB all_reduce (the per-layer bucket size): does RCCL launch anything at all, and how long does a call take?
    python tools/probes/rccl_two_ranks_one_gpu.py            (on the GPU box)
Output: one paragraph per attempt (profiles/r06_n1_readiness.txt)."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def two_ranks(rank, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        t = torch.ones(1 << 20, device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        q.put((rank, "ok", float(t[0])))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error", f"{type(e).__name__}: {str(e)[:600]}"))


def one_rank(port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(405 * (1 << 20) // 2, dtype=torch.bfloat16, device="cuda:0")
    for _ in range(3):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"(b) one-rank group, all_reduce of {t.numel() * 2 / 2**20:.0f} MB x 50: {dt * 1e6:.1f} us per call, value unchanged: {float(t[0]) == 1.0} "
          f"-> {'RCCL moves nothing in a group of one rank (no kernel worth the name runs)' if dt < 200e-6 else 'a device copy / kernel runs per call'}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.set_start_method("spawn")
    q = mp.Queue()
    ps = [mp.Process(target=two_ranks, args=(r, 29641, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = []
    t_end = time.time() + 120
    while len(res) < 2 and time.time() < t_end:
        try:
            res.append(q.get(timeout=5))
        except Exception:  # noqa: BLE001
            if not any(p.is_alive() for p in ps):
                break
    for p in ps:
        if p.is_alive():
            p.kill()
        p.join(5)
    print("(a) two ranks on cuda:0, backend nccl (RCCL):", flush=True)
    if not res:
        print("    no result within 120 s (hung or both workers died)")
    for r in sorted(res):
        print(f"    rank {r[0]}: {r[1]}: {r[2]}")
    one_rank(29642)

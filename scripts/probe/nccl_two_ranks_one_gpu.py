"""Does RCCL accept two ranks on ONE device (so that the cross-rank RCCL path could be exercised on a one-GPU box)?  Spawns two
children (fresh processes; the parent never touches the GPU), 127.0.0.1 rendezvous, bounded by a timeout."""
import os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, torch.distributed as dist
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda:0"))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce ok, value {t[0].item()}", flush=True)
        dist.destroy_process_group()
    except Exception as exc:   # noqa: BLE001
        print(f"rank {rank}: {type(exc).__name__}: {str(exc)[:300]}", flush=True)
        sys.exit(3)
    sys.exit(0)
env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child"], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
t0 = time.time()
while time.time() - t0 < 60 and any(p.poll() is None for p in procs):
    time.sleep(0.5)
for p in procs:
    if p.poll() is None:
        p.kill()
        print("killed a rank after 60 s", flush=True)
print("exit codes", [p.wait() for p in procs])

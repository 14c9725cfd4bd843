import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.zeros(1, dtype=torch.float64, device="cuda"); t[0] = 1.5
dist.all_reduce(t); print("f64 sum", t.item())
dist.all_reduce(t, op=dist.ReduceOp.MAX); print("f64 max", t.item())
i = torch.arange(5, dtype=torch.int64, device="cuda"); dist.all_reduce(i); print("i64", i.tolist())
u = torch.full((4, 8), 3, dtype=torch.uint8, device="cuda"); dist.all_reduce(u, op=dist.ReduceOp.MAX); print("u8 max", int(u.max()))
g = [torch.empty((4, 8), dtype=torch.uint8, device="cuda")]
dist.gather(u, g, dst=0); print("gather ok", int(g[0].sum()))
dist.barrier(); dist.destroy_process_group(); print("nccl world-1 ok")

import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)
    torch.cuda.set_device(0); x = (torch.randn(190000) * (10.0 ** torch.randint(-6, 3, (190000,)).float())).cuda()
    dist.all_reduce(x)
    outs = [torch.zeros_like(x) for _ in range(world)]
    dist.all_gather(outs, x)
    if rank == 0:
        print(world, "ranks: identical across ranks:", all(torch.equal(outs[0], o) for o in outs), "max diff", max(float((outs[0]-o).abs().max()) for o in outs))
    dist.destroy_process_group()
if __name__ == "__main__":
    import socket
    for world in (2, 3, 4, 8):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(w, args=(world, port), nprocs=world, join=True)

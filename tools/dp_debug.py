import os, sys, random
import numpy as np, torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, port):
    import torch.distributed as dist
    from scat_amd import synth
    from tests.test_gpu_model import make_encoder
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0", SCAT_DIST_BACKEND="gloo")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    net = make_encoder(41)
    x = T(synth.images(500 + rank, 2)).cuda(); cot = T(synth.normal_like(600 + rank, "cot", (2, 66))).cuda()
    os.environ["WORLD_SIZE"] = "1"
    twin = make_encoder(41); random.seed(7); (twin(x)[0] * cot).sum().backward()
    os.environ["WORLD_SIZE"] = str(world)
    random.seed(7); (net(x)[0] * cot).sum().backward()
    torch.cuda.synchronize()
    b = net._dp_buckets
    for n in ["regressor.weight", "main_encoder.fc1.weight", "main_encoder.layer4.2.conv3.weight", "main_encoder.layer1.0.conv3.weight", "main_encoder.conv1.weight"]:
        p = dict(net.named_parameters())[n]; q = dict(twin.named_parameters())[n]
        g = q.grad.detach().cpu(); loc = g.clone(); dist.all_reduce(g); g /= world
        pg = p.grad.cpu()
        print(rank, n, "vs_avg", float((pg - g).abs().max() / g.abs().max()), "vs_local", float((pg - loc).abs().max() / loc.abs().max()), "vs_sum", float((pg - 2 * g).abs().max() / g.abs().max()), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join(300) for p in ps]

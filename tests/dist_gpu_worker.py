"""Worker of tests/test_gpu_dist.py: one rank of a sample-sharded run with the REAL engine (libhqt on cuda:0, gloo rendezvous --
several ranks share the one visible GPU; on an 8-GPU node each rank would own a device and the backend would be nccl/RCCL, as in
bench.py).  Started as a fresh child process per rank; rank 0 writes the gathered result."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, gb, steps, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    os.environ['HQT_PERSIST'] = '0'        # the ranks of this test share ONE GPU: the persistent AR chain wants the device to itself (bench.py does the same)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hqtransformer_amd import synth
    from hqtransformer_amd.config import load_config
    from hqtransformer_amd.dist import sample_and_decode_sharded, shard_bounds
    from hqtransformer_amd.models import ImageGPT2
    from hqtransformer_amd.pipeline import InflightSampler
    from hqtransformer_amd.sampling import sampling_ihqgpt
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model = ImageGPT2(load_config(os.path.join(ROOT, 'configs', 'tiny-cls.yaml')), seed=5).to(dev)
    cond = torch.from_numpy(synth.class_ids(7, gb, model.stage2.spec.n_classes))
    fast = mode == 'fast'

    def sample_fn(batch, cond_slice, seed, offset):
        return sampling_ihqgpt(model.stage2, num_candidates=batch, cond=cond_slice, top_k_top=50, top_p_top=0.9, top_k_bot=None, top_p_bot=None,
                               softmax_temperature=[1.0, 0.9], use_fp16=fast, is_tqdm=False, max_seq_len=steps, seed=seed, sample_offset=offset)

    def decode_fn(ct, cb):
        return model.stage1.decode_sequences(ct, cb, precision='fast' if fast else 'exact')
    res = sample_and_decode_sharded(sample_fn, decode_fn, gb, cond, seed=1234, gather='pixels')
    # host cost of keeping 3 lanes busy on this rank (informational: what 8 processes x 3 lanes would each spend per step)
    lo, hi = shard_bounds(gb, world, rank)
    pipe = InflightSampler(model, lanes=3, device=dev)
    for i in range(3):
        pipe.submit(hi - lo, int(cond[lo]), seed=i, max_seq_len=steps, use_fp16=True, sample_offset=lo)
    pipe.drain()
    t0 = time.perf_counter()
    n = 12
    for i in range(n):
        pipe.submit(hi - lo, int(cond[lo]), seed=10 + i, max_seq_len=steps, use_fp16=True, sample_offset=lo)
    host_ms = (time.perf_counter() - t0) * 1e3 / n
    pipe.drain()
    torch.cuda.synchronize()
    hosts = [None] * world
    dist.all_gather_object(hosts, host_ms)
    if rank == 0:
        np.savez(out_path, codes_top=res[0].cpu().numpy(), codes_bot=res[1].cpu().numpy(), pixels=res[2].cpu().numpy())
        with open(out_path + '.json', 'w') as fp:
            json.dump({'host_ms_per_submit_3_lanes': hosts}, fp)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

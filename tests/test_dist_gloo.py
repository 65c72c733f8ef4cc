"""The sharded sampling path on 2 CPU processes over gloo: slice bounds, global-index noise/cond keying and the
rank-0 gather must reproduce the unsharded run of the same global batch.  The per-rank engine is played by the CPU
oracle here (test infrastructure standing in for the GPU), so this covers exactly the N > 1 host logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hqtransformer_amd import synth
from hqtransformer_amd.dist import gather_to_rank0, sample_and_decode_sharded, shard_bounds
from hqtransformer_amd.spec import Stage1Spec, Stage2Spec
from oracle.hqt_oracle import OracleStage1, OracleStage2, rearrange_codes

S2 = Stage2Spec(embed_dim=64, n_layers=1, n_heads=2, n_layers_depth=1, vocab_top=64, vocab_bot=64, vocab_txt=16,
                ctx_len_img=16, ctx_len_txt=4, n_classes=10, cond=1, embedding=0)
S1 = Stage1Spec(ch=32, ch_mult=[1], num_res_blocks=1, attn_resolutions=[4], resolution=8, z_channels=32, embed_dim=16,
                n_embed=64, use_init_downsample=True)
GB, STEPS = 5, 4          # global batch 5 over 2 ranks: ragged 3 + 2


def _fns():
    o2 = OracleStage2(S2, synth.stage2_weights(S2, 1, 'fixture'))
    o1 = OracleStage1(S1, synth.stage1_weights(S1, 2, 'fixture'))
    noise = synth.exp_noise(3, STEPS, GB, S2.vocab_top)

    def sample_fn(batch, cond, seed, offset):
        ct, cb = o2.sample(cond.numpy(), batch, STEPS, noise[:, :, offset:offset + batch])
        return torch.from_numpy(ct), torch.from_numpy(cb)

    def decode_fn(ct, cb):
        gt, gb = rearrange_codes(ct.numpy(), cb.numpy(), 2)
        return torch.from_numpy(o1.decode_code(gt, gb))
    return sample_fn, decode_fn


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sample_fn, decode_fn = _fns()
    cond = torch.from_numpy(synth.class_ids(7, GB, 10))
    out = sample_and_decode_sharded(sample_fn, decode_fn, GB, cond, seed=0, gather='pixels')
    if rank == 0:
        q.put([t.numpy() for t in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_the_batch():
    for gb, w in ((512, 8), (5, 2), (3, 4), (64, 1)):
        spans = [shard_bounds(gb, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == gb
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_sharded_run_equals_unsharded():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sample_fn, decode_fn = _fns()
    cond = torch.from_numpy(synth.class_ids(7, GB, 10))
    ct, cb = sample_fn(GB, cond, 0, 0)
    px = decode_fn(ct, cb)
    assert (got[0] == ct.numpy()).all() and (got[1] == cb.numpy()).all()
    np.testing.assert_allclose(got[2], px.numpy(), atol=2e-5)   # BLAS blocking differs with the batch size

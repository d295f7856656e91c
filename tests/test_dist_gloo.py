"""N > 1 path: two gloo ranks (CPU), each solving its shard on the emulated kernels, one gather."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, emu_lib, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qpalm_amd.dist import solve_sharded
    from qpalm_amd.problems import random_qp
    from qpalm_amd.solver import Context, QpalmBatch
    ctx = Context(0, lib_path=emu_lib)
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    probs = [random_qp(24, 48, seed=700 + k, density_A=0.15, density_M=0.1) for k in range(5)]
    res = solve_sharded(probs, lambda ps: QpalmBatch(ctx, ps, ctx.default_settings(**st)), dist=dist)
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(emu_lib):
    from oracle import binding as ob
    from qpalm_amd.dist import shard_indices
    from qpalm_amd.problems import random_qp
    assert sorted(np.concatenate([shard_indices(5, 2, r) for r in range(2)]).tolist()) == list(range(5))
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, emu_lib, q)) for r in range(2)]
    for p in procs:
        p.start()
    X, Y, I = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    for k in range(5):
        p = random_qp(24, 48, seed=700 + k, density_A=0.15, density_M=0.1)
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        o.solve()
        assert int(I[k, 2]) == o.status_val and int(I[k, 0]) == int(o.info.iter)
        assert np.max(np.abs(X[k] - o.x)) <= 1e-9 * max(1.0, np.max(np.abs(o.x)))
        assert np.max(np.abs(Y[k] - o.y)) <= 1e-9 * max(1.0, np.max(np.abs(o.y)))
